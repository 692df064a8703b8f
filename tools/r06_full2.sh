#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do python -m pytest tests -m gpu -q > gpurun_out/r06_gputests_$i.log 2>&1; echo "gpu tests run $i rc=$?"; tail -2 gpurun_out/r06_gputests_$i.log; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/bench_r06c.json 2> gpurun_out/bench_r06c.err; echo "bench rc=$?"; wc -c gpurun_out/bench_r06c.json
grep "^\[bench\] [a-z0-9_]* ms=" gpurun_out/bench_r06c.err
