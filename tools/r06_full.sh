#!/bin/bash
# the round-end sequence on one box: every GPU test, smoke(), the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r06_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r06_gputests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/bench_r06b.json 2> gpurun_out/bench_r06b.err; echo "bench rc=$?"; wc -c gpurun_out/bench_r06b.json
grep "^\[bench\] [a-z0-9_]* ms=" gpurun_out/bench_r06b.err
python -c "
import json; d=json.load(open('gpurun_out/bench_r06b.json')); print(d['ms_per_step'], d['roofline']['frac'], d['parity_check'], d['config']['value_contract'])"
