#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_spmm.py -q -x 2>&1 | tail -3
one() { python bench.py --workload spmm_banded --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), 'ms', d['parity_check'])"; }
one "default"
python bench.py --workload spmm --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3', round(d['ms_per_step'],3), 'ms', d['parity_check'])"
python tools/spmm_density.py > gpurun_out/r06_spmm_density.txt; cat gpurun_out/r06_spmm_density.txt
