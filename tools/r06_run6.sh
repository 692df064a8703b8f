#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_transpose.py tests/test_gpu_dropin.py -q -x 2>&1 | tail -3
python bench.py --workload csc_spmv --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('csc', round(d['ms_per_step'],3), d['parity_check'], d['config']['plan_bytes_over_matrix'], d['config']['inspect_ms'], d['roofline']['traffic'])"
python bench.py --workload spmm_banded --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('banded', round(d['ms_per_step'],3), d['parity_check'], d['roofline'])"
