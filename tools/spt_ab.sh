#!/bin/bash
# transpose after a change: every test that transposes (stand-alone, CSC operands, SpGEMM formats, drop-in), fuzz, bench
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_transpose.py tests/test_gpu_spmv.py tests/test_gpu_spmm.py tests/test_gpu_spgemm.py tests/test_gpu_dropin.py tests/test_gpu_cpp.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --workload transpose --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1
