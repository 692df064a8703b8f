#!/bin/bash
# cfg5 SpGEMM after a change of the reuse path: bench line, SpGEMM/add/drop-in tests, fuzz, per-kernel stats, HBM traffic.
mkdir -p gpurun_out
timeout 600 python bench.py --workload spgemm --no-cpu-baseline --steps 20 --warmup 5 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_spgemm.py tests/test_gpu_configs.py tests/test_gpu_add.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -3
timeout 600 python tools/fuzz_spgemm.py 2>&1 | tail -2
TOPN=12 tools/quick_stats.sh spgq --workload spgemm
tools/pmc_one.sh spgf --workload spgemm
