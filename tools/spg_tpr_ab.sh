#!/bin/bash
# team width of the bin-2 hash kernels (symbolic + first fill) on cfg5
mkdir -p gpurun_out
for t in 64 32 16; do
  echo -n "hash2 TPR $t: "
  SPBLAS_GFX950_SPG_HASH2_TPR=$t timeout 600 python bench.py --workload spgemm --no-cpu-baseline --steps 5 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(d['ms_per_step'], c['multiply_compute_ms_untimed'], c['first_fill_ms_untimed'], c['second_fill_ms_untimed_records_ranks'])"
done
for t in 32 16; do SPBLAS_GFX950_SPG_HASH2_TPR=$t timeout 900 python -m pytest tests/test_gpu_add.py tests/test_gpu_spgemm.py -x -q 2>&1 | tail -2; done
