#!/bin/bash
# same-box A/B: previous library (tools/tmp_old) vs the current one on the cfg5 fill
mkdir -p gpurun_out
L=spblas-reference_amd/lib/libspblas_gfx950.so
cp $L /tmp/new.so
one() { timeout 600 python bench.py --workload $1 --no-cpu-baseline --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_avg_ms'])"; }
for rep in 1 2; do
  cp tools/tmp_old/libspblas_gfx950.so $L; echo -n "old spgemm: "; one spgemm
  cp /tmp/new.so $L; echo -n "new spgemm: "; one spgemm
done
echo -n "add: "; one add
timeout 900 python -m pytest tests/test_gpu_add.py tests/test_gpu_spgemm.py tests/test_gpu_configs.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -2
timeout 600 python tools/fuzz_spgemm.py 2>&1 | tail -2
