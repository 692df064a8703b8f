#!/bin/bash
# sweep the reduce variants on the headline workload.  SHARED=0: wave-owned bins (RW wave-bins per workgroup,
# C prefetched chunks, GR runs per LDS group); SHARED=8/16: one bin per workgroup with CAS accumulation.
for rep in 1 2; do
for cfg in ${SWEEP:-"0 4 2" "16 1 1" "16 1 2" "8 1 1" "8 1 2"}; do
  set -- $cfg
  echo -n "SHARED=$1 RW=$2 C=$3: "
  SPBLAS_GFX950_PB_SHARED=$1 SPBLAS_GFX950_PB_RWAVES=$2 SPBLAS_GFX950_PB_RCHUNKS=$3 python bench.py --no-cpu-baseline --steps 300 --warmup 50 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), d['roofline']['kernel_min_ms'], d['config']['plan']['rows_per_bin'], d['config']['inspect_ms_untimed'])"
done; done
