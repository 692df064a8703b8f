#!/bin/bash
# sweep the reduce shape on the headline workload: wave-bins per workgroup, prefetched chunks, runs per LDS group
for rep in 1 2; do
for cfg in ${SWEEP:-"4 2 1" "4 2 2" "4 2 4" "8 1 2" "8 1 4"}; do
  set -- $cfg
  echo -n "RW=$1 C=$2 GR=$3: "
  SPBLAS_GFX950_PB_RWAVES=$1 SPBLAS_GFX950_PB_RCHUNKS=$2 SPBLAS_GFX950_PB_RGROUP=$3 python bench.py --no-cpu-baseline --steps 300 --warmup 50 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_avg_ms'],4))"
done; done
