#!/bin/bash
# tools/extra_stats.sh <tag>: kernel-trace stats of the secondary workloads -> gpurun_out/extra_<tag>.md
TAG=$1
OUT=gpurun_out/extra_$TAG.md
echo '```' > $OUT
export SPBLAS_GFX950_TRSV_COOP=1  # under rocprofv3 the solve would otherwise fall back to one launch per level (sptrsv.hip)
for w in spmm spmm_banded spgemm add transpose sptrsv spmv_rmat; do
  echo "## $w" >> $OUT
  TOPN=6 bash tools/quick_stats.sh ${TAG}_$w --workload $w 2>&1 | grep -v '"value"' >> $OUT
done
echo '```' >> $OUT
