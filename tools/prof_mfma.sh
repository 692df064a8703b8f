#!/bin/bash
# tools/prof_mfma.sh <tag>: matrix-core utilisation of the SpMM panel kernel on the banded variant of cfg3 (north_star: "rocprof
# reports ... MFMA utilisation for SpMM"), at HEAD: one kernel-trace pass and one SQ counter pass of
# `bench.py --workload spmm_banded`; writes profiles/<tag>_mfma.json and stamps profiles/pmc_traffic.json
# ("spmm_banded_mfma": what bench.py prints as roofline.mfma_util, with the hash of csrc/spmm.hip).
#   mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / (SQ instances = 32)
#   (SQ_BUSY_CYCLES is summed over the 32 shader engines' SQs, the MFMA busy cycles over the 1 024 SIMDs)
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/mfma_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --workload spmm_banded --steps 5 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $BENCH > $OUT/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/pmc -o p -- $BENCH > $OUT/pmc.log 2>&1
cd $ROOT
python3 tools/mfma_summary.py "$TAG" "$OUT"
