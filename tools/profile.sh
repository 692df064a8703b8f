#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):  tools/profile.sh <tag> [bench args...]
#   pass 0: rocprofv3 --kernel-trace --stats            -> per-kernel durations
#   pass 1..4: rocprofv3 --kernel-trace --pmc <set>     -> HBM/L2 counters (separate passes:
#              FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2; MI355X_MICROARCH.md)
#   calibration: tools/calib_copy.py under the same counter sets (known byte counts)
# Outputs land in gpurun_out/prof_<tag>/ ; tools/pmc_summary.py condenses them into
# profiles/<tag>_*.{md,json} which are committed.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary $*"
CALIB="python3 $ROOT/tools/calib_copy.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum" \
           "TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_READ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc$i -o pmc -- $BENCH > $OUT/pmc$i.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/cal$i -o cal -- $CALIB > $OUT/cal$i.log 2>&1
done
cd $ROOT
python3 tools/pmc_summary.py $OUT $TAG
