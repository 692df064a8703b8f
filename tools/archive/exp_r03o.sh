#!/bin/bash
# r03o: plain vs non-temporal product stores (all of them), interleaved four times on one box; bench.py's own
# ms_per_step as well as the per-kernel averages.  Run on several boxes: the sign of the difference decides.
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3 4; do
bash tools/kstats.sh base$rep
bash tools/kstats.sh nt100$rep SPBLAS_GFX950_LIB=$PWD/tools/ab/libnt100.so
done
for rep in 1 2 3; do
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('bench base  ', round(d['ms_per_step']*1e3,1), 'us')"
SPBLAS_GFX950_LIB=$PWD/tools/ab/libnt100.so python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('bench nt100 ', round(d['ms_per_step']*1e3,1), 'us')"
done
