#!/bin/bash
# r03n: non-temporal product stores for the last N % of each expand workgroup's blocks (A/B builds, one box, three
# rounds because single runs on one box fall into two modes ~3 % apart): does the reduce's after-expand penalty come
# from the end of the expand?
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do
bash tools/kstats.sh base$rep
for v in nt10 nt25 nt50 nt100; do
  bash tools/kstats.sh $v$rep SPBLAS_GFX950_LIB=$PWD/tools/ab/lib$v.so
done
done
bash tools/kstats.sh base4
