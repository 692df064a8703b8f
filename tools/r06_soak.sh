#!/bin/bash
# the whole -m gpu suite N times in a row on one box (flake hunt): gpurun_out/r06_soak.log holds one line per run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r06_soak.log
for i in $(seq 1 ${1:-6}); do
  timeout 900 python -m pytest tests/ -q -m gpu -p no:cacheprovider > gpurun_out/r06_soak_$i.txt 2>&1
  echo "run $i rc=$? $(tail -1 gpurun_out/r06_soak_$i.txt)" >> gpurun_out/r06_soak.log
  grep -E "^FAILED|^ERROR" gpurun_out/r06_soak_$i.txt >> gpurun_out/r06_soak.log
done
cat gpurun_out/r06_soak.log
