#!/bin/bash
# round 6, second GPU call: the two new 8(f) bench records
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --workload csc_spmv --steps 10 --warmup 3 > gpurun_out/r06_csc.json 2> gpurun_out/r06_csc.err; echo "csc rc=$?"
python bench.py --workload spgemm4 --steps 10 --warmup 3 > gpurun_out/r06_spgemm4.json 2> gpurun_out/r06_spgemm4.err; echo "spgemm4 rc=$?"
cp bench_secondary_csc_spmv.json bench_secondary_spgemm4.json gpurun_out/ 2>/dev/null
tail -3 gpurun_out/r06_csc.err gpurun_out/r06_spgemm4.err
cat gpurun_out/r06_csc.json gpurun_out/r06_spgemm4.json
