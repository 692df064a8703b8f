#!/bin/bash
# round 6, first GPU call: the advisor fixes (new tests, and the widest-window test against the library of round 5 -- it must
# fail there), the triangular-solve tests, the default bench line (compact headline)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_spmv.py -q -x -k "widest_window or second_build_fails" > gpurun_out/r06_t1.log 2>&1; echo "new tests rc=$?"
SPBLAS_GFX950_LIB=$PWD/tools/ab/libold_vfcap.so python -m pytest tests/test_gpu_spmv.py -q -k "widest_window" > gpurun_out/r06_t1_old.log 2>&1; echo "old lib rc=$? (nonzero expected)"
python -m pytest tests/test_gpu_sptrsv.py -q > gpurun_out/r06_t2.log 2>&1; echo "sptrsv rc=$?"
python bench.py > gpurun_out/bench_r06a.json 2> gpurun_out/bench_r06a.err; echo "bench rc=$?"
wc -c gpurun_out/bench_r06a.json
tail -3 gpurun_out/r06_t1.log; tail -3 gpurun_out/r06_t1_old.log; tail -3 gpurun_out/r06_t2.log
