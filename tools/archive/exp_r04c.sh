#!/bin/bash
# round 4: hot-column kernel, 4 vs 8 entries per lane (same box); per-kernel times
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_spmv.py -x -q -k "hot_column" 2>&1 | tail -3
SPBLAS_GFX950_LIB=$PWD/tools/ab/libepl8.so python -m pytest tests/test_gpu_spmv.py -x -q -k "hot_column" 2>&1 | tail -3
bash tools/kall.sh r04c_epl4 -- --workload spmv_rmat1 2>&1 | grep "pb_expand\|pb_reduce\|pb_hot"
bash tools/kall.sh r04c_epl8 SPBLAS_GFX950_LIB=$PWD/tools/ab/libepl8.so -- --workload spmv_rmat1 2>&1 | grep "pb_expand\|pb_reduce\|pb_hot"
