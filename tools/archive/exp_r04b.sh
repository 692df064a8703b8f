#!/bin/bash
# round 4: hot-column split, per-kernel times with and without it (same box)
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_spmv.py -x -q -k "hot_column" 2>&1 | tail -5
bash tools/kall.sh r04b_hot0 SPBLAS_GFX950_PB_HOT=0 -- --workload spmv_rmat1 2>&1 | cut -c1-400
bash tools/kall.sh r04b_hot1 -- --workload spmv_rmat1 2>&1 | cut -c1-400
