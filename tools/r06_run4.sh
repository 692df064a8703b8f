#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_add.py tests/test_gpu_spgemm.py -q -x 2>&1 | tail -3
python -m pytest tests/test_gpu_spmv.py -q -x -k "value_free" 2>&1 | tail -3
python -m pytest tests/test_gpu_fused_sharding.py -q -x -k "test_fused_sharded_spmv_multiprocess" 2>&1 | tail -8
