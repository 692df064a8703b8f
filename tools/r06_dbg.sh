#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 FUSED_STRIPES=1 FUSED_CHUNKS=4 FUSED_TRIANGULAR=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29698 tests/mp_fused_worker.py f32 > gpurun_out/r06_dbg.out 2> gpurun_out/r06_dbg.err; echo rc=$?
grep -n "Error\|assert\|Traceback\|line " gpurun_out/r06_dbg.err | head -30
tail -5 gpurun_out/r06_dbg.out
