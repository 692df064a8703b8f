#!/bin/bash
# debug helper: run the multi-process fused-sharding worker directly (all ranks on cuda:0)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 FUSED_STRIPES=${FUSED_STRIPES:-1}
python -m torch.distributed.run --nnodes=1 --nproc-per-node=${1:-2} --master-addr 127.0.0.1 --master-port 29711 tests/mp_fused_worker.py ${2:-f32} 2>&1 | grep -v "^\[Gloo\]\|amdgpu.ids" | grep -v "elastic\|^  File\|^    " | head -${3:-40}
