#!/bin/bash
# round 4: chunked dependent chain -- tests (both publication modes), then the one-GPU cost of the N = 8 shard step:
# barrier step / chunked chain (combine publishes) / chunked chain (stripes), one rank emulating rank 0 of 8
cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python -m pytest tests/test_gpu_fused_sharding.py -x -q -k "chunk_flags" 2>&1 | tail -3
SPBLAS_GFX950_CHUNK_STRIPES=1 timeout 600 python -m pytest tests/test_gpu_fused_sharding.py -x -q -k "chunk_flags" 2>&1 | tail -3
timeout 300 python tools/chunk_overhead.py
