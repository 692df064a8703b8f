#!/bin/bash
# banded SpMM: persistent workgroups with double-buffered windows, the next window's LDS-DMA unseen by the compiler's wait counts
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_spmm.py -q -x 2>&1 | tail -2
one() { python bench.py --workload spmm_banded --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), 'ms', d['parity_check'])"; }
one "dbuf (unseen DMA), 8 waves, 1 wg/cu"
SPBLAS_GFX950_SPMM_BAND_WAVES=16 one "dbuf (unseen DMA), 16 waves"
SPBLAS_GFX950_SPMM_BAND_DBUF=0 one "no dbuf, one wg per block"
SPBLAS_GFX950_SPMM_BAND_WAVES=16 SPBLAS_GFX950_SPMM_DBG=1 one "dbuf 16 waves, no contraction"
