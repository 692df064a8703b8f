#!/bin/bash
# SpGEMM family after a change: tests, fuzz, bench lines (spgemm, add), bin-3 timing
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_add.py tests/test_gpu_spgemm.py tests/test_gpu_configs.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -3
timeout 600 python tools/fuzz_spgemm.py 2>&1 | tail -2
timeout 600 python bench.py --workload spgemm --no-cpu-baseline --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('fill', d['ms_per_step'], 'compute', c['multiply_compute_ms_untimed'], 'first fill', c['first_fill_ms_untimed'], 'recording fill', c['second_fill_ms_untimed_records_ranks'])"
timeout 600 python tools/spg_bin3.py 2>&1 | tail -2
timeout 600 python tools/spg_bin3.py 300000 32 2>&1 | tail -2
