#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --workload spmv_rmat_shards --steps 10 --warmup 3 > gpurun_out/r06_shards.json 2> gpurun_out/r06_shards.err; echo "shards rc=$?"
python - <<'PY'
import json
d=json.load(open("bench_secondary_spmv_rmat_shards.json"))
c=d["config"]
for k in ("shard_ms","shard_rows","shard_nnz","shard_plan_alg","shard_inspect_ms","max_ms","mean_ms","max_over_mean","sum_ms"): print(k, c[k])
print(d["parity_check"])
PY
python bench.py --workload spmv_rmat1 --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4 1gpu', round(d['ms_per_step'],3), 'ms', d['parity_check'], d['config']['inspect_ms'])"
python -m pytest tests/test_gpu_spmv.py -q -x -k "skewed or auto" 2>&1 | tail -2
python -m pytest tests/test_gpu_fused_sharding.py tests/test_gpu_configs.py -q -x 2>&1 | tail -2
