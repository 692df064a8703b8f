#!/bin/bash
# round 4: hot-column split -- the new test, then cfg4 with and without the split in one call (same box)
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_spmv.py -x -q -k "hot_column" 2>&1 | tail -15
for HOT in 0 -1; do
  SPBLAS_GFX950_PB_HOT=$HOT python bench.py --workload spmv_rmat1 --steps 20 --warmup 5 > gpurun_out/r04_rmat_hot$HOT.json 2> gpurun_out/r04_rmat_hot$HOT.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_rmat_hot$HOT.json").read().strip().splitlines()[-1])
c=d["config"]
print("HOT=$HOT", round(d["ms_per_step"],4), "ms", round(d["roofline"]["frac"],4), d.get("parity_check"), "inspect", round(c["inspect_ms_untimed"],1), "ms", c["plan"].get("sliced",{}).get("hot_split"), c["plan"]["device_bytes"])
PY
done
