#!/bin/bash
# r06 GPU job 9: whole GPU suite on the tree with the fused-compaction rework, the proven short division and the asynchronous completion; bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06i; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; tail -6 $OUT/pytest_all.log
timeout 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06i/bench.log").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"])
for k, v in d["extra"]["points_fused_maskbits_and_counts"]["by_frames"].items(): print("fused", k, round(v["launch_ms"], 4), round(v["fps"]), round(v["roofline_frac"], 3))
for k, v in d["extra"]["points_batch_sweep"].items(): print("plain", k, round(v["launch_ms"], 4), round(v["fps"]), round(v["roofline_frac"], 3))
print(json.dumps(d["extra"]["infill_mask_completion"])[:1200])
for k in ("mesh", "product_default", "mesh_convergence", "mesh_infill_mask", "c4_4k_pose_points", "c4_4k_pose_mesh"): print(k, round(d["extra"][k]["fps"]))
PY
