#!/bin/bash
# r06 GPU job 11: the dense per-wave counts + vectorised reduce, the HIP-graph capture of render + asynchronous completion; bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06k; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_render.py -x -q -m gpu -k "compact or fused or hip_graph or host_wait" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err; tail -1 $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06k/bench.log").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"], "this box:", d["roofline"]["traffic_source_is_this_box"])
for k, v in d["extra"]["points_fused_maskbits_and_counts"]["by_frames"].items(): print("fused", k, round(v["launch_ms"], 4), round(v["fps"]), round(v["roofline_frac"], 3))
for k, v in d["extra"]["points_batch_sweep"].items(): print("plain", k, round(v["launch_ms"], 4), round(v["fps"]), round(v["roofline_frac"], 3))
PY
