#!/bin/bash
# r06 GPU job 6: the fused mask compaction (DPP nibble combine, barrier-free counts, optional byte mask): tests, then the bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_render.py -x -q -m gpu -k "compact or fused or raw_abi" > $OUT/pytest_fused.log 2>&1; tail -5 $OUT/pytest_fused.log
timeout 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06f/bench.log").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"], d["roofline"]["traffic_source"], d["roofline"]["traffic_source_is_this_box"])
print(json.dumps(d["extra"]["points_batch_sweep"], indent=0)[:900])
print(json.dumps(d["extra"]["points_fused_maskbits_and_counts"]["by_frames"], indent=0)[:2500])
print(json.dumps(d["cpu_baseline"], indent=0)[:3000])
PY
