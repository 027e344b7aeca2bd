#!/bin/bash
# r06 GPU job 24: last check of the round's tree -- whole GPU suite, smoke, the default bench run (wall time of the whole command)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06t; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; tail -3 $OUT/pytest_all.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
T0=$(date +%s); timeout 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "bench.py wall $(( $(date +%s) - T0 )) s"; tail -1 $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06t/bench.log").read().strip().splitlines()[-1])
print("value", round(d["value"]), "frac", round(d["roofline"]["frac"], 3), "keys", sorted(d.keys()))
print(json.dumps(d["extra"]["product_default_with_finished_infill_mask"])[:300])
PY
