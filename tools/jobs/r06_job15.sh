#!/bin/bash
# r06 GPU job 15: SQ counters of the specialised fused variant (product library), the bench line of the round's library
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
MDVT_LIB_VARIANT= bash tools/pmc_mix.sh r06_points_fused_spec --bits --counts --frames 128 > gpurun_out/pmcmix_r06_points_fused_spec.log 2>&1
grep -h "k_points_rows_fast" gpurun_out/pmcmix_r06_points_fused_spec/summary.txt | grep "INSTS_VALU \|INSTS_SALU\|ACTIVE_INST_SCA\|ACTIVE_INST_VALU\|duration\|SQ_WAVES\|BUSY_CYC" | cut -c1-140
mkdir -p gpurun_out/r06n
timeout 900 python bench.py > gpurun_out/r06n/bench.log 2> gpurun_out/r06n/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06n/bench.log").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"])
for k, v in d["extra"]["points_fused_maskbits_and_counts"]["by_frames"].items(): print("fused", k, round(v["launch_ms"], 4), round(v["fps"]), round(v["roofline_frac"], 3))
for k, v in d["extra"]["points_batch_sweep"].items(): print("plain", k, round(v["launch_ms"], 4), round(v["fps"]), round(v["roofline_frac"], 3))
PY
