#!/bin/bash
# The fresh-context stress of profiles/r04_soak_summary.md as an A/B over the tuning library's workspace hooks:
#   tools/fresh_context_ab.sh <out dir> <iters per process> <config> [<config> ...]
# a config is a comma-separated list of VAR=value settings (no spaces), e.g.
#   MDVT_WS_POOL=off,MDVT_WS_FRESH=none,MDVT_WS_LAYOUT=joint      the r04 tree that failed (47b4117)
# Each config: 12 processes share the GPU, each renders the 100 x 31 mesh + convergence case of seed 504249 with a FRESH context per
# render (tests/dbg_stress_case.py, which diagnoses a differing render before its context goes).
out=$1; iters=$2; shift 2
mkdir -p "$out"
export MDVT_LIB_VARIANT=tuning
n=0
for cfg in "$@"; do
  n=$((n + 1))
  log="$out/cfg${n}.log"
  echo "== config $n: $cfg (${NPROC:-12} x $iters fresh contexts)" | tee "$log"
  t0=$(date +%s)
  for k in $(seq 1 ${NPROC:-12}); do
    ( IFS=,; for kv in $cfg; do export "$kv"; done
      MDVT_SWEEP_SEED=${SWEEP_SEED:-504249} MDVT_SWEEP_CASES=400 CASE=${SWEEP_CASE:-231} FRESH=${FRESH:-1} ITERS=$iters \
        timeout 600 python tests/dbg_stress_case.py > "$out/cfg${n}_p${k}.log" 2>&1; echo "exit $?" >> "$out/cfg${n}_p${k}.log" ) &
  done
  wait
  t1=$(date +%s)
  grep -h -v "amdgpu.ids" "$out"/cfg${n}_p*.log | grep -v "^exit 0" >> "$log"
  bad=$(grep -h "^case " "$out"/cfg${n}_p*.log | awk '{s += $NF} END {print s + 0}')
  done_=$(grep -h "^case " "$out"/cfg${n}_p*.log | wc -l)
  aborted=$(grep -L "^case " "$out"/cfg${n}_p*.log | wc -l)
  echo "== config $n result: $bad bad first renders in $done_ x $iters contexts, $aborted processes without a result, $((t1 - t0)) s" | tee -a "$log"
done
