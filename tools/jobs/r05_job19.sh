#!/bin/bash
# r05 GPU job 19: where the queue walk's time goes -- per-kernel durations of single launch sets (no overlap) with and without the
# fragments' posts (libmdvt_hip_nopost.so: a timing ablation build, outputs wrong by design)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05s; mkdir -p $OUT
for v in "" nopost; do
  echo "== C4 mesh, 4 frames per call, '${v:-product}'"; MDVT_LIB_VARIANT=$v bash tools/kstat.sh --mesh --c4 --width 3840 --height 2160 --frames 4
  echo "== mesh under a pose, 8 frames per call, '${v:-product}'"; MDVT_LIB_VARIANT=$v bash tools/kstat.sh --mesh --pose --frames 8
  echo "== mesh + convergence, 8 frames per call, '${v:-product}'"; MDVT_LIB_VARIANT=$v bash tools/kstat.sh --mesh --conv 2.5 --frames 8
done 2>&1 | tee $OUT/kstat.log
