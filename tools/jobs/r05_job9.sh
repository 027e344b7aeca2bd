#!/bin/bash
# r05 GPU job 9: removed-triangle flags fetched before the vertex programme; rows of cells per workgroup with edge removal: 4 ("new") or 1
# (libmdvt_hip_e1.so) against r4 (flags fetched late) and q (one row per workgroup everywhere)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05i; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
ab() {
  tag=$1; shift
  for v in q r4 "" e1 q r4 "" e1; do
    echo "== $tag on '${v:-new}': $(MDVT_LIB_VARIANT=$v python tools/kbench.py default --rounds 5 --calls 5 "$@" 2>&1 | tail -1)"
  done
}
ab product_default --mesh --infill --conv 2.5 --frames 32 | tee $OUT/ab.log
ab product_default_8 --mesh --infill --conv 2.5 --frames 8 | tee -a $OUT/ab.log
ab mesh_pose_edges --mesh --pose --infill --frames 32 | tee -a $OUT/ab.log
ab c4_mesh_edges --mesh --infill --c4 --width 3840 --height 2160 --frames 8 | tee -a $OUT/ab.log
