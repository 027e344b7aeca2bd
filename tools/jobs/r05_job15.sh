#!/bin/bash
# r05 GPU job 15: edge-key list appends aggregated per wave and source row ("new": + list splat for convergence-only launches, convergence-only
# vertex programme) against the commit before (libmdvt_hip_d0.so)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05o; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
ab() {
  tag=$1; shift
  for v in d0 "" d0 ""; do
    echo "== $tag on '${v:-new}': $(MDVT_LIB_VARIANT=$v python tools/kbench.py default --rounds 5 --calls 5 "$@" 2>&1 | tail -1)"
  done
}
ab product_default --mesh --infill --conv 2.5 --frames 32 | tee $OUT/ab.log
ab product_default_1 --mesh --infill --conv 2.5 --frames 1 | tee -a $OUT/ab.log
ab mesh_pose_edges --mesh --pose --infill --frames 32 | tee -a $OUT/ab.log
ab c4_mesh_edges --mesh --infill --c4 --width 3840 --height 2160 --frames 8 | tee -a $OUT/ab.log
ab points_conv_edges --infill --conv 2.5 --frames 32 | tee -a $OUT/ab.log
ab points_pose_edges --infill --pose --frames 32 | tee -a $OUT/ab.log
MDVT_LIB_VARIANT= bash tools/profile_kbench.sh r05_pd8_agg --mesh --infill --conv 2.5 --frames 8
