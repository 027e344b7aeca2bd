#!/bin/bash
# r05 GPU job 4: one queue append per workgroup -- suite + A/B against the previous commit's library (libmdvt_hip_prev.so)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05d; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
ab() {
  tag=$1; shift
  for v in prev "" prev ""; do
    echo "== $tag on '${v:-new}': $(MDVT_LIB_VARIANT=$v python tools/kbench.py default --rounds 5 --calls 5 "$@" 2>&1 | tail -1)"
  done
}
ab product_default --mesh --infill --conv 2.5 --frames 32 | tee $OUT/ab.log
ab mesh_conv --mesh --conv 2.5 --frames 32 | tee -a $OUT/ab.log
ab mesh_pose --mesh --pose --frames 32 | tee -a $OUT/ab.log
ab mesh_pose_edges --mesh --pose --infill --frames 32 | tee -a $OUT/ab.log
ab c4_mesh --mesh --c4 --width 3840 --height 2160 --frames 8 | tee -a $OUT/ab.log
export MDVT_LIB_VARIANT=
bash tools/profile_kbench.sh r05c_c4_mesh --mesh --c4 --width 3840 --height 2160 --frames 8
