#!/bin/bash
# r05 GPU job 8: posed mesh runs that fit one launch set split over the two banks (C4: 8 frames of 4K) -- suite, A/B against
# libmdvt_hip_r4.so (same kernels, no split); the product default's kernels WITHOUT overlap (8 frames = one launch set, no banks),
# one row of cells per workgroup (libmdvt_hip_q.so) against four
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05h; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
ab() {
  tag=$1; shift
  for v in r4 "" r4 ""; do
    echo "== $tag on '${v:-new}': $(MDVT_LIB_VARIANT=$v python tools/kbench.py default --rounds 5 --calls 5 "$@" 2>&1 | tail -1)"
  done
}
ab c4_mesh --mesh --c4 --width 3840 --height 2160 --frames 8 | tee $OUT/ab.log
ab c4_mesh_edges --mesh --infill --c4 --width 3840 --height 2160 --frames 8 | tee -a $OUT/ab.log
for v in q ""; do
  echo "== product default, 8 frames per call on '${v:-new}': $(MDVT_LIB_VARIANT=$v python tools/kbench.py default --rounds 5 --calls 5 --mesh --infill --conv 2.5 --frames 8 2>&1 | tail -1)" | tee -a $OUT/ab.log
done
MDVT_LIB_VARIANT=q bash tools/profile_kbench.sh r05_pd8_q --mesh --infill --conv 2.5 --frames 8
MDVT_LIB_VARIANT= bash tools/profile_kbench.sh r05_pd8_r4 --mesh --infill --conv 2.5 --frames 8
