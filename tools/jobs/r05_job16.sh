#!/bin/bash
# r05 GPU job 16: k_mesh_raster_conv with its block column / frame opaque per row of cells (fewer hoisted per-lane addresses): "new"; the same
# with 4 / 2 rows of cells per workgroup with edge removal (o4 / o2) and with conv<0> held to 80 VGPRs (w6); against the commit before (d1)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r05p; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
ab() {
  tag=$1; shift
  for v in d1 "" o4 o2 w6 d1 "" o4 o2 w6; do
    echo "== $tag on '${v:-new}': $(MDVT_LIB_VARIANT=$v python tools/kbench.py default --rounds 5 --calls 5 "$@" 2>&1 | tail -1)"
  done
}
ab product_default --mesh --infill --conv 2.5 --frames 32 | tee $OUT/ab.log
ab mesh_conv --mesh --conv 2.5 --frames 32 | tee -a $OUT/ab.log
