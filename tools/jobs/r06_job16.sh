#!/bin/bash
# r06 GPU job 16: timing experiment -- the convergence-only vertex programme with ONE reciprocal per vertex and eye (other bits: libmdvt_hip_rcpv.so,
# built by hand with -DMDVT_EXP_RCP_VERTEX) against the product library, product default and mesh + convergence, same box
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
OUT=gpurun_out/r06o; mkdir -p $OUT
{
for rep in 1 2 3; do
  for v in "" rcpv; do
    echo "== product default, variant '$v'"; MDVT_LIB_VARIANT=$v timeout 120 python tools/kbench.py default --mesh --infill --conv 2.5 --frames 32 --rounds 7 --calls 5 2>&1 | tail -1
    echo "== mesh + convergence, variant '$v'"; MDVT_LIB_VARIANT=$v timeout 120 python tools/kbench.py default --mesh --conv 2.5 --frames 32 --rounds 7 --calls 5 2>&1 | tail -1
  done
done
} 2>&1 | grep -v "amdgpu.ids\|library:" > $OUT/ab.log
cat $OUT/ab.log
