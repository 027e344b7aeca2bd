#!/bin/bash
# r05 GPU job 26: SQ counters of the product default's kernels (single launch sets of 8 frames, product library)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
export MDVT_LIB_VARIANT=
bash tools/pmc_kbench.sh r05_pd8 --mesh --infill --conv 2.5 --frames 8 > gpurun_out/pmc_pd8.log 2>&1
grep -E "k_mesh_raster_conv|k_edge_points_splat_list|k_resolve_general" gpurun_out/pmck_r05_pd8/summary.txt | grep -E "SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_VALU|SQ_WAVES |GRBM_GUI_ACTIVE|TCC_ATOMIC"
