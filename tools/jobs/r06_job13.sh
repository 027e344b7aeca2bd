#!/bin/bash
# r06 GPU job 13: SQ counters of the headline kernel, plain and fused (product library): VALU instructions per lane, VALU busy
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
MDVT_LIB_VARIANT= bash tools/pmc_mix.sh r06_points --frames 128 > gpurun_out/pmcmix_r06_points.log 2>&1
MDVT_LIB_VARIANT= bash tools/pmc_mix.sh r06_points_fused --bits --counts --frames 128 > gpurun_out/pmcmix_r06_points_fused.log 2>&1
grep -h "k_points_rows_fast" gpurun_out/pmcmix_r06_points/summary.txt | cut -c1-140
echo ----
grep -h "k_points_rows_fast" gpurun_out/pmcmix_r06_points_fused/summary.txt | cut -c1-140
