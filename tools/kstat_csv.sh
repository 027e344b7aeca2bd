#!/bin/bash
# rocprofv3 --kernel-trace --stats of any command; the kernel_stats.csv goes to gpurun_out/<tag>_kernel_stats.csv.
#   bash tools/kstat_csv.sh <tag> <command ...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
D=$(mktemp -d /tmp/kstat.XXXX)
( cd $ROOT && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- "$@" ) > $D/cmd.out 2>&1
grep -v "^[WE]2026\|amdgpu.ids" $D/cmd.out | tail -3
mkdir -p $ROOT/gpurun_out
cp "$(find $D -name '*kernel_stats.csv' | head -1)" $ROOT/gpurun_out/${TAG}_kernel_stats.csv
head -14 $ROOT/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150
