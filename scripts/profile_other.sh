#!/bin/bash
# rocprofv3 kernel trace of the kernels bench.py does not launch (scripts/other_kernels_run.py).
# Writes gpurun_out/profiles_<tag>/<tag>_other_kernel_stats.csv (copy into profiles/).  Usage (GPU box): scripts/profile_other.sh r02
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_other_$TAG
mkdir -p $OUT $ROOT/gpurun_out/profiles_$TAG
export TMPDIR=/tmp
cd /tmp
python $ROOT/scripts/other_kernels_run.py 2>&1 | tail -5
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o other -- python $ROOT/scripts/other_kernels_run.py > /dev/null 2>&1
cd $ROOT
f=$(find $OUT -name "other_kernel_stats.csv" | head -1)
grep -v "at::native\|rocclr" "$f" > gpurun_out/profiles_$TAG/${TAG}_other_kernel_stats.csv
cat gpurun_out/profiles_$TAG/${TAG}_other_kernel_stats.csv | cut -c1-200
