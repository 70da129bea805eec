#!/bin/bash
# A/B of the ensemble step's payload: fused (pcl_eval_jac_merit_dev) vs separate kernels, kernel trace of each
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ens_ab; mkdir -p $O
for v in fused sep; do
  fl=""; [ $v = sep ] && fl="--separate-payload"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o t -- python $R/bench.py --workload ensemble $fl --no-extras --no-cpu-baseline --steps 100 > $O/$v.json 2> $O/$v.err
  f=$(find $O/$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; cut -c1-200 $O/$v.json | head -1; head -12 "$f" | cut -d, -f1-8
done
