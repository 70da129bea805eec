cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for B in 8 64; do
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rpre_$B -o rpre -- env HC_ONLY=1 HC_OPTS="hess_rpre=1" python lab/probes/hess_cols_ab.py 8,10 $B > gpurun_out/prof_rpre_$B.log 2>&1
done
find gpurun_out/prof_rpre_8 gpurun_out/prof_rpre_64 -name "*kernel_stats.csv" | while read f; do echo "== $f"; cat "$f" | cut -d, -f1-8 | head -8; done
