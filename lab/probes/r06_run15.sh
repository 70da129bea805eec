cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
HC_ONLY=1 ROUNDS=2 bash lab/probes/ab_jit_headers.sh "python lab/probes/hess_cols_ab.py 8,10 8,64" hc_base hc_gthoist hc_y14b hc_y27b > gpurun_out/r06_hess_batches_15.log 2>&1
for r in 1 2; do for gd in 7 14 32; do echo "== gdot cols $gd (round $r)"; PCL_JIT_CACHE=0 HC_ONLY=1 PCL_HC_GDOT_COLS=$gd python lab/probes/hess_cols_ab.py 8,10 8,64 2>&1 | grep -v amdgpu.ids; done; done >> gpurun_out/r06_hess_batches_15.log 2>&1
cat gpurun_out/r06_hess_batches_15.log
