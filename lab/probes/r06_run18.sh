cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
HC_ONLY=1 ROUNDS=2 bash lab/probes/ab_jit_headers.sh "python lab/probes/hess_cols_ab.py 8,10 8,64" hc_base hc_prio_prod hc_prio_mem > gpurun_out/r06_hess_prio_18.log 2>&1
cat gpurun_out/r06_hess_prio_18.log
