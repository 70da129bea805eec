cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
HC_ONLY=1 ROUNDS=2 bash lab/probes/ab_jit_headers.sh "python lab/probes/hess_cols_ab.py 8,10 1" hp_base hp_norchain hp_nogather hp_nogdot > gpurun_out/r06_pair_ablations_29.log 2>&1
cat gpurun_out/r06_pair_ablations_29.log
