cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "column_group or every_pade_order or golden or hess" 2>&1 | tail -4 > gpurun_out/r06_hess_tests_4.log
HC_ONLY=1 ROUNDS=2 bash lab/probes/ab_jit_headers.sh "python lab/probes/hess_cols_ab.py 8 8,64" hc_r5 hc_bfe hc_new abl_noC abl_nogather abl_nogdot abl_noprod abl_noreadback abl_noacc abl_noY abl_norchain > gpurun_out/r06_hess_variants_4.log 2>&1
cat gpurun_out/r06_hess_tests_4.log; cat gpurun_out/r06_hess_variants_4.log
