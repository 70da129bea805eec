cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "column_group or every_pade_order or golden or hess or r_chain" 2>&1 | tail -4 > gpurun_out/r06_hess_tests_13.log
HC_ONLY=1 ROUNDS=3 bash lab/probes/ab_jit_headers.sh "python lab/probes/hess_cols_ab.py 8,10 1,8,64" hc_ws1 hc_ws2 > gpurun_out/r06_hess_wstride_13.log 2>&1
cat gpurun_out/r06_hess_tests_13.log; cat gpurun_out/r06_hess_wstride_13.log
