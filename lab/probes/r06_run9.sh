cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python lab/probes/hess_rpre_diag.py 6 3 60 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_rpre_diag_9.log
python lab/probes/hess_rpre_diag.py 10 8 100 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_rpre_diag_9.log
python -m pytest tests -m gpu -x -q -k "column_group or every_pade_order or golden or hess or r_chain" 2>&1 | tail -15 > gpurun_out/r06_hess_tests_9.log
for r in 1 2; do
 for o in "hess_rpre=0" "hess_rpre=1"; do echo "== $o (round $r)"; HC_ONLY=1 HC_OPTS="$o" python lab/probes/hess_cols_ab.py 6,8,10 8,64 2>&1 | grep -v amdgpu.ids; done
done > gpurun_out/r06_hess_rpre_9.log 2>&1
cat gpurun_out/r06_rpre_diag_9.log; cat gpurun_out/r06_hess_tests_9.log; cat gpurun_out/r06_hess_rpre_9.log
