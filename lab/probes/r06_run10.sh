cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "column_group or every_pade_order or golden or hess or r_chain" 2>&1 | tail -6 > gpurun_out/r06_hess_tests_10.log
for r in 1 2; do
 for o in "hess_rpre=0" "hess_rpre=1" "hess_rpre=2"; do echo "== $o (round $r)"; HC_ONLY=1 HC_OPTS="$o" python lab/probes/hess_cols_ab.py 8,10 8,64 2>&1 | grep -v amdgpu.ids; done
done > gpurun_out/r06_hess_rpre_10.log 2>&1
cat gpurun_out/r06_hess_tests_10.log; cat gpurun_out/r06_hess_rpre_10.log
