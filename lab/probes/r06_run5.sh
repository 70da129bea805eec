cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "column_group or every_pade_order or golden or hess or r_chain" 2>&1 | tail -6 > gpurun_out/r06_hess_tests_5.log
for r in 1 2; do
 for o in "hess_rpre=0" "hess_rpre=1"; do echo "== $o (round $r)"; HC_ONLY=1 HC_OPTS="$o" python lab/probes/hess_cols_ab.py 6,8,10 1,8,64 2>&1 | grep -v amdgpu.ids; done
done > gpurun_out/r06_hess_rpre_5.log 2>&1
cat gpurun_out/r06_hess_tests_5.log; cat gpurun_out/r06_hess_rpre_5.log
