cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_round6_gpu.py -x -q 2>&1 | tail -3 > gpurun_out/r06_tests_27.log
python -m pytest tests -m gpu -x -q -k "hess or Hess or column_group or golden or smoke" 2>&1 | grep -E "passed|failed" >> gpurun_out/r06_tests_27.log
for r in 1 2; do for o in "hess_rpre=0 hess_pair=0" "hess_rpre=0 hess_pair=1" "hess_rpre=1 hess_pair=0"; do echo "== $o (round $r)"; HC_ONLY=1 HC_OPTS="$o" python lab/probes/hess_cols_ab.py 8,10 2,8 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/r06_hess_pair_batch_27.log 2>&1
cat gpurun_out/r06_tests_27.log gpurun_out/r06_hess_pair_batch_27.log
