cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PCL_JIT_CACHE=0 PCL_HC_NOWAIT=1 python -m pytest tests -m gpu -x -q -k "column_group or r_chain" 2>&1 | tail -4 > gpurun_out/r06_nowait_tests_17.log
for r in 1 2 3; do for nw in 0 1; do echo "== nowait $nw (round $r)"; PCL_JIT_CACHE=0 HC_ONLY=1 PCL_HC_NOWAIT=$nw python lab/probes/hess_cols_ab.py 8,10 1,8,64 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/r06_hess_nowait_17.log 2>&1
cat gpurun_out/r06_nowait_tests_17.log gpurun_out/r06_hess_nowait_17.log
