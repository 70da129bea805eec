set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "hess or Hess or column_group or smoke or golden" 2>&1 | tail -5 > gpurun_out/r06_hess_tests_2.log
for r in 1 2; do
  echo "== trackers off (round $r)"; PCL_JIT_OPTS="-mllvm -amdgpu-use-amdgpu-trackers=0" python lab/probes/hess_cols_ab.py 8,10 1,8 2>&1 | grep -v amdgpu.ids
  echo "== trackers on (round $r)"; python lab/probes/hess_cols_ab.py 8,10 1,8 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r06_hess_flag_ab.log 2>&1
timeout 600 lab/probes/alloc_probe 4 8 > gpurun_out/r06_alloc_probe_8.log 2>&1
tail -3 gpurun_out/r06_hess_tests_2.log; cat gpurun_out/r06_hess_flag_ab.log; tail -40 gpurun_out/r06_alloc_probe_8.log
