cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_round6_gpu.py -x -q -k "chain_wave_and_a_contribution" 2>&1 | tail -12 > gpurun_out/r06_pair_tests_24.log
for r in 1 2 3; do for o in "hess_pair=0" "hess_pair=1"; do echo "== $o (round $r)"; HC_ONLY=1 HC_OPTS="$o" python lab/probes/hess_cols_ab.py 2,4,6,8,10 1 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/r06_hess_pair_24.log 2>&1
cat gpurun_out/r06_pair_tests_24.log gpurun_out/r06_hess_pair_24.log
