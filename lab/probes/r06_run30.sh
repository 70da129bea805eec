cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_round6_gpu.py -x -q -k "chain_wave_and_a_contribution" 2>&1 | tail -2 > gpurun_out/r06_pair_tests_30.log
HC_ONLY=1 ROUNDS=3 bash lab/probes/ab_jit_headers.sh "python lab/probes/hess_cols_ab.py 4,6,8,10 1" hp_nbuf2 hp_nbuf3 > gpurun_out/r06_pair_nbuf_30.log 2>&1
cat gpurun_out/r06_pair_tests_30.log gpurun_out/r06_pair_nbuf_30.log
