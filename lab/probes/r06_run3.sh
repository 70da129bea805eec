cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "column_group or every_pade_order or golden or hess" 2>&1 | tail -4 > gpurun_out/r06_hess_tests_3.log
HC_ONLY=1 ROUNDS=2 bash lab/probes/ab_jit_headers.sh "python lab/probes/hess_cols_ab.py 8,10 1,8,64" hc_r5 hc_new hc_gch9 hc_lessfence hc_y27 > gpurun_out/r06_hess_variants_3.log 2>&1
echo "== hc_r5 with the trackers off" >> gpurun_out/r06_hess_variants_3.log
cp piccolo.jl_amd/csrc/pcl_kernel_hess_cols.hpp /tmp/keep.hpp; cp gpurun_ab/hc_r5/pcl_kernel_hess_cols.hpp piccolo.jl_amd/csrc/
HC_ONLY=1 PCL_JIT_CACHE=0 PCL_JIT_OPTS="-mllvm -amdgpu-use-amdgpu-trackers=0" python lab/probes/hess_cols_ab.py 8,10 1,8,64 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_hess_variants_3.log
cp /tmp/keep.hpp piccolo.jl_amd/csrc/pcl_kernel_hess_cols.hpp
timeout 900 lab/probes/alloc_probe 4 8 > gpurun_out/r06_alloc_probe_8b.log 2>&1
cat gpurun_out/r06_hess_tests_3.log; cat gpurun_out/r06_hess_variants_3.log; grep -A12 "Part C" gpurun_out/r06_alloc_probe_8b.log
