#!/bin/bash
# A/B of kernel-header variants on ONE box: gpurun_ab/<variant>/*.hpp are copied over piccolo.jl_amd/csrc in turn, the library rebuilt and
# the command run in a fresh process, alternating for ROUNDS rounds.  usage: ab_headers.sh "<command>" variantA variantB [...]
cmd="$1"; shift
rounds=${ROUNDS:-3}
mkdir -p gpurun_ab/_orig && cp piccolo.jl_amd/csrc/*.hpp gpurun_ab/_orig/
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp gpurun_ab/_orig/*.hpp piccolo.jl_amd/csrc/ && cp gpurun_ab/$v/*.hpp piccolo.jl_amd/csrc/
    python -c "import piccolo_jl_amd as pa; pa.build_library(force=True)" > /dev/null 2>&1 || echo "build failed for $v"
    echo "== round $r variant $v"
    eval "$cmd" 2>&1 | grep -v amdgpu.ids
  done
done
cp gpurun_ab/_orig/*.hpp piccolo.jl_amd/csrc/
