#!/bin/bash
# A/B of variants of the run-time compiled kernel headers (no library rebuild: the headers are read at run time) on ONE box, fresh processes alternating.
# usage: ab_jit_headers.sh "<command>" variantA variantB ...   (gpurun_ab/<variant>/*.hpp; PCL_JIT_CACHE=0 keeps stale code objects out)
cmd="$1"; shift
rounds=${ROUNDS:-2}
mkdir -p gpurun_ab/_orig && cp piccolo.jl_amd/csrc/*.hpp gpurun_ab/_orig/
export PCL_JIT_CACHE=0
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp gpurun_ab/_orig/*.hpp piccolo.jl_amd/csrc/ && cp gpurun_ab/$v/*.hpp piccolo.jl_amd/csrc/
    echo "== round $r variant $v"
    eval "$cmd" 2>&1 | grep -v amdgpu.ids
  done
done
cp gpurun_ab/_orig/*.hpp piccolo.jl_amd/csrc/
