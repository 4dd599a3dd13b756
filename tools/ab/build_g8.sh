#!/bin/bash
# Development tool: builds variants of the 8-phase NT GEMM (oneprot_amd/csrc/gemm_nt8.hip with extra -D flags) into tools/ab/libg8_<name>.so
# (git-ignored, shipped by gpurun): gemm_nt8.hip + the product's gemm_nt.o, i.e. a library that exports oneprot_gemm_bf16_nt and the tuning hooks.
# usage: build_g8.sh name[,-DFLAG...] ...
set -e
cd "$(dirname "$0")"
CS=../../oneprot_amd/csrc
for spec in "$@"; do
  name="${spec%%,*}"; defs=""
  [ "$spec" != "$name" ] && defs="${spec#*,}"
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -I$CS ${defs//,/ } -c $CS/gemm_nt8.hip -o /tmp/g8_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o libg8_$name.so /tmp/g8_$name.o $CS/gemm_nt.o
  echo built libg8_$name.so
done
