#!/bin/bash
# Development tool: builds attention sources into tools/ab/libattn_v<name>.so (git-ignored, shipped by gpurun) for tools/ab/attn_ab.py.
# usage: build_attn_variants.sh name=source.hip[,-DFLAG...] ...
set -e
cd "$(dirname "$0")"
PIDS=""
for spec in "$@"; do
  name="${spec%%=*}"; rest="${spec#*=}"; src="${rest%%,*}"; defs=""
  [ "$rest" != "$src" ] && defs="${rest#*,}"
  ( hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -I../../oneprot_amd/csrc ${defs//,/ } -shared "$src" -o "libattn_v$name.so" ) &
  PIDS="$PIDS $!"
done
for p in $PIDS; do wait $p || { echo "compile failed"; exit 1; }; done
echo built "$@"
