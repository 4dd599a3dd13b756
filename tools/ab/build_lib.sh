#!/bin/bash
# Development tool: a VARIANT build of the whole kernel library for tools/ab/libs_ab.py.  The listed translation units are compiled with the
# extra flags (or taken from another path: unit=path/to/copy.hip), every other object is the product's (oneprot_amd/csrc/*.o, build.sh first).
# usage: build_lib.sh <name> <unit[=path]>[,<unit[=path]>...] [-DFLAG ...]      e.g.  build_lib.sh gelu5 gemm_nt8,gemm_nt -DGELU_POLY=5
# -> tools/ab/lib_<name>.so (git-ignored, shipped to the GPU box by gpurun)
set -e
cd "$(dirname "$0")"
CS=../../oneprot_amd/csrc
name="$1"; units="$2"; shift 2
ALL="rowops gemm_nt gemm_nt8 gemm_nt_ln gemm_tn sgemm attention featops"
objs=""; pids=""
for u in $ALL; do
  src=""
  for spec in ${units//,/ }; do
    un="${spec%%=*}"
    if [ "$un" = "$u" ]; then src="$CS/$u.hip"; [ "$spec" != "$un" ] && src="${spec#*=}"; fi
  done
  if [ -n "$src" ]; then
    ( hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -I$CS -I$CS/../../include "$@" -Rpass-analysis=kernel-resource-usage -c "$src" -o /tmp/lib_${name}_$u.o 2> /tmp/lib_${name}_$u.remarks \
        || { grep -v "remark:" /tmp/lib_${name}_$u.remarks >&2; exit 1; } ) &
    pids="$pids $!"
    objs="$objs /tmp/lib_${name}_$u.o"
  else
    objs="$objs $CS/$u.o"
  fi
done
for p in $pids; do wait $p || { echo "compile failed"; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC -o lib_$name.so $objs
for u in $ALL; do [ -f /tmp/lib_${name}_$u.remarks ] && python3 $CS/check_resources.py /tmp/lib_${name}_$u.remarks; done
echo built tools/ab/lib_$name.so
