#!/bin/bash
# Development tool (GPU): kernel-library builds ranked inside the cfg-2 training step, one process per run, alternating on one box.
# usage: step_libs_ab.sh <rounds> name=path.so [name=path.so ...]     (name=product: oneprot_amd/liboneprot_hip.so)
set -e
cd "$(dirname "$0")/../.."
rounds=$1; shift
declare -A acc
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    name="${spec%%=*}"; path="${spec#*=}"
    if [ "$path" = product ]; then out=$(python3 tools/ab/step_once.py 12 2>/dev/null | tail -1); else out=$(G8_LIB="$path" python3 tools/ab/step_once.py 12 2>/dev/null | tail -1); fi
    echo "round $r  $name: $out"
    acc[$name]="${acc[$name]} ${out%% *}"
  done
done
for spec in "$@"; do
  name="${spec%%=*}"
  python3 -c "import sys,statistics; v=[float(x) for x in sys.argv[2:]]; print(f'{sys.argv[1]:24s} median {statistics.median(v):8.3f} ms/step   runs: ' + ' '.join(f'{x:.2f}' for x in v))" "$name" ${acc[$name]}
done
