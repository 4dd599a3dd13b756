#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the FFN-1 launch with the default and the non-temporal output-store policy (GPU box).  usage: nt_fetch.sh [ffn1|ffn1fwd]
W=${1:-ffn1}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/nt_fetch; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pol in ${NT_POLS:-0 512}; do
  export GEMM_TUNE=$pol
  [ $pol = 0 ] && unset GEMM_TUNE
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${c}_$pol -o p -- python3 $ROOT/tools/gemm_only.py -1 $W 3 > /dev/null 2> $OUT/${c}_$pol.log
    python3 - <<PY
import csv, collections
per = collections.defaultdict(float)
for r in csv.DictReader(open("$OUT/${c}_$pol/p_counter_collection.csv")):
    if r["Counter_Name"] == "$c" and "k_gemm" in r["Kernel_Name"]: per[r["Dispatch_Id"]] += float(r["Counter_Value"])
v = sorted(per.values()); print("policy $pol $c KB median", v[len(v) // 2], "launches", len(v))
PY
  done
done
