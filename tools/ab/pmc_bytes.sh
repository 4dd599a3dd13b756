#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of the kernels matching a pattern, for any python tool (GPU box; separate --pmc passes, program directly after `--`).
# usage: pmc_bytes.sh <kernel-name-regex> <python script> [args...]      raw KB medians (FETCH_SIZE needs the gfx950 x2 correction for wide reads)
PAT=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/pmc_bytes; rm -rf $OUT; mkdir -p $OUT
SCRIPT=$(cd "$(dirname "$1")" && pwd)/$(basename "$1"); shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- python3 $SCRIPT "$@" > $OUT/$c.out 2> $OUT/$c.log
  python3 - <<PY
import csv, collections, re
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("$OUT/$c/p_counter_collection.csv")):
    if r["Counter_Name"] == "$c" and re.search(r"""$PAT""", r["Kernel_Name"]): per[re.sub(r"\(.*", "", r["Kernel_Name"])[:80]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k, d in per.items():
    v = sorted(d.values()); print("$c KB median %12.1f  min %12.1f  launches %3d  %s" % (v[len(v) // 2], v[0], len(v), k))
PY
done
