#!/bin/bash
# Development tool (GPU): LDS bank-conflict counters of the 8-phase weight-gradient GEMM (variant 3) on the cfg-2 shapes.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/tn8_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc ${TN8_PMC:-FETCH_SIZE} --output-format csv -d $OUT/p1 -o p -- python3 $ROOT/tools/tn_ab.py 3 > $OUT/p1.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, re, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k, c in agg.items():
    if "tn8" not in k: continue
    print(k)
    for name, v in sorted(c.items()): print(f"   {name:34s} {v / max(1, len(n[(k, name)])):16.0f} per launch")
PY
