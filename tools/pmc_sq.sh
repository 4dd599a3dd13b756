#!/bin/bash
# SQ counter passes (8 counters each) of one program: tools/pmc_sq.sh <outdir> <program args...>   (GPU box)
set -e
OUT=$1; shift
mkdir -p $OUT
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/p1 -o p -- "$@" > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p2 -o p -- "$@" > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $OUT/p3 -o p -- "$@" > $OUT/p3.log 2>&1 || echo "pass 3 failed (counter names)"
python3 - $OUT <<'PY'
import csv, glob, re, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k, c in agg.items():
    print(k)
    for name, v in sorted(c.items()): print(f"   {name:34s} {v / max(1, len(n[(k, name)])):16.0f} per launch")
PY
