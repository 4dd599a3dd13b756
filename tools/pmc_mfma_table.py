#!/usr/bin/env python3
"""rocprofv3 counter_collection.csv (pass: --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace)
-> per-kernel MFMA-pipe busy fraction.  MfmaBusy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CUs * 4 SIMDs)  (gfx94x MfmaUtil formula;
GRBM_GUI_ACTIVE is reported summed over the 8 XCDs).  Clocks under the profiler are lower than in an un-profiled run: compare busy fractions, not times.
usage: pmc_mfma_table.py counter_collection.csv out.csv"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for r in rows:
    k = re.sub(r"\(.*", "", r["Kernel_Name"])
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
out = []
for k, c in agg.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8
    if gui > 0:
        out.append((gui, k, len(calls[k]), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 256 * 4), c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)))
tot = sum(o[0] for o in out)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f); w.writerow(["kernel", "calls", "share_of_gpu_active_cycles_pct", "mfma_pipe_busy_pct", "SQ_INSTS_VALU_MFMA_MOPS_BF16"])
    for gui, k, n, util, mops in sorted(out, reverse=True):
        w.writerow([k, n, round(100 * gui / tot, 2), round(100 * util, 2), int(mops)])
busy = sum(o[0] * o[3] for o in out) / tot
print(f"whole run: MFMA pipe busy {100 * busy:.1f} % of GPU-active cycles")
