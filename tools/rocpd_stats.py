#!/usr/bin/env python3
"""rocprofv3 (rocpd sqlite output) -> per-kernel stats CSV in the layout of `rocprofv3 --stats` kernel_stats.csv.
usage: rocpd_stats.py results.db out.csv"""
import csv, os, sqlite3, statistics, subprocess, sys


def demangle(names):
    for exe in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([exe], input="\n".join(n[:-3] if n.endswith(".kd") else n for n in names), capture_output=True, text=True, check=True).stdout.split("\n")
            return dict(zip(names, out))
        except Exception:
            continue
    return {n: n for n in names}


con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]; kd = [t for t in tabs if "kernel_dispatch" in t][0]
rows = {}
for name, dur in cur.execute(f"select s.kernel_name, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id"):
    rows.setdefault(name, []).append(dur)
total = sum(sum(v) for v in rows.values())
nice = demangle(list(rows))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([nice[name], len(v), sum(v), round(sum(v) / len(v), 3), round(100.0 * sum(v) / total, 4), min(v), max(v), round(statistics.pstdev(v), 3)])
print("kernels", len(rows), "total ms", total / 1e6)
