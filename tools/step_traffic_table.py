#!/usr/bin/env python3
"""gpurun_out/r06_step_bytes.txt (tools/ab/pmc_bytes.sh "." bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras: FETCH_SIZE and WRITE_SIZE per launch of every
kernel of the cfg-2 step, separate --pmc passes) -> fabric traffic per training step by kernel: GB per step = (2 x FETCH_SIZE + WRITE_SIZE) x launches per step (the x2 is
the guide's gfx950 correction for wide reads; the fetch side counts L2 misses, Infinity-Cache hits included: an upper bound on HBM bytes).
usage: step_traffic_table.py <pmc_bytes output> <steps profiled (warm-up included)> <ms per step>"""
import collections, re, sys
path, steps, ms = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
f = collections.defaultdict(dict)
for l in open(path):
    m = re.match(r"(FETCH_SIZE|WRITE_SIZE) KB median\s+([\d.]+)\s+min\s+([\d.]+)\s+launches\s+(\d+)\s+(.*)", l)
    if m:
        f[m.group(5).strip()][m.group(1)] = (float(m.group(2)), int(m.group(4)))
rows = []
for k, d in f.items():
    fe, n1 = d.get("FETCH_SIZE", (0, 0)); wr, n2 = d.get("WRITE_SIZE", (0, 0)); n = max(n1, n2) / steps
    rows.append(((2 * fe + wr) * 1024 * n / 1e9, k, 2 * fe * 1024 / 1e6, wr * 1024 / 1e6, n))
rows.sort(reverse=True)
T = sum(r[0] for r in rows)
print(f"fabric traffic of one cfg-2 training step: {T:.0f} GB = {T / ms:.2f} TB/s averaged over a {ms:.1f} ms step ({T / ms / 8 * 100:.0f} % of the 8 TB/s HBM peak; upper bound: Infinity-Cache hits count)")
print(f"{'GB/step':>8s} {'share':>6s} {'fetch MB':>9s} {'write MB':>9s} {'launches':>8s}  kernel (median launch)")
for t, k, fe, wr, n in rows:
    if t >= 0.5:
        print(f"{t:8.1f} {100 * t / T:5.1f}% {fe:9.0f} {wr:9.0f} {n:8.0f}  {k[:100]}")
