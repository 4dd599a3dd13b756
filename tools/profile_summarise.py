#!/usr/bin/env python3
"""gpurun_out/prof_<round>/ (written by tools/profile_round.sh on the GPU box) -> committed summaries under profiles/<round>_*:
   <round>_kernel_stats.csv       rocprofv3 --kernel-trace --stats of the bench.py command
   <round>_bench_profiled.json    the JSON line of that profiled run
   <round>_pmc_traffic.json       FETCH_SIZE (x2: gfx950 correction for wide coalesced reads) + WRITE_SIZE per launch of the dominant kernel
   <round>_encoder_fwd_mfma_busy.json / <round>_mfma_busy.csv   MFMA-pipe busy fractions (encoder forward; whole step per kernel)
usage: profile_summarise.py r02"""
import collections, csv, json, os, re, shutil, subprocess, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", f"prof_{R}")
DST = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(SRC, "stats", "bench_kernel_stats.csv"), os.path.join(DST, f"{R}_kernel_stats.csv"))
line = [l for l in open(os.path.join(SRC, "bench_profiled.json")) if l.startswith("{")][-1]
open(os.path.join(DST, f"{R}_bench_profiled.json"), "w").write(line)


def counter_per_dispatch(d, counter, pattern):
    rows = [r for r in csv.DictReader(open(os.path.join(SRC, d, "p_counter_collection.csv"))) if r["Counter_Name"] == counter and re.search(pattern, r["Kernel_Name"])]
    per = collections.defaultdict(float)
    for r in rows:
        per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    vals = sorted(per.values())
    return vals, (rows[0]["Kernel_Name"] if rows else None)


traffic = {}
for what, label in (("ffn1", "two outputs (bf16 gelu and one-byte gelu' codes: trainable tower)"), ("ffn1fwd", "one bf16 output (frozen tower)")):
    f, kn = counter_per_dispatch(f"fetch_{what}", "FETCH_SIZE", "k_gemm")
    w, _ = counter_per_dispatch(f"write_{what}", "WRITE_SIZE", "k_gemm")
    fmed, wmed = f[len(f) // 2], w[len(w) // 2]
    alg = 131072 * 640 * 2 + 2560 * 640 * 2 + 131072 * 2560 * (3 if what == "ffn1" else 2)      # bf16 gelu(z) (+ the one-byte gelu' codes since round 5)
    traffic[what] = {"kernel": kn, "what": "FFN-1 [131072,640]x[2560,640]^T + bias + GELU, " + label, "FETCH_SIZE_KB_raw_median": fmed, "WRITE_SIZE_KB_raw_median": wmed,
                     "fetch_bytes": fmed * 1024 * 2, "write_bytes": wmed * 1024, "traffic_bytes_per_launch": fmed * 1024 * 2 + wmed * 1024, "algorithmic_bytes_per_launch": alg,
                     "ratio": round((fmed * 1024 * 2 + wmed * 1024) / alg, 3), "launches": len(f)}
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE --output-format csv -- python3 tools/gemm_only.py -1 ffn1|ffn1fwd 3  (separate passes, tools/profile_round.sh)",
       "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced streaming reads -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact for 16-B/lane stores",
       "note": "fetch side counts L2 fabric requests incl. Infinity-Cache hits, not only HBM", "traffic_bytes_per_launch": traffic["ffn1"]["traffic_bytes_per_launch"], **traffic}
json.dump(out, open(os.path.join(DST, f"{R}_pmc_traffic.json"), "w"), indent=1)
print("traffic", {k: (v["traffic_bytes_per_launch"] / 1e9, v["ratio"]) for k, v in traffic.items()})


def mfma_table(d):
    rows = list(csv.DictReader(open(os.path.join(SRC, d, "p_counter_collection.csv"))))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
    for r in rows:
        k = re.sub(r"\(.*", "", r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
    tab = []
    for k, c in agg.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8
        if gui > 0:
            tab.append((gui, k, len(calls[k]), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 256 * 4)))
    tot = sum(t[0] for t in tab)
    return tab, tot, sum(t[0] * t[3] for t in tab) / tot


tab, tot, busy = mfma_table("mfma_fwd")
json.dump({"mfma_busy_pct": round(100 * busy, 1), "workload": {"model": "facebook/esm2_t30_150M_UR50D", "batch": 256, "seq_len": 512}, "what": "ESM-2-150M sequence-encoder forward (tools/encoder_fwd_only.py: embedding + 30 layers, 256 x L=512), "
           "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CUs * 4 SIMDs), GPU-active-cycle weighted over its kernels",
           "command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/encoder_fwd_only.py 2",
           "per_kernel": [{"kernel": k[:90], "launches": n, "share_pct": round(100 * g / tot, 1), "mfma_busy_pct": round(100 * u, 1)} for g, k, n, u in sorted(tab, reverse=True)[:10]]},
          open(os.path.join(DST, f"{R}_encoder_fwd_mfma_busy.json"), "w"), indent=1)
print("encoder fwd MFMA busy", round(100 * busy, 1))
tab, tot, busy = mfma_table("mfma_step")
with open(os.path.join(DST, f"{R}_mfma_busy.csv"), "w", newline="") as f:
    w = csv.writer(f); w.writerow(["kernel", "calls", "share_of_gpu_active_cycles_pct", "mfma_pipe_busy_pct"])
    for g, k, n, u in sorted(tab, reverse=True):
        w.writerow([k, n, round(100 * g / tot, 2), round(100 * u, 2)])
print("whole step MFMA busy", round(100 * busy, 1))
