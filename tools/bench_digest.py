#!/usr/bin/env python3
"""Prints the figures of a bench.py JSON line that a round is steered by."""
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print(f"{j['value']} pairs/s  {j['ms_per_step']} ms/step  step {j.get('step_tflops_per_gpu')} TFLOP/s | FFN-1 {r['avg_launch_ms']} ms frac {r['frac']}")
e = j.get("encoder_fwd")
if e: print(f"encoder_fwd {e['ms']} ms {e['achieved']} TFLOP/s frac {e['frac']}  mfma busy {e.get('mfma_busy_pct')} %  at the measured clock ({(e.get('board') or {}).get('sclk_mhz_mean')} MHz): {e.get('frac_at_that_clock')}")
for k in j.get("kernels", []): print(f"  {k['avg_ms']:8.4f} ms x{k['launches']:4d}  {k['rate']:8.1f} {k['unit']:8s} {k['frac_of_peak']:.3f}  {k['kernel']}")
for k, v in (j.get("other_workloads") or {}).items(): print(f"  {k}: {v['value']} pairs/s {v['ms_per_step']} ms/step loss {v['loss']}")
c = j.get("cpu_baseline")
if c: print(f"cpu {c['value']} pairs/s on {c['cores']} cores {c.get('host_cores')} | cfg1 {c['cfg1']['value']}")
b = j.get("board")
if b: print(f"board {b['power_w_mean']:.0f} W of {b['power_cap_w']} cap, sclk {b['sclk_mhz_mean']:.0f} MHz ({b['sclk_mhz_min']:.0f}-{b['sclk_mhz_max']:.0f}) -> MFMA peak at that clock {b['bf16_mfma_peak_at_that_clock_tflops']:.0f} TFLOP/s: FFN-1 {b['roofline_frac_at_that_clock']}, step {b['step_frac_at_that_clock']}")
