#!/usr/bin/env python3
"""LayerNorm forward / backward launch times at the cfg-2 shape (T = 131072, d = 640), GB/s of algorithmic bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oneprot_amd import hip
if os.environ.get("ONEPROT_LIB"): hip.LIB_PATH = os.path.abspath(os.environ["ONEPROT_LIB"])
T, d = 131072, int(sys.argv[1]) if len(sys.argv) > 1 else 640
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
def bench(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
x = rnd(T, d); dy = rnd(T, d).to(torch.bfloat16); gamma = torch.ones(d, device="cuda"); beta = torch.zeros(d, device="cuda")
mean = x.mean(1).contiguous(); rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
dx = rnd(T, d); dxb = torch.empty(T, d, dtype=torch.bfloat16, device="cuda"); dg = torch.zeros(d, device="cuda"); dbt = torch.zeros(d, device="cuda")
ws = torch.empty(hip.query("oneprot_layernorm_bwd_workspace", d), dtype=torch.uint8, device="cuda")
yb = torch.empty(T, d, dtype=torch.bfloat16, device="cuda")
tb = bench(lambda: hip.call("oneprot_layernorm_bwd", dy, 0, None, 0, x, 0, gamma, mean, rstd, dx, dx, dxb, dg, dbt, ws, T, d, 1))
tf = bench(lambda: hip.call("oneprot_layernorm_fwd", x, 0, gamma, beta, yb, None, mean, rstd, T, d, 1e-5))
print(f"d={d}: LN bwd {tb * 1e3:.1f} us ({T * d * 16 / tb / 1e6:.0f} GB/s of 16 B/elem)   LN fwd {tf * 1e3:.1f} us ({T * d * 6 / tf / 1e6:.0f} GB/s)")
