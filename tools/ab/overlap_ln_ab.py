#!/usr/bin/env python3
"""Development tool (GPU): do an HBM-bound kernel (LayerNorm backward / forward) or the VALU-bound attention forward overlap with an MFMA-bound
GEMM (weight gradient k_gemm_tn, or a data-gradient k_gemm_nt) when the two are issued on two HIP streams?  Prints each kernel alone, the pair
back to back on one stream, and the pair on two streams."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oneprot_amd import hip
B, L, H, hd = 256, 512, 20, 32
T, d, f = B * L, 640, 2560
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
side = torch.cuda.Stream()
def bench(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
# LayerNorm backward as in the layer backward: dy bf16, x fp32, residual gradient accumulated in place, bf16 copy out, dgamma / dbeta
x = rnd(T, d); dy = rnd(T, d).to(torch.bfloat16); gamma = torch.ones(d, device="cuda"); mean = x.mean(1).contiguous(); rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
dx = rnd(T, d); dxb = torch.empty(T, d, dtype=torch.bfloat16, device="cuda"); dg = torch.zeros(d, device="cuda"); dbt = torch.zeros(d, device="cuda")
lnws = torch.empty(hip.query("oneprot_layernorm_bwd_workspace", d), dtype=torch.uint8, device="cuda")
ln_bwd = lambda: hip.call("oneprot_layernorm_bwd", dy, 0, None, 0, x, 0, gamma, mean, rstd, dx, dx, dxb, dg, dbt, lnws, T, d, 1)
yb = torch.empty(T, d, dtype=torch.bfloat16, device="cuda"); beta = torch.zeros(d, device="cuda")
ln_fwd = lambda: hip.call("oneprot_layernorm_fwd", x, 0, gamma, beta, yb, None, mean, rstd, T, d, 1e-5)
# attention forward
mk = lambda: (torch.randn(B, H, L, hd, device="cuda", generator=g) * 0.7).to(torch.bfloat16)
q, k, v = mk(), mk(), mk()
ctx = torch.empty(T, H * hd, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, L, device="cuda")
attn_fwd = lambda: hip.call("oneprot_attn_fwd", q, k, v, None, ctx, lse, B, H, L, hd)
# GEMMs
dY = rnd(T, f).to(torch.bfloat16); X = rnd(T, d).to(torch.bfloat16)
dW = torch.empty(f, d, device="cuda"); db = torch.empty(f, device="cuda")
ws = torch.empty(hip.query("oneprot_gemm_bf16_tn_workspace", f, d), dtype=torch.uint8, device="cuda")
wgrad = lambda: hip.call("oneprot_gemm_bf16_tn", dY, X, T, f, d, f, d, dW, db, ws, ws.numel(), 0)
Wt = (rnd(d, f) * 0.05).to(torch.bfloat16); dX = torch.empty(T, d, dtype=torch.bfloat16, device="cuda")
dgrad = lambda: hip.call("oneprot_gemm_bf16_nt", dY, Wt, T, d, f, f, f, hip.EPI_BF16, None, dX, None, None, None, None, None, 1.0, 0, 0, 0)
def pair(a, b):
    def seq():
        a(); b()
    def par():
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            a()
            done = torch.cuda.Event(); done.record()
        b()
        torch.cuda.current_stream().wait_event(done)
    return seq, par
for na, a, nb, b in (("ln_bwd", ln_bwd, "wgrad ffn1", wgrad), ("ln_bwd", ln_bwd, "dgrad ffn1", dgrad), ("ln_fwd", ln_fwd, "wgrad ffn1", wgrad), ("attn_fwd", attn_fwd, "wgrad ffn1", wgrad),
                     ("attn_fwd", attn_fwd, "dgrad ffn1", dgrad), ("attn_fwd", attn_fwd, "ln_bwd", ln_bwd)):
    seq, par = pair(a, b)
    r = {kk: [] for kk in ("a", "b", "seq", "par")}
    for rep in range(3):
        r["a"].append(bench(a)); r["b"].append(bench(b)); r["seq"].append(bench(seq)); r["par"].append(bench(par))
    m = {kk: statistics.median(vv) for kk, vv in r.items()}
    print(f"{na:9s} {m['a']:.3f} ms   {nb:11s} {m['b']:.3f} ms   one stream {m['seq']:.3f}   two streams {m['par']:.3f}   (saved {100 * (1 - m['par'] / m['seq']):.0f} %)", flush=True)
