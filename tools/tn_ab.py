#!/usr/bin/env python3
"""Development tool (GPU): in-process A/B of the weight-gradient (TN) GEMM variants on the training shapes.  usage: tn_ab.py [variants...]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oneprot_amd import hip
if os.environ.get("G8_LIB"): hip.LIB_PATH = os.path.abspath(os.environ["G8_LIB"])
T, d, f = 131072, 640, 2560
variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2]
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, N, K in (("qkv  dW[1920,640]", 3 * d, d), ("out  dW[640,640]", d, d), ("ffn1 dW[2560,640]", f, d), ("ffn2 dW[640,2560]", d, f)):
    dY = torch.randn(T, N, device="cuda", generator=g).to(torch.bfloat16)
    X = torch.randn(T, K, device="cuda", generator=g).to(torch.bfloat16)
    dW, db = torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda")
    ws = torch.empty(hip.query("oneprot_gemm_bf16_tn_workspace", N, K), dtype=torch.uint8, device="cuda")
    fn = lambda: hip.call("oneprot_gemm_bf16_tn", dY, X, T, N, K, N, K, dW, db, ws, ws.numel(), 0)
    res, outs = {v: [] for v in variants}, {}
    for rep in range(3):
        for v in variants:
            hip.query("oneprot_gemm_tn_variant", v)
            res[v].append(timeit(fn))
            outs[v] = (dW.clone(), db.clone())
    hip.query("oneprot_gemm_tn_variant", -1)
    fl = 2.0 * T * N * K
    ref = outs[variants[0]]
    chk = " ".join(f"v{v}:maxdiff {float((outs[v][0] - ref[0]).abs().max()):.2e}/{float((outs[v][1] - ref[1]).abs().max()):.2e}" for v in variants[1:])
    print(f"{name:18s} " + "  ".join(f"v{v}:{statistics.median(t):.0f}us({fl / statistics.median(t) / 1e6:.0f}TF)" for v, t in res.items()) + "   " + chk, flush=True)
