#!/usr/bin/env python3
"""Development tool (GPU): why a kernel takes longer inside the training step than in a loop of its own.  The attention backward / forward at the cfg-2 shape,
event-timed (a) back to back, (b) with n dense bf16 GEMM launches (the step's dgrad shape, K = 2560) in front of every launch -- the chip then sits at the clock
a sustained MFMA stream leaves it --, (c) with a 1.3 GB copy in front of every launch (caches cold, no MFMA load).  usage: in_step_probe.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oneprot_amd import hip
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, L, H, hd = 256, 512, 20, 32
T, d, f = B * L, 640, 2560
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda: (torch.randn(B, H, L, hd, device="cuda", generator=g) * 0.7).to(torch.bfloat16)
q, k, v = mk(), mk(), mk()
key_bias = torch.zeros(B, L, device="cuda")
ctx = torch.empty(T, H * hd, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, L, device="cuda")
dctx = (torch.randn(T, H * hd, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
dqkv = torch.empty(T, 3 * H * hd, dtype=torch.bfloat16, device="cuda")
ws = torch.empty(hip.query("oneprot_attn_bwd_workspace", B, H, L), dtype=torch.uint8, device="cuda")
cos = torch.rand(L, hd // 2, device="cuda"); sin = torch.rand(L, hd // 2, device="cuda")
A = torch.randn(T, f, device="cuda", generator=g).to(torch.bfloat16)
W = (torch.randn(d, f, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
o0 = torch.empty(T, d, dtype=torch.bfloat16, device="cuda")
big_a, big_b = torch.empty(650_000_000, dtype=torch.uint8, device="cuda"), torch.empty(650_000_000, dtype=torch.uint8, device="cuda")
fwd = lambda: hip.call("oneprot_attn_fwd", q, k, v, key_bias, ctx, lse, B, H, L, hd)
bwd = lambda: hip.call("oneprot_attn_bwd", q, k, v, key_bias, ctx, dctx, lse, cos, sin, hd ** -0.5, dqkv, ws, B, H, L, hd)
gemm = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, d, f, f, f, hip.EPI_BF16, None, o0, None, None, None, None, None, 1.0, 0, 0, 0)
copy = lambda: big_b.copy_(big_a)


def timed(fn, before=None, n_before=0):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(iters):
        for _ in range(n_before):
            before()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e3


for name, fn in (("attention backward", bwd), ("attention forward", fwd), ("dgrad GEMM K=2560", gemm)):
    a = timed(fn)
    b1 = timed(fn, gemm, 4)
    b2 = timed(fn, gemm, 16)
    c = timed(fn, copy, 1)
    print(f"{name:22s}: alone {a:7.1f} us | behind 4 GEMMs {b1:7.1f} | behind 16 GEMMs {b2:7.1f} | behind a 1.3 GB copy {c:7.1f}", flush=True)
