#!/usr/bin/env python3
"""Development tool (GPU): L2 super-tile shapes of the per-tile NT GEMM kernels on the training shapes (in-process, interleaved, medians)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oneprot_amd import hip
T, d, f = 131072, 640, 2560
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
cases = []
for name, N, K, epi, shape in (("ffn1 gelu s4", f, d, 2, 4), ("ffn1 gelu s20", f, d, 2, 20), ("ffn1 bf16 s4", f, d, 0, 4), ("qkv bf16(rope-less) s1", 3 * d, d, 0, 1), ("ffn1_dgrad bf16 s3", d, f, 0, 3), ("ffn1_dgrad bf16 s4", d, f, 0, 4)):
    A = rnd(T, K).to(torch.bfloat16); W = (rnd(N, K) * 0.05).to(torch.bfloat16); bias = rnd(N)
    o0 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda")
    cases.append((name, shape, lambda A=A, W=W, bias=bias, o0=o0, N=N, K=K, epi=epi: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias, o0, None, None, None, None, None, 1.0, 0, 0, 0), 2.0 * T * N * K))
tunes = [(4, 10), (4, 5), (8, 5), (2, 10), (4, 20), (8, 10), (2, 5), (16, 5), (8, 3), (4, 3)]
res = {}
for rep in range(3):
    for name, shape, fn, fl in cases:
        hip.query("oneprot_gemm_force_shape", shape)
        for t in tunes:
            hip.query("oneprot_gemm_tune", *t)
            res.setdefault((name, t), []).append(timeit(fn))
hip.query("oneprot_gemm_tune", 4, 10); hip.query("oneprot_gemm_force_shape", -1)
for name, shape, fn, fl in cases:
    best = min(tunes, key=lambda t: statistics.median(res[(name, t)]))
    print(f"{name:26s} " + " ".join(f"{t[0]}x{t[1]}:{statistics.median(res[(name, t)]):.3f}{'*' if t == best else ' '}" for t in tunes), flush=True)
