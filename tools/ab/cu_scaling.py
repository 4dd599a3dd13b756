#!/usr/bin/env python3
"""Development tool (GPU): is the NT GEMM bound per CU or chip-wide?  The persistent ping-pong form run on 32 / 16 / 8 work-groups per XCD
(ONEPROT_PP_G8, one process each): per-CU-bound -> time x2 per halving; chip-wide-bound -> less."""
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
hip.query("oneprot_gemm_force_shape", 32)
out = []
for name, N, K, epi in (("ffn1 bf16", f, d, 0), ("ffn1_dgrad bf16 K2560", d, f, 0), ("ffn1 gelu", f, d, 2)):
    A = rnd(T, K).to(torch.bfloat16); W = (rnd(N, K) * 0.05).to(torch.bfloat16); bias = rnd(N)
    o0 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda")
    fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias, o0, None, None, None, None, None, 1.0, 0, 0, 0)
    out.append(f"{name}: {statistics.median(timeit(fn) for _ in range(3)):.3f} ms")
print(f"g8={os.environ.get('ONEPROT_PP_G8', '32')}  " + "   ".join(out))
