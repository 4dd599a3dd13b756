#!/usr/bin/env python3
"""In-process A/B of the NT-GEMM block shapes on the training shapes (interleaved rounds, median) -- devices differ by >10 % in wall time,
so shapes are only ever compared inside one process.  usage: gemm_ab.py [rounds]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oneprot_amd import hip
B, L, H, hd = 256, 512, 20, 32
d, f, T = 640, 2560, 256 * 512
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
shapes = [int(x) for x in os.environ.get("AB_SHAPES", "0,1,2,3,4,5").split(",")]
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
cos = torch.rand(L, hd // 2, device="cuda"); sin = torch.rand(L, hd // 2, device="cuda")
cases = {}
def mk(name, N, K, epi):
    A = rnd(T, K).to(torch.bfloat16); W = (rnd(N, K) * 0.05).to(torch.bfloat16); bias = rnd(N)
    if epi == hip.EPI_QKV_ROPE:
        o = [torch.empty(B, H, L, hd, dtype=torch.bfloat16, device="cuda") for _ in range(3)]
        fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias, o[0], o[1], o[2], None, cos, sin, hd ** -0.5, L, H, hd)
    elif epi == hip.EPI_BIAS_RESID:
        o0 = rnd(T, N)
        fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias, o0, None, None, o0, None, None, 1.0, 0, 0, 0)
    elif epi == hip.EPI_BIAS_GELU:
        o0 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda"); o1 = torch.empty_like(o0)
        fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias, o0, o1, None, None, None, None, 1.0, 0, 0, 0)
    elif epi == hip.EPI_GELU_BWD:
        o0 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda"); aux = rnd(T, N).to(torch.bfloat16)
        fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, None, o0, None, None, aux, None, None, 1.0, 0, 0, 0)
    else:
        o0 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda")
        fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, None, o0, None, None, None, None, None, 1.0, 0, 0, 0)
    cases[name] = (fn, 2.0 * T * N * K)
mk("qkv_fwd   [T,640]x[1920,640] rope", 3 * d, d, hip.EPI_QKV_ROPE)
mk("out_fwd   [T,640]x[640,640] resid", d, d, hip.EPI_BIAS_RESID)
mk("ffn1_fwd  [T,640]x[2560,640] gelu", f, d, hip.EPI_BIAS_GELU)
mk("ffn2_fwd  [T,2560]x[640,2560] resid", d, f, hip.EPI_BIAS_RESID)
mk("ffn2_dgrad[T,640]x[2560,640] gelu'", f, d, hip.EPI_GELU_BWD)
mk("ffn1_dgrad[T,2560]x[640,2560] bf16", d, f, hip.EPI_BF16)
mk("out_dgrad [T,640]x[640,640] bf16", d, d, hip.EPI_BF16)
mk("qkv_dgrad [T,1920]x[640,1920] bf16", d, 3 * d, hip.EPI_BF16)
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
res = {(c, s): [] for c in cases for s in shapes}
# AB_NT=1: shapes >= 1000 mean "shape - 1000 with non-temporal output stores"
for r in range(rounds):
    for c, (fn, fl) in cases.items():
        for s in shapes:
            hip.query("oneprot_gemm_tune", 256 * (2 if s >= 1000 else 1), 0)
            hip.query("oneprot_gemm_force_shape", s - 1000 if s >= 1000 else s)
            res[(c, s)].append(timeit(fn))
hip.query("oneprot_gemm_tune", 256, 0)
hip.query("oneprot_gemm_force_shape", -1)
print("median ms per launch (TFLOP/s); shapes: 0=128x128 1=256x128 2=256x256bk32x4 3=128x128bk64 4=256x256bk64x2 5=256x256 nopipe")
for c, (fn, fl) in cases.items():
    row = []
    best = min(shapes, key=lambda s: statistics.median(res[(c, s)]))
    for s in shapes:
        m = statistics.median(res[(c, s)])
        row.append(f"s{s}:{m:.3f}({fl / m / 1e9:.0f}){'*' if s == best else ' '}")
    print(f"{c:40s} " + " ".join(row))
