#!/usr/bin/env python3
"""Development tool (GPU): the 8-phase NT GEMM forms (oneprot_gemm_force_shape 40 = 256x256, 41 = 256x320) against the per-tile heuristic on the
training shapes, interleaved rounds in one process, medians.  `noepi` = the same launch with the epilogue skipped (store policy 77: timing only).
usage: g8_ab.py [rounds]   env G8_SHAPES=-1,40,41  G8_LIB=<other liboneprot_hip.so>"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oneprot_amd import hip
if os.environ.get("G8_LIB"):      # another build of the whole library (e.g. a -D variant made with build.sh into another file)
    hip.LIB_PATH = os.path.abspath(os.environ["G8_LIB"])
B, L, H, hd = 256, 512, 20, 32
d, f, T = 640, 2560, 256 * 512
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
shapes = [int(x) for x in os.environ.get("G8_SHAPES", "-1,40,41").split(",")]
import ctypes
_dph = hip.lib().oneprot_gemm8_dephase; _dph.argtypes = [ctypes.c_int, ctypes.c_int]; _dph.restype = None
DPH = [tuple(int(v) for v in x.split(":")) for x in os.environ.get("G8_DPH", "1:0").split(",")]      # groups:sleeps variants of the 8-phase columns
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
cos = torch.rand(L, hd // 2, device="cuda"); sin = torch.rand(L, hd // 2, device="cuda")
cases = {}
def mk(name, N, K, epi, two=True):
    A = rnd(T, K).to(torch.bfloat16); W = (rnd(N, K) * 0.05).to(torch.bfloat16); bias = rnd(N)
    if epi == hip.EPI_QKV_ROPE:
        o = [torch.empty(B, H, L, hd, dtype=torch.bfloat16, device="cuda") for _ in range(3)]
        fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias, o[0], o[1], o[2], None, cos, sin, hd ** -0.5, L, H, hd)
    elif epi == hip.EPI_BIAS_RESID:
        o0 = rnd(T, N)
        fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias, o0, None, None, o0, None, None, 1.0, 0, 0, 0)
    elif epi == hip.EPI_BIAS_GELU:
        o0 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda"); o1 = torch.empty_like(o0) if two else None
        fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias, o0, o1, None, None, None, None, 1.0, 0, 0, 0)
    elif epi == hip.EPI_GELU_BWD:
        o0 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda"); aux = rnd(T, N).to(torch.bfloat16)
        fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, None, o0, None, None, aux, None, None, 1.0, 0, 0, 0)
    else:
        o0 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda")
        fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, None, o0, None, None, None, None, None, 1.0, 0, 0, 0)
    cases[name] = (fn, 2.0 * T * N * K)
mk("qkv_fwd    N1920 K640  rope", 3 * d, d, hip.EPI_QKV_ROPE)
mk("out_fwd    N640  K640  resid", d, d, hip.EPI_BIAS_RESID)
mk("ffn1_fwd   N2560 K640  gelu+gelu'", f, d, hip.EPI_BIAS_GELU)
mk("ffn1_fwd   N2560 K640  gelu only", f, d, hip.EPI_BIAS_GELU, two=False)
mk("plain      N2560 K640  bf16", f, d, hip.EPI_BF16)
mk("ffn2_fwd   N640  K2560 resid", d, f, hip.EPI_BIAS_RESID)
mk("ffn2_dgrad N2560 K640  gelu'", f, d, hip.EPI_GELU_BWD)
mk("ffn1_dgrad N640  K2560 bf16", d, f, hip.EPI_BF16)
mk("out_dgrad  N640  K640  bf16", d, d, hip.EPI_BF16)
mk("qkv_dgrad  N640  K1920 bf16", d, 3 * d, hip.EPI_BF16)
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
cols = [(s, False, DPH[0]) for s in shapes] + [(s, False, dp) for s in shapes if s >= 40 for dp in DPH[1:]] + [(s, True, DPH[0]) for s in shapes if s >= 40]
res = {(c, k): [] for c in cases for k in cols}
for r in range(rounds):
    for c, (fn, fl) in cases.items():
        for (s, noepi, dp) in cols:
            hip.query("oneprot_gemm_tune", 256 * (78 if noepi else (2 if os.environ.get("G8_NT") == "1" and s >= 40 else 1)), 0)
            hip.query("oneprot_gemm_force_shape", s)
            _dph(*dp)
            res[(c, (s, noepi, dp))].append(timeit(fn))
hip.query("oneprot_gemm_tune", 256, 0)
hip.query("oneprot_gemm_force_shape", -1)
_dph(1, 0)
print("median ms per launch (TFLOP/s)")
for c, (fn, fl) in cases.items():
    row = []
    for k in cols:
        m = statistics.median(res[(c, k)])
        row.append(f"s{k[0]}{'noepi' if k[1] else ''}{'' if k[2] == DPH[0] else '/d%d:%d' % k[2]}:{m:.3f}({fl / m / 1e9:.0f})")
    print(f"{c:36s} " + " ".join(row), flush=True)
