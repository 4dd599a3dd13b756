#!/usr/bin/env python3
"""Runs a handful of launches of one GEMM configuration (for rocprofv3 --pmc passes).  usage: gemm_only.py <shape> <name>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oneprot_amd import hip
shape = int(sys.argv[1]); name = sys.argv[2]; iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
B, L, H, hd = 256, 512, 20, 32
d, f, T = 640, 2560, 256 * 512
cfgs = {"qkv": (3 * d, d, hip.EPI_QKV_ROPE), "out": (d, d, hip.EPI_BIAS_RESID), "ffn1": (f, d, hip.EPI_BIAS_GELU), "ffn1fwd": (f, d, hip.EPI_BIAS_GELU), "ffn2": (d, f, hip.EPI_BIAS_RESID),
        "plain": (f, d, hip.EPI_BF16),
        "plain640": (d, d, hip.EPI_BF16)}
N, K, epi = cfgs[name]
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(T, K, device="cuda", generator=g).to(torch.bfloat16)
W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
bias = torch.randn(N, device="cuda", generator=g)
cos = torch.rand(L, hd // 2, device="cuda"); sin = torch.rand(L, hd // 2, device="cuda")
hip.query("oneprot_gemm_force_shape", shape)
if os.environ.get("GEMM_TUNE"):      # e.g. 512 = non-temporal output stores (sup_m field 256 * (1 + policy))
    hip.query("oneprot_gemm_tune", int(os.environ["GEMM_TUNE"]), 0)
if epi == hip.EPI_QKV_ROPE:
    o = [torch.empty(B, H, L, hd, dtype=torch.bfloat16, device="cuda") for _ in range(3)]
    fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias, o[0], o[1], o[2], None, cos, sin, hd ** -0.5, L, H, hd)
elif epi == hip.EPI_BIAS_RESID:
    o0 = torch.randn(T, N, device="cuda")
    fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias, o0, None, None, o0, None, None, 1.0, 0, 0, 0)
elif epi == hip.EPI_BIAS_GELU:
    o0 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda"); o1 = torch.empty_like(o0) if name == "ffn1" else None      # ffn1fwd: frozen tower, no derivative output
    fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias, o0, o1, None, None, None, None, 1.0, 0, 0, 0)
else:
    o0 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda")
    fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, None, o0, None, None, None, None, None, 1.0, 0, 0, 0)
for _ in range(iters):
    fn()
torch.cuda.synchronize()
