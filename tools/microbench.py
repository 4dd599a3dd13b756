#!/usr/bin/env python3
"""Per-kernel timing at the cfg-2 shapes (ESM-2-150M, B=256, L=512 => T=131072 tokens). Development tool, GPU only."""
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oneprot_amd import hip

DEV = "cuda"


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B, L, H, hd = 256, 512, 20, 32
    d, f = H * hd, 2560
    T = B * L
    if len(sys.argv) > 1:
        B = int(sys.argv[1]); T = B * L
    g = torch.Generator(device=DEV).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
    h = rnd(T, d).to(torch.bfloat16)
    x = rnd(T, d)
    res = {}
    only = os.environ.get("MB_ONLY", "")
    # ---- GEMMs
    for shape in ([-1] if not os.environ.get("MB_SHAPES") else [int(x) for x in os.environ["MB_SHAPES"].split(",")]):
      hip.query("oneprot_gemm_force_shape", shape)
      for name, N, K, epi in (("qkv", 3 * d, d, hip.EPI_QKV_ROPE), ("out", d, d, hip.EPI_BIAS_RESID), ("ffn1", f, d, hip.EPI_BIAS_GELU), ("ffn2", d, f, hip.EPI_BIAS_RESID),
                            ("plain_ffn1", f, d, hip.EPI_BF16)):
          A = rnd(T, K).to(torch.bfloat16)
          W = (rnd(N, K) * 0.05).to(torch.bfloat16)
          bias = rnd(N)
          cos = torch.rand(L, hd // 2, device=DEV); sin = torch.rand(L, hd // 2, device=DEV)
          if epi == hip.EPI_QKV_ROPE:
              o0, o1, o2 = (torch.empty(B, H, L, hd, dtype=torch.bfloat16, device=DEV) for _ in range(3))
              fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias, o0, o1, o2, None, cos, sin, hd ** -0.5, L, H, hd)
          elif epi == hip.EPI_BIAS_RESID:
              o0 = rnd(T, N)
              fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias, o0, None, None, o0, None, None, 1.0, 0, 0, 0)
          elif epi == hip.EPI_BIAS_GELU:
              o0 = torch.empty(T, N, dtype=torch.bfloat16, device=DEV); o1 = torch.empty_like(o0)
              fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias, o0, o1, None, None, None, None, 1.0, 0, 0, 0)
          else:
              o0 = torch.empty(T, N, dtype=torch.bfloat16, device=DEV)
              fn = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias, o0, None, None, None, None, None, 1.0, 0, 0, 0)
          ms = timeit(fn)
          res[f"{name}[s{shape}]"] = (ms, 2.0 * T * N * K / ms / 1e9)
          del A, W
    hip.query("oneprot_gemm_force_shape", -1)
    if only == "gemm":
        return _report(res)
    # ---- wgrad
    for name, N, K in (("wgrad_qkv", 3 * d, d), ("wgrad_ffn1", f, d), ("wgrad_ffn2", d, f), ("wgrad_out", d, d)):
        dY = rnd(T, N).to(torch.bfloat16); X = rnd(T, K).to(torch.bfloat16)
        dW = torch.empty(N, K, device=DEV)
        w = torch.empty(hip.query("oneprot_gemm_bf16_tn_workspace", N, K), dtype=torch.uint8, device=DEV)
        for variant in (0, 1, 0, 1):
            hip.query("oneprot_gemm_tn_variant", variant)
            ms = timeit(lambda: hip.call("oneprot_gemm_bf16_tn", dY, X, T, N, K, N, K, dW, None, w, w.numel(), 0))
            key = f"{name}[v{variant}]"
            if key not in res or ms < res[key][0]:
                res[key] = (ms, 2.0 * T * N * K / ms / 1e9)
        hip.query("oneprot_gemm_tn_variant", -1)
        del dY, X
    # ---- attention
    q, k, v = (rnd(B, H, L, hd).to(torch.bfloat16) for _ in range(3))
    ctx = torch.empty(T, d, dtype=torch.bfloat16, device=DEV); lse = torch.empty(B, H, L, device=DEV)
    ms = timeit(lambda: hip.call("oneprot_attn_fwd", q, k, v, None, ctx, lse, B, H, L, hd))
    res["attn_fwd"] = (ms, 4.0 * B * H * L * L * hd / ms / 1e9)
    dctx = rnd(T, d).to(torch.bfloat16); dqkv = torch.empty(T, 3 * d, dtype=torch.bfloat16, device=DEV)
    w = torch.empty(hip.query("oneprot_attn_bwd_workspace", B, H, L), dtype=torch.uint8, device=DEV)
    cos = torch.rand(L, hd // 2, device=DEV); sin = torch.rand(L, hd // 2, device=DEV)
    ms = timeit(lambda: hip.call("oneprot_attn_bwd", q, k, v, None, ctx, dctx, lse, cos, sin, hd ** -0.5, dqkv, w, B, H, L, hd))
    res["attn_bwd"] = (ms, 8.0 * B * H * L * L * hd / ms / 1e9)
    # ---- LN
    gamma, beta = torch.ones(d, device=DEV), torch.zeros(d, device=DEV)
    y = torch.empty(T, d, dtype=torch.bfloat16, device=DEV); mean = torch.empty(T, device=DEV); rstd = torch.empty(T, device=DEV)
    ms = timeit(lambda: hip.call("oneprot_layernorm_fwd", x, 0, gamma, beta, y, None, mean, rstd, T, d, 1e-5))
    res["ln_fwd"] = (ms, T * d * 6 / ms / 1e6)   # GB/s
    dx = torch.empty(T, d, device=DEV); dg = torch.zeros(d, device=DEV); db = torch.zeros(d, device=DEV)
    w = torch.empty(hip.query("oneprot_layernorm_bwd_workspace", d), dtype=torch.uint8, device=DEV)
    ms = timeit(lambda: hip.call("oneprot_layernorm_bwd", y, 0, None, 0, x, 0, gamma, mean, rstd, dx, dx, y, dg, db, w, T, d, 0))
    res["ln_bwd"] = (ms, T * d * (2 + 4 + 4 + 4) / ms / 1e6)
    _report(res)


def _report(res):
    for k_, (ms, rate) in res.items():
        unit = "GB/s" if k_.startswith("ln") else "TFLOP/s"
        val = rate if k_.startswith("ln") else rate / 1e3
        print(f"{k_:12s} {ms:8.3f} ms   {val:9.1f} {unit}")


if __name__ == "__main__":
    main()
