#!/usr/bin/env python3
"""Development tool (GPU): fused GEMM + residual + LayerNorm (oneprot_gemm_bf16_nt_resid_ln) against the pair it replaces
(oneprot_gemm_bf16_nt bias+residual, then oneprot_layernorm_fwd) on the cfg-2 shapes; interleaved rounds, medians."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oneprot_amd import hip
if os.environ.get("ONEPROT_LIB"): hip.LIB_PATH = os.path.abspath(os.environ["ONEPROT_LIB"])
T, N = 131072, 640
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name, K in (("epilogue-only K=64", 64), ("out-proj K=640", 640), ("FFN-2   K=2560", 2560)):
    A = rnd(T, K).to(torch.bfloat16); W = (rnd(N, K) * 0.05).to(torch.bfloat16); bias = rnd(N); gamma = rnd(N); beta = rnd(N)
    Wp = torch.empty(N * K, dtype=torch.bfloat16, device="cuda")
    hip.call("oneprot_gemm_ln_pack_weight", W, Wp, N, K)
    x = rnd(T, N); h = torch.empty(T, N, dtype=torch.bfloat16, device="cuda"); mean = torch.empty(T, device="cuda"); rstd = torch.empty(T, device="cuda")
    for inplace in (True, False):
        xo = x if inplace else torch.empty_like(x)
        def pair():
            hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, hip.EPI_BIAS_RESID, bias, xo, None, None, x, None, None, 1.0, 0, 0, 0)
            hip.call("oneprot_layernorm_fwd", xo, 0, gamma, beta, h, None, mean, rstd, T, N, 1e-5)
        fused = lambda: hip.call("oneprot_gemm_bf16_nt_resid_ln", A, Wp, T, N, K, K, bias, x, xo, gamma, beta, 1e-5, h, mean, rstd)
        gemm_only = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, hip.EPI_BIAS_RESID, bias, xo, None, None, x, None, None, 1.0, 0, 0, 0)
        res = {"pair": [], "fused": [], "gemm": []}
        for r in range(3):
            res["pair"].append(timeit(pair)); res["fused"].append(timeit(fused)); res["gemm"].append(timeit(gemm_only))
        m = {k: statistics.median(v) for k, v in res.items()}
        gb = (T * K * 2 + T * N * 4 * 2 + T * N * 2) / 1e9
        print(f"{name} {'in place' if inplace else 'out of place'}: GEMM+LN pair {m['pair']:.3f} ms (GEMM alone {m['gemm']:.3f})  fused {m['fused']:.3f} ms "
              f"({2.0 * T * N * K / m['fused'] / 1e9:.0f} TFLOP/s, {gb / m['fused'] * 1e3:.0f} GB/s of {gb:.2f} GB)", flush=True)
