#!/usr/bin/env python3
"""Development tool (GPU): where does a slot of the ping-pong NT GEMM go?  The product source built with ablation macros (timing only, results
are wrong by construction): base / NODMA (loader issues nothing) / HALFDMA (half of the LDS-DMA pieces) / NOMFMA (consumer reads fragments but
issues no MFMA) / NOREAD (MFMAs without fragment reads).   build: tools/ab/build_attn_variants.sh pp_base=../../oneprot_amd/csrc/gemm_nt.hip
pp_nodma=...,-DPP_ABL_NODMA ... ;  run: python tools/ab/pp_ablate.py"""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
here = os.path.dirname(os.path.abspath(__file__))
T, d, f = 131072, 640, 2560
P, I, F, L64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64
libs = {}
for n in ("pp_base", "pp_nodma", "pp_halfdma", "pp_nomfma", "pp_noread"):
    path = os.path.join(here, f"libattn_v{n}.so")
    if not os.path.exists(path):
        continue
    lib = ctypes.CDLL(path)
    lib.oneprot_gemm_bf16_nt.argtypes = [P, P, L64, I, I, I, I, I, P, P, P, P, P, P, P, F, I, I, I, P]
    lib.oneprot_gemm_force_shape.argtypes = [I]
    libs[n] = lib
g = torch.Generator(device="cuda").manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
ptr = lambda t: t.data_ptr() if t is not None else None
shape = int(os.environ.get("PP_SHAPE", "32"))
for (name, N, K, epi, grad) in (("ffn1 gelu+grad", f, d, 2, True), ("ffn1 gelu", f, d, 2, False), ("ffn1 bf16", f, d, 0, False), ("ffn1_dgrad bf16 K2560", d, f, 0, False), ("qkv_dgrad bf16 K1920", d, 3 * d, 0, False)):
    A = torch.randn(T, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    o0 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda")
    o1 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda") if grad else None
    res = {}
    for rep in range(3):
        for n, lib in libs.items():
            lib.oneprot_gemm_force_shape(shape)
            fn = lambda: lib.oneprot_gemm_bf16_nt(ptr(A), ptr(W), T, N, K, K, K, epi, ptr(bias), ptr(o0), ptr(o1), None, None, None, None, 1.0, 0, 0, 0, st)
            res.setdefault(n, []).append(timeit(fn))
    fl = 2.0 * T * N * K
    tiles = (T // 256) * (N // 128); slots = (tiles / 256 + 1) * (K // 32)
    print(f"{name:24s} " + "  ".join(f"{n[3:]}:{statistics.median(v):.0f}us({statistics.median(v) * 1e-6 / slots * 1.95e9:.0f}cyc/slot)" for n, v in res.items()), flush=True)
