#!/usr/bin/env python3
"""Development tool (GPU): does putting co-resident workgroups out of phase hide the epilogue?  (gemm_exp.hip stagger hook)"""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
here = os.path.dirname(os.path.abspath(__file__))
T, d, f = 131072, 640, 2560
P, I, F, L64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64
lib = ctypes.CDLL(os.path.join(here, "libattn_vg_base.so"))
lib.oneprot_gemm_bf16_nt.argtypes = [P, P, L64, I, I, I, I, I, P, P, P, P, P, P, P, F, I, I, I, P]
lib.oneprot_gemm_force_shape.argtypes = [I]
lib.oneprot_gemm_set_stagger.argtypes = [I, I, I]
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
N, K, epi = f, d, 2
A = torch.randn(T, K, device="cuda", generator=g).to(torch.bfloat16)
W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
bias = torch.randn(N, device="cuda", generator=g)
o0 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda"); o1 = torch.empty_like(o0)
fn = lambda: lib.oneprot_gemm_bf16_nt(ptr(A), ptr(W), T, N, K, K, K, epi, ptr(bias), ptr(o0), ptr(o1), None, None, None, None, 1.0, 0, 0, 0, st)
for shape, first in ((1, 512), (3, 512), (0, 768)):
    lib.oneprot_gemm_force_shape(shape)
    out = []
    for mode in (0, 1, 2, 3):
        for sleeps in ((0,) if mode == 0 else (1, 2, 4, 8)):
            lib.oneprot_gemm_set_stagger(mode, first, sleeps)
            t = statistics.median(timeit(fn) for _ in range(3))
            out.append(f"m{mode}s{sleeps}:{t:.0f}")
    print(f"ffn1 gelu shape {shape}: " + " ".join(out), flush=True)
