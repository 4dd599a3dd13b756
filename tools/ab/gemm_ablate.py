#!/usr/bin/env python3
"""Development tool (GPU): where does an NT-GEMM launch spend its time?  Same kernel built three ways (build_attn_variants.sh
g_base / g_small (every store redirected into the first 2 MB of the output: store instructions stay, HBM write traffic goes) /
g_nostore (epilogue runs, global stores never execute) / g_noepi (main loop only)) on the FFN-1 and FFN-2 shapes."""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
here = os.path.dirname(os.path.abspath(__file__))
T, d, f = 131072, 640, 2560
P, I, F, L64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64
libs = {}
for n in ("g_base", "g_small", "g_nostore", "g_noepi"):
    lib = ctypes.CDLL(os.path.join(here, f"libattn_v{n}.so"))
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
for (name, N, K, epi) in (("ffn1 gelu", f, d, 2), ("ffn1 bf16", f, d, 0), ("ffn2 resid", d, f, 3), ("ffn2 bf16", d, f, 0)):
    A = torch.randn(T, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    o0 = torch.empty(T, N, dtype=torch.float32 if epi == 3 else torch.bfloat16, device="cuda")
    o1 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda") if epi == 2 else None
    aux = torch.randn(T, N, device="cuda", generator=g) if epi == 3 else None
    for shape in (1, 3, 4):
        res = {}
        for rep in range(3):
            for n, lib in libs.items():
                lib.oneprot_gemm_force_shape(shape)
                fn = lambda: lib.oneprot_gemm_bf16_nt(ptr(A), ptr(W), T, N, K, K, K, epi, ptr(bias), ptr(o0), ptr(o1), None, ptr(aux), None, None, 1.0, 0, 0, 0, st)
                res.setdefault(n, []).append(timeit(fn))
        fl = 2.0 * T * N * K
        print(f"{name:11s} shape {shape}: " + "  ".join(f"{n[2:]}:{statistics.median(v):.0f}us({fl / statistics.median(v) / 1e6:.0f}TF)" for n, v in res.items()), flush=True)
