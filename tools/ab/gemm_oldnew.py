#!/usr/bin/env python3
"""Development tool (GPU): in-process A/B of a saved copy of gemm_nt.hip (libattn_vg_old.so) against the product library on the training shapes."""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oneprot_amd import hip
here = os.path.dirname(os.path.abspath(__file__))
P, I, F, L64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64
old = ctypes.CDLL(os.path.join(here, "libattn_vg_old.so"))
new = hip.lib()
for lib in (old, new):
    lib.oneprot_gemm_bf16_nt.argtypes = [P, P, L64, I, I, I, I, I, P, P, P, P, P, P, P, F, I, I, I, P]
    lib.oneprot_gemm_bf16_nt.restype = I
T, d, f = 131072, 640, 2560
g = torch.Generator(device="cuda").manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
ptr = lambda t: t.data_ptr() if t is not None else None
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B_, L_, H_, hd_ = 256, 512, 20, 32
cos = torch.rand(L_, hd_ // 2, device="cuda"); sin = torch.rand(L_, hd_ // 2, device="cuda")
for (name, N, K, epi) in (("qkv rope", 3 * d, d, 4), ("ffn1 gelu", f, d, 2), ("ffn1 bf16", f, d, 0), ("ffn2 resid", d, f, 3), ("ffn1_dgrad bf16", d, f, 0), ("out resid", d, d, 3), ("qkv_dgrad bf16", d, 3 * d, 0)):
    A = torch.randn(T, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    o0 = torch.empty(T, N, dtype=torch.float32 if epi == 3 else torch.bfloat16, device="cuda")
    o1 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda") if epi == 2 else None
    if epi == 4:
        o0, o1, o2 = (torch.empty(B_, H_, L_, hd_, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    aux = torch.randn(T, N, device="cuda", generator=g) if epi == 3 else None
    res = {"old": [], "new": []}
    outs = {}
    for rep in range(3):
        for n, lib in (("old", old), ("new", new)):
            if epi == 4:
                fn = lambda: lib.oneprot_gemm_bf16_nt(ptr(A), ptr(W), T, N, K, K, K, epi, ptr(bias), ptr(o0), ptr(o1), ptr(o2), None, ptr(cos), ptr(sin), hd_ ** -0.5, L_, H_, hd_, st)
            else:
                fn = lambda: lib.oneprot_gemm_bf16_nt(ptr(A), ptr(W), T, N, K, K, K, epi, ptr(bias), ptr(o0), ptr(o1), None, ptr(aux), None, None, 1.0, 0, 0, 0, st)
            res[n].append(timeit(fn))
            if epi != 3: outs[n] = o0.clone()
    fl = 2.0 * T * N * K
    same = "" if epi == 3 else f"  identical: {bool(torch.equal(outs['old'], outs['new']))}"
    print(f"{name:16s} " + "  ".join(f"{n}:{statistics.median(v):.0f}us({fl / statistics.median(v) / 1e6:.0f}TF)" for n, v in res.items()) + same, flush=True)
