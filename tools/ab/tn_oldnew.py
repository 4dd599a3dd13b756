#!/usr/bin/env python3
"""Development tool (GPU): in-process A/B of a saved copy of gemm_tn.hip (libattn_vtn_old.so) against the product library on the training shapes."""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oneprot_amd import hip
here = os.path.dirname(os.path.abspath(__file__))
P, I, L64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
old = ctypes.CDLL(os.path.join(here, "libattn_vtn_old.so")); new = hip.lib()
for lib in (old, new):
    lib.oneprot_gemm_bf16_tn.argtypes = [P, P, L64, I, I, I, I, P, P, P, I, P]; lib.oneprot_gemm_bf16_tn.restype = I
    lib.oneprot_gemm_bf16_tn_workspace.argtypes = [I, I]; lib.oneprot_gemm_bf16_tn_workspace.restype = ctypes.c_size_t
T, d, f = 131072, 640, 2560
g = torch.Generator(device="cuda").manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, N, K in (("qkv  dW[1920,640]", 3 * d, d), ("out  dW[640,640]", d, d), ("ffn1 dW[2560,640]", f, d), ("ffn2 dW[640,2560]", d, f)):
    dY = torch.randn(T, N, device="cuda", generator=g).to(torch.bfloat16); X = torch.randn(T, K, device="cuda", generator=g).to(torch.bfloat16)
    dW, db = torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda")
    ws = torch.empty(max(old.oneprot_gemm_bf16_tn_workspace(N, K), new.oneprot_gemm_bf16_tn_workspace(N, K)), dtype=torch.uint8, device="cuda")
    res, outs = {"old": [], "new": []}, {}
    for rep in range(3):
        for n, lib in (("old", old), ("new", new)):
            fn = lambda: lib.oneprot_gemm_bf16_tn(dY.data_ptr(), X.data_ptr(), T, N, K, N, K, dW.data_ptr(), db.data_ptr(), ws.data_ptr(), 0, st)
            res[n].append(timeit(fn)); outs[n] = (dW.clone(), db.clone())
    fl = 2.0 * T * N * K
    print(f"{name:18s} " + "  ".join(f"{n}:{statistics.median(v):.0f}us({fl / statistics.median(v) / 1e6:.0f}TF)" for n, v in res.items()) +
          f"  identical: {bool(torch.equal(outs['old'][0], outs['new'][0]) and torch.equal(outs['old'][1], outs['new'][1]))}", flush=True)
