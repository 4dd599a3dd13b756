#!/usr/bin/env python3
"""Development tool (GPU): do the weight-gradient GEMM (k_gemm_tn) and the data-gradient GEMM (k_gemm_nt) of one Linear overlap when they are
issued on two HIP streams?  Both read dY; neither depends on the other.  Prints sequential vs concurrent time for the FFN / QKV shapes."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oneprot_amd import hip
T, d, f = 131072, 640, 2560
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
side = torch.cuda.Stream()
def bench(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name, N, K, shapes in (("ffn2 (dY[T,640], X=u[T,2560])", d, f, (-1, 3, 1, 0)), ("ffn1 (dY[T,2560], X=h[T,640])", f, d, (-1, 3, 1, 0)), ("qkv (dY[T,1920], X=h[T,640])", 3 * d, d, (-1, 3, 1, 0))):
    dY = rnd(T, N).to(torch.bfloat16); X = rnd(T, K).to(torch.bfloat16)
    Wt = (rnd(K, N) * 0.05).to(torch.bfloat16)            # dgrad: dX[T,K] = dY[T,N] * Wt[K,N]^T
    dW = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
    ws = torch.empty(hip.query("oneprot_gemm_bf16_tn_workspace", N, K), dtype=torch.uint8, device="cuda")
    dX = torch.empty(T, K, dtype=torch.bfloat16, device="cuda")
    wgrad = lambda: hip.call("oneprot_gemm_bf16_tn", dY, X, T, N, K, N, K, dW, db, ws, ws.numel(), 0)
    dgrad = lambda: hip.call("oneprot_gemm_bf16_nt", dY, Wt, T, K, N, N, N, hip.EPI_BF16, None, dX, None, None, None, None, None, 1.0, 0, 0, 0)
    for shape in shapes:
        hip.query("oneprot_gemm_force_shape", shape)
        def seq():
            wgrad(); dgrad()
        def par():
            ev = torch.cuda.Event(); ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                wgrad()
                done = torch.cuda.Event(); done.record()
            dgrad()
            torch.cuda.current_stream().wait_event(done)
        r = {k: [] for k in ("w", "d", "seq", "par")}
        for rep in range(3):
            r["w"].append(bench(wgrad)); r["d"].append(bench(dgrad)); r["seq"].append(bench(seq)); r["par"].append(bench(par))
        m = {k: statistics.median(v) for k, v in r.items()}
        print(f"{name:34s} nt shape {shape:2d}: wgrad {m['w']:.3f}  dgrad {m['d']:.3f}  sequential {m['seq']:.3f}  two streams {m['par']:.3f} ms", flush=True)
hip.query("oneprot_gemm_force_shape", -1)
