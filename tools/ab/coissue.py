#!/usr/bin/env python3
"""Development probe (GPU): run the FFN-1 GEMM (plain bf16 epilogue) and a VALU-only kernel alone and concurrently on two streams.
If VALU instructions of one wave issue under the MFMA stream of co-resident waves, t(both) ~ max(t_gemm, t_valu); if they serialise, ~ sum."""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oneprot_amd import hip
here = os.path.dirname(os.path.abspath(__file__))
spin = ctypes.CDLL(os.path.join(here, "libattn_vspin.so"))
spin.valu_spin.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
T, d, f = 131072, 640, 2560
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(T, d, device="cuda", generator=g).to(torch.bfloat16); W = (torch.randn(f, d, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
o0 = torch.empty(T, f, dtype=torch.bfloat16, device="cuda"); dummy = torch.zeros(16, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def gemm(): 
    with torch.cuda.stream(s1): hip.call("oneprot_gemm_bf16_nt", A, W, T, f, d, d, d, hip.EPI_BF16, None, o0, None, None, None, None, None, 1.0, 0, 0, 0)
def valu(blocks, iters):
    spin.valu_spin(dummy.data_ptr(), blocks, iters, s2.cuda_stream)
def wall(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        s1.wait_event(e0); s2.wait_event(e0)
        fn()
        e1s1, e1s2 = torch.cuda.Event(), torch.cuda.Event()
        e1s1.record(s1); e1s2.record(s2)
        torch.cuda.current_stream().wait_event(e1s1); torch.cuda.current_stream().wait_event(e1s2); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
for shape in (4, 1):
    hip.query("oneprot_gemm_force_shape", shape)
    tg = wall(gemm)
    for blocks in (256, 1024):           # one / four VALU waves per SIMD
        iters = 1
        while wall(lambda: valu(blocks, iters)) < 0.8 * tg: iters *= 2
        tv = wall(lambda: valu(blocks, iters))
        tb = wall(lambda: (valu(blocks, iters), gemm()))
        print(f"shape {shape}: gemm alone {tg:.0f} us, valu alone ({blocks} blocks x {iters} iters) {tv:.0f} us, both {tb:.0f} us  (max {max(tg, tv):.0f}, sum {tg + tv:.0f})", flush=True)
hip.query("oneprot_gemm_force_shape", -1)
