#!/usr/bin/env python3
"""Development tool (GPU): two persistent 8-phase GEMMs side by side, each on HALF the CUs (work-group cap 16 per XCD), on two streams, against the
same two launches back to back on the whole chip.  Question: do the output bursts of one hide under the K loops of the other when the two kernels
are not in lock step?  usage: split_chip_ab.py"""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oneprot_amd import hip
T, d, f = 131072, 640, 2560
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
_dph = hip.lib().oneprot_gemm8_dephase; _dph.argtypes = [ctypes.c_int, ctypes.c_int]; _dph.restype = None
def mk(N, K, epi, dual=False, skew=0):
    A = rnd(T + skew, K).to(torch.bfloat16)[skew:]; W = (rnd(N, K) * 0.05).to(torch.bfloat16); bias = rnd(N)
    o0 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda"); o1 = torch.empty_like(o0) if dual else None
    return lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias if epi == hip.EPI_BIAS_GELU else None, o0, o1, None, None, None, None, 1.0, 0, 0, 0)
cases = {"ffn1 gelu+gelu' || same": (mk(f, d, hip.EPI_BIAS_GELU, True), mk(f, d, hip.EPI_BIAS_GELU, True)),
         "ffn1 gelu || qkv-sized plain N1920": (mk(f, d, hip.EPI_BIAS_GELU), mk(3 * d, d, hip.EPI_BF16)),
         "ffn1_dgrad K2560 plain || ffn1 gelu+gelu'": (mk(d, f, hip.EPI_BF16), mk(f, d, hip.EPI_BIAS_GELU, True))}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn, iters=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name, (a, b) in cases.items():
    def serial():
        _dph(1, 0); a(); b()
    def split():
        _dph((16 << 16) | 1, 0)
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1): a()
        with torch.cuda.stream(s2): b()
        cur.wait_stream(s1); cur.wait_stream(s2)
    def half_alone():
        _dph((16 << 16) | 1, 0); a(); b()
    res = {"serial": [], "split": [], "half": []}
    for r in range(3):
        res["serial"].append(timed(serial)); res["split"].append(timed(split)); res["half"].append(timed(half_alone))
    m = {k: statistics.median(v) for k, v in res.items()}
    print(f"{name}: whole chip back to back {m['serial']:.3f} ms | half chip each, two streams {m['split']:.3f} ms | half chip each, back to back {m['half']:.3f} ms", flush=True)
_dph(1, 0)
