#!/usr/bin/env python3
"""Runs a handful of attention forward + backward launches at the cfg-2 shape (for rocprofv3 passes) and prints their event-timed durations.
usage: attn_only.py [iters] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oneprot_amd import hip
if os.environ.get("G8_LIB"): hip.LIB_PATH = os.path.abspath(os.environ["G8_LIB"])
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
L, H, hd = int(os.environ.get("ATTN_L", "512")), 20, 32
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda: (torch.randn(B, H, L, hd, device="cuda", generator=g) * 0.7).to(torch.bfloat16)
q, k, v = mk(), mk(), mk()
lens = torch.randint(L // 2, L + 1, (B,), device="cuda", generator=g)
key_bias = torch.where(torch.arange(L, device="cuda")[None, :] < lens[:, None], 0.0, float("-inf")).float().contiguous()
ctx = torch.empty(B * L, H * hd, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, L, device="cuda")
dctx = (torch.randn(B * L, H * hd, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
dqkv = torch.empty(B * L, 3 * H * hd, dtype=torch.bfloat16, device="cuda")
ws = torch.empty(hip.query("oneprot_attn_bwd_workspace", B, H, L), dtype=torch.uint8, device="cuda")
cos = torch.rand(L, hd // 2, device="cuda"); sin = torch.rand(L, hd // 2, device="cuda")
fwd = lambda: hip.call("oneprot_attn_fwd", q, k, v, key_bias, ctx, lse, B, H, L, hd)
bwd = lambda: hip.call("oneprot_attn_bwd", q, k, v, key_bias, ctx, dctx, lse, cos, sin, hd ** -0.5, dqkv, ws, B, H, L, hd)
def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
tf, tb = timeit(fwd), timeit(bwd)
fl = 4.0 * B * H * L * L * hd
print(f"attn fwd {tf * 1e3:.1f} us ({fl / tf / 1e9:.0f} TFLOP/s)   bwd {tb * 1e3:.1f} us ({2.5 * fl / tb / 1e9:.0f} TFLOP/s)")
if os.environ.get("ATTN_FWD_AB"):
    # both forward kernels in one process (box-to-box spread is larger than the difference between variants), with and without key padding
    nob = torch.zeros_like(key_bias)
    for name, kb in (("ragged padding (lens L/2..L)", key_bias), ("no padding", nob)):
        f2 = lambda: hip.call("oneprot_attn_fwd", q, k, v, kb, ctx, lse, B, H, L, hd)
        res = {0: [], 1: [], 2: []}
        for rep in range(3):
            for path in (0, 1, 2):
                hip.query("oneprot_attn_force_fwd_path", path)
                res[path].append(timeit(f2))
        hip.query("oneprot_attn_force_fwd_path", -1)
        a, b_, c = sorted(res[0])[1], sorted(res[1])[1], sorted(res[2])[1]
        print(f"fwd, {name}: row-max kernel {a * 1e3:.1f} us, no-max persistent LDS-DMA kernel {b_ * 1e3:.1f} us ({fl / b_ / 1e9:.0f} TFLOP/s), no-max chunked kernel {c * 1e3:.1f} us")
dphs = [int(x) for x in os.environ.get("ATTN_DPH", "").split(",") if x]
if dphs:
    # (ablation build) start offset of the second wave of each SIMD in the 8-wave fused backward, in units of 64 cycles; interleaved rounds
    hip.query("oneprot_attn_force_bwd_path", 2)
    rr = {d_: [] for d_ in dphs}
    for rep in range(5):
        for d_ in dphs:
            hip.lib().oneprot_attn_debug_ablate(d_)
            rr[d_].append(timeit(bwd))
    hip.lib().oneprot_attn_debug_ablate(0)
    print("fused 8-wave bwd, second wave of a SIMD d x 64 cycles late (median of 5 interleaved rounds): " + "  ".join(f"d={d_}: {sorted(v)[2] * 1e3:.1f} us" for d_, v in rr.items()))
for abl in [int(x) for x in os.environ.get("ATTN_ABL", "").split(",") if x]:
    # needs a library built with -DONEPROT_ATTN_ABLATE (the hook is not part of the shipped C-ABI: the ablated kernel computes wrong results)
    hip.query("oneprot_attn_force_bwd_path", 1); hip.lib().oneprot_attn_debug_ablate(abl)
    print(f"fused bwd, ablation mask {abl}: {timeit(bwd) * 1e3:.1f} us")
res = {0: [], 1: [], 2: []}
for rep in range(3):
    for path in (0, 1, 2):
        hip.query("oneprot_attn_force_bwd_path", path)
        res[path].append(timeit(bwd))
hip.query("oneprot_attn_force_bwd_path", -1)
med = {p_: sorted(v)[1] * 1e3 for p_, v in res.items()}
print(f"bwd: split kernels {med[0]:.1f} us, fused 16-wave kernel {med[1]:.1f} us ({2.5 * fl / (med[1] * 1e-3) / 1e9:.0f} TFLOP/s), fused 8-wave kernel (64 keys per wave) {med[2]:.1f} us ({2.5 * fl / (med[2] * 1e-3) / 1e9:.0f} TFLOP/s)")
