#!/usr/bin/env python3
"""Development tool (GPU): context / LSE error of the three forward attention kernels against fp32 torch on the same bf16 inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oneprot_amd import hip
def run(B, H, L, hd, scale, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    q = (torch.randn(B, H, L, hd, device="cuda", generator=g) * scale * hip.LOG2E).to(torch.bfloat16)
    k = torch.randn(B, H, L, hd, device="cuda", generator=g).to(torch.bfloat16)
    v = torch.randn(B, H, L, hd, device="cuda", generator=g).to(torch.bfloat16)
    lens = torch.randint(max(1, L // 3), L + 1, (B,), device="cuda", generator=g)
    bias = torch.where(torch.arange(L, device="cuda")[None, :] < lens[:, None], 0.0, float("-inf")).float().contiguous()
    s = (q.double() / hip.LOG2E) @ k.double().transpose(-1, -2) + bias[:, None, None, :].double()
    p = torch.softmax(s, -1)
    ref = (p @ v.double()).permute(0, 2, 1, 3).reshape(B * L, H * hd)
    lse_ref = torch.logsumexp(s, -1)
    out = []
    for path in (0, 1, 2):
        hip.query("oneprot_attn_force_fwd_path", path)
        ctx = torch.empty(B * L, H * hd, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, L, device="cuda")
        hip.call("oneprot_attn_fwd", q, k, v, bias, ctx, lse, B, H, L, hd)
        e = (ctx.double() - ref)
        out.append(f"path {path}: ctx rms err {e.pow(2).mean().sqrt().item():.3e} max {e.abs().max().item():.3e} | lse rms {(lse.double() - lse_ref).pow(2).mean().sqrt().item():.3e}")
    hip.query("oneprot_attn_force_fwd_path", -1)
    # the error floor: the exact result rounded to bf16
    fl = (ref.to(torch.bfloat16).double() - ref).pow(2).mean().sqrt().item()
    print(f"B={B} H={H} L={L} hd={hd} q-scale {scale}: output rounding floor {fl:.3e}\n  " + "\n  ".join(out))
for args in [(3, 20, 24, 64, 0.125), (4, 2, 37, 32, 0.18), (8, 20, 512, 32, 0.18), (8, 20, 512, 32, 0.7), (8, 20, 128, 16, 0.25), (2, 20, 512, 64, 0.125)]:
    run(*args)
