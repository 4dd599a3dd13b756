#!/usr/bin/env python3
"""Development tool (GPU): stage lengths of the 8-wave fused attention backward (k_attn_bwd_fused64) at the cfg-2 shape, from s_memtime stamps of
the eight waves of one work-group (library built with -DBWD64_STAMP into tools/ab/libattn_bwd64stamp.so).  Stamps per step (= query block, two
tiles): 0 top, 1 fragment reads + score / dP chains issued, 2 the PREVIOUS step's dQ MFMAs issued + its ticketed read-add-write done, 3 both key
blocks: exponentials, packs, dV / dK MFMAs, dS^T written.  Every stamp drains lgkmcnt itself."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
here = os.path.dirname(os.path.abspath(__file__))
B, H, L, hd = 256, 20, 512, 32
g = torch.Generator(device="cuda").manual_seed(0)
q = (torch.randn(B, H, L, hd, device="cuda", generator=g) * hd ** -0.5 * 1.4426950408889634).to(torch.bfloat16)
k = torch.randn(B, H, L, hd, device="cuda", generator=g).to(torch.bfloat16)
v = torch.randn(B, H, L, hd, device="cuda", generator=g).to(torch.bfloat16)
ctx = torch.randn(B * L, H * hd, device="cuda", generator=g).to(torch.bfloat16); dctx = torch.randn(B * L, H * hd, device="cuda", generator=g).to(torch.bfloat16)
lse = torch.randn(B, H, L, device="cuda", generator=g) + 6.0
dqkv = torch.empty(B * L, 3 * H * hd, dtype=torch.bfloat16, device="cuda"); wsp = torch.empty(B * H * L, device="cuda")
P, I = ctypes.c_void_p, ctypes.c_int
lib = ctypes.CDLL(os.path.join(here, "libattn_bwd64stamp.so"))
lib.oneprot_attn_bwd.argtypes = [P] * 9 + [ctypes.c_float, P, P, I, I, I, I, P]
lib.oneprot_attn_force_bwd_path.argtypes = [I]
lib.oneprot_attn_force_bwd_path(2)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    assert lib.oneprot_attn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), None, ctx.data_ptr(), dctx.data_ptr(), lse.data_ptr(), None, None, hd ** -0.5,
                                dqkv.data_ptr(), wsp.data_ptr(), B, H, L, hd, st) == 0
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 16 * 8))()
assert lib.oneprot_attn_debug_bwd64_stamps(buf) == 0
names = ["frag reads + S/dP chains", "deferred dQ (MFMAs, ticket, read-add-write)", "kb0 + kb1: exp .. dS^T", "-> next top"]
print("cycles per stage, median over the 16 steps of each wave:  " + " | ".join(names))
for w in range(8):
    rows = []
    for stp in range(16):
        t = [buf[(w * 16 + stp) * 8 + j] for j in range(4)]
        nxt = buf[(w * 16 + stp + 1) * 8] if stp < 15 else t[3]
        rows.append([t[j + 1] - t[j] for j in range(3)] + [nxt - t[3]])
    med = [sorted(r[j] for r in rows)[8] for j in range(4)]
    print(f"wave {w}: " + " ".join(f"{m:6d}" for m in med) + f"   step total {sum(med):6d}   walk {buf[(w * 16 + 15) * 8 + 3] - buf[(w * 16) * 8]:7d}")
