#!/usr/bin/env python3
"""Development tool (GPU): phase lengths of k_attn_fwd2 at the cfg-2 shape from s_memtime stamps (library built with -DFWD2_STAMP by
build_attn_variants.sh, e.g. `stamp=attn_cur.hip,-DFWD2_STAMP`).  Stamps: 0 kernel entry (q fragments requested), 1 after the opening barrier,
2 K/V committed to LDS, 3 bias entries written, 4 after the second barrier, 5 tile loop done, 6 fallback check done, 7 outputs stored."""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
here = os.path.dirname(os.path.abspath(__file__))
B, H, L, hd = 256, 20, 512, 32
g = torch.Generator(device="cuda").manual_seed(0)
q = (torch.randn(B, H, L, hd, device="cuda", generator=g) * hd ** -0.5 * 1.4426950408889634).to(torch.bfloat16)
k = torch.randn(B, H, L, hd, device="cuda", generator=g).to(torch.bfloat16)
v = torch.randn(B, H, L, hd, device="cuda", generator=g).to(torch.bfloat16)
bias = torch.zeros(B, L, device="cuda")
ctx = torch.empty(B * L, H * hd, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, L, device="cuda")
P = ctypes.c_void_p
lib = ctypes.CDLL(os.path.join(here, f"libattn_v{sys.argv[1] if len(sys.argv) > 1 else 'stamp'}.so"))
lib.oneprot_attn_fwd.argtypes = [P, P, P, P, P, P] + [ctypes.c_int] * 4 + [P]
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    assert lib.oneprot_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), bias.data_ptr(), ctx.data_ptr(), lse.data_ptr(), B, H, L, hd, st) == 0
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 16 * 8))()
assert lib.oneprot_attn_debug_stamps(buf) == 0
names = ["entry->barrier1", "stage K/V (load+commit)", "bias entries", "wait barrier2", "tile loop", "fallback check", "store"]
if os.environ.get("FWD3"):      # persistent kernel, 4th slab of every 4th work-group: 0 loop top, 1 own DMA landed, 2 barrier passed, 3 next slab's DMA / stores / q issued, 4 tiles done, 5 outputs packed
    names = ["wait own DMA (vmcnt 0)", "barrier", "issue next DMA + stores + q", "classes + tile loop", "fallback check + pack", "-", "-"]
if os.environ.get("FWD3"):
    for w in range(16):
        rows = []
        for i in range(64):
            t = [buf[(i * 16 + w) * 8 + j] for j in range(8)]
            if t[0] == 0 or t[5] <= t[0]: continue
            rows.append([t[j + 1] - t[j] for j in range(5)])
        med = lambda j: sorted(r[j] for r in rows)[len(rows) // 2]
        print(f"wave {w:2d}: " + "  ".join(f"{n.split(' ')[0]} {med(j):6d}" for j, n in enumerate(names[:5])))
    sys.exit(0)
for w, wn in ((0, "wave 0"), (1, "wave 5")):
    rows = []
    for i in range(64):
        t = [buf[(i * 2 + w) * 8 + j] for j in range(8)]
        last = 5 if os.environ.get("FWD3") else 7
        if t[0] == 0 or t[last] <= t[0]: continue
        rows.append([max(0, t[j + 1] - t[j]) if j < last else 0 for j in range(7)] + [t[last] - t[0]])
    print(f"{wn}: {len(rows)} work-groups sampled; median cycles per phase")
    for j, n in enumerate(names + ["TOTAL"]):
        col = sorted(r[j] for r in rows)
        print(f"  {n:28s} median {col[len(col) // 2]:8d}   p10 {col[len(col) // 10]:8d}   p90 {col[len(col) * 9 // 10]:8d}")
