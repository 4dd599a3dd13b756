#!/usr/bin/env python3
"""Development tool (GPU): event-timed launches of the two per-step row kernels outside the layer loop at the cfg-2 shape: the embedding backward
(k_embed_bwd_small) and the final LayerNorm + pooling forward (k_lnpool_fwd).  env G8_LIB=<other liboneprot_hip.so> for an A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oneprot_amd import hip
if os.environ.get("G8_LIB"): hip.LIB_PATH = os.path.abspath(os.environ["G8_LIB"])
B, L, d, V = 256, 512, 640, 33
g = torch.Generator(device="cuda").manual_seed(0)
ids = torch.randint(4, 24, (B, L), device="cuda", generator=g)
x = torch.randn(B * L, d, device="cuda", generator=g)
rs = torch.ones(B, device="cuda")
dW = torch.empty(V, d, device="cuda")
ws = torch.empty(hip.query("oneprot_esm_embed_bwd_workspace", B * L, d, V), dtype=torch.uint8, device="cuda")
gamma, beta = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
pooled, mean, rstd, wrow = torch.empty(B, d, device="cuda"), torch.empty(B * L, device="cuda"), torch.empty(B * L, device="cuda"), torch.empty(B * L, device="cuda")
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print(f"embed bwd {timeit(lambda: hip.call('oneprot_esm_embed_bwd', ids, x, rs, dW, ws, B, L, d, V, 1, 32, 1, 0)):.1f} us   "
      f"lnpool fwd {timeit(lambda: hip.call('oneprot_lnpool_fwd', x, ids, 1, gamma, beta, pooled, mean, rstd, wrow, None, None, B, L, d, 1e-5, 0)):.1f} us")

# cfg-5 anchor head (ESM-2-650M width, attention1d pooling): final LayerNorm that also writes the normalised hidden state, then the pooling kernels
B, L, d = 128, 512, 1280
ids = torch.randint(4, 24, (B, L), device="cuda", generator=g)
x = torch.randn(B * L, d, device="cuda", generator=g)
hidden = torch.empty(B, L, d, device="cuda")
gamma, beta = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
pooled, mean, rstd, wrow, attn = torch.empty(B, d, device="cuda"), torch.empty(B * L, device="cuda"), torch.empty(B * L, device="cuda"), torch.empty(B * L, device="cuda"), torch.empty(B, L, device="cuda")
w, bias, dp = torch.randn(d, device="cuda", generator=g) * 0.05, torch.zeros(1, device="cuda"), torch.randn(B, d, device="cuda", generator=g)
dw, db = torch.empty(d, device="cuda"), torch.empty(1, device="cuda")
ws = torch.empty(hip.query("oneprot_attnpool_bwd_workspace", B, d), dtype=torch.uint8, device="cuda")
print(f"128 x 512 x 1280: lnpool fwd (+hidden) {timeit(lambda: hip.call('oneprot_lnpool_fwd', x, ids, 1, gamma, beta, pooled, mean, rstd, wrow, None, hidden, B, L, d, 1e-5, 0)):.1f} us   "
      f"attnpool fwd {timeit(lambda: hip.call('oneprot_attnpool_fwd', hidden, ids, 1, w, bias, pooled, attn, B, L, d)):.1f} us   "
      f"attnpool bwd (no dx) {timeit(lambda: hip.call('oneprot_attnpool_bwd', hidden, attn, w, dp, dw, db, None, ws, B, L, d)):.1f} us   "
      f"(with dx) {timeit(lambda: hip.call('oneprot_attnpool_bwd', hidden, attn, w, dp, dw, db, hidden.new_empty(B, L, d), ws, B, L, d)):.1f} us")
