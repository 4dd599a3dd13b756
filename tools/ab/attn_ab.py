#!/usr/bin/env python3
"""Development tool (GPU): in-process A/B timing of attention kernel variants built by build_attn_variants.sh at the cfg-2 shape."""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
here = os.path.dirname(os.path.abspath(__file__))
B, H, L, hd = 256, 20, 512, int(os.environ.get("HD", "32"))
variants = sys.argv[1:] or ["0"]
g = torch.Generator(device="cuda").manual_seed(0)
q = (torch.randn(B, H, L, hd, device="cuda", generator=g) * hd ** -0.5).to(torch.bfloat16)
q2 = (q.float() * 1.4426950408889634).to(torch.bfloat16)      # variants named mf* (and the product kernels since round 1): q pre-multiplied by log2(e)
k = torch.randn(B, H, L, hd, device="cuda", generator=g).to(torch.bfloat16)
v = torch.randn(B, H, L, hd, device="cuda", generator=g).to(torch.bfloat16)
bias = torch.zeros(B, L, device="cuda")
ctx = torch.empty(B * L, H * hd, dtype=torch.bfloat16, device="cuda")
dctx = torch.randn(B * L, H * hd, device="cuda", generator=g).to(torch.bfloat16)
lse = torch.empty(B, H, L, device="cuda")
dqkv = torch.empty(B * L, 3 * H * hd, dtype=torch.bfloat16, device="cuda")
cos = torch.rand(L, hd // 2, device="cuda"); sin = torch.rand(L, hd // 2, device="cuda")
P = ctypes.c_void_p
libs = {}
for n in variants:
    lib = ctypes.CDLL(os.path.join(here, f"libattn_v{n}.so"))
    lib.oneprot_attn_fwd.argtypes = [P, P, P, P, P, P] + [ctypes.c_int] * 4 + [P]
    lib.oneprot_attn_bwd.argtypes = [P] * 9 + [ctypes.c_float, P, P] + [ctypes.c_int] * 4 + [P]
    lib.oneprot_attn_bwd_workspace.restype = ctypes.c_size_t
    lib.oneprot_attn_bwd_workspace.argtypes = [ctypes.c_int] * 3
    libs[n] = lib
ws = torch.empty(libs[variants[0]].oneprot_attn_bwd_workspace(B, H, L), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def fwd(lib, qq=None): 
    rc = lib.oneprot_attn_fwd((qq if qq is not None else q).data_ptr(), k.data_ptr(), v.data_ptr(), bias.data_ptr(), ctx.data_ptr(), lse.data_ptr(), B, H, L, hd, st); assert rc == 0, rc
def bwd(lib, qq=None):
    rc = lib.oneprot_attn_bwd((qq if qq is not None else q).data_ptr(), k.data_ptr(), v.data_ptr(), bias.data_ptr(), ctx.data_ptr(), dctx.data_ptr(), lse.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                              hd ** -0.5, dqkv.data_ptr(), ws.data_ptr(), B, H, L, hd, st); assert rc == 0, rc
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
what = os.environ.get("WHAT", "fwd,bwd").split(",")
ref = None
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    if name not in what: continue
    res = {n: [] for n in variants}
    for rep in range(3):
        for n in variants:
            res[n].append(timeit((lambda: fn(libs[n], q2)) if n.startswith("mf") else (lambda: fn(libs[n]))))
    print(name, "  ".join(f"v{n}:{statistics.median(t):.0f}us" for n, t in res.items()), flush=True)
# correctness of each variant's forward vs variant 0 (ablations differ by construction)
fwd(libs[variants[0]], q2 if variants[0].startswith("mf") else None); c0 = ctx.float().clone(); l0 = lse.clone()
for n in variants[1:]:
    fwd(libs[n], q2 if n.startswith("mf") else None); print(f"v{n} ctx max diff vs v{variants[0]}:", float((ctx.float() - c0).abs().max()), "lse diff", float((lse - l0).abs().max()))
