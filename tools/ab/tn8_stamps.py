#!/usr/bin/env python3
"""Development tool (GPU): slot lengths of the 8-phase weight-gradient GEMM.  Needs tools/ab/libtn8_stamp.so:
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DTN8_STAMP -Ioneprot_amd/csrc -shared oneprot_amd/csrc/gemm_tn.hip -o tools/ab/libtn8_stamp.so
usage: tn8_stamps.py [N] [K]"""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libtn8_stamp.so"))
P, I, L64, SZ = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t
lib.oneprot_gemm_bf16_tn.argtypes = [P, P, L64, I, I, I, I, P, P, P, SZ, I, P]; lib.oneprot_gemm_bf16_tn.restype = I
lib.oneprot_gemm_bf16_tn_workspace.argtypes = [I, I]; lib.oneprot_gemm_bf16_tn_workspace.restype = SZ
lib.oneprot_gemm_tn_variant.argtypes = [I]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
K = int(sys.argv[2]) if len(sys.argv) > 2 else 640
T = 131072
g = torch.Generator(device="cuda").manual_seed(0)
dY = torch.randn(T, N, device="cuda", generator=g).to(torch.bfloat16); X = torch.randn(T, K, device="cuda", generator=g).to(torch.bfloat16)
dW, db = torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda")
ws = torch.empty(lib.oneprot_gemm_bf16_tn_workspace(N, K), dtype=torch.uint8, device="cuda")
lib.oneprot_gemm_tn_variant(3)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    assert lib.oneprot_gemm_bf16_tn(dY.data_ptr(), X.data_ptr(), T, N, K, N, K, dW.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), 0, st) == 0
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 1024)()
assert lib.oneprot_tn8_debug_read(buf) == 0
for grp in range(2):
    print(f"group {grp}: cycles per 32-token unit, in time order [gathers issued | gathers landed (lgkmcnt 0) | unit p+1 landed (vmcnt) | opening barrier | 40 MFMAs + 5 LDS-DMA issued | bias MFMAs | closing barrier | SUM]")
    prev = None
    for p in range(8, 20):
        r = [buf[(grp * 64 + p) * 8 + i] for i in range(8)]
        if prev is not None:
            d = [r[2] - prev, r[4] - r[2], r[5] - r[4], r[0] - r[5], r[3] - r[0], r[6] - r[3], r[1] - r[6]]
            print(f"  p{p:2d}: " + " ".join(f"{v:6d}" for v in d) + f"   {sum(d):6d}")
        prev = r[1]
