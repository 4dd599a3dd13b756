#!/usr/bin/env python3
"""Development tool (GPU): timeline of one work-group of the 8-phase NT GEMM.  Needs tools/ab/libg8_stamp.so (build_g8.sh stamp,-DG8_STAMP): the first
wave of each wave group of work-group 0 records s_memtime after every barrier of its first 32 K-tiles (8 stamps per K-tile).  Prints, per K-tile,
the slot lengths in cycles of the 100 MHz... no: s_memtime counts shader cycles.  usage: g8_stamps.py [N] [K] [epi] [shape]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, os.environ.get("G8_STAMP_LIB", "libg8_stamp.so")))
P, I, F, L64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64
lib.oneprot_gemm_bf16_nt.argtypes = [P, P, L64, I, I, I, I, I, P, P, P, P, P, P, P, F, I, I, I, P]
lib.oneprot_gemm_bf16_nt.restype = I
lib.oneprot_gemm_force_shape.argtypes = [I]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
K = int(sys.argv[2]) if len(sys.argv) > 2 else 640
epi = int(sys.argv[3]) if len(sys.argv) > 3 else 0
shape = int(sys.argv[4]) if len(sys.argv) > 4 else 41
T = 131072
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(T, K, device="cuda", generator=g).to(torch.bfloat16)
W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
bias = torch.randn(N, device="cuda", generator=g)
o0 = torch.empty(T, N, dtype=torch.float32 if epi == 3 else torch.bfloat16, device="cuda")
o1 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda") if epi == 2 else None
aux = torch.randn(T, N, device="cuda", generator=g).to(torch.float32 if epi == 3 else torch.bfloat16) if epi in (3, 5) else None
stamps = torch.zeros(2048, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
ptr = lambda t: t.data_ptr() if t is not None else None
lib.oneprot_gemm_force_shape(shape)
cap = int(os.environ.get("G8_CAP", "0"))
lib.oneprot_gemm8_dephase.argtypes = [I, I]
dg, ds = int(os.environ.get("G8_DG", "1")), int(os.environ.get("G8_DS", "0"))
lib.oneprot_gemm8_dephase((cap << 16) | dg, ds)
for _ in range(3):
    rc = lib.oneprot_gemm_bf16_nt(ptr(A), ptr(W), T, N, K, K, K, epi, ptr(bias), ptr(o0), ptr(o1), ptr(stamps), ptr(aux), None, None, 1.0, 0, 0, 0, st)
    assert rc == 0, rc
torch.cuda.synchronize()
e = stamps.cpu()[512:640].view(2, 8, 8)
es = stamps.cpu()[768:800].view(2, 16)
s = stamps.cpu()[:512].view(2, 32, 8)
nk = K // 64
for grp in range(2):
    t0 = int(s[grp, 0, 0])
    print(f"group {grp}: per K-tile [start offset | 8 slot lengths after barriers 1..8: read0+wait, mfma0, read1, mfma1, read2, mfma2, read3+vmcnt, mfma3]; K-tiles per output tile: {nk}")
    prev = None
    for kt in range(32):
        row = [int(v) for v in s[grp, kt]]
        if row[0] == 0: break
        gaps = [row[0] - prev if prev is not None else 0] + [row[i] - row[i - 1] for i in range(1, 8)]
        prev = row[7]
        mark = " <- first K-tile of an output tile (gap[0] contains the epilogue)" if kt % nk == 0 and kt > 0 else ""
        print(f"  kt {kt:2d} @{row[0] - t0:8d}: " + " ".join(f"{v:5d}" for v in gaps) + mark)

m = stamps.cpu()[1024:1536].view(2, 32, 4, 2)
for grp in range(2):
    print(f"group {grp}: MFMA runs per K-tile, 4 phases: [cycles from the stamp before the first MFMA to the stamp after the last one]")
    for kt in range(4, 12):
        if int(m[grp, kt, 0, 0]) == 0: break
        print(f"  kt {kt:2d}: " + " ".join(f"{int(m[grp, kt, ph, 1] - m[grp, kt, ph, 0]):5d}" for ph in range(4)))

for grp in range(2):
    print(f"group {grp} tile ends: [K-loop end -> resync barrier -> X1 issued -> epilogue issued -> accumulators zeroed]")
    for q in range(8):
        row = [int(v) for v in e[grp, q][:5]]
        if row[0] == 0: break
        print(f"  tile {q}: " + " ".join(f"{row[i] - row[i - 1]:6d}" for i in range(1, 5)))

for grp in range(2):
    row = [int(v) for v in es[grp][:8]]
    print(f"group {grp} last tile, pair-map epilogue: cycles per column group (4-8 stores each): " + " ".join(f"{row[i] - row[i - 1]:6d}" for i in range(1, 8) if row[i]))
