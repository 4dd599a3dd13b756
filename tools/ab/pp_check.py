#!/usr/bin/env python3
"""Development tool (GPU): correctness of the ping-pong NT GEMM at each ring depth, with the position of the wrong elements."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oneprot_amd import hip
for (M, N, K) in ((1024, 640, 640), (8192, 640, 640), (256, 128, 1024), (4096, 256, 512)):
    g = torch.Generator().manual_seed(3)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16).cuda()
    ref = A.float() @ W.float().t()
    for ring in (4, 5, 6):
        hip.query("oneprot_gemm_force_shape", 32 + 64 * ring)
        for rep in range(3):
            out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
            hip.call("oneprot_gemm_bf16_nt", A, W, M, N, K, K, K, hip.EPI_BF16, None, out, None, None, None, None, None, 1.0, 0, 0, 0)
            torch.cuda.synchronize()
            bad = (out.float() - ref).abs() > 0.02 + 0.01 * ref.abs()
            if bad.any():
                rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
                tiles = sorted({(int(r) // 256, int(c) // 128) for r, c in bad.nonzero()[:: max(1, int(bad.sum()) // 2000)].tolist()})
                print(f"M{M} N{N} K{K} ring {ring} rep {rep}: {int(bad.sum())} bad; rows {int(rows.min())}..{int(rows.max())} cols {int(cols.min())}..{int(cols.max())}; tiles {tiles[:12]}")
            else:
                print(f"M{M} N{N} K{K} ring {ring} rep {rep}: ok")
hip.query("oneprot_gemm_force_shape", -1)
