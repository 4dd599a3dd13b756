"""Development tool (GPU): run the vendor bf16 GEMM (torch.matmul -> hipBLASLt) on the hot-path shapes so that
`rocprofv3 --kernel-trace --stats` shows which tile configuration the vendor library picks on this chip (reference point only;
the product never calls it)."""
import torch
T, d, f = 131072, 640, 2560
for (N, K) in ((f, d), (d, f), (d, d), (3 * d, d)):
    A = torch.randn(T, K, device="cuda").to(torch.bfloat16)
    W = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        C = A @ W.t()
    torch.cuda.synchronize()
# TN shape: dW[N,K] = dY^T X
for (N, K) in ((f, d), (d, f), (d, d)):
    dY = torch.randn(T, N, device="cuda").to(torch.bfloat16)
    X = torch.randn(T, K, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        C = dY.t() @ X
    torch.cuda.synchronize()
