import sys, os
sys.path.insert(0, "/root/repo")
import torch
from oneprot_amd import hip
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K, tA, bkn) in [(256, 832, 640, 0, 0), (256, 1024, 832, 0, 0), (832, 640, 256, 1, 1), (256, 640, 832, 0, 1), (256, 256, 1024, 0, 0), (1024, 832, 256, 1, 1)]:
    A = torch.randn(K * M, device="cuda"); B = torch.randn(K * N, device="cuda"); C = torch.zeros(M, N, device="cuda")
    print(M, N, K, tA, bkn, f"{timeit(lambda: hip.call('oneprot_sgemm', A, B, C, M, N, K, tA, bkn, 1.0, 0)):.1f} us")
