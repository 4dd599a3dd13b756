import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oneprot_amd import hip
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
n = 256 * 1024 * 1024   # floats = 1 GiB
x = torch.randn(n, device="cuda"); y = torch.empty_like(x)
ms = timeit(lambda: y.copy_(x)); print(f"torch copy 1GiB->1GiB: {ms:.3f} ms  {2*n*4/ms/1e9:.2f} TB/s")
ms = timeit(lambda: y.fill_(1.0)); print(f"torch fill 1GiB: {ms:.3f} ms  {n*4/ms/1e9:.2f} TB/s")
ms = timeit(lambda: x.sum()); print(f"torch sum 1GiB: {ms:.3f} ms  {n*4/ms/1e9:.2f} TB/s")
yb = torch.empty(n, dtype=torch.bfloat16, device="cuda")
ms = timeit(lambda: hip.call("oneprot_cast_f32_to_bf16", x, yb, n)); print(f"cast f32->bf16 (our kernel) : {ms:.3f} ms  {n*6/ms/1e9:.2f} TB/s")
T, d = 131072, 640
xx = torch.randn(T, d, device="cuda"); g = torch.ones(d, device="cuda"); b = torch.zeros(d, device="cuda"); yy = torch.empty(T, d, dtype=torch.bfloat16, device="cuda")
ms = timeit(lambda: hip.call("oneprot_layernorm_fwd", xx, 0, g, b, yy, None, None, None, T, d, 1e-5)); print(f"LN fwd: {ms:.3f} ms  {T*d*6/ms/1e9:.2f} TB/s")
ms = timeit(lambda: torch.nn.functional.layer_norm(xx, (d,), g, b)); print(f"torch LN fwd fp32->fp32: {ms:.3f} ms  {T*d*8/ms/1e9:.2f} TB/s")
a16 = torch.randn(T, 2560, device="cuda").to(torch.bfloat16); w16 = torch.randn(640, 2560, device="cuda").to(torch.bfloat16)
ms = timeit(lambda: torch.matmul(a16, w16.t())); print(f"torch(hipBLASLt) bf16 [T,2560]x[2560,640]: {ms:.3f} ms  {2*T*640*2560/ms/1e9:.1f} TF/s")
a16 = torch.randn(T, 640, device="cuda").to(torch.bfloat16); w16 = torch.randn(2560, 640, device="cuda").to(torch.bfloat16)
ms = timeit(lambda: torch.matmul(a16, w16.t())); print(f"torch(hipBLASLt) bf16 [T,640]x[640,2560]: {ms:.3f} ms  {2*T*640*2560/ms/1e9:.1f} TF/s")
w16 = torch.randn(640, 640, device="cuda").to(torch.bfloat16)
ms = timeit(lambda: torch.matmul(a16, w16.t())); print(f"torch(hipBLASLt) bf16 [T,640]x[640,640]: {ms:.3f} ms  {2*T*640*640/ms/1e9:.1f} TF/s")
