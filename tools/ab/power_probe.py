#!/usr/bin/env python3
"""Development tool (GPU): board power and shader clock while the cfg-2 training step runs, sampled from the amdgpu hwmon / sysfs files by a helper thread
(no root needed), next to the same while only GEMMs / only attention kernels run.  Is the step running at the board's power cap?  usage: power_probe.py [seconds]"""
import glob, os, sys, threading, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ["ONEPROT_ALLOW_RANDOM_INIT"] = "1"
import torch
import bench
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0


def rd(p):
    try:
        return open(p).read().strip()
    except Exception:
        return None


cards = []
for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
    if rd(hw + "/power1_average") is not None or rd(hw + "/power1_input") is not None:
        cards.append(hw)
print(f"{len(cards)} hwmon nodes with a power reading (every GPU of the host is visible; the one this process drives is found by a calibration burst)", flush=True)


class Sampler:
    def __init__(self):
        self.stop = threading.Event(); self.rows = []
    def run(self):
        while not self.stop.is_set():
            row = []
            for hw in cards:
                p = rd(hw + "/power1_average") or rd(hw + "/power1_input")
                f = rd(hw + "/freq1_input")
                row.append((float(p) / 1e6 if p else float("nan"), float(f) / 1e6 if f else float("nan")))
            self.rows.append(row)
            time.sleep(0.02)
    def __enter__(self):
        self.t = threading.Thread(target=self.run); self.t.start(); return self
    def __exit__(self, *a):
        self.stop.set(); self.t.join()
    def per_card(self):
        return [([r[i][0] for r in self.rows[len(self.rows) // 5:]], [r[i][1] for r in self.rows[len(self.rows) // 5:]]) for i in range(len(cards))]
    def summary(self, i):
        ps, fs = self.per_card()[i]
        return (f"power {statistics.mean(ps):6.0f} W (min {min(ps):.0f}, max {max(ps):.0f})   sclk {statistics.mean(fs):5.0f} MHz (min {min(fs):.0f}, max {max(fs):.0f})   "
                f"bf16 MFMA peak at that clock {2500 * statistics.mean(fs) / 2400:5.0f} TFLOP/s")


dev = torch.device("cuda:0")
_argv = sys.argv; sys.argv = [_argv[0], "--no-cpu-baseline", "--no-extras"]; args = bench.parse_args(); sys.argv = _argv
work = bench.build_workload(args, dev, 0)
module, batch = work["module"], work["batch"]
from oneprot_amd import hip
T, d, f = 131072, 640, 2560
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(T, f, device="cuda", generator=g).to(torch.bfloat16); W = (torch.randn(d, f, device="cuda", generator=g) * 0.05).to(torch.bfloat16); o0 = torch.empty(T, d, dtype=torch.bfloat16, device="cuda")
gemm = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, T, d, f, f, f, hip.EPI_BF16, None, o0, None, None, None, None, None, 1.0, 0, 0, 0)
B, L, H, hd = 256, 512, 20, 32
mk = lambda: (torch.randn(B, H, L, hd, device="cuda", generator=g) * 0.7).to(torch.bfloat16)
q, k, v = mk(), mk(), mk(); kb = torch.zeros(B, L, device="cuda"); ctx = torch.empty(T, H * hd, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, L, device="cuda")
attn = lambda: hip.call("oneprot_attn_fwd", q, k, v, kb, ctx, lse, B, H, L, hd)
xs = torch.randn(T, d, device="cuda"); ys = torch.empty_like(xs)
copy = lambda: ys.copy_(xs)


def loop(fn, n_inner):
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(n_inner):
            fn()
        torch.cuda.synchronize(); n += n_inner
    return n / (time.time() - t0)


with Sampler() as s0:
    loop(lambda: time.sleep(0.01), 1)
with Sampler() as s1:
    loop(gemm, 200)
idle, busy = s0.per_card(), s1.per_card()
mine = max(range(len(cards)), key=lambda i: statistics.mean(busy[i][0]) - statistics.mean(idle[i][0]))
print(f"this process drives {cards[mine]}: power cap {float(rd(cards[mine] + '/power1_cap')) / 1e6:.0f} W; idle {s0.summary(mine)}", flush=True)
dctx = (torch.randn(T, H * hd, device="cuda", generator=g) * 0.1).to(torch.bfloat16); dqkv = torch.empty(T, 3 * H * hd, dtype=torch.bfloat16, device="cuda")
wsb = torch.empty(hip.query("oneprot_attn_bwd_workspace", B, H, L), dtype=torch.uint8, device="cuda"); cos = torch.rand(L, hd // 2, device="cuda"); sin = torch.rand(L, hd // 2, device="cuda")
attn_b = lambda: hip.call("oneprot_attn_bwd", q, k, v, kb, ctx, dctx, lse, cos, sin, hd ** -0.5, dqkv, wsb, B, H, L, hd)
A1 = torch.randn(T, d, device="cuda", generator=g).to(torch.bfloat16); W1 = (torch.randn(f, d, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
b1 = torch.randn(f, device="cuda", generator=g); o1 = torch.empty(T, f, dtype=torch.bfloat16, device="cuda")
ffn1 = lambda: hip.call("oneprot_gemm_bf16_nt", A1, W1, T, f, d, d, d, hip.EPI_BIAS_GELU, b1, o1, None, None, None, None, None, 1.0, 0, 0, 0)
flop = {"dgrad GEMM K=2560 only": 2.0 * T * d * f, "FFN-1 GEMM + GELU only": 2.0 * T * d * f, "attention forward only": 4.0 * B * H * L * L * hd, "attention backward only": 10.0 * B * H * L * L * hd}
for name, fn, n_inner in (("training step (cfg-2)", lambda: module.training_step(batch, 0), 2), ("dgrad GEMM K=2560 only", gemm, 200), ("FFN-1 GEMM + GELU only", ffn1, 200),
                          ("attention forward only", attn, 200), ("attention backward only", attn_b, 100), ("fp32 copy only", copy, 400)):
    with Sampler() as s:
        rate = loop(fn, n_inner)
    extra = f"   {1e6 / rate:7.1f} us per call = {flop[name] * rate / 1e12:6.0f} TFLOP/s" if name in flop else (f"   {1e3 / rate:7.1f} ms per step" if "step" in name else "")
    print(f"{name:24s}: {s.summary(mine)}{extra}", flush=True)
