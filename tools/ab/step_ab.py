#!/usr/bin/env python3
"""Development tool (GPU): the run-time switches and kernel-path hooks ranked INSIDE the cfg-2 training step (the step runs at the board's power cap: what a
kernel costs in a loop of its own at burst clocks does not rank variants reliably -- profiles/r06_power_probe.txt).  One workload, every configuration timed for
`--steps` steps between events, `--rounds` interleaved rounds, median ms per step and the difference to the first configuration.
usage: step_ab.py [--steps 8] [--rounds 3] [--only name,name]"""
import argparse, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ["ONEPROT_ALLOW_RANDOM_INIT"] = "1"
import torch
import bench
from oneprot_amd import hip
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=8); ap.add_argument("--rounds", type=int, default=3); ap.add_argument("--only", default="")
ap.add_argument("--windows", type=int, default=0, help="co-resident windows per step (tests/cu_spin.hip on a side stream: the stand-in for the RCCL channels of an overlapped all-reduce)")
ap.add_argument("--window-ms", type=float, default=0.7); ap.add_argument("--hold-cus", type=int, default=16)
a = ap.parse_args()
_argv = sys.argv; sys.argv = [_argv[0], "--no-cpu-baseline", "--no-extras"]; args = bench.parse_args(); sys.argv = _argv
dev = torch.device("cuda:0")
work = bench.build_workload(args, dev, 0)
module, batch = work["module"], work["batch"]
Q = hip.query


def env(k, v):
    def f():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    return f


def dyn(on):
    def f():
        ws = hip.sched_workspace(131072)
        hip.lib().oneprot_dynamic_tiles(*(ws if on else (None, 0)))
    return f


# name -> (set, unset)
CONFIGS = {
    "default": (lambda: None, lambda: None),
    "attention backward: fused 16-wave kernel": (lambda: Q("oneprot_attn_force_bwd_path", 1), lambda: Q("oneprot_attn_force_bwd_path", -1)),
    "attention backward: split dQ / dK-dV kernels": (lambda: Q("oneprot_attn_force_bwd_path", 0), lambda: Q("oneprot_attn_force_bwd_path", -1)),
    "attention forward: chunked kernel (fwd2)": (lambda: Q("oneprot_attn_force_fwd_path", 2), lambda: Q("oneprot_attn_force_fwd_path", -1)),
    "attention forward: round-1 kernel (row maximum)": (lambda: Q("oneprot_attn_force_fwd_path", 0), lambda: Q("oneprot_attn_force_fwd_path", -1)),
    "weight-gradient GEMM: round-2 form (variant 2)": (lambda: Q("oneprot_gemm_tn_variant", 2), lambda: Q("oneprot_gemm_tn_variant", -1)),
    "out-projection + LN: four-wave form": (lambda: Q("oneprot_gemm_ln_form", 1), lambda: Q("oneprot_gemm_ln_form", 0)),
    "ONEPROT_FUSED_LN=0 (out-projection, then LayerNorm)": (env("ONEPROT_FUSED_LN", "0"), env("ONEPROT_FUSED_LN", None)),
    "ONEPROT_FFN2_LN=0 (FFN-2, then LayerNorm)": (env("ONEPROT_FFN2_LN", "0"), env("ONEPROT_FFN2_LN", None)),
    "tiles from the work queues": (dyn(True), dyn(False)),
    "weight-gradient GEMM with a 16-CU reserve": (lambda: Q("oneprot_cu_reserve", 16), lambda: Q("oneprot_cu_reserve", 0)),
    "work queues + 16-CU reserve (the multi-rank defaults)": (lambda: (dyn(True)(), Q("oneprot_cu_reserve", 16)), lambda: (dyn(False)(), Q("oneprot_cu_reserve", 0))),
}
names = [n for n in CONFIGS if not a.only or any(o in n for o in a.only.split(",")) or n == "default"]


if a.windows:
    import ctypes, threading, time
    spin = ctypes.CDLL(os.path.join(ROOT, "tests", "libcu_spin.so"))
    spin.cu_spin_launch.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()


def timed():
    module.training_step(batch, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stop = None
    if a.windows:                                           # windows spread evenly over the wall time of a step (~0.215 s), for as long as the timed steps run
        stop = threading.Event()
        def feeder():
            gap = 0.215 / a.windows
            while not stop.is_set():
                time.sleep(gap)
                spin.cu_spin_launch(a.hold_cus, int(a.window_ms * 1e3), sink.data_ptr(), side.cuda_stream)
        th = threading.Thread(target=feeder); th.start()
    e0.record()
    for _ in range(a.steps):
        module.training_step(batch, 0)
    e1.record(); torch.cuda.synchronize()
    if stop is not None:
        stop.set(); th.join(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.steps


for _ in range(3):
    module.training_step(batch, 0)
res = {n: [] for n in names}
for r in range(a.rounds):
    for n in names:
        CONFIGS[n][0]()
        try:
            res[n].append(timed())
        finally:
            CONFIGS[n][1]()
base = statistics.median(res["default"])
print(f"cfg-2 sub-step, median of {a.rounds} interleaved rounds of {a.steps} steps; sched error word {hip.sched_error()}"
      + (f"; beside {a.windows} windows per step of {a.hold_cus} held CUs for {a.window_ms} ms each" if a.windows else ""))
for n in names:
    m = statistics.median(res[n])
    print(f"  {n:58s} {m:8.2f} ms  ({(m / base - 1) * 100:+5.1f} %)   rounds: " + " ".join(f"{x:.1f}" for x in res[n]))
