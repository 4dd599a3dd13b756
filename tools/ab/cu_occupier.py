#!/usr/bin/env python3
"""Development tool (GPU): what a co-resident kernel that holds k compute units (tests/cu_spin.hip: the stand-in for the RCCL channels of an
overlapped gradient all-reduce) does to the persistent one-work-group-per-CU kernels of the training step.  Runs the cfg-2 sub-step of bench.py
with k in {0, 4, 8, 16, 32} CUs held on a side stream (a) for the whole step and (b) for `--window-ms` windows inside the backward, as an
all-reduce of one ~118 MB gradient range would (0.7 ms at the xGMI ring rate; five per tower and step), and prints ms per step.
usage: cu_occupier.py [--steps 5] [--window-ms 0.7] [--windows 5]        (tests/libcu_spin.so is built by __graft_entry__.build())"""
import argparse, ctypes, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ["ONEPROT_ALLOW_RANDOM_INIT"] = "1"
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--window-ms", type=float, default=0.7)
ap.add_argument("--windows", type=int, default=5)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--only", default="", help="whole | windows: only that part")
ap.add_argument("--ks", default="4,8,16,32")
ap.add_argument("--soak-seconds", type=float, default=0.0, help="instead of the timing tables: train for this long with k = 8 / 16 windows arriving at random moments; finite loss, counters at the end")
a = ap.parse_args()
spin = ctypes.CDLL(os.path.join(ROOT, "tests", "libcu_spin.so"))
spin.cu_spin_launch.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
_argv = sys.argv
sys.argv = [_argv[0], "--batch", str(a.batch), "--no-cpu-baseline", "--no-extras"]
args = bench.parse_args()
sys.argv = _argv
work = bench.build_workload(args, dev, 0)
module, batch = work["module"], work["batch"]
sink = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()


def step():
    module.training_step(batch, 0)


def timed(hold_cus, whole_step_us=0, windows=0, window_us=0, gap_ms=0.0):
    """one step with CUs held: whole_step_us > 0 -> one spin launched just before the step; windows > 0 -> `windows` spins of window_us spread over the step"""
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if whole_step_us:
        spin.cu_spin_launch(hold_cus, whole_step_us, sink.data_ptr(), side.cuda_stream)
        time.sleep(0.002)                                   # the spin is resident before the step's first kernel
    e0.record()
    if windows:
        # the windows are queued on the side stream behind host-timed gaps: launch them from a helper thread while the step is being enqueued
        import threading
        def feeder():
            for _ in range(windows):
                time.sleep(gap_ms * 1e-3)
                spin.cu_spin_launch(hold_cus, window_us, sink.data_ptr(), side.cuda_stream)
        th = threading.Thread(target=feeder); th.start()
        step(); th.join()
    else:
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for _ in range(3):
    step()
torch.cuda.synchronize()
if a.soak_seconds > 0:
    import random, threading
    from oneprot_amd import hip
    stop = threading.Event()
    fired = [0]
    def feeder():
        rng = random.Random(7)
        while not stop.is_set():
            time.sleep(rng.uniform(0.002, 0.05))
            spin.cu_spin_launch(rng.choice((4, 8, 16, 32)), int(rng.uniform(200, 1500)), sink.data_ptr(), side.cuda_stream)
            fired[0] += 1
    th = threading.Thread(target=feeder); th.start()
    t0, n, worst = time.time(), 0, 0.0
    try:
        while time.time() - t0 < a.soak_seconds:
            for _ in range(25):
                loss = module.training_step(batch, 0)
                n += 1
            v = float(loss.detach())                        # (synchronises)
            if not (v == v and abs(v) < 1e4):                # a poisoned launch (sched_error != 0) or a training run that diverged on its one fixed batch
                print(f"  loss {v} after {n} steps; sched error word {hip.sched_error()}", flush=True)
                break
            print(f"  {n:6d} steps, {fired[0]:6d} windows, {time.time() - t0:6.0f} s, loss {v:.5f}", flush=True)
    finally:
        stop.set(); th.join()
    torch.cuda.synchronize()
    print(f"soak: {n} steps beside {fired[0]} windows of 4-32 held CUs (0.2-1.5 ms each) in {time.time() - t0:.0f} s; late ticket draws {hip.sched_late_draws()}, "
          f"bounded waits that ran out / tile-count mismatches {hip.sched_error()}, dynamic tiles {hip.dynamic_tiles_wanted()}", flush=True)
    sys.exit(0)
base = statistics.median(timed(0) for _ in range(a.steps))
print(f"cfg-2 sub-step, batch {a.batch}: {base:.1f} ms with nothing else on the GPU", flush=True)
KS = [int(x) for x in a.ks.split(",")]
print("k CUs held for the WHOLE step (spin resident before the first kernel):", flush=True)
for k in (KS if a.only in ("", "whole") else []):
    t = statistics.median(timed(k, whole_step_us=int(base * 1.6e3)) for _ in range(a.steps))
    print(f"  k = {k:3d}: {t:7.1f} ms  ({t / base:.3f} x)", flush=True)
    torch.cuda.synchronize(); time.sleep(0.3)
win_us = int(a.window_ms * 1e3)
gap = base * 0.6 / max(a.windows, 1)                        # spread over the backward (the last ~2/3 of the step)
print(f"k CUs held for {a.windows} windows of {a.window_ms} ms inside the step (an overlapped all-reduce of five gradient ranges); exposed instead they would cost {a.windows * a.window_ms:.1f} ms:", flush=True)
for k in (KS if a.only in ("", "windows") else []):
    t = statistics.median(timed(k, windows=a.windows, window_us=win_us, gap_ms=gap) for _ in range(a.steps))
    print(f"  k = {k:3d}: {t:7.1f} ms  (+{t - base:.1f} ms)", flush=True)

from oneprot_amd import hip
print(f"sched workspace after the run: late ticket draws {hip.sched_late_draws()}, bounded waits that ran out {hip.sched_error()}", flush=True)
