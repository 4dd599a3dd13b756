#!/usr/bin/env python3
"""Development tool (GPU): ms per cfg-2 training step with the kernel library named by G8_LIB (default: the product's) -- one process per library, so that
builds that differ at compile time can be ranked INSIDE the step (tools/ab/step_libs_ab.sh alternates them on one box).  usage: step_once.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ["ONEPROT_ALLOW_RANDOM_INIT"] = "1"
import warnings
warnings.filterwarnings("ignore")
import torch
from oneprot_amd import hip
if os.environ.get("G8_LIB"):
    hip.LIB_PATH = os.path.abspath(os.environ["G8_LIB"])
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
_argv = sys.argv; sys.argv = [_argv[0], "--no-cpu-baseline", "--no-extras"]; args = bench.parse_args(); sys.argv = _argv
work = bench.build_workload(args, torch.device("cuda:0"), 0)
module, batch = work["module"], work["batch"]
if os.environ.get("STEP_RAGGED"):      # sequence lengths L/2 .. L (the reference pads a batch to its longest row): <eos> at the end of each row, <pad> = 1 behind it
    g = torch.Generator().manual_seed(5)
    for m, (s_ids, x_ids, _, _) in batch.items():
        B, L = s_ids.shape
        lens = torch.randint(L // 2, L + 1, (B,), generator=g)
        for ids, eos in ((s_ids, 2), (x_ids, 2)):
            for b in range(B):
                n = int(lens[b])
                ids[b, n - 1] = eos
                ids[b, n:] = 1
for _ in range(4):
    module.training_step(batch, 0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    loss = module.training_step(batch, 0)
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / steps:.3f} {float(loss.detach()):.6f}")
