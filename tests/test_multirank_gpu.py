"""N>1 path on real kernels.  (a) RCCL itself: the collectives the product issues on backend "nccl", on one GPU (world 1) and -- when the box
has them -- across two GPUs.  (b) Two ranks (both on the one GPU of the test box, gloo transport) each take half of the batch; with
local_loss + gather_with_grad and mean gradient all-reduce the result must equal the single-process run on the whole batch
(mean over ranks of the per-rank losses == global loss; reduced gradient == global gradient)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, out_dir, golden, port, loss_name="CLIP", transport="gloo", env=None):
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gpu_rank_worker.py"), str(r), str(world), str(port), out_dir, golden, loss_name, transport],
                              env=dict(os.environ, **(env or {})))
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0


def _run_rccl(world, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_rccl_rank_worker.py"), str(r), str(world), str(port)], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0


def test_rccl_collectives_single_gpu():
    """backend "nccl" (= RCCL) on the one GPU of the test box, world size 1: the product's reduce_scatter_tensor / all_gather_into_tensor /
    asynchronous ReduceOp.AVG calls go through RCCL itself (gloo has none of them and runs fallbacks instead)."""
    _run_rccl(1, 29751)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_rccl_collectives_two_gpus():
    """the same calls across two GPUs over xGMI, plus the SigLIP neighbour ring; expectations are asserted inside the rank workers"""
    _run_rccl(2, 29752)


@pytest.mark.parametrize("loss_name,port", [("CLIP", 29741), ("SIGLIP", 29745)])
def test_two_ranks_equal_single_process(golden_dir, tmp_path, loss_name, port):
    """CLIP: packed all-gather + reduce-scatter.  SIGLIP: the direct peer exchange of the sequence chunks (here one step, staged through the host
    because gloo moves host memory only) with the HIP block kernels, and the one-collective backward -- both losses are means over the LOCAL rows,
    so the mean over ranks equals the single-process loss on the whole batch and the mean-reduced gradient equals its gradient."""
    golden = os.path.join(golden_dir, "esm_pair_hd16.pt")       # batch 6 -> 3 pairs per rank
    out = str(tmp_path)
    _run(1, out, golden, port, loss_name)
    _run(2, out, golden, port + 1, loss_name)
    one = torch.load(os.path.join(out, f"{loss_name}_w1_rank0.pt"), weights_only=False)
    r0 = torch.load(os.path.join(out, f"{loss_name}_w2_rank0.pt"), weights_only=False)
    r1 = torch.load(os.path.join(out, f"{loss_name}_w2_rank1.pt"), weights_only=False)
    assert r0["overlap_calls"] >= 2 and one["overlap_calls"] == 0          # arena gradient reduced in ranges from inside the backward
    mean_loss = 0.5 * (r0["loss"] + r1["loss"])
    assert abs(mean_loss - one["loss"]) / one["loss"] < 1e-3, (r0["loss"], r1["loss"], one["loss"])
    assert abs(r0["gnorm"] - one["gnorm"]) / one["gnorm"] < 2e-2 and abs(r0["gnorm"] - r1["gnorm"]) < 1e-5 * one["gnorm"] + 1e-6
    for k in ("w", "emb"):
        assert torch.equal(r0[k], r1[k]), k                      # replicas stay bit-identical after the all-reduced step
        d = (r0[k] - one[k]).abs()
        assert d.max() < 2.1e-3 and (d > 5e-4).float().mean() < 0.02, (k, d.max())    # same Adam step up to bf16-noise sign flips on tiny gradients


def test_two_ranks_with_the_gradient_overlap_switched_off(golden_dir, tmp_path):
    """ONEPROT_GRAD_OVERLAP=0: no all-reduce is issued from inside the backward (nothing co-resident with its persistent kernels); the arena gradients
    are reduced in buckets afterwards -- the same numbers as the overlapped form."""
    golden = os.path.join(golden_dir, "esm_pair_hd16.pt")
    out_on, out_off = str(tmp_path / "on"), str(tmp_path / "off")
    os.makedirs(out_on); os.makedirs(out_off)
    _run(2, out_on, golden, 29781)
    _run(2, out_off, golden, 29783, env={"ONEPROT_GRAD_OVERLAP": "0"})
    on = [torch.load(os.path.join(out_on, f"CLIP_w2_rank{r}.pt"), weights_only=False) for r in range(2)]
    off = [torch.load(os.path.join(out_off, f"CLIP_w2_rank{r}.pt"), weights_only=False) for r in range(2)]
    assert all(r["overlap_calls"] >= 2 for r in on) and all(r["overlap_calls"] == 0 for r in off)
    for a, b in zip(on, off):
        assert a["loss"] == b["loss"] and abs(a["gnorm"] - b["gnorm"]) <= 1e-6 * a["gnorm"]
        assert torch.equal(a["w"], b["w"]) and torch.equal(a["emb"], b["emb"])          # the same mean of the same two gradients
    assert torch.equal(off[0]["w"], off[1]["w"])


@pytest.mark.parametrize("loss_name,port", [("CLIP", 29771), ("SIGLIP", 29775)])
def test_four_ranks_equal_single_process(golden_dir, tmp_path, loss_name, port):
    """The same at world size 4 (12 pairs of the head-dim-24 fixture, 3 per rank, four processes sharing the one GPU over gloo): CLIP label offsets at
    ranks 2 and 3; bidirectional SigLIP = one two-way step followed by the remainder step, i.e. a second transfer posted while the first step's
    blocks run (ref loss.py:72-83,260-309).  Five processes touch the card (the box allows six)."""
    golden = os.path.join(golden_dir, "esm_pair_hd24.pt")
    out = str(tmp_path)
    _run(1, out, golden, port, loss_name)
    _run(4, out, golden, port + 1, loss_name)
    one = torch.load(os.path.join(out, f"{loss_name}_w1_rank0.pt"), weights_only=False)
    rs = [torch.load(os.path.join(out, f"{loss_name}_w4_rank{r}.pt"), weights_only=False) for r in range(4)]
    assert all(r["overlap_calls"] >= 2 for r in rs)
    mean_loss = sum(r["loss"] for r in rs) / 4
    assert abs(mean_loss - one["loss"]) / one["loss"] < 1e-3, ([r["loss"] for r in rs], one["loss"])
    assert abs(rs[0]["gnorm"] - one["gnorm"]) / one["gnorm"] < 2e-2
    for r in rs[1:]:
        assert abs(r["gnorm"] - rs[0]["gnorm"]) < 1e-5 * one["gnorm"] + 1e-6
        for k in ("w", "emb"):
            assert torch.equal(r[k], rs[0][k]), k                 # replicas stay bit-identical after the all-reduced step
    for k in ("w", "emb"):
        d = (rs[0][k] - one[k]).abs()
        assert d.max() < 2.1e-3 and (d > 5e-4).float().mean() < 0.02, (k, d.max())


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
@pytest.mark.parametrize("loss_name,port", [("CLIP", 29761), ("SIGLIP", 29765)])
def test_whole_substep_over_rccl_two_gpus(golden_dir, tmp_path, loss_name, port):
    """The sub-step as a node runs it: rank r on GPU r, backend "nccl".  Overlapped arena-gradient all-reduce ranges issued from inside the backward,
    the packed feature exchange (or the SigLIP peer exchange) and the head-parameter reduce all go through RCCL; result vs the single-process run on
    the whole batch, replicas bit-identical afterwards.  Skipped on the one-GPU test box; the one-GPU variant above covers the same code with gloo."""
    golden = os.path.join(golden_dir, "esm_pair_hd16.pt")
    out = str(tmp_path)
    _run(1, out, golden, port, loss_name)
    _run(2, out, golden, port + 1, loss_name, "nccl")
    one = torch.load(os.path.join(out, f"{loss_name}_w1_rank0.pt"), weights_only=False)
    r0 = torch.load(os.path.join(out, f"{loss_name}_nccl_w2_rank0.pt"), weights_only=False)
    r1 = torch.load(os.path.join(out, f"{loss_name}_nccl_w2_rank1.pt"), weights_only=False)
    assert r0["overlap_calls"] >= 2
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - one["loss"]) / one["loss"] < 1e-3
    assert abs(r0["gnorm"] - one["gnorm"]) / one["gnorm"] < 2e-2 and abs(r0["gnorm"] - r1["gnorm"]) < 1e-5 * one["gnorm"] + 1e-6
    for k in ("w", "emb"):
        assert torch.equal(r0[k], r1[k]), k
        d = (r0[k] - one[k]).abs()
        assert d.max() < 2.1e-3 and (d > 5e-4).float().mean() < 0.02, (k, d.max())



def test_bench_two_ranks_prints_per_rank_exchange(tmp_path):
    """`python bench.py --gpus 2` as the driver starts it (the parent launches its ranks through torch.distributed.run on 127.0.0.1), here with both ranks on
    the one GPU of the test box over gloo and a small encoder: ONE JSON line from rank 0, whole-job value, weak scaling, and the per-rank `exchange`
    object (feature all-gather, its reduce-scatter, exposed part of the gradient all-reduce) that a scaling curve is read against."""
    import json
    env = dict(os.environ, ONEPROT_DIST_BACKEND="gloo", ONEPROT_ALLOW_RANDOM_INIT="1", MASTER_PORT="29781")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--model", "facebook/esm2_t6_8M_UR50D", "--batch", "8",
           "--seq-len", "128", "--no-cpu-baseline", "--no-extras"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 16 and d["value"] > 0
    per = d["exchange"]["per_rank"]
    assert [p["rank"] for p in per] == [0, 1]
    for p in per:
        assert set(p["parts"]) >= {"feature_all_gather", "feature_reduce_scatter", "grad_allreduce_exposed"}, p
        assert p["exchange_ms"] > 0
