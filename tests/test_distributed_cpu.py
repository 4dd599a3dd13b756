"""world_size 2 / 3 / 4 / 8 tests of the N>1 path on CPU (gloo): the packed feature all-gather (+ reduce-scatter backward), the SigLIP ring
exchange and the bucketed gradient all-reduce -- checked against goldens the reference produced on real gloo ranks."""
import json
import os

import pytest
import torch
import torch.multiprocessing as mp

from tests import _dist_workers as W


def _run(fn, world, port, *args):
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=fn, args=(r, world, port) + args) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    for p in procs:
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"


def _close(a, b, tol=2e-6):
    assert (a - b).abs().max() <= tol + 1e-5 * b.abs().max(), (a - b).abs().max()


# world 4: bidirectional SigLIP = one two-way step + the remainder step, CLIP label offsets at rank 3; world 8 = the node size of BASELINE
# cfg-3..5: three two-way steps + remainder, more than one transfer posted ahead (ref loss.py:72-83,260-309)
@pytest.mark.parametrize("world,port", [(2, 29721), (3, 29722), (4, 29731), (8, 29732)])
def test_gather_features_matches_reference_ranks(golden_dir, tmp_path, world, port):
    gp = os.path.join(golden_dir, f"loss_world{world}.pt")
    _run(W.clip_gather_worker, world, port, gp, str(tmp_path))
    g = torch.load(gp, weights_only=False)
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), f"clip_rank{r}.pt"), weights_only=False)
        assert "clip_ll1_gwg1_tensor" in got
        for key, (loss, gm, gs, *extra) in got.items():
            rl, rgm, rgs, *rextra = g["per_rank"][r][key]
            assert abs(loss - rl) / abs(rl) < 1e-5, key
            _close(gm, rgm); _close(gs, rgs)
            assert len(extra) == len(rextra)
            for a, b in zip(extra, rextra):                      # gradient of a tensor logit scale
                assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (key, a, b)


@pytest.mark.parametrize("world,port", [(2, 29723), (3, 29724), (4, 29733), (8, 29734)])
def test_siglip_ring_matches_reference_ranks(golden_dir, tmp_path, world, port):
    gp = os.path.join(golden_dir, f"loss_world{world}.pt")
    _run(W.siglip_ring_worker, world, port, gp, str(tmp_path))
    g = torch.load(gp, weights_only=False)
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), f"siglip_rank{r}.pt"), weights_only=False)
        assert {"siglip_bidir0_tensor", "siglip_bidir1_tensor"} <= set(got)
        for key, (loss, gm, gs, *extra) in got.items():
            rl, rgm, rgs, *rextra = g["per_rank"][r][key]
            assert abs(loss - rl) / abs(rl) < 1e-5, key
            _close(gm, rgm, 1e-5); _close(gs, rgs, 1e-5)
            assert len(extra) == len(rextra)
            for a, b in zip(extra, rextra):                      # gradients of the tensor logit scale / bias (ref loss.py:241-245)
                assert abs(a - b) <= 2e-5 * max(1.0, abs(b)), (key, a, b)


def test_bucketed_gradient_allreduce(tmp_path):
    _run(W.allreduce_worker, 2, 29725, str(tmp_path))
    a = torch.load(os.path.join(str(tmp_path), "ar_rank0.pt"), weights_only=False)
    b = torch.load(os.path.join(str(tmp_path), "ar_rank1.pt"), weights_only=False)
    for i in range(3):
        mean = (a["mine"][i] + b["mine"][i]) / 2
        _close(a["reduced"][i], mean); _close(b["reduced"][i], mean)


def test_bf16_wire_format_of_the_gradient_allreduce(tmp_path):
    """ONEPROT_GRAD_COMM_DTYPE=bf16: the mean of the ranks' gradients to bf16 precision for the large tensors (buckets and overlapped arena ranges),
    exactly as before for the small ones."""
    _run(W.allreduce_bf16_worker, 2, 29727, str(tmp_path))
    a = torch.load(os.path.join(str(tmp_path), "arbf_rank0.pt"), weights_only=False)
    b = torch.load(os.path.join(str(tmp_path), "arbf_rank1.pt"), weights_only=False)
    for i in range(3):
        mean = (a["mine"][i] + b["mine"][i]) / 2
        assert torch.equal(a["reduced"][i], b["reduced"][i])
        if i == 1:
            _close(a["reduced"][i], mean)                                        # small parameters: fp32 on the wire
        else:
            err = (a["reduced"][i] - mean).abs()
            assert float(err.max()) <= 2 ** -7 * float(mean.abs().max()) and float(err.max()) > 1e-6       # bf16 on the wire (two roundings: each rank's share, the sum)
            assert torch.equal(a["reduced"][i], a["reduced"][i].to(torch.bfloat16).float())              # what came back is a bf16 value


def test_validation_loss_is_global_before_the_plateau_scheduler(tmp_path):
    """without a Trainer the monitored validation loss was rank-local, so ReduceLROnPlateau could decide differently per rank."""
    _run(W.val_plateau_worker, 2, 29726, str(tmp_path))
    a = torch.load(os.path.join(str(tmp_path), "val_rank0.pt"), weights_only=False)
    b = torch.load(os.path.join(str(tmp_path), "val_rank1.pt"), weights_only=False)
    assert a["seen"] == b["seen"] == [3.0, 3.0, 2.0]            # best-so-far of the GLOBAL epoch means (3, 3, 2), not of 3, 1, 2 / 3, 5, 2
    assert a["lrs"] == b["lrs"] == [1.0, 0.5, 0.5]              # epoch 1 is a plateau for everyone (patience 0), epoch 2 an improvement


def test_first_node_and_env_mapping(golden_dir, monkeypatch):
    from oneprot_amd import distributed as D
    with open(os.path.join(golden_dir, "distributed_cases.json")) as f:
        cases = json.load(f)
    for c in cases["first_node"]:
        monkeypatch.setenv("SLURM_JOB_NODELIST", c["nodelist"])
        assert D._get_first_node() == c["first"], c
    for c in cases["env"]:
        for k, v in dict(SLURM_JOB_NODELIST="jwb[0097,0101]", SLURM_NTASKS="8", SLURM_PROCID="3", SLURM_LOCALID="1", SYSTEMNAME=c["SYSTEMNAME"]).items():
            monkeypatch.setenv(k, v)
        D.init_distributed_mode(port=23456)
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            assert os.environ[k] == c[k], (k, c)
    import src.distributed as SD      # the reference's import path
    assert SD.init_distributed_mode is D.init_distributed_mode


def test_gradient_overlap_switch(monkeypatch):
    """ONEPROT_GRAD_OVERLAP: 1 (default) attaches distributed.GradOverlap to the towers, 0 leaves the arena gradients to the bucketed all-reduce after
    the backward; anything else is refused."""
    from oneprot_amd import distributed as D
    monkeypatch.delenv("ONEPROT_GRAD_OVERLAP", raising=False)
    assert D.grad_overlap_enabled() is True
    monkeypatch.setenv("ONEPROT_GRAD_OVERLAP", "0")
    assert D.grad_overlap_enabled() is False
    monkeypatch.setenv("ONEPROT_GRAD_OVERLAP", "yes")
    with pytest.raises(ValueError):
        D.grad_overlap_enabled()
