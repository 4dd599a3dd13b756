"""Worker functions for the world_size>1 CPU (gloo) tests.  Kept in an importable module so `spawn` can find them."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from oneprot_amd import distributed as D
    r, w, _ = D.setup_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    return D


def clip_gather_worker(rank, world, port, golden_path, out_dir):
    """gather_features (packed all-gather, reduce-scatter backward) composed with the oracle's CLIP arithmetic must reproduce what the
    reference produced on real gloo ranks, for all four (local_loss, gather_with_grad) combinations."""
    D = _init(rank, world, port)
    from oneprot_amd.loss import gather_features
    from oracle import oneprot_oracle as O
    g = torch.load(golden_path, weights_only=False)
    res = {}
    for ll in (False, True):
        for gwg in (False, True):
            m = g["m"][rank].clone().requires_grad_(True)
            s = g["s"][rank].clone().requires_grad_(True)
            all_m, all_s = gather_features(m, s, local_loss=ll, gather_with_grad=gwg, rank=rank, world_size=world)
            loss = O.clip_loss(m, s, 1.0, all_m, all_s, rank, world, ll)
            loss.backward()
            res[f"clip_ll{int(ll)}_gwg{int(gwg)}"] = (loss.detach(), m.grad.clone(), s.grad.clone())
    torch.save(res, os.path.join(out_dir, f"clip_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def siglip_ring_worker(rank, world, port, golden_path, out_dir):
    """SigLipLoss' neighbour-exchange ring (uni- and bi-directional, with autograd) with the block arithmetic swapped for the oracle's."""
    D = _init(rank, world, port)
    from oneprot_amd import loss as L
    from oracle import oneprot_oracle as O
    g = torch.load(golden_path, weights_only=False)
    res = {}
    for bidir in (False, True):
        fn = L.SigLipLoss(cache_labels=True, rank=rank, world_size=world, bidir=bidir)
        fn._loss = lambda m, s, scale, bias=None, negative_only=False: O.siglip_block(m, s, scale, bias, negative_only)
        m = g["m"][rank].clone().requires_grad_(True)
        s = g["s"][rank].clone().requires_grad_(True)
        loss = fn(m, s, logit_scale=1.0)
        loss.backward()
        res[f"siglip_bidir{int(bidir)}"] = (loss.detach(), m.grad.clone(), s.grad.clone())
    torch.save(res, os.path.join(out_dir, f"siglip_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def allreduce_worker(rank, world, port, out_dir):
    D = _init(rank, world, port)
    torch.manual_seed(rank)
    big = torch.nn.Parameter(torch.zeros(600_000))
    small = [torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(3, 5))]
    unused = torch.nn.Parameter(torch.zeros(4))
    for p in [big] + small:
        p.grad = torch.randn_like(p)
    mine = [p.grad.clone() for p in [big] + small]
    D.allreduce_gradients([big] + small + [unused], bucket_bytes=1 << 20)
    assert unused.grad is None
    torch.save({"mine": mine, "reduced": [p.grad.clone() for p in [big] + small]}, os.path.join(out_dir, f"ar_rank{rank}.pt"))
    assert D.get_rank() == rank and D.get_world_size() == world and D.is_main_process() == (rank == 0)
    dist.barrier()
    dist.destroy_process_group()
