"""RCCL (backend "nccl") rank worker for tests/test_multirank_gpu.py: one process per GPU (`world` ranks on `world` different GPUs; world 1 runs
the same calls through RCCL on the single GPU of the test box).  Exercises exactly the product calls that gloo cannot:
`reduce_scatter_tensor` + `all_gather_into_tensor` (gather_features with gradient, ref loss.py:31-33), asynchronous `ReduceOp.AVG` ranges
(distributed.GradOverlap), `allreduce_gradients`, and the SigLIP neighbour exchange (world > 1).  Every rank re-creates all ranks' seeded
tensors, so expectations are computed locally and asserted in the worker; a non-zero exit code fails the test."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from oneprot_amd import distributed as D
    from oneprot_amd.loss import gather_features, neighbour_exchange_with_grad, neighbour_exchange_bidir_with_grad
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    if world == 1:
        dist.init_process_group(backend="nccl", rank=0, world_size=1)       # setup_process_group only initialises for world > 1
    else:
        r, w, _ = D.setup_process_group()
        assert (r, w) == (rank, world)
    assert dist.get_backend() == "nccl"
    B, Dm = 5, 16
    feats = [(torch.randn(B, Dm, generator=torch.Generator().manual_seed(10 + r)), torch.randn(B, Dm, generator=torch.Generator().manual_seed(20 + r)))
             for r in range(world)]
    # ---- gather_features with gradient: packed all_gather_into_tensor forward, reduce_scatter_tensor backward
    m = feats[rank][0].clone().to(dev).requires_grad_(True)
    s = feats[rank][1].clone().to(dev).requires_grad_(True)
    all_m, all_s = gather_features(m, s, local_loss=True, gather_with_grad=True, rank=rank, world_size=world)
    exp_m, exp_s = torch.cat([f[0] for f in feats]), torch.cat([f[1] for f in feats])
    assert torch.equal(all_m.detach().cpu(), exp_m) and torch.equal(all_s.detach().cpu(), exp_s)
    # every rank back-propagates (rank+1) * w . gathered: the reduce-scatter must hand each owner the SUM over ranks of its slice's gradient
    wm, ws_ = torch.randn(world * B, Dm, generator=torch.Generator().manual_seed(30)), torch.randn(world * B, Dm, generator=torch.Generator().manual_seed(31))
    ((all_m * wm.to(dev)).sum() * (rank + 1) + (all_s * ws_.to(dev)).sum() * (rank + 1)).backward()
    tot = sum(r + 1 for r in range(world))
    assert torch.allclose(m.grad.cpu(), tot * wm[rank * B:(rank + 1) * B], rtol=1e-6, atol=1e-6)
    assert torch.allclose(s.grad.cpu(), tot * ws_[rank * B:(rank + 1) * B], rtol=1e-6, atol=1e-6)
    # ---- gather without gradient (dist.all_gather semantics, ref loss.py:35-44)
    all_m2, _ = gather_features(m.detach(), s.detach(), local_loss=False, gather_with_grad=False, rank=rank, world_size=world)
    assert torch.equal(all_m2.cpu(), exp_m)
    # ---- asynchronous mean all-reduce of arena-gradient ranges (ReduceOp.AVG) + bucketed reduce of the rest
    n = 300_000
    grads = [torch.randn(n, generator=torch.Generator().manual_seed(40 + r)) for r in range(world)]
    mean = sum(grads) / world
    p = torch.nn.Parameter(torch.zeros(n, device=dev))
    gflat = grads[rank].clone().to(dev)
    ov = D.GradOverlap()
    assert ov.use_avg
    ov.world = max(ov.world, 1)
    ov.reduce_range(p, gflat, n // 2, n)
    ov.reduce_range(p, gflat, 0, n // 2)
    p.grad = gflat
    small = torch.nn.Parameter(torch.zeros(7, device=dev))
    small.grad = torch.full((7,), float(rank + 1), device=dev)
    big = torch.nn.Parameter(torch.zeros(400_000, device=dev))
    big.grad = torch.full((400_000,), float(rank + 1), device=dev)
    if world > 1:
        D.allreduce_gradients([p, small, big], bucket_bytes=1 << 20)
    else:                                   # allreduce_gradients returns early for one rank: wait for the range handles directly
        for h in p._oneprot_pending_reduce:
            h.wait()
    torch.cuda.synchronize()
    assert torch.allclose(p.grad.cpu(), mean, rtol=1e-6, atol=1e-6), "ReduceOp.AVG ranges"
    if world > 1:
        avg = tot / world
        assert torch.allclose(small.grad.cpu(), torch.full((7,), avg)) and torch.allclose(big.grad.cpu(), torch.full((400_000,), avg))
        # ---- SigLIP ring exchanges with autograd (ref loss.py:116-201)
        left, right = (rank - 1) % world, (rank + 1) % world
        t = feats[rank][1].clone().to(dev).requires_grad_(True)
        got = neighbour_exchange_with_grad(left, right, t)
        assert torch.equal(got.detach().cpu(), feats[left][1])
        (got * (rank + 1)).sum().backward()                      # the gradient travels the opposite way: from the right neighbour
        assert torch.allclose(t.grad.cpu(), torch.full((B, Dm), float(right + 1)))
        a = feats[rank][0].clone().to(dev)
        fr, fl = neighbour_exchange_bidir_with_grad(left, right, a, a)
        assert torch.equal(fr.cpu(), feats[right][0]) and torch.equal(fl.cpu(), feats[left][0])
    # ---- the same exchange through the C-ABI RCCL wrappers (include/oneprot_comm.h, oneprot_amd/comm.py): own communicator, no torch.distributed
    from oneprot_amd import comm as C
    from oneprot_amd import loss as LM
    box = [C.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    cm = C.RcclComm(world, rank, box[0])
    LM.set_feature_comm(cm)
    m2 = feats[rank][0].clone().to(dev).requires_grad_(True)
    s2 = feats[rank][1].clone().to(dev).requires_grad_(True)
    am, as_ = gather_features(m2, s2, local_loss=True, gather_with_grad=True, rank=rank, world_size=world)
    assert torch.equal(am.detach().cpu(), exp_m) and torch.equal(as_.detach().cpu(), exp_s)
    ((am * wm.to(dev)).sum() * (rank + 1) + (as_ * ws_.to(dev)).sum() * (rank + 1)).backward()
    assert torch.allclose(m2.grad.cpu(), tot * wm[rank * B:(rank + 1) * B], rtol=1e-6, atol=1e-6)
    assert torch.allclose(s2.grad.cpu(), tot * ws_[rank * B:(rank + 1) * B], rtol=1e-6, atol=1e-6)
    # ---- point-to-point exchange of the C ABI (oneprot_comm_send_recv): to the right neighbour, from the left one (world 1: a self-exchange)
    a = (torch.arange(8, dtype=torch.float32) + 100 * rank).to(dev)
    b = torch.empty_like(a)
    cm.send_recv(a, (rank + 1) % world, b, (rank - 1) % world)
    torch.cuda.synchronize()
    assert torch.equal(b.cpu(), torch.arange(8, dtype=torch.float32) + 100 * ((rank - 1) % world)), "send_recv"
    c2, d2 = torch.empty_like(a), torch.empty_like(a)
    cm.exchange([(a, (rank + 1) % world, c2, (rank - 1) % world), (a * 2, (rank - 1) % world, d2, (rank + 1) % world)])      # both directions in one group
    torch.cuda.synchronize()
    assert torch.equal(c2.cpu(), b.cpu()) and torch.equal(d2.cpu(), 2 * (torch.arange(8, dtype=torch.float32) + 100 * ((rank + 1) % world))), "grouped exchange"
    if world > 1:
        # ---- the whole multi-rank SigLIP loss (direct peer exchanges + reduce-scatter backward) with the HIP block kernels, on both transports,
        # against the oracle's blocks evaluated for ALL ranks on the host; rank r back-propagates (r + 1) * loss_r (per-rank upstream scaling)
        from oracle import oneprot_oracle as O
        ms = [f[0].clone().requires_grad_(True) for f in feats]
        ss = [f[1].clone().requires_grad_(True) for f in feats]
        per = [O.siglip_block(ms[r], ss[r], 1.0, None, False) + sum(O.siglip_block(ms[r], ss[j], 1.0, None, True) for j in range(world) if j != r) for r in range(world)]
        sum((r + 1) * per[r] for r in range(world)).backward()
        for transport in ("rccl_c_abi", "torch_distributed"):
            LM.set_feature_comm(cm if transport == "rccl_c_abi" else None)
            for bidir in (True, False):
                fn = LM.SigLipLoss(cache_labels=True, rank=rank, world_size=world, bidir=bidir)
                m3 = feats[rank][0].clone().to(dev).requires_grad_(True)
                s3 = feats[rank][1].clone().to(dev).requires_grad_(True)
                l3 = fn(m3, s3)
                (l3 * (rank + 1)).backward()
                torch.cuda.synchronize()
                assert abs(float(l3) - float(per[rank])) < 1e-4 * abs(float(per[rank])), (transport, bidir, float(l3), float(per[rank]))
                assert torch.allclose(m3.grad.cpu(), ms[rank].grad, rtol=1e-4, atol=1e-5), (transport, bidir, "dm")
                assert torch.allclose(s3.grad.cpu(), ss[rank].grad, rtol=1e-4, atol=1e-5), (transport, bidir, "ds")
        LM.set_feature_comm(cm)
    gbuf = grads[rank].clone().to(dev)
    cm.all_reduce_(gbuf, average=True)
    hb = torch.arange(16, dtype=torch.float32).to(torch.bfloat16).to(dev) * (rank + 1)
    cm.all_reduce_(hb)
    torch.cuda.synchronize()
    assert torch.allclose(gbuf.cpu(), mean, rtol=1e-6, atol=1e-6)
    assert torch.equal(hb.float().cpu(), torch.arange(16, dtype=torch.float32) * tot)
    LM.set_feature_comm(None)
    cm.destroy()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rccl worker rank {rank}/{world} ok", flush=True)


if __name__ == "__main__":
    main()
