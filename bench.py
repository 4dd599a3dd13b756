#!/usr/bin/env python3
"""bench.py -- OneProt contrastive-alignment training sub-step on MI355X.

Metric (BASELINE.json): protein-pairs/sec/node, seq + struct-token, L=512, ESM-2-150M, at 1/2/4/8 GPUs.
One "step" = one iteration of the loop body of OneProtLitModule.training_step (ref oneprot_module.py:92-107) for the
seq<->struct_token pair: forward sequence encoder, forward struct-token encoder, zero_grad, CLIP loss (+0.01*L1),
backward, (gradient all-reduce when N>1), clip-norm 1.0, Adam step -- on synthetic ids resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]             (N>1: launched by torch.distributed.run, one rank/GPU)

Workload at every N: cfg-2 per GPU (B=256 pairs, L=512, ESM-2-150M x2, output_dim 1024, random-init weights, reference
default model flags: sequence encoder frozen, struct-token encoder trainable, loss CLIP local_loss+gather_with_grad,
use_l1_regularization) => weak scaling, global batch 256*N (cfg-3 at N=8).

Extra objects on the JSON line:
  roofline     dominant kernel = the bf16 MFMA NT GEMM (k_gemm_nt, FFN-1 launch [T x 640] x [2560 x 640]^T + bias + GELU):
               algorithmic FLOPs per launch (2*T*N*K) / mean launch duration measured with HIP events on the launch stream
               inside the timed region; peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md).
  cpu_baseline the CPU oracle (oracle/oneprot_oracle.py, fp32 torch restatement of the reference) timed on this box's host
               cores on a bounded sample of the same workload (reduced batch), rank 0 at N=1 only.
"""
import argparse
import functools
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

CFG150 = dict(layers=30, hidden=640, heads=20, ffn=2560, pad=1, mask=32, eps=1e-5)
PEAK_BF16_TFLOPS = 2500.0


def synth_ids(B, L, lo, hi, gen, device):
    ids = torch.randint(lo, hi + 1, (B, L), generator=gen)
    ids[:, 0] = 0
    ids[:, -1] = 2
    return ids.to(device)


def encoder_flops_fwd(cfg, L):
    d, f, n = cfg["hidden"], cfg["ffn"], cfg["layers"]
    return n * (8 * L * d * d + 4 * L * d * f + 4 * L * L * d)


def cpu_baseline(sample_pairs, L, threads):
    """Time the CPU oracle's training sub-step (same arithmetic, fp32, torch CPU) on `sample_pairs` pairs."""
    from oracle import oneprot_oracle as O
    torch.set_num_threads(threads)
    gen = torch.Generator().manual_seed(1881)

    def rand_sd(vocab, head):
        d, f, n = CFG150["hidden"], CFG150["ffn"], CFG150["layers"]
        sd = {"transformer.embeddings.word_embeddings.weight": torch.randn(vocab, d, generator=gen) * 0.02}
        for i in range(n):
            p = f"transformer.encoder.layer.{i}."
            for nm, shp in (("attention.self.query", (d, d)), ("attention.self.key", (d, d)), ("attention.self.value", (d, d)), ("attention.output.dense", (d, d)),
                            ("intermediate.dense", (f, d)), ("output.dense", (d, f))):
                sd[p + nm + ".weight"] = torch.randn(*shp, generator=gen) * 0.02
                sd[p + nm + ".bias"] = torch.zeros(shp[0])
            for nm in ("attention.LayerNorm", "LayerNorm"):
                sd[p + nm + ".weight"], sd[p + nm + ".bias"] = torch.ones(d), torch.zeros(d)
        sd["transformer.encoder.emb_layer_norm_after.weight"], sd["transformer.encoder.emb_layer_norm_after.bias"] = torch.ones(d), torch.zeros(d)
        sd["proj.0.weight"], sd["proj.0.bias"] = torch.ones(d), torch.zeros(d)
        if head == "linear":
            sd["proj.1.weight"] = torch.randn(1024, d, generator=gen) * 0.03
            sd["norm.1.log_logit_scale"] = torch.log(torch.tensor(1 / 0.07))
        else:
            h = (d + 1024) // 2
            sd["proj.1.weight"] = torch.randn(h, d, generator=gen) * 0.03
            sd["proj.3.weight"], sd["proj.3.bias"] = torch.ones(h), torch.zeros(h)
            sd["proj.4.weight"] = torch.randn(1024, h, generator=gen) * 0.03
        return sd

    sd_seq, sd_st = rand_sd(33, "mlp"), rand_sd(54, "linear")
    seq_ids = synth_ids(sample_pairs, L, 4, 23, gen, "cpu")
    st_ids = synth_ids(sample_pairs, L, 33, 52, gen, "cpu")
    spec_seq = dict(kind="esm", pooling="mean", proj_type="mlp", use_logit_scale=False)
    spec_st = dict(kind="esm", pooling="mean", proj_type="linear", use_logit_scale=True)
    run = lambda: O.train_substep(seq_ids, st_ids, sd_seq, sd_st, CFG150, CFG150, spec_seq, spec_st, use_l1=True, frozen_seq=True)
    t0 = time.perf_counter(); run(); t_first = time.perf_counter() - t0
    t0 = time.perf_counter(); run(); t = time.perf_counter() - t0
    return dict(value=round(sample_pairs / t, 4), unit="protein-pairs/sec", cores=threads, kind="port",
                sample=f"oracle train sub-step (fp32 torch CPU), ESM-2-150M x2, L={L}, batch {sample_pairs} pairs, frozen sequence encoder; "
                       f"1 warm-up ({t_first:.1f}s) + 1 timed run ({t:.1f}s)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="pairs per GPU")
    ap.add_argument("--seq-len", type=int, default=512)
    ap.add_argument("--model", default="facebook/esm2_t30_150M_UR50D")
    ap.add_argument("--train-seq", action="store_true", help="also train the sequence encoder (reference default: frozen)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-pairs", type=int, default=4)
    args = ap.parse_args()

    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("ONEPROT_ALLOW_RANDOM_INIT", "1")
    from oneprot_amd import distributed as D
    from oneprot_amd import hip
    rank, world, local = D.setup_process_group()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: for N>1 launch with python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    dev = torch.device("cuda", torch.cuda.current_device())
    hip.lib()      # fail loudly if the HIP library is missing

    import warnings
    warnings.filterwarnings("ignore", message=".*no weight file.*")
    from src.models.components.sequence_encoder import SequenceEncoder
    from src.models.components.struct_token_encoder import StructTokenEncoder
    from src.models.oneprot_module import OneProtLitModule
    from oneprot_amd.optim import FusedAdam

    torch.manual_seed(1881)        # identical weights on every rank
    seq = SequenceEncoder(args.model, output_dim=1024, pooling_type="mean", proj_type="mlp", use_lora=False, frozen=not args.train_seq)
    st = StructTokenEncoder(args.model, output_dim=1024, pooling_type="mean", proj_type="linear", use_logit_scale=True, learnable_logit_scale=False)
    module = OneProtLitModule(components={"sequence": seq, "struct_token": st}, optimizer=functools.partial(FusedAdam, lr=1e-3, weight_decay=0.0),
                              loss_fn="CLIP", use_l1_regularization=True, local_loss=True, gather_with_grad=True).to(dev)
    module.train()
    B, L = args.batch, args.seq_len
    gen = torch.Generator().manual_seed(1881 + rank)
    seq_ids = synth_ids(B, L, 4, 23, gen, dev)
    st_ids = synth_ids(B, L, 33, 52, gen, dev)
    batch = {"struct_token": (seq_ids, st_ids, "struct_token", None)}

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        module.training_step(batch, 0)
    barrier()
    # live per-launch timing of the dominant kernel (FFN-1 GEMM, bias+GELU epilogue) with events on the launch stream
    hip.profile_begin("oneprot_gemm_bf16_nt", epilogue=hip.EPI_BIAS_GELU)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = module.training_step(batch, 0)
    barrier()
    elapsed = time.perf_counter() - t0
    launches_ms = hip.profile_end()
    el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(el)
    loss_val = float(loss)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        T = B * L
        cfg = dict(CFG150) if "150M" in args.model else None
        d, f = seq.transformer.d, seq.transformer.f
        gemm_flops = 2.0 * T * f * d
        gemm_ms = sum(launches_ms) / max(len(launches_ms), 1)
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        fwd = encoder_flops_fwd(dict(hidden=d, ffn=f, layers=seq.transformer.n_layers), L)
        step_flops = B * fwd * ((3 if args.train_seq else 1) + 3)      # sequence encoder fwd (+2x bwd if trained) + struct-token encoder fwd + bwd
        # HBM/fabric bytes per launch of the dominant kernel: PMC passes cannot run inside this process, so the figure measured with
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, gfx950 x2 correction on FETCH_SIZE) is read from profiles/.
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                traffic = json.load(f)["traffic_bytes_per_launch"] / 1e9
        except Exception:
            pass
        out = {
            "metric": "protein-pairs/sec/node (seq+struct-token, L=512, ESM-2-150M)", "value": round(value, 2), "unit": "protein-pairs/sec/node",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"cfg-{'2' if world == 1 else '3-shaped'}: seq<->struct_token sub-step, {args.model} x2, L={L}, {B} pairs/GPU, global batch {B * world}, "
                                   f"output_dim 1024, CLIP local_loss+gather_with_grad, L1 0.01, Adam 1e-3, clip 1.0, "
                                   f"sequence encoder {'trainable' if args.train_seq else 'frozen (reference default)'}, random-init weights",
                       "global_batch": B * world, "seq_len": L, "parallelism": f"dp{world}", "frozen_sequence_encoder": not args.train_seq,
                       "loss": round(loss_val, 5)},
            "step_tflops_per_gpu": round(step_flops / (ms_step * 1e-3) / 1e12, 1),
            "roofline": {"bound": "mfma", "kernel": "k_gemm_nt<BIAS_GELU> FFN-1 [T,640]x[2560,640]^T", "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_unit": "GB/launch (rocprofv3 PMC, profiles/r01_pmc_traffic.json; algorithmic 1.51)", "launches_timed": len(launches_ms),
                         "avg_launch_ms": round(gemm_ms, 4), "flops_per_launch": gemm_flops},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(len(os.sched_getaffinity(0)), 16)
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_pairs, L, threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
