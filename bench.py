#!/usr/bin/env python3
"""bench.py -- OneProt contrastive-alignment training step on MI355X.

Metric (BASELINE.json): protein-pairs/sec/node, seq + struct-token, L=512, ESM-2-150M, at 1/2/4/8 GPUs.
One "step" = one call of OneProtLitModule.training_step on one CombinedLoader batch, i.e. one iteration of the loop body
ref oneprot_module.py:92-107 per modality in the batch: forward sequence encoder, forward modality encoder, zero_grad,
CLIP loss (+0.01*L1), backward, (gradient all-reduce when N>1), clip-norm 1.0, Adam step -- on synthetic ids resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--pair struct_token|text|roundrobin]

N>1: bench.py starts its N ranks itself (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...`, one rank per GPU over
RCCL) before anything touches the GPU, or runs as one of them when a launcher has already set RANK / WORLD_SIZE.

Workloads (all weak-scaling: `--batch` pairs per GPU per modality, default 256; reference default model flags):
  struct_token (default)  cfg-2 (N=1) / cfg-3 (N=8): ESM-2-150M x2, L=512; sequence encoder frozen, struct-token encoder trainable
  text                    cfg-4: ESM-2-150M (L=512, frozen) <-> BERT-base text tower (T=256, frozen: text.yaml), heads train
  roundrobin              cfg-5: ESM-2-650M anchor (attention1d pooling, linear head, frozen, train_ddp_1.yaml:45-49) against struct_token
                          (ESM-2-35M, trainable), text (BERT-base, frozen) and pocket in one mixed batch -> three optimiser sub-steps per step.
                          The pocket modality runs through StructEncoder (head / normalisation / logit scale on the HIP kernels) around a
                          STAND-IN opaque torch encoder: the reference's ProNet GNN is un-vendored third-party code and is not reproduced

Extra objects on the JSON line:
  roofline     dominant kernel = the bf16 MFMA NT GEMM with the bias+erf-GELU epilogue (FFN-1 launches): algorithmic FLOPs (2*M*N*K of every
               timed launch) / their summed durations, measured with HIP events on the launch stream inside the timed region;
               peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md); traffic = HBM bytes per launch from the committed rocprofv3 PMC passes.
  encoder_fwd  the north_star figure: ESM-2 sequence-encoder forward (frozen tower, embedding + all layers) in TFLOP/s and as a fraction of
               the MFMA peak, timed with events after the timed region; MFMA-busy % from the committed PMC pass.
  kernels      per-kernel-family rates from one extra, event-bracketed step after the timed region: MFMA kernels in TFLOP/s, LayerNorm / Adam
               in HBM GB/s against 8 TB/s.
  board        board power and shader clock of rank 0's GPU over the timed steps (amdgpu hwmon files, a helper thread every 50 ms): the step runs at
               the board's power cap with the clock below its 2.4 GHz maximum; the bf16 MFMA peak at the measured clock and the roofline / step
               fractions against THAT (roofline.peak stays the 2.5 PFLOP/s spec figure).
  other_workloads  cfg-4 (seq <-> text) and the cfg-5-shaped round-robin step (650M anchor, batch 128): pairs/s (median of 5 steps each, 2 warm-ups),
               run after the timed region (default workload, N=1 only).
  cpu_baseline the CPU oracle (oracle/oneprot_oracle.py, fp32 torch restatement of the reference) timed on this box's host cores on a
               bounded sample of the same workload (reduced batch; 1 warm-up + 5 timed runs, median), plus the cfg-1 CPU headline point
               (ESM-2-8M x2, L=128, batch 32); rank 0 at N=1 only.
"""
import argparse
import functools
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PROFILE_ROUND = "r06"         # committed rocprofv3 summaries this file quotes (profiles/<round>_*.json, tools/profile_round.sh)
PEAK_BF16_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0
ESM = {"8M": "facebook/esm2_t6_8M_UR50D", "35M": "facebook/esm2_t12_35M_UR50D", "150M": "facebook/esm2_t30_150M_UR50D", "650M": "facebook/esm2_t33_650M_UR50D"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pair", default="struct_token", choices=["struct_token", "text", "roundrobin"])
    ap.add_argument("--batch", type=int, default=256, help="pairs per GPU (per modality)")
    ap.add_argument("--seq-len", type=int, default=512)
    ap.add_argument("--text-len", type=int, default=256)
    ap.add_argument("--model-seq", default=None, help="sequence encoder (default: ESM-2-150M; roundrobin: ESM-2-650M)")
    ap.add_argument("--model-mod", default=None, help="struct-token encoder (default: ESM-2-150M; roundrobin: ESM-2-35M)")
    ap.add_argument("--model", default=None, help="shorthand: the same ESM-2 model for both towers")
    ap.add_argument("--train-seq", action="store_true", help="also train the sequence encoder (reference default: frozen)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the encoder_fwd / kernels measurements after the timed region")
    ap.add_argument("--cpu-sample-pairs", type=int, default=4)
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the cfg-4 / cfg-5-shaped runs after the timed region")
    ap.add_argument("--no-mfma-busy", action="store_true", help="skip the rocprofv3 child that measures the encoder forward's MFMA-pipe utilisation (N = 1, default workload)")
    return ap.parse_args()


def launch_ranks(args):
    """N>1 and no launcher environment: become the launcher.  Nothing in this process has touched the GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC for RCCL (see oneprot_amd/distributed.py)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode              # rank 0 prints the JSON line on the inherited stdout


def synth_ids(B, L, lo, hi, gen, device, cls=0, eos=2):
    import torch
    ids = torch.randint(lo, hi + 1, (B, L), generator=gen)
    ids[:, 0] = cls
    ids[:, -1] = eos
    return ids.to(device)


def tower_fwd_flops(tr, L):
    """algorithmic forward FLOPs per sequence: n * (8 L d^2 + 4 L d f + 4 L^2 d)"""
    d, f, n = tr.d, tr.f, tr.n_layers
    return n * (8 * L * d * d + 4 * L * d * f + 4 * L * L * d)


def tower_cfg(tr):
    c = tr.config
    cfg = dict(layers=tr.n_layers, hidden=tr.d, heads=tr.H, ffn=tr.f, pad=c.pad_token_id, eps=c.layer_norm_eps)
    if getattr(c, "model_type", "esm") == "bert":
        cfg.update(vocab=c.vocab_size, max_pos=c.max_position_embeddings)
    else:
        cfg.update(mask=c.mask_token_id)
    return cfg


def build_workload(args, dev, rank):
    """-> dict(module, batch, subs=[(modality, seq_ids, mod_ids)], specs, names, desc, metric)"""
    import torch
    from src.models.components.sequence_encoder import SequenceEncoder
    from src.models.components.struct_token_encoder import StructTokenEncoder
    from src.models.components.text_encoder import TextEncoder
    from src.models.oneprot_module import OneProtLitModule
    from oneprot_amd.optim import FusedAdam
    B, L, T = args.batch, args.seq_len, args.text_len
    rr = args.pair == "roundrobin"
    name_seq = args.model_seq or args.model or (ESM["650M"] if rr else ESM["150M"])
    name_mod = args.model_mod or args.model or (ESM["35M"] if rr else ESM["150M"])
    torch.manual_seed(1881)        # identical weights on every rank
    if rr:      # train_ddp_1.yaml:45-49
        seq = SequenceEncoder(name_seq, output_dim=1024, pooling_type="attention1d", proj_type="linear", use_lora=False, frozen=not args.train_seq)
        spec_seq = dict(kind="esm", pooling="attention1d", proj_type="linear", use_logit_scale=False)
    else:       # sequence.yaml
        seq = SequenceEncoder(name_seq, output_dim=1024, pooling_type="mean", proj_type="mlp", use_lora=False, frozen=not args.train_seq)
        spec_seq = dict(kind="esm", pooling="mean", proj_type="mlp", use_logit_scale=False)
    comps, specs = {"sequence": seq}, {"sequence": spec_seq}
    if args.pair in ("struct_token", "roundrobin"):
        comps["struct_token"] = StructTokenEncoder(name_mod, output_dim=1024, pooling_type="mean", proj_type="linear", use_logit_scale=True, learnable_logit_scale=False)
        specs["struct_token"] = dict(kind="esm", pooling="mean", proj_type="linear", use_logit_scale=True)
    if args.pair in ("text", "roundrobin"):
        comps["text"] = TextEncoder("microsoft/BiomedNLP-BiomedBERT-base-uncased-abstract-fulltext", output_dim=1024, pooling_type="cls", proj_type="mlp",
                                    use_logit_scale=True, learnable_logit_scale=False, frozen=True, use_lora=False)
        specs["text"] = dict(kind="bert", pooling="cls", proj_type="mlp", use_logit_scale=True)
    if rr:       # pocket.yaml: StructEncoder(encoder=<opaque GNN>, proj_type linear, logit scale on); the GNN itself is a labelled stand-in
        from src.models.components.struct_graph_encoder import StructEncoder
        from oneprot_amd.data import StandInGraphEncoder
        comps["pocket"] = StructEncoder(StandInGraphEncoder(16, 256, 1024), output_dim=1024, proj_type="linear", use_logit_scale=True, learnable_logit_scale=False, dropout=0.25)
        specs["pocket"] = dict(kind="opaque_graph", proj_type="linear", use_logit_scale=True)
    module = OneProtLitModule(components=comps, optimizer=functools.partial(FusedAdam, lr=1e-3, weight_decay=0.0), loss_fn="CLIP",
                              use_l1_regularization=True, local_loss=True, gather_with_grad=True).to(dev)
    module.train()
    gen = torch.Generator().manual_seed(1881 + rank)
    subs = []
    for m in comps:
        if m == "sequence":
            continue
        seq_ids = synth_ids(B, L, 4, 23, gen, dev)
        if m == "pocket":
            mod_ids = torch.randn(B, 64, 16, generator=gen).to(dev)          # 64 nodes x 16 descriptors per pocket: input of the stand-in encoder
        elif m == "text":
            mod_ids = synth_ids(B, T, 5, 30521, gen, dev, cls=2, eos=3)
        else:
            mod_ids = synth_ids(B, L, 33, 52, gen, dev)
        subs.append((m, seq_ids, mod_ids))
    batch = {m: (s, x, m, None) for m, s, x in subs}
    short = lambda n: n.split("/")[-1]
    if args.pair == "struct_token":
        desc = f"seq<->struct_token sub-step, {name_seq} / {name_mod}, L={L}"
        metric = f"protein-pairs/sec/node (seq+struct-token, L={L}, {short(name_seq).split('_')[2] if 'esm2' in name_seq else short(name_seq)})"
        if (name_seq, name_mod, L) == (ESM["150M"], ESM["150M"], 512):
            metric = "protein-pairs/sec/node (seq+struct-token, L=512, ESM-2-150M)"
    elif args.pair == "text":
        desc = f"seq<->text sub-step, {name_seq} (L={L}) / BERT-base text tower (T={T}), both transformers frozen (sequence.yaml:12, text.yaml:12)"
        metric = f"protein-pairs/sec/node (seq+text, L={L}/T={T}, {short(name_seq)} + BERT-base)"
    else:
        desc = (f"mixed-batch round-robin step = 3 sub-steps (struct_token, text, pocket), anchor {name_seq} attention1d+linear frozen, struct_token {name_mod}, "
                f"text BERT-base frozen, pocket = StructEncoder around a STAND-IN opaque torch encoder (ProNet is un-vendored, not reproduced), L={L}/T={T}")
        metric = f"protein-pairs/sec/node (round-robin seq+struct-token / seq+text / seq+pocket[stand-in], anchor {short(name_seq)})"
    return dict(module=module, batch=batch, subs=subs, specs=specs, names=dict(seq=name_seq, mod=name_mod), desc=desc, metric=metric)


def step_flops(module, subs, L_of):
    """algorithmic FLOPs of one step: per sub-step, each tower's forward (+2x for its backward when its transformer is trainable)"""
    total = 0
    for m, seq_ids, mod_ids in subs:
        for name, ids in (("sequence", seq_ids), (m, mod_ids)):
            if not hasattr(module.network[name], "transformer"):      # opaque stand-in encoder: no transformer FLOPs to count
                continue
            tr = module.network[name].transformer
            mult = 3 if tr.flat.requires_grad else 1
            total += ids.shape[0] * tower_fwd_flops(tr, ids.shape[1]) * mult
    return total


def summarise_kernels(prof):
    """per-kernel-family rates from hip.profile_end() of one step: [(ms, scalar args + (number of tensor args,))] per entry point"""
    prof = {k: [(ms, sc[:-1]) for ms, sc in v] for k, v in prof.items()}          # the trailing tensor count is not a scalar argument
    out = []

    def add(kernel, items, work, unit, peak):
        if not items:
            return
        t = sum(ms for ms, _ in items) * 1e-3
        rate = work / t / (1e12 if unit == "TFLOP/s" else 1e9)
        out.append({"kernel": kernel, "launches": len(items), "avg_ms": round(t / len(items) * 1e3, 4), "rate": round(rate, 1), "unit": unit,
                    "frac_of_peak": round(rate / peak, 4)})
    epi_names = {0: "bf16 (dgrad)", 1: "f32", 2: "bias+GELU (FFN-1)", 3: "bias+residual (out-proj / FFN-2)", 4: "QKV+RoPE", 5: "GELU' (FFN-2 dgrad)"}
    groups = {}
    for ms, sc in prof.get("oneprot_gemm_bf16_nt", []):
        M, N, K, epi = sc[0], sc[1], sc[2], sc[5]
        groups.setdefault((epi, N, K), []).append((ms, 2.0 * M * N * K))
    for (epi, N, K), items in sorted(groups.items()):
        add(f"NT GEMM {epi_names.get(epi, epi)} N={N} K={K}", items, sum(w for _, w in items), "TFLOP/s", PEAK_BF16_TFLOPS)
    for name, label in (("oneprot_gemm_bf16_nt_resid_ln", "NT GEMM + residual + LayerNorm, full-row kernel (out-proj)"),
                        ("oneprot_gemm_bf16_nt_resid_ln8", "NT GEMM + residual + LayerNorm across work-groups (FFN-2 + next LN)")):
        fg = {}
        for ms, sc in prof.get(name, []):
            fg.setdefault((sc[1], sc[2]), []).append((ms, 2.0 * sc[0] * sc[1] * sc[2]))
        for (N, K), items in sorted(fg.items()):
            add(f"{label} N={N} K={K}", items, sum(w for _, w in items), "TFLOP/s", PEAK_BF16_TFLOPS)
    tn = [(ms, 2.0 * sc[0] * sc[1] * sc[2]) for ms, sc in prof.get("oneprot_gemm_bf16_tn", [])]
    add("TN GEMM (weight gradients, incl. slab reduce)", tn, sum(w for _, w in tn), "TFLOP/s", PEAK_BF16_TFLOPS)
    af = [(ms, 4.0 * sc[-4] * sc[-3] * sc[-2] * sc[-2] * sc[-1]) for ms, sc in prof.get("oneprot_attn_fwd", [])]
    add("k_attn_fwd", af, sum(w for _, w in af), "TFLOP/s", PEAK_BF16_TFLOPS)
    ab = [(ms, 8.0 * sc[-4] * sc[-3] * sc[-2] * sc[-2] * sc[-1]) for ms, sc in prof.get("oneprot_attn_bwd", [])]
    add("k_attn_bwd (fused short-sequence kernel, or dq + dkv; algorithmic 2x forward)", ab, sum(w for _, w in ab), "TFLOP/s", PEAK_BF16_TFLOPS)
    big = lambda items, idx: [(ms, sc) for ms, sc in items if sc[idx] >= 4096]          # the [T, d] launches, not the [B, d] head ones
    lf = [(ms, 6.0 * sc[-3] * sc[-2]) for ms, sc in big(prof.get("oneprot_layernorm_fwd", []), -3)]
    add("k_layernorm_fwd (fp32 in, bf16 out: 6 B/elem)", lf, sum(w for _, w in lf), "GB/s", PEAK_HBM_GBS)
    lb = [(ms, 14.0 * sc[-3] * sc[-2]) for ms, sc in big(prof.get("oneprot_layernorm_bwd", []), -3)]
    add("k_layernorm_bwd (14 B/elem)", lb, sum(w for _, w in lb), "GB/s", PEAK_HBM_GBS)
    ad = [(ms, 28.0 * sc[0]) for ms, sc in prof.get("oneprot_adam_step", []) if sc[0] >= (1 << 20)]
    add("k_adam (28 B/param)", ad, sum(w for _, w in ad), "GB/s", PEAK_HBM_GBS)
    return out


def cpu_baseline(work, sample_pairs, threads):
    """Time the CPU oracle (same arithmetic, fp32, torch CPU) on `sample_pairs` pairs of this workload: 1 warm-up + 5 timed, median."""
    import torch
    from oracle import oneprot_oracle as O
    torch.set_num_threads(threads)
    module, subs, specs = work["module"], work["subs"], work["specs"]
    sds = {k: {n: v.detach().cpu() for n, v in enc.state_dict().items()} for k, enc in module.network.items()}
    cfgs = {k: (tower_cfg(enc.transformer) if hasattr(enc, "transformer") else None) for k, enc in module.network.items()}
    frozen = tuple(k for k, enc in module.network.items() if hasattr(enc, "transformer") and not enc.transformer.flat.requires_grad)
    n = sample_pairs
    batches = [{m: (s[:n].cpu(), x[:n].cpu()) for m, s, x in subs}]
    run = lambda: O.train_round_robin(batches, sds, cfgs, specs, use_l1=True, frozen=frozen)      # one sub-step per modality, clip 1.0, Adam
    t0 = time.perf_counter(); run(); t_first = time.perf_counter() - t0
    times = []
    for _ in range(5):
        t0 = time.perf_counter(); run(); times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    out = dict(value=round(n * len(subs) / med, 4), unit="protein-pairs/sec", cores=threads, kind="port",
               sample=f"oracle training step (fp32 torch CPU) of this workload at batch {n} pairs per modality; 1 warm-up ({t_first:.1f}s) + 5 timed runs, "
                      f"median {med:.2f}s (min {min(times):.2f}, max {max(times):.2f})")
    return out


def cpu_cfg1(threads):
    """BASELINE cfg-1, the reference's own CPU-runnable case: ESM-2-8M x2, L=128, batch 32, full training sub-step on the host cores."""
    import torch
    from oracle import oneprot_oracle as O
    from src.models.components.sequence_encoder import SequenceEncoder
    from src.models.components.struct_token_encoder import StructTokenEncoder
    torch.set_num_threads(threads)
    torch.manual_seed(1881)
    seq = SequenceEncoder(ESM["8M"], output_dim=1024, pooling_type="mean", proj_type="mlp", use_lora=False, frozen=True)
    st = StructTokenEncoder(ESM["8M"], output_dim=1024, pooling_type="mean", proj_type="linear", use_logit_scale=True)
    sd_seq, sd_st = ({k: v.detach().clone() for k, v in e.state_dict().items()} for e in (seq, st))
    cfg = tower_cfg(seq.transformer)
    gen = torch.Generator().manual_seed(1881)
    a, b = synth_ids(32, 128, 4, 23, gen, "cpu"), synth_ids(32, 128, 33, 52, gen, "cpu")
    run = lambda: O.train_substep(a, b, sd_seq, sd_st, cfg, cfg, dict(kind="esm", pooling="mean", proj_type="mlp", use_logit_scale=False),
                                  dict(kind="esm", pooling="mean", proj_type="linear", use_logit_scale=True), use_l1=True, frozen_seq=True)
    for _ in range(3):
        run()
    times = []
    for _ in range(5):
        t0 = time.perf_counter(); run(); times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    return dict(value=round(32 / med, 2), unit="protein-pairs/sec", cores=threads, kind="port",
                sample=f"cfg-1: ESM-2-8M x2, L=128, batch 32, frozen sequence encoder, oracle training sub-step; 3 warm-up + 5 timed, median {med:.3f}s")


def host_cpu_share():
    """(threads to use, description): the CPU time this process can actually get.  The affinity mask of a GPU box shows every core of the machine
    (256), but the container's cgroup quota is a fraction of it (16 CPUs per GPU on the pool this was written on): a torch CPU run on 256 threads
    inside a 16-CPU quota does not finish.  cgroup v2 cpu.max / v1 cfs quota, else the pool's documented share of 16."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    share = int(quota) if quota and quota >= 1 else None
    threads = max(1, min(aff, share if share is not None else 16))
    return threads, {"affinity_mask": aff, "os_cpu_count": os.cpu_count(), "cgroup_cpu_quota": quota,
                     "threads_used": threads, "rule": "min(affinity, cgroup quota)" if share is not None else "min(affinity, 16): no cgroup quota visible, the pool's CPU share per GPU"}


def _steps_timed(mod, bt, steps):
    """2 warm-up steps, then `steps` steps between HIP events on the current stream (no host synchronisation inside): (median seconds per step, each step's ms,
    last loss).  The median, with every step listed beside it: these side workloads run few steps, and one step that meets a host hiccup would otherwise own the figure."""
    import statistics
    import torch
    for _ in range(2):
        mod.training_step(bt, 0)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        ls = mod.training_step(bt, 0)
        ev[i + 1].record()
    torch.cuda.synchronize()
    each = [round(ev[i].elapsed_time(ev[i + 1]), 2) for i in range(steps)]
    return statistics.median(each) * 1e-3, each, ls


def other_workloads(args, dev, steps=5):
    """pairs/s of `--pair text` (cfg-4 shape, B pairs) and `--pair roundrobin --batch 128` (cfg-5 shape): 2 warm-up + `steps` timed steps each, median step"""
    import copy
    import gc
    import torch
    res = {}
    for tag, pair, batch in (("cfg4_text", "text", args.batch), ("cfg5_roundrobin", "roundrobin", min(args.batch, 128))):
        a = copy.copy(args)
        a.pair, a.batch, a.model, a.model_seq, a.model_mod, a.train_seq = pair, batch, None, None, None, False
        w = build_workload(a, dev, 0)
        _log(f"{tag} built")
        mod, bt = w["module"], w["batch"]
        dt, per_step, ls = _steps_timed(mod, bt, steps)
        res[tag] = {"value": round(batch * len(w["subs"]) / dt, 1), "unit": "protein-pairs/sec (1 GPU)", "ms_per_step": round(dt * 1e3, 2), "steps": steps,
                    "ms_each_step": per_step,
                    "pairs_per_gpu_per_modality": batch, "sub_steps_per_step": len(w["subs"]), "loss": round(float(ls.detach()), 5), "workload": w["desc"]}
        if "text" in mod.network:
            # the text tower runs HF's train-mode dropout as the reference does (frozen tower included, ref text_encoder.py:56-62); the same step with it
            # switched off (transformer.train_dropout = False: what rounds 1-4 measured) beside it
            res[tag]["text_tower_dropout"] = "on (reference behaviour; ONEPROT_BERT_DROPOUT=0 / transformer.train_dropout = False switch it off)"
            mod.network["text"].transformer.train_dropout = False
            dt0, _, _ = _steps_timed(mod, bt, steps)
            res[tag]["value_text_dropout_off"] = round(batch * len(w["subs"]) / dt0, 1)
            res[tag]["ms_per_step_text_dropout_off"] = round(dt0 * 1e3, 2)
        del mod, bt, w, ls
        gc.collect(); torch.cuda.empty_cache()
    return res


def mfma_busy_child(iters=2, limit_s=420):
    """north_star's "rocprof-reported MFMA utilisation" of the ESM-2-150M encoder forward, measured BY THIS RUN: a child process
    `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/encoder_fwd_only.py` (the program itself behind `--`), started
    before this process has touched the GPU, summarised as tools/pmc_mfma_table.py does: busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * CUs * 4),
    weighted by each kernel's GPU-active cycles.  Returns None (the caller then quotes the committed profile, labelled) when rocprofv3 is missing, when this
    process itself runs under a profiler, or when the child fails."""
    import collections
    import csv
    import glob
    import re
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None                                          # profiled already: no profiler inside a profiler
    out = tempfile.mkdtemp(prefix="oneprot_mfma_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = [exe, "--kernel-trace", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "--output-format", "csv", "-d", out, "-o", "p", "--",
           sys.executable, os.path.join(ROOT, "tools", "encoder_fwd_only.py"), str(iters)]
    try:
        r = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=limit_s)
        files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            _log(f"mfma_busy_child: rocprofv3 rc {r.returncode}, {len(files)} counter files: {r.stderr.decode(errors='replace')[-300:]}")
            return None
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        calls = collections.defaultdict(set)
        for fn in files:
            with open(fn) as f:
                for row in csv.DictReader(f):
                    k = re.sub(r"\(.*", "", row["Kernel_Name"])
                    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                    calls[k].add(row["Dispatch_Id"])
        n_cu = 256
        rows = []
        for k, c in agg.items():
            gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8              # reported summed over the 8 XCDs
            if gui > 0:
                rows.append((gui, k, len(calls[k]), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * n_cu * 4)))
        tot = sum(r_[0] for r_ in rows)
        if tot <= 0:
            return None
        busy = sum(g * u for g, _, _, u in rows) / tot
        top = [{"kernel": k[:60], "calls": n, "share_of_gpu_active_cycles_pct": round(100 * g / tot, 1), "mfma_busy_pct": round(100 * u, 1)}
               for g, k, n, u in sorted(rows, reverse=True)[:6]]
        return {"mfma_busy_pct": round(100 * busy, 1), "kernels": top,
                "source": f"this run: child `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/encoder_fwd_only.py {iters}` "
                          "(256 x L=512 forward of the frozen ESM-2-150M tower; busy = MFMA-busy cycles / (GPU-active cycles x 256 CUs x 4 SIMDs), "
                          "clocks under the profiler are lower than in the timed region: a utilisation, not a time)"}
    except Exception as e:                                   # noqa: BLE001  (a measurement leg: never fails the bench)
        _log(f"mfma_busy_child: {type(e).__name__}: {e}")
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


_T0 = time.perf_counter()


def _log(msg):
    """progress on stderr (the JSON line is the only thing on stdout)"""
    print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


class BoardSampler:
    """Board power and shader clock of the GPU this rank drives while the timed steps run: the amdgpu hwmon files (power1_average / power1_input, freq1_input,
    power1_cap; readable without root), sampled every 50 ms by a helper thread.  The card is the one whose PCI address torch reports for the device.  Why it is in
    the bench line: the step runs AT the board's power cap with the shader clock pulled below its 2.4 GHz maximum (tools/ab/power_probe.py), so
    the bf16 MFMA peak the silicon offers during the run is 2.5 PFLOP/s x sclk / 2.4 GHz, not 2.5."""

    def __init__(self, torch_device):
        import glob
        import threading
        import torch
        self.hw, self.rows, self._stop, self._thread = None, [], threading.Event(), None
        try:
            pr = torch.cuda.get_device_properties(torch_device)
            want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
            for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
                if want in os.path.realpath(os.path.join(hw, "..", "..")) + "/" or want in (self._rd(os.path.join(hw, "..", "..", "uevent")) or ""):
                    self.hw = hw
        except Exception:
            self.hw = None

    @staticmethod
    def _rd(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            p = self._rd(self.hw + "/power1_average") or self._rd(self.hw + "/power1_input")
            f = self._rd(self.hw + "/freq1_input")
            if p and f:
                self.rows.append((float(p) / 1e6, float(f) / 1e6))
            self._stop.wait(0.05)

    def start(self):
        import threading
        if self.hw is not None:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
        if len(self.rows) < 3:
            return None
        ps, fs = [r[0] for r in self.rows], [r[1] for r in self.rows]
        cap = self._rd(self.hw + "/power1_cap")
        sclk = sum(fs) / len(fs)
        return {"power_w_mean": round(sum(ps) / len(ps), 0), "power_w_max": round(max(ps), 0), "power_cap_w": round(float(cap) / 1e6, 0) if cap else None,
                "sclk_mhz_mean": round(sclk, 0), "sclk_mhz_min": round(min(fs), 0), "sclk_mhz_max": round(max(fs), 0), "samples": len(ps),
                "bf16_mfma_peak_at_that_clock_tflops": round(PEAK_BF16_TFLOPS * sclk / 2400.0, 0),
                "source": "amdgpu hwmon (power1_average, freq1_input) of this rank's card, every 50 ms over the timed steps; the spec peak of 2.5 PFLOP/s is 4096 FLOP/clk/CU x 256 CUs at 2.4 GHz"}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    # N = 1, default workload: the MFMA-pipe utilisation of the encoder forward from a rocprofv3 child, BEFORE this process initialises HIP
    default_workload = (args.pair == "struct_token" and args.batch == 256 and args.seq_len == 512 and not (args.model or args.model_seq or args.model_mod))
    mfma_live = None
    if args.gpus == 1 and default_workload and not args.no_extras and not args.no_mfma_busy and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        _log("encoder-forward MFMA-busy pass (rocprofv3 child) ...")
        mfma_live = mfma_busy_child()
        _log(f"... {mfma_live['mfma_busy_pct'] if mfma_live else 'not available'}")

    import torch
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("ONEPROT_ALLOW_RANDOM_INIT", "1")
    from oneprot_amd import distributed as D
    from oneprot_amd import hip
    rank, world, local = D.setup_process_group()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    dev = torch.device("cuda", torch.cuda.current_device())
    hip.lib()      # fail loudly if the HIP library is missing

    import warnings
    warnings.filterwarnings("ignore", message=".*no weight file.*")
    warnings.filterwarnings("ignore", message=".*requires_grad=True to a scalar.*")
    work = build_workload(args, dev, rank)
    _log("workload built")
    module, batch, subs = work["module"], work["batch"], work["subs"]
    B = args.batch

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        module.training_step(batch, 0)
    barrier()
    xt = D.ExchangeTimer.enable() if world > 1 else None
    # live per-launch timing of the dominant kernel (FFN-1 GEMM, bias+GELU epilogue) with events on the launch stream
    hip.profile_begin({"oneprot_gemm_bf16_nt": hip.EPI_BIAS_GELU})
    board = BoardSampler(dev) if (rank == 0 and not args.no_extras) else None
    if board is not None:
        board.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = module.training_step(batch, 0)
    barrier()
    elapsed = time.perf_counter() - t0
    board_stats = board.stop() if board is not None else None
    launches = hip.profile_end()["oneprot_gemm_bf16_nt"]
    _log(f"timed region done: {elapsed / args.steps * 1e3:.1f} ms/step")
    exchange = None
    if xt is not None:        # per-rank exchange time per step: feature all-gather + its reduce-scatter + the un-hidden part of the gradient all-reduce
        parts = {k: round(v / args.steps, 3) for k, v in xt.totals_ms().items()}
        D.ExchangeTimer.disable()
        mine = {"rank": rank, "exchange_ms": round(sum(parts.values()), 3), "parts": parts}
        box = [None] * world
        torch.distributed.all_gather_object(box, mine)
        exchange = box
    el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(el)
    loss_val = float(loss.detach())

    extras = {}
    if rank == 0 and not args.no_extras:
        # ---- north_star figure: ESM-2 encoder forward (the frozen sequence tower: embedding + every layer), events on the launch stream
        seq_tr = module.network["sequence"].transformer
        seq_ids = subs[0][1]
        with torch.no_grad():
            seq_tr.run_layers(seq_ids, save=False)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10                                       # ~0.5 s: long enough for the board sampler (50 ms) to see the clock the forward runs at
            fwd_board = BoardSampler(dev)
            fwd_board.start()
            e0.record()
            for _ in range(reps):
                seq_tr.run_layers(seq_ids, save=False)
            e1.record()
            torch.cuda.synchronize()
            fwd_board = fwd_board.stop()
        fwd_ms = e0.elapsed_time(e1) / reps
        fwd_tf = seq_ids.shape[0] * tower_fwd_flops(seq_tr, seq_ids.shape[1]) / (fwd_ms * 1e-3) / 1e12
        # MFMA-busy % comes from a committed rocprofv3 PMC pass (counters cannot be read from inside this process): quoted only when this run IS the
        # profiled workload (model, batch, L), and labelled with its source
        mfma_busy, mfma_src, mfma_kernels = None, None, None
        if mfma_live is not None:
            mfma_busy, mfma_src, mfma_kernels = mfma_live["mfma_busy_pct"], mfma_live["source"], mfma_live["kernels"]
        else:
          try:
              with open(os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_encoder_fwd_mfma_busy.json")) as f:
                  mj = json.load(f)
              wl = mj.get("workload", {"model": ESM["150M"], "batch": 256, "seq_len": 512})
              if (wl["model"], wl["batch"], wl["seq_len"]) == (work["names"]["seq"], seq_ids.shape[0], seq_ids.shape[1]):
                  mfma_busy = mj["mfma_busy_pct"]
                  mfma_src = f"profiles/{PROFILE_ROUND}_encoder_fwd_mfma_busy.json (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE of tools/encoder_fwd_only.py; not measured by this run)"
          except Exception:
              pass
        extras["encoder_fwd"] = {"what": f"{work['names']['seq']} forward, {seq_ids.shape[0]} x L={seq_ids.shape[1]} (embedding + {seq_tr.n_layers} layers, no pooling head)",
                                 "ms": round(fwd_ms, 3), "achieved": round(fwd_tf, 1), "unit": "TFLOP/s", "peak": PEAK_BF16_TFLOPS,
                                 "frac": round(fwd_tf / PEAK_BF16_TFLOPS, 4), "mfma_busy_pct": mfma_busy, "mfma_busy_source": mfma_src, "mfma_busy_kernels": mfma_kernels,
                                 "target_frac": 0.40}
        if fwd_board is not None:
            # the forward runs at the board's power cap too: its FLOP rate against the MFMA peak at the clock it ran at.  (The busy counter above counts every
            # MFMA cycle, useful or not: it DROPPED by ~2 points in round 6 when the attention forward's all-ones row-sum MFMAs were replaced by v_dot2c -- and
            # the forward got faster.  MFMAs now do the dense contractions only.)
            extras["encoder_fwd"]["board"] = {k: fwd_board[k] for k in ("power_w_mean", "power_cap_w", "sclk_mhz_mean", "samples", "bf16_mfma_peak_at_that_clock_tflops")}
            extras["encoder_fwd"]["frac_at_that_clock"] = round(fwd_tf / max(fwd_board["bf16_mfma_peak_at_that_clock_tflops"], 1.0), 4)
    if not args.no_extras:
        # ---- one extra step with every kernel family bracketed by events (outside the timed region: the brackets cost launch time)
        hip.profile_begin({k: None for k in ("oneprot_gemm_bf16_nt", "oneprot_gemm_bf16_nt_resid_ln", "oneprot_gemm_bf16_nt_resid_ln8", "oneprot_gemm_bf16_tn",
                                             "oneprot_attn_fwd", "oneprot_attn_bwd", "oneprot_layernorm_fwd",
                                             "oneprot_layernorm_bwd", "oneprot_adam_step")})
        module.training_step(batch, 0)
        barrier()
        prof = hip.profile_end()
        if rank == 0:
            extras["kernels"] = summarise_kernels(prof)

    # the FFN-2 + LayerNorm launches wait, bounded, for their neighbours' partial statistics: a wait that ran out means wrong numbers, not a slow step
    ln8_errors = hip.sched_error()
    if ln8_errors != 0:
        raise RuntimeError("oneprot_gemm_bf16_nt_resid_ln8: a wait for the neighbouring column tiles ran out during the timed steps")
    if rank == 0:
        extras["ffn2_ln"] = {"what": "FFN-2 GEMM with the next LayerNorm finished across work-groups (ONEPROT_FFN2_LN=0 keeps the pair of launches)",
                             "enabled": os.environ.get("ONEPROT_FFN2_LN", "1") != "0", "waits_run_out": ln8_errors}
        # how the persistent kernels share the chip with co-resident kernels (the RCCL channels of the overlapped all-reduce when N > 1): DESIGN section 5
        extras["work_queues"] = {"dynamic_tiles": hip.dynamic_tiles_wanted(), "cu_reserve_weight_gradient_gemm": hip.cu_reserve_wanted(),
                                 "late_ticket_draws": hip.sched_late_draws(), "what": "ONEPROT_DYNAMIC_TILES / ONEPROT_CU_RESERVE (defaults: on / 16 when WORLD_SIZE > 1): NT GEMM tiles and "
                                 "attention-forward slabs drawn from per-XCD work queues, weight-gradient GEMM cut into (CUs - reserve) work items; rank 0's diagnostic counters"}
        ms_step = elapsed / args.steps * 1e3
        pairs_per_step = B * len(subs)
        value = world * pairs_per_step * args.steps / elapsed
        gemm_flops = sum(2.0 * sc[0] * sc[1] * sc[2] for _, sc in launches)
        gemm_s = sum(ms for ms, _ in launches) * 1e-3
        achieved = gemm_flops / gemm_s / 1e12 if gemm_s > 0 else 0.0
        # algorithmic bytes of an FFN-1 launch: bf16 A [M,K] + bf16 W [N,K] + bf16 gelu(z) [M,N] (+ the one-byte gelu'(z) codes [M,N] when the tower trains: 5 tensor arguments)
        alg_bytes = sum(2.0 * (sc[0] * sc[2] + sc[1] * sc[2] + sc[0] * sc[1]) + (1.0 * sc[0] * sc[1] if sc[-1] >= 5 else 0.0) for _, sc in launches)
        shapes = sorted({(sc[0], sc[1], sc[2]) for _, sc in launches})
        flops = step_flops(module, subs, None)
        # HBM/fabric bytes per launch of the dominant kernel: PMC passes cannot run inside this process, so the figure measured with
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, gfx950 x2 correction on FETCH_SIZE) is read from profiles/.
        # The timed launches are of two kinds -- with the GELU' output (trainable tower: 5 tensor arguments) and without (frozen tower: 4) --
        # measured separately; `traffic` is their launch-weighted mean.
        traffic, traffic_src, traffic_alg = None, None, None
        if args.pair == "struct_token" and shapes == [(B * args.seq_len, 2560, 640)]:
            try:
                with open(os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc_traffic.json")) as f:
                    tj = json.load(f)
                n2 = sum(1 for _, sc in launches if sc[-1] >= 5)
                n1 = len(launches) - n2
                traffic = (n2 * tj["ffn1"]["traffic_bytes_per_launch"] + n1 * tj["ffn1fwd"]["traffic_bytes_per_launch"]) / max(len(launches), 1) / 1e9
                traffic_alg = (n2 * tj["ffn1"]["algorithmic_bytes_per_launch"] + n1 * tj["ffn1fwd"]["algorithmic_bytes_per_launch"]) / max(len(launches), 1) / 1e9
                traffic_src = f"{PROFILE_ROUND}_pmc_traffic.json; {n2} launches with two outputs (bf16 gelu + one-byte gelu' codes) at {tj['ffn1']['traffic_bytes_per_launch'] / 1e9:.2f} GB, {n1} with one at {tj['ffn1fwd']['traffic_bytes_per_launch'] / 1e9:.2f} GB"
            except Exception:
                pass
        cfg_tag = {"struct_token": "cfg-2" if world == 1 else "cfg-3-shaped", "text": "cfg-4-shaped", "roundrobin": "cfg-5-shaped"}[args.pair]
        out = {
            "metric": work["metric"], "value": round(value, 2), "unit": "protein-pairs/sec/node",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{cfg_tag}: {work['desc']}, {B} pairs/GPU per modality, global batch {B * world}, output_dim 1024, CLIP local_loss+gather_with_grad, "
                                   f"L1 0.01, Adam 1e-3, clip 1.0, sequence encoder {'trainable' if args.train_seq else 'frozen (reference default)'}, random-init weights",
                       "global_batch": B * world, "seq_len": args.seq_len, "parallelism": f"dp{world}", "frozen_sequence_encoder": not args.train_seq,
                       "sub_steps_per_step": len(subs), "loss": round(loss_val, 5)},
            "step_tflops_per_gpu": round(flops / (ms_step * 1e-3) / 1e12, 1),
            "roofline": {"bound": "mfma", "kernel": "g8::k_gemm8<256x320 tile, BIAS_GELU> FFN-1 launches " + ", ".join(f"[{m}x{k}]x[{n}x{k}]^T" for m, n, k in shapes),
                         "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                         "traffic": round(traffic, 3) if traffic else None,
                         "traffic_unit": f"GB/launch, from a committed profile, not this run (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/{traffic_src}; algorithmic {traffic_alg:.2f} GB/launch)" if traffic else None,
                         "launches_timed": len(launches), "avg_launch_ms": round(gemm_s / max(len(launches), 1) * 1e3, 4),
                         "flops_per_launch": gemm_flops / max(len(launches), 1),
                         # the same launches against the HBM roof: A + W read once, one bf16 output (two with the GELU' output) written once.  With two
                         # outputs the HBM bound (1.51 GB / 8 TB/s = 0.189 ms) is above the MFMA bound (429.5 GFLOP / 2.5 PF = 0.172 ms) of this launch.
                         "hbm_view": {"achieved": round(alg_bytes / gemm_s / 1e9, 1) if gemm_s > 0 else 0.0, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "frac": round(alg_bytes / gemm_s / 1e9 / PEAK_HBM_GBS, 4) if gemm_s > 0 else 0.0,
                                      "algorithmic_GB_per_launch": round(alg_bytes / max(len(launches), 1) / 1e9, 3)}},
        }
        if board_stats is not None:
            # the dominant kernel against the peak the clock of THIS run allows (the step sits at the board's power cap: the clock is what gives)
            board_stats["roofline_frac_at_that_clock"] = round(achieved / max(board_stats["bf16_mfma_peak_at_that_clock_tflops"], 1.0), 4)
            board_stats["step_frac_at_that_clock"] = round(flops / (ms_step * 1e-3) / 1e12 / max(board_stats["bf16_mfma_peak_at_that_clock_tflops"], 1.0), 4)
            out["board"] = board_stats
        out.update(extras)
        if exchange is not None:
            from oneprot_amd.distributed import grad_overlap_enabled
            out["exchange"] = {"what": "per rank, ms per step on the launch stream: packed feature all-gather + reduce-scatter backward + the part of the gradient all-reduce not hidden under the backward",
                               "grad_overlap": int(grad_overlap_enabled()), "grad_overlap_switch": "ONEPROT_GRAD_OVERLAP=0/1 (1: arena ranges all-reduced from inside the backward; 0: bucketed after it)",
                               "per_rank": exchange}
        if world == 1 and not args.no_cpu_baseline:
            threads, share = host_cpu_share()
            _log(f"cpu baseline on {threads} threads ({share})")
            out["cpu_baseline"] = cpu_baseline(work, args.cpu_sample_pairs, threads)
            _log("cpu baseline cfg-1")
            out["cpu_baseline"]["host_cores"] = share
            out["cpu_baseline"]["cfg1"] = cpu_cfg1(threads)
        if world == 1 and not args.no_extras and args.pair == "struct_token" and not args.no_other_workloads:
            # the other single-GPU BASELINE shapes, driver-visible: cfg-4 (seq <-> text) and the cfg-5-shaped 4-modality round-robin step, a few steps
            # each AFTER the timed region and after the main workload's memory has been released
            del module, batch, subs, loss
            work.clear()
            import gc
            gc.collect(); torch.cuda.empty_cache()
            _log("other workloads")
            out["other_workloads"] = other_workloads(args, dev)
            _log("done")
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
