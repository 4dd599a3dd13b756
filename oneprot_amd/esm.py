"""ESM-2 encoder ("transformer" member of the Sequence / StructToken encoders) running on the HIP kernels.

Replaces, for the OneProt hot path, what the reference gets from HF `AutoModel.from_pretrained(...)`
(ref sequence_encoder.py:8-19,51-55; struct_token_encoder.py:26-27) -- i.e. transformers' EsmModel
(modeling_esm.py:198-760).  State-dict key names are identical to EsmModel's, so OneProt / HF checkpoints load.

Memory layout
  * all encoder parameters live in ONE fp32 arena (`flat`, the only nn.Parameter) in the order the kernels consume
    them (q,k,v weights adjacent => one fused [3d,d] QKV operand); named tensors are views into it;
  * a bf16 mirror of the arena (GEMM operands) plus transposed bf16 weight copies for the dgrad GEMMs are refreshed
    when the arena's version counter changes (i.e. after an optimizer step / checkpoint load);
  * the gradient w.r.t. the arena is produced as one flat fp32 tensor by the hand-written backward below, so
    clip-norm and Adam are single launches over contiguous memory.
"""
import json
import math
import os
import warnings
from collections import OrderedDict

import torch
import torch.nn as nn

from . import hip

KNOWN_ESM = {   # public facts of the ESM-2 checkpoints (SURVEY.md section 8 header)
    "esm2_t6_8M_UR50D": dict(num_hidden_layers=6, hidden_size=320, intermediate_size=1280),
    "esm2_t12_35M_UR50D": dict(num_hidden_layers=12, hidden_size=480, intermediate_size=1920),
    "esm2_t30_150M_UR50D": dict(num_hidden_layers=30, hidden_size=640, intermediate_size=2560),
    "esm2_t33_650M_UR50D": dict(num_hidden_layers=33, hidden_size=1280, intermediate_size=5120),
}
ESM_DEFAULTS = dict(model_type="esm", vocab_size=33, pad_token_id=1, mask_token_id=32, num_attention_heads=20, layer_norm_eps=1e-5,
                    token_dropout=True, position_embedding_type="rotary", emb_layer_norm_before=False, max_position_embeddings=1026,
                    initializer_range=0.02, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)


class ModelConfig:
    """Minimal stand-in for transformers.PretrainedConfig (attribute access + to_dict)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to_dict(self):
        return dict(self.__dict__)

    def __repr__(self):
        return f"ModelConfig({self.__dict__})"


def resolve_config(model_name_or_path, defaults_by_type=None):
    """Local directory with config.json, or a known hub id (no network in this environment).
    Mirrors the failure mode of AutoConfig.from_pretrained: OSError when it cannot be resolved."""
    path = str(model_name_or_path)
    cfg_file = os.path.join(path, "config.json")
    if os.path.isfile(cfg_file):
        with open(cfg_file) as f:
            raw = json.load(f)
        base = dict(ESM_DEFAULTS) if raw.get("model_type", "esm") == "esm" else {}
        base.update(raw)
        return ModelConfig(**base), path
    short = path.split("/")[-1]
    if short in KNOWN_ESM:
        base = dict(ESM_DEFAULTS)
        base.update(KNOWN_ESM[short])
        return ModelConfig(**base), None
    raise OSError(f"{model_name_or_path} is not a local folder with a config.json and is not a known ESM-2 model identifier")


def load_weight_file(path):
    """Tensors of a HF-style checkpoint directory (model.safetensors or pytorch_model.bin), or None."""
    if path is None:
        return None
    st = os.path.join(path, "model.safetensors")
    if os.path.isfile(st):
        from safetensors.torch import load_file
        return load_file(st)
    pt = os.path.join(path, "pytorch_model.bin")
    if os.path.isfile(pt):
        return torch.load(pt, map_location="cpu", weights_only=True)
    return None


def _pad8(n):
    return (n + 7) // 8 * 8


class _Out:
    def __init__(self, last_hidden_state):
        self.last_hidden_state = last_hidden_state
        self.pooler_output = None

    def __getitem__(self, i):
        return (self.last_hidden_state, self.pooler_output)[i]


# experiment hook: ONEPROT_GELU_CODE_DROP_BITS=n rounds the saved one-byte gelu'(z) codes to 8 - n bits (n = 1: twice the quantisation step).  Used by
# tests/test_baseline_shapes_gpu.py to attribute the backward's deviation from the fp32 oracle; 0 (default) leaves the codes alone.
_GELU_CODE_DROP_BITS = int(os.environ.get("ONEPROT_GELU_CODE_DROP_BITS", "0"))


class ArenaModule(nn.Module):
    """Parameters of a whole encoder in one fp32 arena (`flat`) with named views, a bf16 mirror for the MFMA GEMMs, and
    state-dict hooks that expose / accept the HF key names."""

    # Counter-based dropout streams (Philox key = seed, counter high words = stream id): every consumer gets a domain of its own so that no two
    # of them can draw the same mask -- bits 60..63 the purpose (1: BERT hidden / attention dropout, 2: LoRA dropout), bits 44..59 the tower
    # (construction order within the process until OneProtLitModule re-numbers it by modality name; saved / restored with the stream state), bits 0..43 the consumer's own (call, layer, site) numbering.
    RNG_DOMAIN_BERT, RNG_DOMAIN_LORA = 1, 2
    _next_rng_uid = 0

    def _init_arena(self):
        self._spec = OrderedDict()
        self._cursor = 0
        self._extra = OrderedDict()
        self._rng_uid = ArenaModule._next_rng_uid & 0xFFFF
        ArenaModule._next_rng_uid += 1

    def _rng_stream(self, domain, local):
        return (int(domain) << 60) | (self._rng_uid << 44) | (int(local) & ((1 << 44) - 1))

    def rng_state(self):
        """Seeds and call counters of the dropout streams (plain ints): what a checkpoint has to carry for a resumed run to continue the mask
        sequence instead of replaying it from call 0 (OneProtLitModule.on_save_checkpoint / on_load_checkpoint)."""
        st = {k: int(getattr(self, k)) for k in ("_lora_seed", "_lora_calls", "_drop_seed", "_drop_calls") if getattr(self, k, None) is not None}
        if st:
            st["_rng_uid"] = int(self._rng_uid)              # the tower id inside the stream ids: a resumed process continues THESE streams
        return st

    def set_rng_state(self, state):
        for k in ("_lora_seed", "_lora_calls", "_drop_seed", "_drop_calls", "_rng_uid"):
            if k in state:
                setattr(self, k, int(state[k]))

    def _finish_arena(self):
        self._total = self._cursor
        self.flat = nn.Parameter(torch.zeros(self._total))
        self.extra = nn.ParameterDict({k.replace(".", "__"): nn.Parameter(torch.zeros(s)) for k, s in self._extra.items()})
        self._register_state_dict_hook(self._sd_hook)
        self._register_load_state_dict_pre_hook(self._load_hook)
        self._bf16 = None
        self._bf16_T = {}
        self._bf16_version = None

    def _transpose_layer_weights(self, key, name, R, C):
        """self._bf16_T[(i, key)] = bf16 [C, R] transposed copy of the fp32 [R, C] arena block that starts at tensor name.format(i=i), for
        every layer i, in ONE launch (the layers sit at a constant arena pitch); the copies are views of one [n_layers, C, R] buffer."""
        offs = [self._spec[name.format(i=i)][0] for i in range(self.n_layers)]
        pitch = offs[1] - offs[0] if self.n_layers > 1 else 0
        buf = self._bf16_T.get(("all", key))
        if buf is None:
            buf = self._bf16_T[("all", key)] = torch.empty(self.n_layers, C, R, dtype=torch.bfloat16, device=self.flat.device)
            for i in range(self.n_layers):
                self._bf16_T[(i, key)] = buf[i]
        w = self.flat.data
        if pitch >= 0 and all(o == offs[0] + i * pitch for i, o in enumerate(offs)):
            hip.call("oneprot_transpose_cast_f32_to_bf16_batched", w[offs[0]:], buf, R, C, pitch, R * C, self.n_layers)
        else:
            for i, o in enumerate(offs):
                hip.call("oneprot_transpose_cast_f32_to_bf16", w[o:o + R * C], buf[i], R, C)

    def _add(self, name, shape):
        n = 1
        for s in shape:
            n *= s
        self._spec[name] = (self._cursor, n, tuple(shape))
        self._cursor += _pad8(n)

    def view(self, name, src=None):
        off, n, shape = self._spec[name]
        src = self.flat if src is None else src
        if isinstance(src, nn.Parameter):
            src = src.data
        return src[off:off + n].view(shape)

    def span(self, first, last):
        """contiguous arena range covering tensors first..last (used for the fused QKV operand)"""
        o0 = self._spec[first][0]
        o1, n1, _ = self._spec[last]
        return o0, o1 + n1 - o0

    def named_views(self):
        return OrderedDict((k, self.view(k)) for k in self._spec)

    def _w16(self, name):
        return self.view(name, self._bf16)

    _sd_extra_buffers = ()
    _load_ignore_suffixes = ()

    # ------------------------------------------------------------------------------------------------- LoRA (peft 0.5.0 semantics)
    # ref sequence_encoder.py:61-74 / text_encoder.py:39-52: get_peft_model(transformer, LoraConfig(r, lora_alpha, lora_dropout,
    # target_modules=[query,key,value], bias="all")).  peft's lora.Linear computes  y = x W^T + b + (alpha/r) * (dropout(x) A^T) B^T  with
    # A ~ kaiming_uniform(a=sqrt(5)), B = 0; only lora_A / lora_B and every parameter whose name contains "bias" stay trainable; the wrapped
    # model's keys gain the prefix "base_model.model." and the adapters are "<linear>.lora_{A,B}.default.weight".
    # Here: the adapters are two stacked parameters.  In eval mode (and with lora_dropout = 0) the bf16 GEMM operand of a target is the merged
    # W + (alpha/r) B A -- peft's forward at dropout 0 -- and the hand-written backward yields d(W_eff), from which dA = s B^T dW and
    # dB = s dW A^T follow.  In TRAIN mode with lora_dropout > 0 the two branches stay apart, as in peft: the fused QKV GEMM runs on the
    # K-concatenated operands [h | dropout(h) A^T] x [W | s B]^T (one launch, the rotary epilogue unchanged), the mask comes from a counter-based
    # generator (oneprot_dropout_bf16: a pure function of seed, call, layer, element) and is regenerated in the backward, where
    # du = dqkv (sB), dB = s dqkv^T u, dA = du^T dropout(h) and the layer-input gradient gains mask * (du A) / keep.
    # Either way the arena gradient is masked down to its "bias" entries (peft bias="all").
    LORA_TARGETS = ("query", "key", "value")
    _lora = None

    def enable_lora(self, r, alpha, target_modules, dropout=0.0):
        targets = [t for t in self.LORA_TARGETS if t in list(target_modules)]
        if len(targets) != len(list(target_modules)):
            raise NotImplementedError(f"LoRA target modules {list(target_modules)}: only {list(self.LORA_TARGETS)} (the reference's list) are built")
        n, d = self.n_layers, self.d
        A = torch.empty(n, len(targets), r, d)
        A.uniform_(-1.0 / math.sqrt(d), 1.0 / math.sqrt(d))              # kaiming_uniform_(a=sqrt(5)) on [r, d]
        self.lora_A = nn.Parameter(A)
        self.lora_B = nn.Parameter(torch.zeros(n, len(targets), d, r))
        self._lora = dict(r=r, alpha=alpha, scaling=alpha / r, targets=targets, dropout=dropout)
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError(f"lora_dropout must be in [0, 1), got {dropout}")
        self._lora_seed = int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF      # masks follow torch.manual_seed; every call / layer draws its own stream
        self._lora_calls = 0
        for p in self.parameters():
            p.requires_grad = False
        self.lora_A.requires_grad = True
        self.lora_B.requires_grad = True
        self.flat.requires_grad = True                                   # bias="all": only the "bias" entries receive a gradient (mask below)
        idx = torch.cat([torch.arange(off, off + cnt) for name, (off, cnt, _) in self._spec.items() if "bias" in name])
        self.register_buffer("_lora_bias_index", idx, persistent=False)
        self._bf16_version = None

    def _lora_key(self, prefix, i, t, which):
        return f"{prefix}encoder.layer.{i}.attention.self.{t}.lora_{which}.default.weight"

    def _sd_hook(self, module, state_dict, prefix, local_metadata):
        flat = state_dict.pop(prefix + "flat")
        base = prefix + ("base_model.model." if self._lora else "")
        for name in self._spec:
            state_dict[base + name] = self.view(name, flat)
        for k in list(self._extra):
            state_dict[base + k] = state_dict.pop(prefix + "extra." + k.replace(".", "__"))
        for key, fn in self._sd_extra_buffers:
            state_dict[base + key] = fn(self)
        if self._lora:
            A, B = state_dict.pop(prefix + "lora_A"), state_dict.pop(prefix + "lora_B")
            for i in range(self.n_layers):
                for ti, t in enumerate(self._lora["targets"]):
                    state_dict[self._lora_key(base, i, t, "A")] = A[i, ti]
                    state_dict[self._lora_key(base, i, t, "B")] = B[i, ti]
        return state_dict

    def _load_hook(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        peft = prefix + "base_model.model."
        for k in [k for k in state_dict if k.startswith(peft)]:          # a PeftModel's keys load into either form
            state_dict[prefix + k[len(peft):]] = state_dict.pop(k)
        if self._lora:
            A, B = self.lora_A.detach().clone(), self.lora_B.detach().clone()
            for i in range(self.n_layers):
                for ti, t in enumerate(self._lora["targets"]):
                    for which, dst in (("A", A), ("B", B)):
                        key = self._lora_key(prefix, i, t, which)
                        if key in state_dict:
                            dst[i, ti] = state_dict.pop(key).to(dst)
                        elif prefix + "lora_" + which not in state_dict:
                            missing_keys.append(key)
            state_dict.setdefault(prefix + "lora_A", A)
            state_dict.setdefault(prefix + "lora_B", B)
        flat = self.flat.detach().clone()
        for name, (off, n, shape) in self._spec.items():
            key = prefix + name
            if key in state_dict:
                t = state_dict.pop(key)
                if tuple(t.shape) != shape:
                    error_msgs.append(f"size mismatch for {key}: checkpoint {tuple(t.shape)} vs model {shape}")
                    continue
                flat[off:off + n] = t.reshape(-1).to(flat)
            elif prefix + "flat" not in state_dict:
                missing_keys.append(key)
        if prefix + "flat" not in state_dict:
            state_dict[prefix + "flat"] = flat
        for k in self._extra:
            key = prefix + k
            if key in state_dict:
                state_dict[prefix + "extra." + k.replace(".", "__")] = state_dict.pop(key)
        for key in [k for k in state_dict if k.startswith(prefix) and k.endswith(tuple(self._load_ignore_suffixes))] if self._load_ignore_suffixes else []:
            state_dict.pop(key)

    def _refresh_bf16_mirror(self):
        """returns True when the mirror was rebuilt"""
        two = self._lora_two_branch()
        ver = (self.flat._version, self.flat.data_ptr()) + ((self.lora_A._version, self.lora_B._version, two) if self._lora else ())
        if self._bf16_version == ver:
            return False
        dev = self.flat.device
        if self._bf16 is None or self._bf16.device != dev or self._bf16.numel() != self._total:
            self._bf16 = torch.empty(self._total, dtype=torch.bfloat16, device=dev)
            self._bf16_T = {}
        hip.call("oneprot_cast_f32_to_bf16", self.flat.data, self._bf16, self._total)
        if two:                 # train mode with lora_dropout: the mirror keeps the base weights, the adapters ride in the concatenated operands
            self._lora_refresh_branch_operands()
        elif self._lora:        # merged operands W + (alpha/r) B A for the target projections
            for i in range(self.n_layers):
                w = self._qkv_effective_f32(i)
                o, n = self.span(f"encoder.layer.{i}.attention.self.query.weight", f"encoder.layer.{i}.attention.self.value.weight")
                hip.call("oneprot_cast_f32_to_bf16", w, self._bf16[o:o + n], n)
        self._bf16_version = ver
        return True

    def _transpose_qkv_weights(self):
        """the [d, 3d] transposed bf16 copies of the fused q|k|v weights (dgrad operand): one launch over the arena, or per layer from
        the LoRA-merged scratch"""
        d = self.d
        if not self._lora:
            return self._transpose_layer_weights("qkv", "encoder.layer.{i}.attention.self.query.weight", 3 * d, d)
        for i in range(self.n_layers):
            t = self._bf16_T.get((i, "qkv"))
            if t is None:
                t = self._bf16_T[(i, "qkv")] = torch.empty(d, 3 * d, dtype=torch.bfloat16, device=self.flat.device)
            hip.call("oneprot_transpose_cast_f32_to_bf16", self._qkv_effective_f32(i), t, 3 * d, d)

    def _qkv_effective_f32(self, i):
        """fp32 [3d, d] fused q|k|v weight of layer i as the GEMMs must see it: the arena view, or (LoRA) a scratch copy with (alpha/r) B A
        added to the target blocks."""
        o, n = self.span(f"encoder.layer.{i}.attention.self.query.weight", f"encoder.layer.{i}.attention.self.value.weight")
        w = self.flat.data[o:o + n]
        if not self._lora or self._lora_two_branch():
            return w
        d, r = self.d, self._lora["r"]
        if getattr(self, "_lora_scratch", None) is None or self._lora_scratch.device != w.device:
            self._lora_scratch = torch.empty(3 * d * d, device=w.device)
        tmp = self._lora_scratch
        tmp.copy_(w)
        for ti, t in enumerate(self._lora["targets"]):
            blk = self.LORA_TARGETS.index(t)
            hip.call("oneprot_sgemm", self.lora_B.data[i, ti], self.lora_A.data[i, ti], tmp[blk * d * d:(blk + 1) * d * d], d, d, r, 0, 1, self._lora["scaling"], 1)
        return tmp

    def _wgrad(self, dY, X, T, N, K, ldy, ldx, dW, db, ws):
        """weight + bias gradient of one Linear; under LoRA the base weight is frozen, so only the bias gradient (column sums of dY) is formed"""
        if self._lora:
            hip.call("oneprot_colsum_bf16", dY, db, ws, T, N, 0)
        else:
            hip.call("oneprot_gemm_bf16_tn", dY, X, T, N, K, ldy, ldx, dW, db, ws, ws.numel(), 0)

    @staticmethod
    def _tn_workspace(shapes, dev):
        """one workspace large enough for every (N, K) weight gradient of a backward pass (the split count, hence the slab size, differs per shape)"""
        return torch.empty(max(hip.query("oneprot_gemm_bf16_tn_workspace", N, K) for N, K in shapes), dtype=torch.uint8, device=dev)

    def lora_backward(self, gflat):
        """(dA, dB) and `gflat` reduced to its "bias" entries (peft bias="all").  Merged form: from the gradient w.r.t. the merged weights left in
        `gflat`; two-branch form: from the per-layer adapter gradients backward_layers left in self._lora_raw."""
        d, r, s_ = self.d, self._lora["r"], self._lora["scaling"]
        dA, dB = torch.empty_like(self.lora_A), torch.empty_like(self.lora_B)
        raw = getattr(self, "_lora_raw", None)
        if raw is not None:
            self._lora_raw = None
            dA_raw, dB_raw = raw                                             # [n, Rp, d], [n, 3d, Rp]; target ti at rows / columns ti * rp .. + r
            rp = self._lora_ops["rp"]
            for ti, t in enumerate(self._lora["targets"]):
                blk = self.LORA_TARGETS.index(t)
                dA[:, ti] = dA_raw[:, ti * rp:ti * rp + r]
                dB[:, ti] = s_ * dB_raw[:, blk * d:(blk + 1) * d, ti * rp:ti * rp + r]
        else:
            for i in range(self.n_layers):
                for ti, t in enumerate(self._lora["targets"]):
                    dW = self.view(f"encoder.layer.{i}.attention.self.{t}.weight", gflat)
                    hip.call("oneprot_sgemm", dW, self.lora_A.data[i, ti], dB[i, ti], d, r, d, 0, 0, s_, 0)        # dB = s dW A^T
                    hip.call("oneprot_sgemm", self.lora_B.data[i, ti], dW, dA[i, ti], r, d, d, 1, 1, s_, 0)        # dA = s B^T dW
        masked = torch.zeros_like(gflat)
        idx = self._lora_bias_index
        masked[idx] = gflat[idx]
        return dA, dB, masked

    # ---- LoRA in train mode with lora_dropout > 0: peft's two-branch form (see the comment above enable_lora)
    def _lora_two_branch(self):
        if not (bool(self._lora) and self._lora["dropout"] > 0 and self.training):
            return False
        if getattr(self, "_padded", False):      # head_dim 24 models run their QKV operands in a padded layout: the concatenated form is not built for it
            if not getattr(self, "_lora_pad_warned", False):
                self._lora_pad_warned = True
                warnings.warn("lora_dropout is not applied to this padded-head model (head_dim not in 16/32/64): the adapters run merged into the weights, "
                              "which is peft's forward at dropout 0 (INTEGRATION.md, 'Deviations')", stacklevel=3)
            return False
        return True

    def _lora_refresh_branch_operands(self):
        """bf16 operands of the two-branch form for every layer: Wc [n, 3d, Kc] = [W | s B (block-diagonal over the targets) | 0] for the
        K-concatenated QKV GEMM, Acat [n, Rp, d] (the targets' A stacked, each padded to rp = ceil8(r) rows), AtT [n, targets, d, rp] and BsT [n, Rp, 3d] for the backward.
        Runs after every optimizer step (the adapters moved): the buffers are allocated once and refilled in place -- the frozen base weights as ONE
        strided copy out of the bf16 mirror (the q|k|v block sits at the same offset in every layer of the arena), the r-wide adapter columns by
        slice assignment; the zero padding is written once, at allocation."""
        lo, n, d, dev = self._lora, self.n_layers, self.d, self.flat.device
        r, nt, s_ = lo["r"], len(lo["targets"]), lo["scaling"]
        rp = -(-r // 8) * 8
        Rp = nt * rp
        Kc = -(-(d + Rp) // 128) * 128
        bf = torch.bfloat16
        ops = getattr(self, "_lora_ops", None)
        if ops is None or ops["Wc"].device != dev or ops["Kc"] != Kc or ops["rp"] != rp:
            ops = self._lora_ops = dict(rp=rp, Rp=Rp, Kc=Kc, Wc=torch.zeros(n, 3 * d, Kc, dtype=bf, device=dev), Acat=torch.zeros(n, Rp, d, dtype=bf, device=dev),
                                        AtT=torch.zeros(n, nt, d, rp, dtype=bf, device=dev), BsT=torch.zeros(n, Rp, 3 * d, dtype=bf, device=dev))
        o0, cnt = self.span("encoder.layer.0.attention.self.query.weight", "encoder.layer.0.attention.self.value.weight")
        stride = self.span("encoder.layer.1.attention.self.query.weight", "encoder.layer.1.attention.self.value.weight")[0] - o0 if n > 1 else cnt
        assert cnt == 3 * d * d
        ops["Wc"][:, :, :d].copy_(torch.as_strided(self._bf16, (n, 3 * d, d), (stride, d, 1), o0))
        for ti, t in enumerate(lo["targets"]):
            blk = self.LORA_TARGETS.index(t)
            a16 = self.lora_A.data[:, ti].to(bf)                                  # [n, r, d]
            b16 = (s_ * self.lora_B.data[:, ti]).to(bf)                           # [n, d, r]
            ops["Acat"][:, ti * rp:ti * rp + r] = a16
            ops["AtT"][:, ti, :, :r] = a16.transpose(1, 2)
            ops["Wc"][:, blk * d:(blk + 1) * d, d + ti * rp:d + ti * rp + r] = b16
            ops["BsT"][:, ti * rp:ti * rp + r, blk * d:(blk + 1) * d] = b16.transpose(1, 2)

    def _lora_scratch_for(self, T, dev):
        sc = getattr(self, "_lora_branch_scratch", None)
        if sc is None or sc["key"] != (T, dev, self._lora_ops["Kc"]):
            sc = dict(key=(T, dev, self._lora_ops["Kc"]), hd=torch.empty(T, self.d, dtype=torch.bfloat16, device=dev),
                      ut=torch.empty(T, self._lora_ops["rp"], dtype=torch.bfloat16, device=dev),
                      Xc=torch.zeros(T, self._lora_ops["Kc"], dtype=torch.bfloat16, device=dev))      # columns past d + Rp stay zero
            self._lora_branch_scratch = sc
        return sc

    def _lora_stream(self, call_id, i, ti):
        """dropout stream of target ti in layer i of forward call call_id: peft gives every wrapped Linear its own dropout module, so q, k and v
        see independent masks of the same input"""
        return self._rng_stream(self.RNG_DOMAIN_LORA, (call_id * self.n_layers + i) * 4 + ti)

    def _lora_branch_operand(self, i, h, T, call_id):
        """A operand of layer i's QKV GEMM in the two-branch form: [h | dropout_q(h) A_q^T | dropout_k(h) A_k^T | dropout_v(h) A_v^T | 0] bf16
        [T, Kc]; also returns u = the adapter columns [T, Rp] (kept for the backward; the dropped inputs are regenerated there)."""
        ops, d = self._lora_ops, self.d
        rp, Rp = ops["rp"], ops["Rp"]
        sc = self._lora_scratch_for(T, h.device)
        u = torch.empty(T, Rp, dtype=torch.bfloat16, device=h.device)
        for ti in range(len(self._lora["targets"])):
            hip.call("oneprot_dropout_bf16", h, sc["hd"], T * d, float(self._lora["dropout"]), self._lora_seed, self._lora_stream(call_id, i, ti))
            hip.call("oneprot_gemm_bf16_nt", sc["hd"], ops["Acat"][i, ti * rp:(ti + 1) * rp], T, rp, d, d, d, hip.EPI_BF16, None, sc["ut"], None, None, None, None, None,
                     1.0, 0, 0, 0)
            u[:, ti * rp:(ti + 1) * rp].copy_(sc["ut"])
        Xc = sc["Xc"]
        Xc[:, :d].copy_(h.view(T, d))
        Xc[:, d:d + Rp].copy_(u)
        return Xc, u

    def _lora_branch_backward(self, i, h, u, dqkv, T, call_id, ws_tn, raw, dh16=None, dh32=None):
        """adapter gradients of layer i into raw = (dA_raw [n, Rp, d], dB_raw [n, 3d, Rp]) and the branches' share of the layer-input gradient,
        sum_t mask_t * (du_t A_t) / keep, added into dh16 (bf16) or dh32 (fp32)."""
        ops, d = self._lora_ops, self.d
        rp, Rp, dev = ops["rp"], ops["Rp"], dqkv.device
        p_ = float(self._lora["dropout"])
        sc = self._lora_scratch_for(T, dev)
        du = torch.empty(T, Rp, dtype=torch.bfloat16, device=dev)
        hip.call("oneprot_gemm_bf16_nt", dqkv, ops["BsT"][i], T, Rp, 3 * d, 3 * d, 3 * d, hip.EPI_BF16, None, du, None, None, None, None, None, 1.0, 0, 0, 0)
        hip.call("oneprot_gemm_bf16_tn", dqkv, u, T, 3 * d, Rp, 3 * d, Rp, raw[1][i], None, ws_tn, ws_tn.numel(), 0)            # dqkv^T u
        dhd = torch.empty(T, d, dtype=torch.bfloat16, device=dev)
        dut = sc["ut"]
        for ti in range(len(self._lora["targets"])):
            stream_id = self._lora_stream(call_id, i, ti)
            dut.copy_(du[:, ti * rp:(ti + 1) * rp])
            hip.call("oneprot_dropout_bf16", h, sc["hd"], T * d, p_, self._lora_seed, stream_id)                                  # the forward's dropout_t(h) again
            hip.call("oneprot_gemm_bf16_tn", dut, sc["hd"], T, rp, d, rp, d, raw[0][i, ti * rp:(ti + 1) * rp], None, ws_tn, ws_tn.numel(), 0)      # du_t^T dropout_t(h)
            hip.call("oneprot_gemm_bf16_nt", dut, ops["AtT"][i, ti], T, d, rp, rp, rp, hip.EPI_BF16, None, dhd, None, None, None, None, None, 1.0, 0, 0, 0)
            if dh16 is not None:
                hip.call("oneprot_dropout_bwd_add_bf16", dhd, dh16, T * d, p_, self._lora_seed, stream_id)
            else:
                hip.call("oneprot_dropout_bwd_add_f32", dhd, dh32, T * d, p_, self._lora_seed, stream_id)

    def _lora_raw_buffers(self, dev):
        ops = self._lora_ops
        return (torch.zeros(self.n_layers, ops["Rp"], self.d, device=dev), torch.zeros(self.n_layers, 3 * self.d, ops["Rp"], device=dev))

    def save_pretrained(self, path):
        """HF-style directory (config.json + model.safetensors) -- used by ref peft_checkpoint.py:20."""
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump({k: v for k, v in self.config.to_dict().items() if isinstance(v, (int, float, str, bool, type(None), list))}, f, indent=1)
        sd = {k: v.detach().cpu().contiguous().clone() for k, v in self.state_dict().items()}
        save_file(sd, os.path.join(path, "model.safetensors"))


class EsmTransformer(ArenaModule):
    """EsmModel replacement.  `add_pooling_layer` only controls whether the (unused) HF pooler parameters exist,
    as in the reference (SequenceEncoder: False, StructTokenEncoder: True)."""

    def __init__(self, config, add_pooling_layer=True):
        super().__init__()
        if getattr(config, "position_embedding_type", "rotary") != "rotary" or getattr(config, "emb_layer_norm_before", False):
            raise NotImplementedError("only rotary ESM-2 configurations are on the OneProt hot path")
        self.config = config
        d, f, n, V = config.hidden_size, config.intermediate_size, config.num_hidden_layers, config.vocab_size
        self.d, self.f, self.n_layers, self.H = d, f, n, config.num_attention_heads
        self.hd = d // self.H
        # The attention / RoPE kernels are built for head_dim 16/32/64.  Any other even head_dim <= 64 (ESM-2-35M: 24) runs on the
        # next larger one: each head's two RoPE halves are placed at the start of the two halves of a zero-padded head
        # (`_head_pad_maps`), in the bf16 GEMM operands only -- parameters, gradients and checkpoints keep the HF shapes.
        self.hdp = next((c for c in (16, 32, 64) if c >= self.hd), None)
        if self.hdp is None or self.hd % 2 or self.hd * self.H != d or d % 8 or (self.H * self.hdp) % 64:
            raise NotImplementedError(f"head_dim {self.hd} x {self.H} heads (hidden {d}): kernels need an even head_dim <= 64 and heads*padded_head_dim % 64 == 0")
        self.dp = self.H * self.hdp
        self._padded = self.hdp != self.hd
        self._init_arena()
        self._add("embeddings.word_embeddings.weight", (V, d))
        for i in range(n):
            p = f"encoder.layer.{i}."
            for nm in ("query", "key", "value"):
                self._add(p + f"attention.self.{nm}.weight", (d, d))
            for nm in ("query", "key", "value"):
                self._add(p + f"attention.self.{nm}.bias", (d,))
            self._add(p + "attention.output.dense.weight", (d, d))
            self._add(p + "attention.output.dense.bias", (d,))
            self._add(p + "attention.LayerNorm.weight", (d,))
            self._add(p + "attention.LayerNorm.bias", (d,))
            self._add(p + "intermediate.dense.weight", (f, d))
            self._add(p + "intermediate.dense.bias", (f,))
            self._add(p + "output.dense.weight", (d, f))
            self._add(p + "output.dense.bias", (d,))
            self._add(p + "LayerNorm.weight", (d,))
            self._add(p + "LayerNorm.bias", (d,))
        self._add("encoder.emb_layer_norm_after.weight", (d,))
        self._add("encoder.emb_layer_norm_after.bias", (d,))
        # parameters HF carries but the hot path never touches (kept for strict state-dict compatibility)
        if add_pooling_layer:
            self._extra["pooler.dense.weight"] = (d, d)
            self._extra["pooler.dense.bias"] = (d,)
        self._extra["contact_head.regression.weight"] = (1, n * self.H)
        self._extra["contact_head.regression.bias"] = (1,)
        self._finish_arena()
        inv_freq = 1.0 / (10000.0 ** (torch.arange(0, self.hd, 2, dtype=torch.float32) / self.hd))
        self.register_buffer("inv_freq", inv_freq, persistent=False)
        self._rope_cache = {}
        self._pad_ops = {}
        self._wo_packed = {}
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self):
        std = getattr(self.config, "initializer_range", 0.02)
        for name in self._spec:
            v = self.view(name)
            if name.endswith("LayerNorm.weight") or name.endswith("layer_norm_after.weight"):
                v.fill_(1.0)
            elif name.endswith(".bias"):
                v.zero_()
            else:
                v.normal_(0.0, std)
        pad = self.config.pad_token_id
        if pad is not None:
            self.view("embeddings.word_embeddings.weight")[pad].zero_()
        for k, p in self.extra.items():
            if p.dim() > 1:
                p.normal_(0.0, std)

    _sd_extra_buffers = (("rotary_embeddings.inv_freq", lambda self: self.inv_freq.detach().clone()),)
    # rotary buffers: model-level (transformers >= 5) or per-layer (4.33) keys are accepted, values recomputed;
    # hub checkpoints also carry an unused absolute position table that HF ignores as well
    _load_ignore_suffixes = ("rotary_embeddings.inv_freq", "embeddings.position_embeddings.weight", "embeddings.position_ids")

    def resize_token_embeddings(self, new_vocab):
        """ref struct_token_encoder.py:27: append rows (normal(0, initializer_range), as transformers 4.33 did)."""
        old = self.config.vocab_size
        if new_vocab == old:
            return
        old_views = {k: v.clone() for k, v in self.named_views().items()}
        old_spec = self._spec
        self.config.vocab_size = new_vocab
        self._spec, self._cursor = OrderedDict(), 0
        for name, (_, _, shape) in old_spec.items():
            self._add(name, (new_vocab, self.d) if name == "embeddings.word_embeddings.weight" else shape)
        self._total = self._cursor
        req = self.flat.requires_grad
        self.flat = nn.Parameter(torch.zeros(self._total, device=self.flat.device), requires_grad=req)
        with torch.no_grad():
            for name, v in old_views.items():
                dst = self.view(name)
                if name == "embeddings.word_embeddings.weight":
                    dst[:old] = v[:old]
                    dst[old:].normal_(0.0, getattr(self.config, "initializer_range", 0.02))
                else:
                    dst.copy_(v)
        self._bf16_version = None

    def get_input_embeddings_weight(self):
        return self.view("embeddings.word_embeddings.weight")

    # ----------------------------------------------------------------------------------------- device-side caches
    def _refresh_bf16(self):
        if not self._refresh_bf16_mirror():
            return
        dev = self.flat.device
        if self._padded:
            self._refresh_padded_heads()
        # out-projection weights packed for the full-row GEMM that also applies the following LayerNorm (csrc/gemm_nt_ln.hip; d = 640 only)
        if self._fused_ln_ok():
            for i in range(self.n_layers):
                t = self._wo_packed.get(i)
                if t is None or t.device != dev:
                    t = self._wo_packed[i] = torch.empty(self.d * self.dp, dtype=torch.bfloat16, device=dev)
                hip.call("oneprot_gemm_ln_pack_weight", self._w16(f"encoder.layer.{i}.attention.output.dense.weight"), t, self.d, self.dp)
        if self.flat.requires_grad:
            d, f = self.d, self.f
            self._transpose_layer_weights("w1", "encoder.layer.{i}.intermediate.dense.weight", f, d)
            self._transpose_layer_weights("w2", "encoder.layer.{i}.output.dense.weight", d, f)
            if not self._padded:
                self._transpose_layer_weights("o", "encoder.layer.{i}.attention.output.dense.weight", d, d)
                self._transpose_qkv_weights()

    def _fused_ln_ok(self):
        """the full-row GEMM + LayerNorm kernel is built for 640-wide rows with un-padded heads (ESM-2-150M); ONEPROT_FUSED_LN=0 keeps the pair (A/B runs)"""
        return self.d == 640 and not self._padded and os.environ.get("ONEPROT_FUSED_LN", "1") != "0"

    def _head_pad_maps(self):
        """(rowmap [3d], colmap [d]): position of HF row s*d + h*hd + j of the fused QKV weight inside the padded [3*dp] layout, and of
        context column h*hd + j inside [dp].  j < hd/2 -> j ; j >= hd/2 -> hdp/2 + (j - hd/2): the RoPE partner of padded column c is
        c +- hdp/2, as the hd=hdp kernels assume."""
        dev = self.flat.device
        if self._pad_ops.get("maps_dev") != dev:
            half, hhp = self.hd // 2, self.hdp // 2
            j = torch.arange(self.hd)
            within = torch.where(j < half, j, hhp + j - half)
            col = (torch.arange(self.H)[:, None] * self.hdp + within[None, :]).reshape(-1)
            row = (torch.arange(3)[:, None] * self.dp + col[None, :]).reshape(-1)
            self._pad_ops["maps"] = (row.to(dev), col.to(dev))
            self._pad_ops["maps_dev"] = dev
        return self._pad_ops["maps"]

    def _refresh_padded_heads(self):
        """bf16 QKV / out-proj operands (and their transposes for the dgrad GEMMs) in the padded-head layout; a layout permutation
        of the bf16 mirror, redone whenever the mirror is."""
        d, dp, dev = self.d, self.dp, self.flat.device
        rowmap, colmap = self._head_pad_maps()
        for i in range(self.n_layers):
            p = f"encoder.layer.{i}."
            o, n = self.span(p + "attention.self.query.weight", p + "attention.self.value.weight")
            ob, nb = self.span(p + "attention.self.query.bias", p + "attention.self.value.bias")
            ops = self._pad_ops.get(i)
            if ops is None or ops["qkv"].device != dev:
                ops = dict(qkv=torch.zeros(3 * dp, d, dtype=torch.bfloat16, device=dev), bqkv=torch.zeros(3 * dp, device=dev),
                           o=torch.zeros(d, dp, dtype=torch.bfloat16, device=dev))
                self._pad_ops[i] = ops
            ops["qkv"][rowmap] = self._bf16[o:o + n].view(3 * d, d)
            ops["bqkv"][rowmap] = self.flat.data[ob:ob + nb]
            ops["o"][:, colmap] = self._w16(p + "attention.output.dense.weight")
            if self.flat.requires_grad:
                self._bf16_T[(i, "qkv")] = ops["qkv"].t().contiguous()       # [d, 3dp]
                self._bf16_T[(i, "o")] = ops["o"].t().contiguous()           # [dp, d]

    def _qkv_operands(self, i):
        """(W_qkv bf16 [3*dp, d], bias fp32 [3*dp], W_o bf16 [d, dp]) for layer i"""
        p = f"encoder.layer.{i}."
        if self._padded:
            ops = self._pad_ops[i]
            return ops["qkv"], ops["bqkv"], ops["o"]
        o, n = self.span(p + "attention.self.query.weight", p + "attention.self.value.weight")
        ob, nb = self.span(p + "attention.self.query.bias", p + "attention.self.value.bias")
        return self._bf16[o:o + n], self.flat.data[ob:ob + nb], self._w16(p + "attention.output.dense.weight")

    def _rope(self, L):
        key = (L, self.flat.device)
        if key not in self._rope_cache:
            t = torch.arange(L, dtype=torch.float32)
            freqs = torch.outer(t, self.inv_freq.detach().float().cpu())      # hf modeling_esm.py:150-158 (positions = arange(L))
            cos, sin = freqs.cos(), freqs.sin()
            if self._padded:      # padded dims hold zeros; rotate them by angle 0
                extra = self.hdp // 2 - self.hd // 2
                cos = torch.cat([cos, torch.ones(L, extra)], 1)
                sin = torch.cat([sin, torch.zeros(L, extra)], 1)
            self._rope_cache[key] = (cos.contiguous().to(self.flat.device), sin.contiguous().to(self.flat.device))
        return self._rope_cache[key]

    # ----------------------------------------------------------------------------------------- forward / backward
    def run_layers(self, ids, save):
        """Embedding + n layers.  Returns (x_final fp32 [T,d], saved-dict or None)."""
        if not ids.is_cuda:
            raise hip.HipKernelError("OneProt HIP path needs CUDA(ROCm) tensors; there is no CPU fallback")
        self._refresh_bf16()
        cfg = self.config
        B, L = ids.shape
        T, d, f, H, hd, dp = B * L, self.d, self.f, self.H, self.hdp, self.dp        # hd: kernel (padded) head dim
        q_scale = self.hd ** -0.5
        dev = ids.device
        ids = ids.contiguous()
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        b16 = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)
        cos, sin = self._rope(L)
        key_bias = f32(B, L)
        hip.call("oneprot_key_padding_bias", ids, key_bias, T, cfg.pad_token_id)
        x = f32(T, d)
        row_scale = f32(B)
        hip.call("oneprot_esm_embed_fwd", ids, self.view("embeddings.word_embeddings.weight"), x, row_scale, B, L, d, cfg.vocab_size,
                 cfg.pad_token_id, cfg.mask_token_id, 1 if cfg.token_dropout else 0)
        saved = dict(ids=ids, key_bias=key_bias, row_scale=row_scale, layers=[], B=B, L=L) if save else None
        h = b16(T, d)
        q, k, v = b16(B, H, L, hd), b16(B, H, L, hd), b16(B, H, L, hd)
        ctx_ = b16(T, dp)
        u = b16(T, f)
        eps = cfg.layer_norm_eps
        fused_ln = self._fused_ln_ok() and T % 128 == 0
        # FFN-2 + residual AND the next layer's first LayerNorm in one launch of the 8-phase GEMM (row statistics completed across the work-groups of a row
        # panel: oneprot_gemm_bf16_nt_resid_ln8); ONEPROT_FFN2_LN=0 keeps the pair (A/B runs)
        ln8_mode = os.environ.get("ONEPROT_FFN2_LN", "1")    # "0": never; "force": whenever the shape is served (tests: small batches against the oracle); else when it pays
        ln8_min = 0 if ln8_mode == "0" else (1 if ln8_mode == "force" else 2)
        ffn2_ln = ln8_min > 0 and not self._padded and hip.query("oneprot_gemm_resid_ln8_eligible", T, d, f) >= ln8_min
        outproj_ln8 = not fused_ln and ln8_min > 0 and not self._padded and hip.query("oneprot_gemm_resid_ln8_eligible", T, d, dp) >= ln8_min
        sched_ws = hip.sched_workspace(T) if (ffn2_ln or outproj_ln8 or hip.dynamic_tiles_wanted()) else None      # (pointer, bytes): partial row statistics + work queues
        pre = None                                          # (h1, stats [2,T] or None) of this layer, already written by the previous layer's FFN-2 launch
        lora_two = self._lora_two_branch()
        if lora_two:      # every call draws its own dropout masks; the backward regenerates them from (seed, call, layer)
            lora_call = self._lora_calls
            self._lora_calls += 1
            if save:
                saved["lora_call"] = lora_call
        for i in range(self.n_layers):
            p = f"encoder.layer.{i}."
            if save:
                stats1 = pre[1] if pre is not None else f32(2, T)      # mean | rstd of the first LayerNorm
                st = dict(x_in=x, mean1=stats1[0], rstd1=stats1[1], mean2=f32(T), rstd2=f32(T), h1=pre[0] if pre is not None else b16(T, d), q=b16(B, H, L, hd),
                          k=b16(B, H, L, hd), v=b16(B, H, L, hd), ctx=b16(T, dp), lse=f32(B, H, L), h2=b16(T, d), z=torch.empty(T, f, dtype=torch.uint8, device=dev),
                          u=b16(T, f))
                h1, q, k, v, ctx_, h2, u, z = st["h1"], st["q"], st["k"], st["v"], st["ctx"], st["h2"], st["u"], st["z"]
                m1, r1, m2, r2, lse = st["mean1"], st["rstd1"], st["mean2"], st["rstd2"], st["lse"]
            else:
                h1 = h2 = h
                z = m1 = r1 = m2 = r2 = lse = None
            if pre is None:
                hip.call("oneprot_layernorm_fwd", x, 0, self.view(p + "attention.LayerNorm.weight"), self.view(p + "attention.LayerNorm.bias"), h1, None,
                         m1, r1, T, d, eps)
            w_qkv, b_qkv, w_o = self._qkv_operands(i)
            if lora_two:      # peft's two branches in one launch: [h1 | dropout(h1) A^T] x [W | s B]^T  (K = Kc)
                xc, lora_u = self._lora_branch_operand(i, h1, T, lora_call)
                kc = self._lora_ops["Kc"]
                hip.call("oneprot_gemm_bf16_nt", xc, self._lora_ops["Wc"][i], T, 3 * dp, kc, kc, kc, hip.EPI_QKV_ROPE, b_qkv, q, k, v, None, cos, sin,
                         q_scale * hip.LOG2E, L, H, hd)
                if save:
                    st["lora_u"] = lora_u
            else:
                hip.call("oneprot_gemm_bf16_nt", h1, w_qkv, T, 3 * dp, d, d, d, hip.EPI_QKV_ROPE, b_qkv, q, k, v, None, cos, sin, q_scale * hip.LOG2E, L, H, hd)
            hip.call("oneprot_attn_fwd", q, k, v, key_bias, ctx_, lse, B, H, L, hd)
            x_mid = f32(T, d) if save else x
            if fused_ln:
                # out-projection + bias + residual AND the FFN's pre-LayerNorm in one full-row kernel: x_mid is not read back by a LayerNorm launch
                # (measured on cfg-2: 0.28-0.29 ms against 0.31-0.32 ms for the pair; the FFN-2 / next-layer pair goes through the 8-phase GEMM with the
                # statistics finished across work-groups instead, below -- this full-row form ties with the pair there, NOTEBOOK.md section 6c)
                hip.call("oneprot_gemm_bf16_nt_resid_ln", ctx_, self._wo_packed[i], T, d, dp, dp, self.view(p + "attention.output.dense.bias"), x, x_mid,
                         self.view(p + "LayerNorm.weight"), self.view(p + "LayerNorm.bias"), eps, h2, m2, r2)
            elif outproj_ln8:
                # wider rows (ESM-2-650M, d = 1280): the same pair through the 8-phase GEMM with the statistics finished across its four column tiles
                stats2 = f32(2, T) if save else None
                hip.call("oneprot_gemm_bf16_nt_resid_ln8", ctx_, w_o, T, d, dp, dp, dp, self.view(p + "attention.output.dense.bias"), x, x_mid,
                         self.view(p + "LayerNorm.weight"), self.view(p + "LayerNorm.bias"), eps, h2, stats2, *sched_ws)
                if save:
                    st["mean2"], st["rstd2"] = stats2[0], stats2[1]
            else:
                hip.call("oneprot_gemm_bf16_nt", ctx_, w_o, T, d, dp, dp, dp, hip.EPI_BIAS_RESID,
                         self.view(p + "attention.output.dense.bias"), x_mid, None, None, x, None, None, 1.0, 0, 0, 0)
                hip.call("oneprot_layernorm_fwd", x_mid, 0, self.view(p + "LayerNorm.weight"), self.view(p + "LayerNorm.bias"), h2, None, m2, r2, T, d, eps)
            hip.call("oneprot_gemm_bf16_nt", h2, self._w16(p + "intermediate.dense.weight"), T, f, d, d, d, hip.EPI_BIAS_GELU,
                     self.view(p + "intermediate.dense.bias"), u, z, None, None, None, None, 1.0, 0, 0, 0)
            if z is not None and _GELU_CODE_DROP_BITS:      # experiment hook (tests: what the one-byte gelu' codes cost the gradient): keep only the top 8 - n bits
                nb = _GELU_CODE_DROP_BITS
                z.add_(1 << (nb - 1)).bitwise_and_(0xFF & ~((1 << nb) - 1))
            x_out = f32(T, d) if save else x_mid
            if ffn2_ln and i + 1 < self.n_layers:
                pn = f"encoder.layer.{i + 1}.attention.LayerNorm."
                pre = (b16(T, d), f32(2, T)) if save else (h, None)
                hip.call("oneprot_gemm_bf16_nt_resid_ln8", u, self._w16(p + "output.dense.weight"), T, d, f, f, f, self.view(p + "output.dense.bias"), x_mid, x_out,
                         self.view(pn + "weight"), self.view(pn + "bias"), eps, pre[0], pre[1], *sched_ws)
            else:
                pre = None
                hip.call("oneprot_gemm_bf16_nt", u, self._w16(p + "output.dense.weight"), T, d, f, f, f, hip.EPI_BIAS_RESID,
                         self.view(p + "output.dense.bias"), x_out, None, None, x_mid, None, None, 1.0, 0, 0, 0)
            if save:
                st["x_mid"] = x_mid
                saved["layers"].append(st)
            x = x_out
        if save:
            saved["x_final"] = x
        return x, saved

    GRAD_CHUNK_LAYERS = 6      # arena-gradient ranges are handed to `on_ready` every this many layers (overlapped all-reduce)

    def backward_layers(self, saved, g, g16, gflat, on_ready=None):
        """g: fp32 [T,d] gradient w.r.t. the last layer's output (consumed in place), g16: its bf16 copy;
        gflat: fp32 arena gradient (written).  on_ready(lo, hi): called as soon as the arena-gradient range [lo, hi) is final
        (the arena is laid out embeddings | layer 0 .. n-1 | final LayerNorm, and the backward walks the layers downwards), so the
        data-parallel all-reduce of the upper layers runs under the backward of the lower ones."""
        B, L = saved["B"], saved["L"]
        T, d, f, H, hd, dp = B * L, self.d, self.f, self.H, self.hdp, self.dp
        q_scale = self.hd ** -0.5
        dev = g.device
        cfg = self.config
        cos, sin = self._rope(L)
        gv = lambda name: self.view(name, gflat)
        b16 = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)
        ws_ln = torch.empty(hip.query("oneprot_layernorm_bwd_workspace", d), dtype=torch.uint8, device=dev)
        lora_raw = None
        tn_shapes = ((3 * dp, d), (f, d), (d, f), (d, dp))
        if "lora_call" in saved:                          # the forward ran peft's two-branch form (train mode, lora_dropout > 0)
            lora_raw = self._lora_raw = self._lora_raw_buffers(dev)
            tn_shapes += ((3 * d, self._lora_ops["Rp"]), (self._lora_ops["rp"], d))
        ws_tn = self._tn_workspace(tn_shapes, dev)
        ws_at = torch.empty(hip.query("oneprot_attn_bwd_workspace", B, H, L), dtype=torch.uint8, device=dev)
        dz = b16(T, f)
        dh = b16(T, d)
        dctx = b16(T, dp) if self._padded else dh
        dqkv = b16(T, 3 * dp)
        ready_hi = self._total
        if self._padded:      # weight gradients come out in the padded-head layout and are gathered back into the arena gradient
            rowmap, colmap = self._head_pad_maps()
            gw_qkv, gb_qkv, gw_o = torch.empty(3 * dp, d, device=dev), torch.empty(3 * dp, device=dev), torch.empty(d, dp, device=dev)
        for i in reversed(range(self.n_layers)):
            st = saved["layers"][i]
            p = f"encoder.layer.{i}."
            # ---- FFN2: x_out = x_mid + u W2^T + b2        (weight grad + bias grad in one TN launch)
            self._wgrad(g16, st["u"], T, d, f, d, f, gv(p + "output.dense.weight"), gv(p + "output.dense.bias"), ws_tn)
            hip.call("oneprot_gemm_bf16_nt", g16, self._bf16_T[(i, "w2")], T, f, d, d, d, hip.EPI_GELU_BWD, None, dz, None, None, st["z"], None, None,
                     1.0, 0, 0, 0)
            # ---- FFN1: z = h2 W1^T + b1
            self._wgrad(dz, st["h2"], T, f, d, f, d, gv(p + "intermediate.dense.weight"), gv(p + "intermediate.dense.bias"), ws_tn)
            hip.call("oneprot_gemm_bf16_nt", dz, self._bf16_T[(i, "w1")], T, d, f, f, f, hip.EPI_BF16, None, dh, None, None, None, None, None, 1.0, 0, 0, 0)
            # ---- LN2 (input x_mid): g += LN'(dh); also refreshes the bf16 copy g16
            hip.call("oneprot_layernorm_bwd", dh, 0, None, 0, st["x_mid"], 0, self.view(p + "LayerNorm.weight"), st["mean2"], st["rstd2"], g, g, g16,
                     gv(p + "LayerNorm.weight"), gv(p + "LayerNorm.bias"), ws_ln, T, d, 0)
            # ---- out-proj: x_mid = x_in + ctx Wo^T + bo
            hip.call("oneprot_gemm_bf16_tn", g16, st["ctx"], T, d, dp, d, dp, gw_o if self._padded else gv(p + "attention.output.dense.weight"),
                     gv(p + "attention.output.dense.bias"), ws_tn, ws_tn.numel(), 0)
            hip.call("oneprot_gemm_bf16_nt", g16, self._bf16_T[(i, "o")], T, dp, d, d, d, hip.EPI_BF16, None, dctx, None, None, None, None, None, 1.0, 0, 0, 0)
            # ---- attention
            hip.call("oneprot_attn_bwd", st["q"], st["k"], st["v"], saved["key_bias"], st["ctx"], dctx, st["lse"], cos, sin, q_scale, dqkv, ws_at, B, H, L, hd)
            # ---- QKV projection
            o, n = self.span(p + "attention.self.query.weight", p + "attention.self.value.weight")
            ob, nb = self.span(p + "attention.self.query.bias", p + "attention.self.value.bias")
            hip.call("oneprot_gemm_bf16_tn", dqkv, st["h1"], T, 3 * dp, d, 3 * dp, d, gw_qkv if self._padded else gflat[o:o + n],
                     gb_qkv if self._padded else gflat[ob:ob + nb], ws_tn, ws_tn.numel(), 0)
            if self._padded:
                gv(p + "attention.output.dense.weight").copy_(gw_o[:, colmap])
                gflat[o:o + n].view(3 * d, d).copy_(gw_qkv[rowmap])
                gflat[ob:ob + nb].copy_(gb_qkv[rowmap])
            hip.call("oneprot_gemm_bf16_nt", dqkv, self._bf16_T[(i, "qkv")], T, d, 3 * dp, 3 * dp, 3 * dp, hip.EPI_BF16, None, dh, None, None, None, None, None,
                     1.0, 0, 0, 0)
            if lora_raw is not None:      # two-branch LoRA: adapter gradients, and dh += mask * (du A) / keep
                self._lora_branch_backward(i, st["h1"], st["lora_u"], dqkv, T, saved["lora_call"], ws_tn, lora_raw, dh16=dh)
            # ---- LN1 (input x_in)
            hip.call("oneprot_layernorm_bwd", dh, 0, None, 0, st["x_in"], 0, self.view(p + "attention.LayerNorm.weight"), st["mean1"], st["rstd1"], g, g, g16,
                     gv(p + "attention.LayerNorm.weight"), gv(p + "attention.LayerNorm.bias"), ws_ln, T, d, 0)
            saved["layers"][i] = None      # release this layer's activations
            if on_ready is not None and i > 0 and i % self.GRAD_CHUNK_LAYERS == 0:
                lo = self._spec[f"encoder.layer.{i}.attention.self.query.weight"][0]
                on_ready(lo, ready_hi)
                ready_hi = lo
        V = cfg.vocab_size
        ws_e = torch.empty(hip.query("oneprot_esm_embed_bwd_workspace", T, d, V), dtype=torch.uint8, device=dev)
        hip.call("oneprot_esm_embed_bwd", saved["ids"], g, saved["row_scale"], gv("embeddings.word_embeddings.weight"), ws_e, B, L, d, V,
                 cfg.pad_token_id, cfg.mask_token_id, 1 if cfg.token_dropout else 0, 0)
        if on_ready is not None:
            on_ready(0, ready_hi)

    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, **_):
        """EsmModel-compatible call returning .last_hidden_state (no autograd; the trainable path is
        oneprot_amd.encoders' fused encode)."""
        x, _ = self.run_layers(input_ids, save=False)
        B, L = input_ids.shape
        hidden = torch.empty(B, L, self.d, device=x.device)
        pooled = torch.empty(B, self.d, device=x.device)
        hip.call("oneprot_lnpool_fwd", x, input_ids.contiguous(), self.config.pad_token_id, self.view("encoder.emb_layer_norm_after.weight"),
                 self.view("encoder.emb_layer_norm_after.bias"), pooled, None, None, None, None, hidden, B, L, self.d, self.config.layer_norm_eps, 0)
        return _Out(hidden)

    @classmethod
    def from_pretrained(cls, model_name_or_path, config=None, add_pooling_layer=True, **_):
        cfg, path = resolve_config(model_name_or_path)
        if config is not None:
            cfg = config
        model = cls(cfg, add_pooling_layer=add_pooling_layer)
        sd = load_weight_file(path)
        if sd is None:
            if os.environ.get("ONEPROT_ALLOW_RANDOM_INIT", "0") != "1":
                raise OSError(f"no weights (model.safetensors / pytorch_model.bin) found for {model_name_or_path}; "
                              "set ONEPROT_ALLOW_RANDOM_INIT=1 to build a randomly initialised model of that architecture")
            warnings.warn(f"{model_name_or_path}: no weight file, using random initialisation")
        else:
            sd = {(k[4:] if k.startswith("esm.") else k): v for k, v in sd.items()}
            missing, unexpected = model.load_state_dict(sd, strict=False)
            missing = [m for m in missing if not (m.startswith("pooler.") or m.startswith("contact_head.") or m.startswith("extra."))]
            if missing:
                raise OSError(f"checkpoint {model_name_or_path} lacks tensors: {missing[:5]}...")
        return model
