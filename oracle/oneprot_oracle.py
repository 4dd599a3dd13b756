"""CPU oracle for the OneProt contrastive-alignment training step.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this file.  The shipped path
(oneprot_amd/, src/) never does, and fails loudly without the HIP library.

This is a from-scratch restatement, in plain fp32 torch on the CPU, of the
arithmetic the reference executes on its hot path.  Every function cites the
reference line it follows ("ref:" = /root/reference, "hf:" = the third-party
transformers package the reference delegates to; pinned 4.33.2 in the
reference's requirements.txt:7, 5.15.0 installed where the goldens were made --
same maths).

Pinning: tests/test_oracle_golden.py checks every function here against
fixtures produced by running the reference itself (tests/golden/make_golden.py)
-- the reference's own test-suite holds no numeric vectors for this path.
One branch is PARITY UNPINNED: `lora_proj` restates peft 0.5.0's LoRA linear
(ref requirements.txt:8; peft is not installed here, so no fixture could be made).

Weights are passed as a flat dict with the reference's state-dict key names
("transformer.encoder.layer.0.attention.self.query.weight", "proj.1.weight",
"norm.1.log_logit_scale", ...), all fp32 torch tensors.
"""
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------
def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    """nn.LayerNorm over the last dim, biased variance (hf: modeling_esm.py:429,518,552)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x: Tensor) -> Tensor:
    """x * 0.5 * (1 + erf(x / sqrt(2)))  (hf: modeling_esm.py:82-86; nn.GELU() default in ref base_encoder.py:161)."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    y = x @ w.t()
    return y if b is None else y + b



def lora_proj(x: Tensor, sd: Dict[str, Tensor], key: str, scaling: Optional[float], keep: Optional[Tensor] = None) -> Tensor:
    """A query/key/value projection, optionally wrapped by a LoRA adapter as peft 0.5.0's `lora.Linear.forward` computes it (ref
    sequence_encoder.py:61-74, text_encoder.py:39-52 call `get_peft_model`; peft is a third-party dependency pinned at 0.5.0 in ref
    requirements.txt:8 and ABSENT here, so this branch restates its published algorithm and is *parity unpinned*):
        y = x W^T + b + (lora_alpha / r) * (dropout(x) A^T) B^T
    with A = `<key>lora_A.default.weight` [r, d], B = `<key>lora_B.default.weight` [d, r].  `keep` is this module's dropout mask already
    divided by the keep probability (nn.Dropout in train mode: x * mask / (1 - p)); None = eval mode / p = 0."""
    y = linear(x, sd[key + "weight"], sd[key + "bias"])
    a = sd.get(key + "lora_A.default.weight")
    if a is not None:
        xd = x if keep is None else x * keep
        y = y + scaling * ((xd @ a.t()) @ sd[key + "lora_B.default.weight"].t())
    return y


def strip_peft_prefix(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """PeftModel state-dict keys ("transformer.base_model.model.<hf key>") -> the plain "transformer.<hf key>" this file indexes by."""
    return {k.replace("transformer.base_model.model.", "transformer."): v for k, v in sd.items()}


def rope_tables(L: int, head_dim: int, base: float = 10000.0):
    """cos/sin [L, head_dim] for positions arange(L) (hf: modeling_esm.py:106-160, 733-737).

    inv_freq = 1 / base^(2i/hd); emb = cat(freqs, freqs) -> half-split layout.
    Positions are NOT padding-aware."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    t = torch.arange(L, dtype=torch.float32)
    freqs = torch.outer(t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x: Tensor) -> Tensor:
    """(-x2, x1) with x split in contiguous halves (hf: modeling_esm.py:48-52)."""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """x: [B,H,L,hd] (hf: modeling_esm.py:55-79)."""
    return x * cos[None, None] + rotate_half(x) * sin[None, None]


# --------------------------------------------------------------------------------------
# ESM-2 encoder (third-party arithmetic the reference calls at
# ref: sequence_encoder.py:78, struct_token_encoder.py:31)
# --------------------------------------------------------------------------------------
def esm_embeddings(ids: Tensor, attn_mask: Tensor, W: Tensor, mask_token_id: int, token_dropout: bool = True) -> Tensor:
    """hf: modeling_esm.py:224-271 (rotary variant: no position table, no LN before)."""
    x = W[ids]
    if token_dropout:
        is_mask = ids == mask_token_id
        x = x.masked_fill(is_mask.unsqueeze(-1), 0.0)
        mask_ratio_train = 0.15 * 0.8
        src_len = attn_mask.sum(-1)
        ratio_obs = is_mask.sum(-1).float() / src_len
        x = x * (1 - mask_ratio_train) / (1 - ratio_obs)[:, None, None]
    x = x * attn_mask.unsqueeze(-1).to(x.dtype)
    return x


def additive_key_mask(attn_mask: Tensor) -> Tensor:
    """[B,L] {0,1} -> [B,1,1,L] additive mask, finfo.min at padded keys
    (hf: masking_utils.create_bidirectional_mask / eager softmax, modeling_esm.py:292-317)."""
    neg = torch.finfo(torch.float32).min
    return ((1.0 - attn_mask.float()) * neg)[:, None, None, :]


def esm_layer(x: Tensor, sd: Dict[str, Tensor], pre: str, heads: int, key_mask: Tensor, cos: Tensor, sin: Tensor,
              eps: float, lora_scaling: Optional[float] = None, lora_keep: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """One pre-LN EsmLayer (hf: modeling_esm.py:340-521).  lora_keep: {"query" | "key" | "value": dropout mask / keep probability} of the
    adapters' own dropout modules in train mode (peft 0.5.0: one nn.Dropout per wrapped Linear)."""
    B, L, d = x.shape
    hd = d // heads
    lk = lora_keep or {}
    h = layer_norm(x, sd[pre + "attention.LayerNorm.weight"], sd[pre + "attention.LayerNorm.bias"], eps)
    q = lora_proj(h, sd, pre + "attention.self.query.", lora_scaling, lk.get("query"))
    k = lora_proj(h, sd, pre + "attention.self.key.", lora_scaling, lk.get("key"))
    v = lora_proj(h, sd, pre + "attention.self.value.", lora_scaling, lk.get("value"))
    q = q.view(B, L, heads, hd).transpose(1, 2)
    k = k.view(B, L, heads, hd).transpose(1, 2)
    v = v.view(B, L, heads, hd).transpose(1, 2)
    q = q * hd ** -0.5                       # scale BEFORE rotary (modeling_esm.py:374)
    q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
    s = q @ k.transpose(-1, -2) + key_mask
    p = torch.softmax(s, dim=-1)
    a = (p @ v).transpose(1, 2).reshape(B, L, d)
    x = x + linear(a, sd[pre + "attention.output.dense.weight"], sd[pre + "attention.output.dense.bias"])
    h = layer_norm(x, sd[pre + "LayerNorm.weight"], sd[pre + "LayerNorm.bias"], eps)
    u = gelu_erf(linear(h, sd[pre + "intermediate.dense.weight"], sd[pre + "intermediate.dense.bias"]))
    x = x + linear(u, sd[pre + "output.dense.weight"], sd[pre + "output.dense.bias"])
    return x


def esm_forward(ids: Tensor, sd: Dict[str, Tensor], cfg: dict, pre: str = "transformer.", taps: Optional[dict] = None) -> Tensor:
    """EsmModel.forward -> last_hidden_state [B,L,d] (hf: modeling_esm.py:680-760)."""
    attn_mask = (ids != cfg["pad"]).long()          # ref: sequence_encoder.py:77
    x = esm_embeddings(ids, attn_mask, sd[pre + "embeddings.word_embeddings.weight"], cfg["mask"])
    if taps is not None:
        taps["embeddings"] = x
    L = ids.shape[1]
    cos, sin = rope_tables(L, cfg["hidden"] // cfg["heads"])
    km = additive_key_mask(attn_mask)
    for i in range(cfg["layers"]):
        x = esm_layer(x, sd, f"{pre}encoder.layer.{i}.", cfg["heads"], km, cos, sin, cfg["eps"], cfg.get("lora_scaling"), (cfg.get("lora_keep") or {}).get(i))
        if taps is not None:
            taps[f"layer{i}"] = x
    x = layer_norm(x, sd[pre + "encoder.emb_layer_norm_after.weight"], sd[pre + "encoder.emb_layer_norm_after.bias"], cfg["eps"])
    return x


# --------------------------------------------------------------------------------------
# BERT encoder (third-party arithmetic called at ref: text_encoder.py:59)
# --------------------------------------------------------------------------------------
def bert_forward(ids: Tensor, sd: Dict[str, Tensor], cfg: dict, pre: str = "transformer.", taps: Optional[dict] = None) -> Tensor:
    """BertModel.forward in eval mode (dropout off) -> last_hidden_state
    (hf: modeling_bert.py:53-108 embeddings, 139-204 attention, 354-417 layer).  Post-LN, scale inside attention."""
    B, T = ids.shape
    heads, d, eps = cfg["heads"], cfg["hidden"], cfg["eps"]
    hd = d // heads
    attn_mask = (ids != cfg["pad"]).long()
    e = pre + "embeddings."
    x = sd[e + "word_embeddings.weight"][ids] + sd[e + "token_type_embeddings.weight"][0][None, None] \
        + sd[e + "position_embeddings.weight"][:T][None]
    x = layer_norm(x, sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], eps)
    # train mode (hf modeling_bert.py: BertEmbeddings.dropout, BertSelfAttention.dropout on the probabilities, BertSelfOutput.dropout, BertOutput.dropout;
    # the reference leaves them active whenever the tower is in train mode, text_encoder.py:59): cfg["bert_keep"] = {"emb": m, "layers": [{"attn": m
    # [B, H, T, T], "out1": m, "out2": m}, ...]}, every m a dropout mask already divided by its keep probability; absent = eval mode
    bk = cfg.get("bert_keep")
    if bk is not None:
        x = x * bk["emb"]
    if taps is not None:
        taps["embeddings"] = x
    km = additive_key_mask(attn_mask)
    for i in range(cfg["layers"]):
        p = f"{pre}encoder.layer.{i}."
        lk = (cfg.get("lora_keep") or {}).get(i) or {}      # train-mode dropout masks of the adapters (see lora_proj)
        dk = bk["layers"][i] if bk is not None else None
        q = lora_proj(x, sd, p + "attention.self.query.", cfg.get("lora_scaling"), lk.get("query")).view(B, T, heads, hd).transpose(1, 2)
        k = lora_proj(x, sd, p + "attention.self.key.", cfg.get("lora_scaling"), lk.get("key")).view(B, T, heads, hd).transpose(1, 2)
        v = lora_proj(x, sd, p + "attention.self.value.", cfg.get("lora_scaling"), lk.get("value")).view(B, T, heads, hd).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * hd ** -0.5 + km
        pr = torch.softmax(s, -1)
        if dk is not None:
            pr = pr * dk["attn"]
        a = (pr @ v).transpose(1, 2).reshape(B, T, d)
        a = linear(a, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
        if dk is not None:
            a = a * dk["out1"]
        x = layer_norm(a + x, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"], eps)
        u = gelu_erf(linear(x, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
        o = linear(u, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
        if dk is not None:
            o = o * dk["out2"]
        x = layer_norm(o + x, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)
    return x


# --------------------------------------------------------------------------------------
# BaseEncoder head: pooling -> projection -> L2 normalise [-> logit scale]
# --------------------------------------------------------------------------------------
def mean_pool(x: Tensor, mask: Optional[Tensor]) -> Tensor:
    """ref: base_encoder.py:109-118 -- all non-pad tokens incl. CLS/EOS."""
    if x.dim() == 2:
        return x
    if mask is None:
        return x.mean(1)
    return (x * mask.unsqueeze(2)).sum(1) / mask.sum(1, keepdim=True)


def cls_pool(x: Tensor) -> Tensor:
    """ref: base_encoder.py:125-126."""
    return x[:, 0]


def attention1d_pool(x: Tensor, conv_w: Tensor, conv_b: Tensor, mask: Optional[Tensor]) -> Tensor:
    """ref: base_encoder.py:88-103 (+ MaskedConv1d :40-86, kernel 1 => a per-token dot product).
    NOTE the reference encoders call pooling(x, attention_mask[B,L]); MaskedConv1d only multiplies by the mask when one is
    passed to *it*, which Attention1dPooling never does, so the mask only enters through masked_fill(-inf)."""
    B = x.shape[0]
    attn = (x @ conv_w.view(-1) + conv_b).view(B, -1)
    if mask is not None:
        attn = attn.masked_fill(~mask.view(B, -1).bool(), float("-inf"))
    attn = torch.softmax(attn, dim=-1).view(B, -1, 1)
    return (attn * x).sum(1)


def projection(x: Tensor, sd: Dict[str, Tensor], proj_type: Optional[str]) -> Tensor:
    """ref: base_encoder.py:147-169.  nn.LayerNorm eps 1e-5, bias-free Linears, erf GELU."""
    if proj_type == "linear":
        return linear(layer_norm(x, sd["proj.0.weight"], sd["proj.0.bias"], 1e-5), sd["proj.1.weight"])
    if proj_type == "mlp":
        h = linear(layer_norm(x, sd["proj.0.weight"], sd["proj.0.bias"], 1e-5), sd["proj.1.weight"])
        h = layer_norm(gelu_erf(h), sd["proj.3.weight"], sd["proj.3.bias"], 1e-5)
        return linear(h, sd["proj.4.weight"])
    return x


def l2_normalize(x: Tensor, eps: float = 1e-12) -> Tensor:
    """F.normalize(p=2, dim=-1): x / max(||x||, eps)  (ref: base_encoder.py:11-12)."""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


def logit_scale(x: Tensor, log_logit_scale: Tensor, max_logit_scale: float = 100.0) -> Tensor:
    """ref: base_encoder.py:32-33."""
    return torch.clamp(log_logit_scale.exp(), max=max_logit_scale) * x


def encoder_features(kind: str, ids: Tensor, sd: Dict[str, Tensor], cfg: dict, pooling: str, proj_type: Optional[str],
                     use_logit_scale: bool, taps: Optional[dict] = None) -> Tensor:
    """SequenceEncoder/StructTokenEncoder/TextEncoder.forward
    (ref: sequence_encoder.py:76-81, struct_token_encoder.py:29-34, text_encoder.py:57-62)."""
    mask = (ids != cfg["pad"]).long()
    if any(k.startswith("transformer.base_model.model.") for k in sd):
        sd = strip_peft_prefix(sd)
    hidden = esm_forward(ids, sd, cfg, taps=taps) if kind == "esm" else bert_forward(ids, sd, cfg, taps=taps)
    if taps is not None:
        taps["last_hidden"] = hidden
    if pooling == "mean":
        pooled = mean_pool(hidden, mask)
    elif pooling == "cls":
        pooled = cls_pool(hidden)
    elif pooling == "attention1d":
        pooled = attention1d_pool(hidden, sd["pooling.layer.weight"], sd["pooling.layer.bias"], mask)
    else:
        pooled = hidden
    if taps is not None:
        taps["pooled"] = pooled
    y = projection(pooled, sd, proj_type)
    if taps is not None:
        taps["projected"] = y
    y = l2_normalize(y)
    if use_logit_scale:
        y = logit_scale(y, sd["norm.1.log_logit_scale"])
    return y


def struct_encoder_features(encoded: Tensor, sd: Dict[str, Tensor], proj_type: Optional[str], use_logit_scale: bool) -> Tensor:
    """StructEncoder.forward after the opaque encoder (ref: struct_graph_encoder.py:36-42; dropout is the identity in eval mode / at p = 0):
    proj -> L2 normalise -> logit scale.  `encoded` [B, output_dim] is what the opaque module (ProNet in the reference) returned."""
    y = l2_normalize(projection(encoded, sd, proj_type))
    return logit_scale(y, sd["norm.1.log_logit_scale"]) if use_logit_scale else y


# --------------------------------------------------------------------------------------
# losses (ref: src/models/components/loss.py)
# --------------------------------------------------------------------------------------
def clip_loss(modality_features: Tensor, sequence_features: Tensor, logit_scale_: float = 1.0,
              all_modality: Optional[Tensor] = None, all_sequence: Optional[Tensor] = None,
              rank: int = 0, world_size: int = 1, local_loss: bool = False) -> Tensor:
    """ClipLoss.forward (ref: loss.py:85-114).  For world_size>1 the caller passes the gathered
    features (what gather_features, loss.py:19-46, returns); labels follow loss.py:72-83."""
    if world_size > 1:
        if local_loss:
            lpm = logit_scale_ * modality_features @ all_sequence.t()
            lps = logit_scale_ * sequence_features @ all_modality.t()
        else:
            lpm = logit_scale_ * all_modality @ all_sequence.t()
            lps = lpm.t()
    else:
        lpm = logit_scale_ * modality_features @ sequence_features.t()
        lps = logit_scale_ * sequence_features @ modality_features.t()
    n = lpm.shape[0]
    labels = torch.arange(n, dtype=torch.long)
    if world_size > 1 and local_loss:
        labels = labels + n * rank
    return (F.cross_entropy(lpm, labels) + F.cross_entropy(lps, labels)) / 2


def siglip_block(modality_features: Tensor, sequence_features: Tensor, logit_scale_: float = 1.0,
                 logit_bias: Optional[float] = None, negative_only: bool = False) -> Tensor:
    """SigLipLoss._loss (ref: loss.py:229-255)."""
    logits = logit_scale_ * modality_features @ sequence_features.t()
    if logit_bias is not None:
        logits = logits + logit_bias
    n = modality_features.shape[0]
    labels = -torch.ones(n, n)
    if not negative_only:
        labels = 2 * torch.eye(n) + labels
    return -F.logsigmoid(labels * logits).sum() / n


def siglip_loss_global(all_modality: Tensor, all_sequence: Tensor, rank: int, world_size: int, logit_scale_: float = 1.0,
                       logit_bias: Optional[float] = None) -> Tensor:
    """What rank `rank` obtains from SigLipLoss.forward (ref: loss.py:257-309) once every remote chunk of
    sequence features has circulated: the local block plus negative-only blocks against all other ranks' chunks.
    all_* are [world, B, D]."""
    m = all_modality[rank]
    loss = siglip_block(m, all_sequence[rank], logit_scale_, logit_bias)
    for r in range(world_size):
        if r != rank:
            loss = loss + siglip_block(m, all_sequence[r], logit_scale_, logit_bias, negative_only=True)
    return loss


# --------------------------------------------------------------------------------------
# training sub-step (ref: src/models/oneprot_module.py:92-107) and Adam (torch.optim.Adam, lr 1e-3, wd 0,
# ref: configs/model/default.yaml:2-6)
# --------------------------------------------------------------------------------------
def clip_grad_norm(grads: Dict[str, Tensor], max_norm: float = 1.0):
    """torch.nn.utils.clip_grad_norm_ (global L2; coefficient clamped to 1)  (ref: oneprot_module.py:106)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, {k: g * coef for k, g in grads.items()}


def adam_first_step(p: Tensor, g: Tensor, lr: float = 1e-3, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> Tensor:
    """Adam update from zero state, step 1."""
    m = (1 - b1) * g
    v = (1 - b2) * g * g
    mhat = m / (1 - b1)
    vhat = v / (1 - b2)
    return p - lr * mhat / (vhat.sqrt() + eps)


def train_substep(seq_ids: Tensor, mod_ids: Tensor, sd_seq: Dict[str, Tensor], sd_mod: Dict[str, Tensor], cfg_seq: dict,
                  cfg_mod: dict, seq_spec: dict, mod_spec: dict, use_l1: bool = True, frozen_seq: bool = False,
                  lr: float = 1e-3):
    """One iteration of the loop body at ref: oneprot_module.py:92-107 on one rank:
    fwd seq, fwd mod, loss(+L1), backward (torch autograd over the restated forward), clip 1.0, Adam step 1.
    *_spec = dict(kind, pooling, proj_type, use_logit_scale).  Returns a dict of results."""
    def leafify(sd, trainable):
        out = {}
        for k, v in sd.items():
            t = v.detach().clone()
            if trainable(k) and t.is_floating_point() and k != "norm.1.log_logit_scale" and "inv_freq" not in k:
                t.requires_grad_(True)
            out[k] = t
        return out
    ps = leafify(sd_seq, lambda k: (not frozen_seq) or not k.startswith("transformer."))
    pm = leafify(sd_mod, lambda k: True)
    sf = encoder_features(seq_spec["kind"], seq_ids, ps, cfg_seq, seq_spec["pooling"], seq_spec["proj_type"], seq_spec["use_logit_scale"])
    mf = encoder_features(mod_spec["kind"], mod_ids, pm, cfg_mod, mod_spec["pooling"], mod_spec["proj_type"], mod_spec["use_logit_scale"])
    # ref calls loss_fn(sequence_features, modality_features): argument names swapped, symmetric (oneprot_module.py:100)
    loss_c = clip_loss(sf, mf)
    loss = loss_c + 0.01 * (sf.abs().mean() + mf.abs().mean()) if use_l1 else loss_c
    loss.backward()
    grads = {}
    for pref, d in (("seq.", ps), ("mod.", pm)):
        for k, t in d.items():
            if t.requires_grad and t.grad is not None:
                grads[pref + k] = t.grad
    total, clipped = clip_grad_norm(grads, 1.0)
    new = {}
    for k, g in clipped.items():
        src = ps if k.startswith("seq.") else pm
        new[k] = adam_first_step(src[k[4:]].detach(), g, lr=lr)
    return dict(sequence_features=sf.detach(), modality_features=mf.detach(), loss_clip=loss_c.detach(), loss=loss.detach(),
                grads=grads, grad_total_norm=total, new_params=new)


def train_substep_seqsim(ids_a: Tensor, ids_b: Tensor, sd_seq: Dict[str, Tensor], cfg_seq: dict, seq_spec: dict, use_l1: bool = True, lr: float = 1e-3):
    """The `seqsim` sub-step of ref oneprot_module.py:84-107 with `use_seqsim=True`: BOTH inputs go through the sequence encoder
    (`forward(x, "seqsim")` maps to `network["sequence"]`, oneprot_module.py:67-71), i.e. ONE parameter set is applied twice before one backward;
    its gradient is the sum over the two applications.  Then clip 1.0 and Adam step 1 on that one set."""
    ps = {}
    for k, v in sd_seq.items():
        t = v.detach().clone()
        if t.is_floating_point() and k != "norm.1.log_logit_scale" and "inv_freq" not in k:
            t.requires_grad_(True)
        ps[k] = t
    feats = [encoder_features(seq_spec["kind"], ids, ps, cfg_seq, seq_spec["pooling"], seq_spec["proj_type"], seq_spec["use_logit_scale"]) for ids in (ids_a, ids_b)]
    loss_c = clip_loss(feats[0], feats[1])
    loss = loss_c + 0.01 * (feats[0].abs().mean() + feats[1].abs().mean()) if use_l1 else loss_c
    loss.backward()
    grads = {"seq." + k: t.grad for k, t in ps.items() if t.requires_grad and t.grad is not None}
    total, clipped = clip_grad_norm(grads, 1.0)
    new = {k: adam_first_step(ps[k[4:]].detach(), g, lr=lr) for k, g in clipped.items()}
    return dict(features_a=feats[0].detach(), features_b=feats[1].detach(), loss_clip=loss_c.detach(), loss=loss.detach(), grads=grads,
                grad_total_norm=total, new_params=new)


def train_multi_substeps(seq_ids, mod_ids, sd_seq, sd_mod, cfg_seq, cfg_mod, seq_spec, mod_spec, n_steps, use_l1=True, frozen_seq=False, lr=1e-3):
    """n consecutive iterations of the loop body at ref oneprot_module.py:92-107 with ONE persistent torch.optim.Adam over all
    trainable tensors (ref configure_optimizers :157-170 + configs/model/default.yaml:2-6) and clip-norm 1.0 before every step.
    Returns the list of per-step total losses and the final parameter dicts."""
    def leafify(sd, trainable):
        out = {}
        for k, v in sd.items():
            t = v.detach().clone()
            if trainable(k) and t.is_floating_point() and k != "norm.1.log_logit_scale" and "inv_freq" not in k and not k.startswith("transformer.pooler") \
                    and not k.startswith("transformer.contact_head"):
                t.requires_grad_(True)
            out[k] = t
        return out
    ps = leafify(sd_seq, lambda k: (not frozen_seq) or not k.startswith("transformer."))
    pm = leafify(sd_mod, lambda k: True)
    params = [t for t in list(ps.values()) + list(pm.values()) if t.requires_grad]
    opt = torch.optim.Adam(params, lr=lr, weight_decay=0.0)
    losses = []
    for _ in range(n_steps):
        sf = encoder_features(seq_spec["kind"], seq_ids, ps, cfg_seq, seq_spec["pooling"], seq_spec["proj_type"], seq_spec["use_logit_scale"])
        mf = encoder_features(mod_spec["kind"], mod_ids, pm, cfg_mod, mod_spec["pooling"], mod_spec["proj_type"], mod_spec["use_logit_scale"])
        opt.zero_grad()
        loss = clip_loss(sf, mf)
        if use_l1:
            loss = loss + 0.01 * (sf.abs().mean() + mf.abs().mean())
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        losses.append(float(loss))
    return losses, {k: v.detach() for k, v in ps.items()}, {k: v.detach() for k, v in pm.items()}


def standin_graph_encoder(x: Tensor, sd: Dict[str, Tensor], pre: str = "encoder.") -> Tensor:
    """Functional twin of the bench's stand-in for the opaque pocket encoder (oneprot_amd.data.StandInGraphEncoder -- NOT ProNet, which is
    un-vendored): silu(x W1^T + b1) averaged over the nodes, then W2."""
    h = torch.nn.functional.silu(linear(x, sd[pre + "node.weight"], sd[pre + "node.bias"])).mean(dim=1)
    return linear(h, sd[pre + "out.weight"], sd[pre + "out.bias"])


def _features(spec: dict, ids: Tensor, sd: Dict[str, Tensor], cfg: dict) -> Tensor:
    if spec["kind"] == "opaque_graph":      # StructEncoder (ref struct_graph_encoder.py:36-42) around the stand-in encoder, dropout off
        return struct_encoder_features(standin_graph_encoder(ids, sd), sd, spec["proj_type"], spec["use_logit_scale"])
    return encoder_features(spec["kind"], ids, sd, cfg, spec["pooling"], spec["proj_type"], spec["use_logit_scale"])


def validation_substep(seq_ids, mod_ids, sd_seq, sd_mod, cfg_seq, cfg_mod, seq_spec, mod_spec):
    """OneProtLitModule.validation_step (ref oneprot_module.py:110-121): forward both encoders, plain CLIP loss (no L1 term, logit_scale 1.0).
    Returns (loss, sequence_features, modality_features)."""
    with torch.no_grad():
        sf, mf = _features(seq_spec, seq_ids, sd_seq, cfg_seq), _features(mod_spec, mod_ids, sd_mod, cfg_mod)
        return clip_loss(sf, mf), sf, mf


def test_substep(seq_ids, mod_ids, sd_seq, sd_mod, cfg_seq, cfg_mod, seq_spec, mod_spec):
    """OneProtLitModule.test_step (ref oneprot_module.py:137-146).  NB line 142 hands `network[modality].norm[1].log_logit_scale.exp()` to
    the loss although the modality features already carry that factor: the logits are scaled twice (14.29^2 = 204x).  Restated as written."""
    with torch.no_grad():
        sf, mf = _features(seq_spec, seq_ids, sd_seq, cfg_seq), _features(mod_spec, mod_ids, sd_mod, cfg_mod)
        return clip_loss(sf, mf, float(sd_mod["norm.1.log_logit_scale"].exp())), sf, mf


test_substep.__test__ = False        # not a pytest test


def train_round_robin(batches, sds, cfgs, specs, use_l1=True, frozen=(), train_on_all_modalities_after_step=0, lr=1e-3):
    """The mixed-batch loop (ref oneprot_module.py:80-108 driven by CombinedLoader("min_size") batches, oneprot_datamodule.py:75): ONE
    torch.optim.Adam over every trainable tensor of every component (configure_optimizers :157-170); per batch, per modality in dict order:
    fwd sequence, fwd modality, zero_grad, loss (+L1), backward, clip-norm 1.0 over the parameters that received a gradient, step (Adam skips
    parameters whose grad is None -- the inactive modalities').  While global_step < train_on_all_modalities_after_step only "struct_token"
    trains (:84-86; global_step counts optimiser steps under manual optimisation).
    batches: list of {modality: (seq_ids, mod_ids)}; sds/cfgs/specs: {"sequence": ..., modality: ...}; frozen: component names whose
    `transformer.*` tensors are not trained.  Returns [(modality, loss), ...] in execution order and the final parameter dicts."""
    params = {}
    for name, sd in sds.items():
        out = {}
        for k, v in sd.items():
            t = v.detach().clone()
            trainable = t.is_floating_point() and k != "norm.1.log_logit_scale" and "inv_freq" not in k and not k.startswith("transformer.pooler") \
                and not k.startswith("transformer.contact_head") and not (name in frozen and k.startswith("transformer."))
            if trainable:
                t.requires_grad_(True)
            out[k] = t
        params[name] = out
    leaves = [t for d in params.values() for t in d.values() if t.requires_grad]
    opt = torch.optim.Adam(leaves, lr=lr, weight_decay=0.0)
    log, global_step = [], 0
    for batch in batches:
        mods = ["struct_token"] if global_step < train_on_all_modalities_after_step else [m for m in batch if m != "seqsim"]
        for m in mods:
            seq_ids, mod_ids = batch[m]
            sf = _features(specs["sequence"], seq_ids, params["sequence"], cfgs["sequence"])
            mf = _features(specs[m], mod_ids, params[m], cfgs[m])
            opt.zero_grad()
            loss = clip_loss(sf, mf)
            if use_l1:
                loss = loss + 0.01 * (sf.abs().mean() + mf.abs().mean())
            loss.backward()
            torch.nn.utils.clip_grad_norm_([t for t in leaves if t.grad is not None], 1.0)
            opt.step()
            global_step += 1
            log.append((m, float(loss)))
    return log, {n: {k: v.detach() for k, v in d.items()} for n, d in params.items()}


def retrieval_metrics(sequence_outputs: Tensor, modality_outputs: Tensor, ks=(1, 10, 100)) -> dict:
    """ref retrieval_metric.py:76-102: S @ M^T, descending argsort, position of the diagonal; median rank (floor + 1) and R@k both ways.
    Pinned by tests/golden/retrieval.pt: the reference's RetrievalMetric.compute itself, run by make_golden.py (its torchmetrics base class
    replaced by an import-time placeholder; the arithmetic is the reference's own lines), on features whose similarities are exact in fp32."""
    import numpy as np
    out = {}
    lps = (sequence_outputs @ modality_outputs.t()).detach().cpu()
    gt = torch.arange(len(modality_outputs)).view(-1, 1)
    for name, logit in (("seq_to_mod", lps), ("mod_to_seq", lps.t())):
        ranking = torch.argsort(logit, descending=True)
        preds = torch.where(ranking == gt)[1].numpy()
        out[f"{name}_median_rank"] = float(np.floor(np.median(preds)) + 1)
        for k in ks:
            out[f"{name}_R@{k}"] = float(np.mean(preds < k))
    return out


# --------------------------------------------------------------------------------------
# src/distributed.py:8-38 restated (string logic only)
# --------------------------------------------------------------------------------------
def first_node(nodelist: str) -> str:
    """First hostname of a SLURM nodelist: 'a[1-3,7]' -> 'a1', 'a,b' -> 'a', 'a' -> 'a' (ref: distributed.py:8-38)."""
    lb = nodelist.find("[")
    if lb >= 0 and "]" in nodelist[lb:]:
        inner = nodelist[lb + 1: nodelist.find("]", lb)]
        first = inner.split(",")[0].split("-")[0]
        return nodelist[:lb] + first
    return nodelist.split(",")[0]
