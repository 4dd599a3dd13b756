"""BERT text tower on the HIP kernels -- what the reference obtains from `AutoModel.from_pretrained` in TextEncoder
(ref text_encoder.py:33) i.e. transformers' BertModel (hf modeling_bert.py:53-108 embeddings, 139-204 attention, 354-417 layer;
post-LN, 1/sqrt(hd) inside attention, erf-GELU).  State-dict keys are BertModel's.  Forward and hand-written backward
(`frozen=False`, the TextEncoder signature default).  Dropout follows the reference: HF's four train-mode dropouts (p = 0.1) are active whenever the
module is in train mode, also for the frozen tower (ref text_encoder.py:56-62 never switches it to eval -- SURVEY.md section 8 a7); `.eval()`,
`transformer.train_dropout = False` or ONEPROT_BERT_DROPOUT=0 run p = 0, which is what parity against the eval-mode reference is defined on
(see _train_dropout below)."""
import os
import warnings

import torch

from . import hip
from .esm import ArenaModule, ModelConfig, load_weight_file, resolve_config, _Out

BERT_DEFAULTS = dict(model_type="bert", vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                     max_position_embeddings=512, type_vocab_size=2, pad_token_id=0, layer_norm_eps=1e-12, initializer_range=0.02,
                     hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
KNOWN_BERT = {"BiomedNLP-BiomedBERT-base-uncased-abstract-fulltext": {}, "BiomedNLP-PubMedBERT-base-uncased-abstract-fulltext": {}, "bert-base-uncased": {}}


def resolve_bert_config(model_name_or_path):
    path = str(model_name_or_path)
    if os.path.isfile(os.path.join(path, "config.json")):
        cfg, p = resolve_config(path)
        base = dict(BERT_DEFAULTS)
        base.update(cfg.to_dict())
        return ModelConfig(**base), p
    if path.split("/")[-1] in KNOWN_BERT:
        return ModelConfig(**BERT_DEFAULTS), None
    raise OSError(f"{model_name_or_path} is not a local folder with a config.json and is not a known BERT model identifier")


class BertTransformer(ArenaModule):
    final_layer_norm = False      # post-LN model: pooling reads the last layer's output directly

    def __init__(self, config):
        super().__init__()
        self.config = config
        d, f, n = config.hidden_size, config.intermediate_size, config.num_hidden_layers
        self.d, self.f, self.n_layers, self.H = d, f, n, config.num_attention_heads
        self.hd = d // self.H
        if self.hd not in (16, 32, 64) or d % 64:
            raise NotImplementedError(f"head_dim {self.hd} / hidden {d}: kernels are built for head_dim 16/32/64, hidden % 64 == 0")
        self._init_arena()
        self._add("embeddings.word_embeddings.weight", (config.vocab_size, d))
        self._add("embeddings.position_embeddings.weight", (config.max_position_embeddings, d))
        self._add("embeddings.token_type_embeddings.weight", (getattr(config, "type_vocab_size", 2), d))
        self._add("embeddings.LayerNorm.weight", (d,))
        self._add("embeddings.LayerNorm.bias", (d,))
        for i in range(n):
            p = f"encoder.layer.{i}."
            for nm in ("query", "key", "value"):
                self._add(p + f"attention.self.{nm}.weight", (d, d))
            for nm in ("query", "key", "value"):
                self._add(p + f"attention.self.{nm}.bias", (d,))
            for nm, shp in (("attention.output.dense.weight", (d, d)), ("attention.output.dense.bias", (d,)), ("attention.output.LayerNorm.weight", (d,)),
                            ("attention.output.LayerNorm.bias", (d,)), ("intermediate.dense.weight", (f, d)), ("intermediate.dense.bias", (f,)),
                            ("output.dense.weight", (d, f)), ("output.dense.bias", (d,)), ("output.LayerNorm.weight", (d,)), ("output.LayerNorm.bias", (d,))):
                self._add(p + nm, shp)
        self._extra["pooler.dense.weight"] = (d, d)       # HF pooler: present in checkpoints, unused by the reference
        self._extra["pooler.dense.bias"] = (d,)
        self._finish_arena()
        self._ones_zeros = {}
        self.reset_parameters()

    _load_ignore_suffixes = ("embeddings.position_ids",)

    @torch.no_grad()
    def reset_parameters(self):
        std = getattr(self.config, "initializer_range", 0.02)
        for name in self._spec:
            v = self.view(name)
            if name.endswith("LayerNorm.weight"):
                v.fill_(1.0)
            elif name.endswith(".bias"):
                v.zero_()
            else:
                v.normal_(0.0, std)

    def _norope(self, L, dev):
        key = (L, dev)
        if key not in self._ones_zeros:
            self._ones_zeros[key] = (torch.ones(L, self.hd // 2, device=dev), torch.zeros(L, self.hd // 2, device=dev))
        return self._ones_zeros[key]

    def _refresh_bf16(self):
        if not self._refresh_bf16_mirror():
            return
        if self.flat.requires_grad:          # transposed bf16 weights for the dgrad GEMMs
            d, f = self.d, self.f
            self._transpose_qkv_weights()
            self._transpose_layer_weights("o", "encoder.layer.{i}.attention.output.dense.weight", d, d)
            self._transpose_layer_weights("w1", "encoder.layer.{i}.intermediate.dense.weight", f, d)
            self._transpose_layer_weights("w2", "encoder.layer.{i}.output.dense.weight", d, f)

    # ---- hf's train-mode dropout (hidden_dropout_prob / attention_probs_dropout_prob, 0.1 in bert-base).  The reference never switches its text tower
    # to eval mode (text_encoder.py:59), so even the frozen tower of the shipped configuration is stochastic there -- and so it is here since round 5:
    # ON by default in train mode (ONEPROT_BERT_DROPOUT=0 or `transformer.train_dropout = False` switch it off: the parity tests against the
    # eval-mode reference do).  All four dropouts, for a frozen tower (forward only) and a trainable one (the backward regenerates every mask: the dense
    # outputs' gradients pass through the same hidden masks, the attention backward runs its masked form, oneprot_attn_bwd_dropout).  Hidden masks come
    # from the counter-based generator of the LoRA dropout (Philox4x32-10 of seed, call, layer, site, element), the attention masks from a per-element hash.
    train_dropout = None

    def _train_dropout(self):
        on = self.train_dropout if self.train_dropout is not None else os.environ.get("ONEPROT_BERT_DROPOUT", "1") != "0"
        cfg = self.config
        if not (on and self.training):
            return False
        if not (float(cfg.hidden_dropout_prob) > 0 or float(cfg.attention_probs_dropout_prob) > 0):
            return False
        if getattr(self, "_drop_seed", None) is None:
            self._drop_seed = int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF
            self._drop_calls = 0
        return True

    def _drop_stream(self, call_id, layer, site):
        """stream id of a dropout site: layer -1 = embeddings; site 0 = attention probabilities, 1 = attention output dense, 2 = FFN output dense"""
        return self._rng_stream(self.RNG_DOMAIN_BERT, (call_id * (self.n_layers + 1) + (layer + 1)) * 4 + site)

    @torch.no_grad()
    def run_layers(self, ids, save=False):
        """Embeddings + n post-LN layers.  Returns (last hidden state fp32 [T,d], saved-dict or None)."""
        if not ids.is_cuda:
            raise hip.HipKernelError("OneProt HIP path needs CUDA(ROCm) tensors; there is no CPU fallback")
        self._refresh_bf16()
        cfg = self.config
        B, L = ids.shape
        if L > cfg.max_position_embeddings:
            raise ValueError(f"sequence length {L} exceeds max_position_embeddings {cfg.max_position_embeddings}")
        T, d, f, H, hd = B * L, self.d, self.f, self.H, self.hd
        dev = ids.device
        ids = ids.contiguous()
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        b16 = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)
        one, zero = self._norope(L, dev)
        key_bias = f32(B, L)
        hip.call("oneprot_key_padding_bias", ids, key_bias, T, cfg.pad_token_id)
        x, h, tmp = f32(T, d), b16(T, d), f32(T, d)
        e = "embeddings."
        hip.call("oneprot_bert_embed_fwd", ids, self.view(e + "word_embeddings.weight"), self.view(e + "position_embeddings.weight"),
                 self.view(e + "token_type_embeddings.weight"), self.view(e + "LayerNorm.weight"), self.view(e + "LayerNorm.bias"), x, h, B, L, d,
                 cfg.vocab_size, cfg.layer_norm_eps)
        drop = self._train_dropout()
        if drop:      # hf's train-mode dropouts: embeddings, attention probabilities, the two dense outputs of every layer
            drop_call = self._drop_calls
            self._drop_calls += 1
            p_h, p_a = float(cfg.hidden_dropout_prob), float(cfg.attention_probs_dropout_prob)
            y_drop = f32(T, d)
            hip.call("oneprot_dropout_f32", x, x, T * d, p_h, self._drop_seed, self._drop_stream(drop_call, -1, 0))
            hip.call("oneprot_cast_f32_to_bf16", x, h, T * d)
        saved = dict(ids=ids, key_bias=key_bias, layers=[], B=B, L=L) if save else None
        if drop and save:
            saved["drop_call"] = drop_call
        q, k, v, ctx, u = b16(B, H, L, hd), b16(B, H, L, hd), b16(B, H, L, hd), b16(T, d), b16(T, f)
        eps = cfg.layer_norm_eps
        lora_two = self._lora_two_branch()
        if lora_two:
            lora_call = self._lora_calls
            self._lora_calls += 1
            if save:
                saved["lora_call"] = lora_call
        for i in range(self.n_layers):
            p = f"encoder.layer.{i}."
            if save:     # per layer: input (bf16), attention operands, the two pre-LN sums with their statistics, FFN intermediates
                st = dict(x16=h, q=b16(B, H, L, hd), k=b16(B, H, L, hd), v=b16(B, H, L, hd), ctx=b16(T, d), lse=f32(B, H, L), s1=f32(T, d), mean1=f32(T), rstd1=f32(T),
                          y1=f32(T, d), y16=b16(T, d), z=torch.empty(T, f, dtype=torch.uint8, device=dev), u=b16(T, f), s2=f32(T, d), mean2=f32(T), rstd2=f32(T))
                q, k, v, ctx, u, z, lse = st["q"], st["k"], st["v"], st["ctx"], st["u"], st["z"], st["lse"]
                s1, s2, y1, y16, x_out, h_out = st["s1"], st["s2"], st["y1"], st["y16"], f32(T, d), b16(T, d)
                m1, r1, m2, r2 = st["mean1"], st["rstd1"], st["mean2"], st["rstd2"]
            else:
                z = lse = m1 = r1 = m2 = r2 = None
                s1 = s2 = tmp
                y1, y16, x_out, h_out = x, h, x, h
            o, n = self.span(p + "attention.self.query.weight", p + "attention.self.value.weight")
            ob, nb = self.span(p + "attention.self.query.bias", p + "attention.self.value.bias")
            # rotary tables (1, 0) turn the QKV epilogue into "q *= 1/sqrt(hd), head-major q/k/v" (scores scaled inside attention in HF: same product)
            if lora_two:      # peft's two branches in one launch: [h | dropout(h) A^T] x [W | s B]^T  (esm.py, enable_lora)
                xc, lora_u = self._lora_branch_operand(i, h, T, lora_call)
                kc = self._lora_ops["Kc"]
                hip.call("oneprot_gemm_bf16_nt", xc, self._lora_ops["Wc"][i], T, 3 * d, kc, kc, kc, hip.EPI_QKV_ROPE, self.flat.data[ob:ob + nb], q, k, v, None,
                         one, zero, hd ** -0.5 * hip.LOG2E, L, H, hd)
                if save:
                    st["lora_u"] = lora_u
            else:
                hip.call("oneprot_gemm_bf16_nt", h, self._bf16[o:o + n], T, 3 * d, d, d, d, hip.EPI_QKV_ROPE, self.flat.data[ob:ob + nb], q, k, v, None,
                         one, zero, hd ** -0.5 * hip.LOG2E, L, H, hd)
            if drop:
                hip.call("oneprot_attn_fwd_dropout", q, k, v, key_bias, ctx, lse, B, H, L, hd, p_a, self._drop_seed, self._drop_stream(drop_call, i, 0))
                # s1 = x + dropout(ctx Wo^T + bo): the residual add leaves the GEMM epilogue so that the mask can sit between the two
                hip.call("oneprot_gemm_bf16_nt", ctx, self._w16(p + "attention.output.dense.weight"), T, d, d, d, d, hip.EPI_F32,
                         self.view(p + "attention.output.dense.bias"), y_drop, None, None, None, None, None, 1.0, 0, 0, 0)
                # ... and the LayerNorm that follows reads the sum where it is formed (one kernel; a frozen tower never writes the sum)
                hip.call("oneprot_dropout_add_layernorm_fwd", y_drop, x, s1 if save else None, self.view(p + "attention.output.LayerNorm.weight"),
                         self.view(p + "attention.output.LayerNorm.bias"), y16, y1, m1, r1, T, d, eps, p_h, self._drop_seed, self._drop_stream(drop_call, i, 1))
            else:
                hip.call("oneprot_attn_fwd", q, k, v, key_bias, ctx, lse, B, H, L, hd)
                hip.call("oneprot_gemm_bf16_nt", ctx, self._w16(p + "attention.output.dense.weight"), T, d, d, d, d, hip.EPI_BIAS_RESID,
                         self.view(p + "attention.output.dense.bias"), s1, None, None, x, None, None, 1.0, 0, 0, 0)
                hip.call("oneprot_layernorm_fwd", s1, 0, self.view(p + "attention.output.LayerNorm.weight"), self.view(p + "attention.output.LayerNorm.bias"), y16, y1,
                         m1, r1, T, d, eps)
            hip.call("oneprot_gemm_bf16_nt", y16, self._w16(p + "intermediate.dense.weight"), T, f, d, d, d, hip.EPI_BIAS_GELU,
                     self.view(p + "intermediate.dense.bias"), u, z, None, None, None, None, 1.0, 0, 0, 0)
            if drop:
                hip.call("oneprot_gemm_bf16_nt", u, self._w16(p + "output.dense.weight"), T, d, f, f, f, hip.EPI_F32, self.view(p + "output.dense.bias"),
                         y_drop, None, None, None, None, None, 1.0, 0, 0, 0)
                hip.call("oneprot_dropout_add_layernorm_fwd", y_drop, y1, s2 if save else None, self.view(p + "output.LayerNorm.weight"),
                         self.view(p + "output.LayerNorm.bias"), h_out, x_out, m2, r2, T, d, eps, p_h, self._drop_seed, self._drop_stream(drop_call, i, 2))
            else:
                hip.call("oneprot_gemm_bf16_nt", u, self._w16(p + "output.dense.weight"), T, d, f, f, f, hip.EPI_BIAS_RESID, self.view(p + "output.dense.bias"),
                         s2, None, None, y1, None, None, 1.0, 0, 0, 0)
                hip.call("oneprot_layernorm_fwd", s2, 0, self.view(p + "output.LayerNorm.weight"), self.view(p + "output.LayerNorm.bias"), h_out, x_out, m2, r2, T, d, eps)
            if save:
                saved["layers"].append(st)
            x, h = x_out, h_out
        if save:
            saved["x_final"] = x
        return x, saved

    def backward_layers(self, saved, g, g16, gflat, on_ready=None):
        """g: fp32 [T,d] gradient w.r.t. the last layer's output (consumed), g16 unused (post-LN layers start with a LayerNorm backward);
        gflat: fp32 arena gradient (written).  Per layer, backwards (hf modeling_bert.py:354-417):
          y2 = LN2(s2), s2 = y1 + gelu(y1 W1^T + b1) W2^T + b2 ;  y1 = LN1(s1), s1 = x + attn(x) Wo^T + bo."""
        B, L = saved["B"], saved["L"]
        T, d, f, H, hd = B * L, self.d, self.f, self.H, self.hd
        dev = g.device
        cfg = self.config
        gv = lambda name: self.view(name, gflat)
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        b16 = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)
        ws_ln = torch.empty(hip.query("oneprot_layernorm_bwd_workspace", d), dtype=torch.uint8, device=dev)
        lora_raw = None
        tn_shapes = ((3 * d, d), (f, d), (d, f), (d, d))
        if "lora_call" in saved:                          # the forward ran peft's two-branch form (train mode, lora_dropout > 0)
            lora_raw = self._lora_raw = self._lora_raw_buffers(dev)
            tn_shapes += ((3 * d, self._lora_ops["Rp"]), (self._lora_ops["rp"], d))
        ws_tn = self._tn_workspace(tn_shapes, dev)
        ws_at = torch.empty(hip.query("oneprot_attn_bwd_workspace", B, H, L), dtype=torch.uint8, device=dev)
        ds, ds16, gy = f32(T, d), b16(T, d), f32(T, d)
        dz, dctx, dqkv = b16(T, f), b16(T, d), b16(T, 3 * d)
        drop_call = saved.get("drop_call")                 # the forward ran hf's train-mode dropouts: every mask is regenerated from (seed, call, layer, site)
        if drop_call is not None:
            p_h, p_a = float(cfg.hidden_dropout_prob), float(cfg.attention_probs_dropout_prob)
            dm16 = b16(T, d)                                # gradient of a dense output = the pre-LN sum's gradient through that output's mask
        for i in reversed(range(self.n_layers)):
            st = saved["layers"][i]
            p = f"encoder.layer.{i}."
            # ---- LN2: ds = LN2'(g)  (fp32 + bf16 copy)
            hip.call("oneprot_layernorm_bwd", g, 1, None, 0, st["s2"], 0, self.view(p + "output.LayerNorm.weight"), st["mean2"], st["rstd2"], None, ds, ds16,
                     gv(p + "output.LayerNorm.weight"), gv(p + "output.LayerNorm.bias"), ws_ln, T, d, 0)
            # ---- FFN2 (weight + bias grads in one TN launch), then du * gelu'(z) in the dgrad epilogue
            do16 = ds16
            if drop_call is not None:
                hip.call("oneprot_dropout_bf16", ds16, dm16, T * d, p_h, self._drop_seed, self._drop_stream(drop_call, i, 2))
                do16 = dm16
            self._wgrad(do16, st["u"], T, d, f, d, f, gv(p + "output.dense.weight"), gv(p + "output.dense.bias"), ws_tn)
            hip.call("oneprot_gemm_bf16_nt", do16, self._bf16_T[(i, "w2")], T, f, d, d, d, hip.EPI_GELU_BWD, None, dz, None, None, st["z"], None, None, 1.0, 0, 0, 0)
            # ---- FFN1; gy = ds (residual branch) + dz W1
            self._wgrad(dz, st["y16"], T, f, d, f, d, gv(p + "intermediate.dense.weight"), gv(p + "intermediate.dense.bias"), ws_tn)
            hip.call("oneprot_gemm_bf16_nt", dz, self._bf16_T[(i, "w1")], T, d, f, f, f, hip.EPI_BIAS_RESID, None, gy, None, None, ds, None, None, 1.0, 0, 0, 0)
            # ---- LN1: ds = LN1'(gy)
            hip.call("oneprot_layernorm_bwd", gy, 1, None, 0, st["s1"], 0, self.view(p + "attention.output.LayerNorm.weight"), st["mean1"], st["rstd1"], None, ds, ds16,
                     gv(p + "attention.output.LayerNorm.weight"), gv(p + "attention.output.LayerNorm.bias"), ws_ln, T, d, 0)
            # ---- out-proj
            da16 = ds16
            if drop_call is not None:
                hip.call("oneprot_dropout_bf16", ds16, dm16, T * d, p_h, self._drop_seed, self._drop_stream(drop_call, i, 1))
                da16 = dm16
            self._wgrad(da16, st["ctx"], T, d, d, d, d, gv(p + "attention.output.dense.weight"), gv(p + "attention.output.dense.bias"), ws_tn)
            hip.call("oneprot_gemm_bf16_nt", da16, self._bf16_T[(i, "o")], T, d, d, d, d, hip.EPI_BF16, None, dctx, None, None, None, None, None, 1.0, 0, 0, 0)
            # ---- attention (no rotary: cos/sin = null)
            if drop_call is not None:
                hip.call("oneprot_attn_bwd_dropout", st["q"], st["k"], st["v"], saved["key_bias"], st["ctx"], dctx, st["lse"], None, None, hd ** -0.5, dqkv, ws_at,
                         B, H, L, hd, p_a, self._drop_seed, self._drop_stream(drop_call, i, 0))
            else:
                hip.call("oneprot_attn_bwd", st["q"], st["k"], st["v"], saved["key_bias"], st["ctx"], dctx, st["lse"], None, None, hd ** -0.5, dqkv, ws_at, B, H, L, hd)
            # ---- QKV projection; g = ds (residual branch) + dqkv Wqkv
            o, n = self.span(p + "attention.self.query.weight", p + "attention.self.value.weight")
            ob, nb = self.span(p + "attention.self.query.bias", p + "attention.self.value.bias")
            hip.call("oneprot_gemm_bf16_tn", dqkv, st["x16"], T, 3 * d, d, 3 * d, d, gflat[o:o + n], gflat[ob:ob + nb], ws_tn, ws_tn.numel(), 0)
            hip.call("oneprot_gemm_bf16_nt", dqkv, self._bf16_T[(i, "qkv")], T, d, 3 * d, 3 * d, 3 * d, hip.EPI_BIAS_RESID, None, g, None, None, ds, None, None,
                     1.0, 0, 0, 0)
            if lora_raw is not None:      # two-branch LoRA: adapter gradients, and g += mask * (du A) / keep
                self._lora_branch_backward(i, st["x16"], st["lora_u"], dqkv, T, saved["lora_call"], ws_tn, lora_raw, dh32=g)
            saved["layers"][i] = None
        if drop_call is not None:                          # x0 = dropout(LN(embeddings))
            hip.call("oneprot_dropout_f32", g, g, T * d, p_h, self._drop_seed, self._drop_stream(drop_call, -1, 0))
        self._embedding_backward(saved["ids"], g, gflat)
        if on_ready is not None:
            on_ready(0, self._total)

    def _embedding_backward(self, ids, g, gflat):
        """x0 = LN(word[id] + pos[l] + type[0]) (hf modeling_bert.py:53-108).  The pre-LN sum is re-gathered (torch indexing: data movement),
        LayerNorm statistics / backward and the table reductions are HIP kernels; equal token ids are summed in sorted order."""
        cfg = self.config
        B, L = ids.shape
        T, d, dev = B * L, self.d, ids.device
        e = "embeddings."
        gv = lambda name: self.view(name, gflat)
        esum = (self.view(e + "word_embeddings.weight")[ids.reshape(-1)] + self.view(e + "position_embeddings.weight")[:L].repeat(B, 1)
                + self.view(e + "token_type_embeddings.weight")[0]).contiguous()
        mean, rstd, y = torch.empty(T, device=dev), torch.empty(T, device=dev), torch.empty(T, d, device=dev)
        hip.call("oneprot_layernorm_fwd", esum, 0, self.view(e + "LayerNorm.weight"), self.view(e + "LayerNorm.bias"), None, y, mean, rstd, T, d, cfg.layer_norm_eps)
        de = torch.empty(T, d, device=dev)
        ws_ln = torch.empty(hip.query("oneprot_layernorm_bwd_workspace", d), dtype=torch.uint8, device=dev)
        hip.call("oneprot_layernorm_bwd", g, 1, None, 0, esum, 0, self.view(e + "LayerNorm.weight"), mean, rstd, None, de, None,
                 gv(e + "LayerNorm.weight"), gv(e + "LayerNorm.bias"), ws_ln, T, d, 0)
        # position rows 0..L-1: sum over the batch; token-type row 0: sum over positions of that
        dpos = gv(e + "position_embeddings.weight")
        hip.call("oneprot_rowsum_f32", de, dpos[:L], B, L * d)
        hip.call("oneprot_rowsum_f32", dpos[:L], gv(e + "token_type_embeddings.weight")[0], L, d)
        # word rows: stable sort of the ids, one block per run of equal ids (padding_idx row gets no gradient, as nn.Embedding)
        sorted_ids, perm = torch.sort(ids.reshape(-1), stable=True)
        rows, counts = torch.unique_consecutive(sorted_ids, return_counts=True)
        starts = (torch.cumsum(counts, 0) - counts).contiguous()
        pad = cfg.pad_token_id if cfg.pad_token_id is not None else -1
        hip.call("oneprot_embed_scatter_sorted", de, perm.contiguous(), starts, rows.contiguous(), T, int(rows.numel()), d, pad, gv(e + "word_embeddings.weight"))

    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, **_):
        x, _ = self.run_layers(input_ids, save=False)
        B, L = input_ids.shape
        return _Out(x.view(B, L, self.d))

    @classmethod
    def from_pretrained(cls, model_name_or_path, **_):
        cfg, path = resolve_bert_config(model_name_or_path)
        model = cls(cfg)
        sd = load_weight_file(path)
        if sd is None:
            if os.environ.get("ONEPROT_ALLOW_RANDOM_INIT", "0") != "1":
                raise OSError(f"no weights (model.safetensors / pytorch_model.bin) found for {model_name_or_path}; "
                              "set ONEPROT_ALLOW_RANDOM_INIT=1 to build a randomly initialised model of that architecture")
            warnings.warn(f"{model_name_or_path}: no weight file, using random initialisation")
        else:
            sd = {(k[5:] if k.startswith("bert.") else k): v for k, v in sd.items() if not k.startswith("cls.")}
            missing, _ = model.load_state_dict(sd, strict=False)
            missing = [m for m in missing if not (m.startswith("pooler.") or m.startswith("extra."))]
            if missing:
                raise OSError(f"checkpoint {model_name_or_path} lacks tensors: {missing[:5]}...")
        return model
