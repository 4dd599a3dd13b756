"""BERT text tower (forward only) on the HIP kernels -- what the reference obtains from `AutoModel.from_pretrained` in
TextEncoder (ref text_encoder.py:33) i.e. transformers' BertModel in eval mode (hf modeling_bert.py:53-108 embeddings,
139-204 attention, 354-417 layer; post-LN, 1/sqrt(hd) inside attention, erf-GELU).  The text tower is frozen in every shipped
OneProt config (configs/model/components/text.yaml:12), so only the forward exists; state-dict keys are BertModel's."""
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

    @torch.no_grad()
    def run_layers(self, ids, save=False):
        if save:
            raise NotImplementedError("BERT backward is not built (the text tower is frozen in every shipped config)")
        if not ids.is_cuda:
            raise hip.HipKernelError("OneProt HIP path needs CUDA(ROCm) tensors; there is no CPU fallback")
        self._refresh_bf16_mirror()
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
        q, k, v, ctx, u = b16(B, H, L, hd), b16(B, H, L, hd), b16(B, H, L, hd), b16(T, d), b16(T, f)
        for i in range(self.n_layers):
            p = f"encoder.layer.{i}."
            o, n = self.span(p + "attention.self.query.weight", p + "attention.self.value.weight")
            ob, nb = self.span(p + "attention.self.query.bias", p + "attention.self.value.bias")
            # rotary tables (1, 0) turn the QKV epilogue into "q *= 1/sqrt(hd), head-major q/k/v" (scores scaled inside attention in HF: same product)
            hip.call("oneprot_gemm_bf16_nt", h, self._bf16[o:o + n], T, 3 * d, d, d, d, hip.EPI_QKV_ROPE, self.flat.data[ob:ob + nb], q, k, v, None,
                     one, zero, hd ** -0.5, L, H, hd)
            hip.call("oneprot_attn_fwd", q, k, v, key_bias, ctx, None, B, H, L, hd)
            hip.call("oneprot_gemm_bf16_nt", ctx, self._w16(p + "attention.output.dense.weight"), T, d, d, d, d, hip.EPI_BIAS_RESID,
                     self.view(p + "attention.output.dense.bias"), tmp, None, None, x, None, None, 1.0, 0, 0, 0)
            hip.call("oneprot_layernorm_fwd", tmp, 0, self.view(p + "attention.output.LayerNorm.weight"), self.view(p + "attention.output.LayerNorm.bias"), h, x,
                     None, None, T, d, cfg.layer_norm_eps)
            hip.call("oneprot_gemm_bf16_nt", h, self._w16(p + "intermediate.dense.weight"), T, f, d, d, d, hip.EPI_BIAS_GELU,
                     self.view(p + "intermediate.dense.bias"), u, None, None, None, None, None, 1.0, 0, 0, 0)
            hip.call("oneprot_gemm_bf16_nt", u, self._w16(p + "output.dense.weight"), T, d, f, f, f, hip.EPI_BIAS_RESID, self.view(p + "output.dense.bias"),
                     tmp, None, None, x, None, None, 1.0, 0, 0, 0)
            hip.call("oneprot_layernorm_fwd", tmp, 0, self.view(p + "output.LayerNorm.weight"), self.view(p + "output.LayerNorm.bias"), h, x, None, None, T, d,
                     cfg.layer_norm_eps)
        return x, None

    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, **_):
        x, _ = self.run_layers(input_ids)
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
