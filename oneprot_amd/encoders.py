"""Encoder plugins of the OneProt hot path on the HIP kernels.

Same constructor signatures, attribute names (`transformer`, `pooling`, `proj`, `norm`, `config`) and state-dict keys
as the reference classes:
    BaseEncoder / Normalize / LearnableLogitScaling / MeanPooling / CLSTokenPooling   ref base_encoder.py:6-194
    SequenceEncoder                                                                   ref sequence_encoder.py:22-81
    StructTokenEncoder                                                                ref struct_token_encoder.py:6-34
    TextEncoder                                                                       ref text_encoder.py:8-62
`forward(input_ids[B,L] int64) -> float32 [B, output_dim]`.

The whole encoder (embedding -> n layers -> final LN + pooling -> projection head -> L2-norm [-> logit scale]) is ONE
autograd node whose forward and backward are sequences of C-ABI kernel launches; torch only owns the memory.
There is no CPU path: calling an encoder with CPU tensors raises.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import hip
from .esm import EsmTransformer, resolve_config, ModelConfig


# ------------------------------------------------------------------------------------------------- small modules
class Normalize(nn.Module):
    def __init__(self, dim: int) -> None:
        super().__init__()
        self.dim = dim

    def forward(self, x):
        """F.normalize(x, dim=self.dim, p=2) (ref base_encoder.py:6-12) on the row-normalise kernel"""
        dim = self.dim if self.dim >= 0 else x.dim() + self.dim
        if dim != x.dim() - 1:
            return _L2NormFn.apply(x.transpose(dim, -1), 1.0).transpose(dim, -1)
        return _L2NormFn.apply(x, 1.0)


class LearnableLogitScaling(nn.Module):
    """clip(exp(log_logit_scale), max) * x  (ref base_encoder.py:15-38).  Key: norm.1.log_logit_scale."""

    def __init__(self, logit_scale_init: float = 1 / 0.07, learnable: bool = True, max_logit_scale: float = 100) -> None:
        super().__init__()
        self.max_logit_scale = max_logit_scale
        self.logit_scale_init = logit_scale_init
        self.learnable = learnable
        log_logit_scale = torch.ones([]) * np.log(self.logit_scale_init)
        if learnable:
            self.log_logit_scale = nn.Parameter(log_logit_scale)
        else:
            self.register_buffer("log_logit_scale", log_logit_scale)

    def scale_value(self) -> float:
        """clip(exp(log_logit_scale), max) as a host float (fixed scale: read once, cached on the tensor's version).  The learnable scale
        never comes through here on the hot path -- it changes every step, so the fused encoder keeps it on the device (`scale_device`)."""
        key = (self.log_logit_scale._version, self.log_logit_scale.data_ptr())
        if getattr(self, "_cached", (None, None))[0] != key:
            self._cached = (key, min(math.exp(float(self.log_logit_scale.detach())), self.max_logit_scale))
        return self._cached[1]

    def scale_device(self):
        """(clip(exp(l), max), d/dl of it) as 1-element device tensors: no host synchronisation"""
        e = self.log_logit_scale.detach().float().exp().reshape(1)
        return torch.clamp(e, max=self.max_logit_scale), e * (e < self.max_logit_scale).float()

    def forward(self, x):
        return torch.clip(self.log_logit_scale.exp(), max=self.max_logit_scale) * x

    def extra_repr(self):
        return f"logit_scale_init={self.logit_scale_init},learnable={self.learnable}, max_logit_scale={self.max_logit_scale}"


def _mask_as_ids(features, input_mask):
    """the pooling kernels take token ids + pad id; a 0/1 mask is the same thing with pad id 0"""
    B, L = features.shape[0], features.shape[1]
    if input_mask is None:
        return torch.ones(B, L, dtype=torch.int64, device=features.device)
    return (input_mask != 0).to(torch.int64).contiguous()


class _PoolFn(torch.autograd.Function):
    """stand-alone pooling of a [B,L,d] fp32 tensor (inside the encoders pooling is fused into the final LayerNorm kernel; this is what a
    caller reaching into `encoder.pooling` gets): mode 0 masked mean, 1 CLS"""

    @staticmethod
    def forward(ctx, features, mask_ids, mode):
        if not features.is_cuda:
            raise hip.HipKernelError("OneProt HIP path needs CUDA(ROCm) tensors; there is no CPU fallback")
        x = features.contiguous().float()
        B, L, d = x.shape
        pooled = torch.empty(B, d, device=x.device)
        hip.call("oneprot_pool_fwd", x, mask_ids, 0, pooled, B, L, d, mode)
        ctx.save_for_backward(mask_ids)
        ctx.shape, ctx.mode = (B, L, d), mode
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        (mask_ids,) = ctx.saved_tensors
        B, L, d = ctx.shape
        g = torch.empty(B, L, d, device=dpooled.device)
        hip.call("oneprot_pool_bwd", dpooled.contiguous().float(), mask_ids, 0, g, None, B, L, d, ctx.mode)
        return g, None, None


class MeanPooling(nn.Module):
    """ref base_encoder.py:105-118: sum(x * m) / sum(m) over all non-pad positions (CLS / EOS included)"""
    mode = 0

    def forward(self, features, input_mask=None):
        return _PoolFn.apply(features, _mask_as_ids(features, input_mask), 0)


class CLSTokenPooling(nn.Module):
    mode = 1

    def forward(self, features, input_mask=None):
        return features[:, 0]


class MaskedConv1d(nn.Conv1d):
    """kernel-1 convolution holder of Attention1dPooling (ref base_encoder.py:40-86); only its parameters are used (fused kernel)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, dilation: int = 1, groups: int = 1, bias: bool = True):
        padding = dilation * (kernel_size - 1) // 2
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, dilation=dilation, groups=groups, bias=bias, padding=padding)


class Attention1dPooling(nn.Module):
    """ref base_encoder.py:88-103.  NB the reference builds it with hidden_size hard-coded to 1280 (base_encoder.py:180), i.e. it only
    matches ESM-2-650M-width encoders; same here."""
    mode = 2

    def __init__(self, hidden_size):
        super().__init__()
        self.layer = MaskedConv1d(hidden_size, 1, 1)

    def forward(self, x, input_mask=None):
        return _AttnPoolFn.apply(x, _mask_as_ids(x, input_mask), self.layer.weight, self.layer.bias)


class _AttnPoolFn(torch.autograd.Function):
    """stand-alone Attention1dPooling.forward (ref base_encoder.py:95-103) on the attnpool kernels"""

    @staticmethod
    def forward(ctx, x, mask_ids, w, b):
        if not x.is_cuda:
            raise hip.HipKernelError("OneProt HIP path needs CUDA(ROCm) tensors; there is no CPU fallback")
        x = x.contiguous().float()
        B, L, d = x.shape
        if w.numel() != d:
            raise RuntimeError(f"Attention1dPooling was built for hidden size {w.numel()} but the input width is {d}")
        pooled, attn = torch.empty(B, d, device=x.device), torch.empty(B, L, device=x.device)
        hip.call("oneprot_attnpool_fwd", x, mask_ids, 0, w, b, pooled, attn, B, L, d)
        ctx.save_for_backward(x, attn, w)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        x, attn, w = ctx.saved_tensors
        B, L, d = x.shape
        dev = x.device
        dw, db, dx = torch.empty(d, device=dev), torch.empty(1, device=dev), torch.empty(B, L, d, device=dev)
        hip.call("oneprot_attnpool_bwd", x, attn, w, dpooled.contiguous().float(), dw, db, dx, _ws(hip.query("oneprot_attnpool_bwd_workspace", B, d), dev), B, L, d)
        return dx, None, dw.view_as(w), db


class _IdentityPooling(nn.Identity):
    """nn.Identity that tolerates the (x, input_mask) call of BaseEncoder.forward"""
    mode = 3

    def forward(self, x, input_mask=None):
        return x


def _ws(nbytes, dev):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)


# ------------------------------------------------------------------------------------------------- projection head
class _Head:
    """fp32 projection head on [B, d] rows: LN -> Linear [-> GELU -> LN -> Linear] -> L2 normalise -> * scale
    (ref base_encoder.py:147-178).  Forward keeps what backward needs."""

    @staticmethod
    def forward(pooled, proj, scale, save, scale_t=None):
        """scale: host float (fixed logit scale, or 1.0); scale_t: 1-element device tensor (learnable logit scale: applied by a second tiny
        launch so that the value never visits the host)"""
        B, d = pooled.shape
        dev = pooled.device
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        st = {}
        x = pooled
        mods = list(proj)
        if len(mods) == 1:          # Identity
            y = x
        else:
            ln0, lin1 = mods[0], mods[1]
            a0, m0, r0 = f32(B, d), f32(B), f32(B)
            hip.call("oneprot_layernorm_fwd", x, 0, ln0.weight, ln0.bias, None, a0, m0, r0, B, d, ln0.eps)
            n1 = lin1.out_features
            y1 = f32(B, n1)
            hip.call("oneprot_sgemm", a0, lin1.weight, y1, B, n1, d, 0, 0, 1.0, 0)
            st.update(a0=a0, m0=m0, r0=r0)
            if len(mods) == 2:
                y = y1
            else:
                ln3, lin4 = mods[3], mods[4]
                g1 = f32(B, n1)
                hip.call("oneprot_gelu_f32", y1, g1, B * n1)
                a3, m3, r3 = f32(B, n1), f32(B), f32(B)
                hip.call("oneprot_layernorm_fwd", g1, 0, ln3.weight, ln3.bias, None, a3, m3, r3, B, n1, ln3.eps)
                n4 = lin4.out_features
                y = f32(B, n4)
                hip.call("oneprot_sgemm", a3, lin4.weight, y, B, n4, n1, 0, 0, 1.0, 0)
                st.update(y1=y1, g1=g1, a3=a3, m3=m3, r3=r3)
        D = y.shape[1]
        feat, inv = f32(B, D), f32(B)
        hip.call("oneprot_l2norm_fwd", y, feat, inv, B, D, scale)
        st.update(inv=inv, feat=feat, pooled=pooled)
        if scale_t is not None:
            out = feat.clone()
            hip.call("oneprot_scale_by_device_scalar", out, out.numel(), scale_t)
            st.update(feat_scaled=out)
            feat = out
        return feat, (st if save else None)

    @staticmethod
    def backward(dfeat, proj, scale, st, scale_t=None):
        """returns (dpooled, [grads of proj parameters in proj.parameters() order]); with a device-side scale `dfeat` is first multiplied by it"""
        if scale_t is not None:
            dfeat = dfeat.clone()
            hip.call("oneprot_scale_by_device_scalar", dfeat, dfeat.numel(), scale_t)
        dev = dfeat.device
        B, D = dfeat.shape
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        dy = f32(B, D)
        hip.call("oneprot_l2norm_bwd", st["feat"], dfeat.contiguous(), st["inv"], dy, B, D, scale, 0.0)
        mods = list(proj)
        if len(mods) == 1:
            return dy, []
        pooled = st["pooled"]
        d = pooled.shape[1]
        grads = []
        ln0, lin1 = mods[0], mods[1]
        n1 = lin1.out_features
        if len(mods) > 2:
            ln3, lin4 = mods[3], mods[4]
            dW4 = f32(D, n1)
            hip.call("oneprot_sgemm", dy, st["a3"], dW4, D, n1, B, 1, 1, 1.0, 0)          # dW = dy^T a3
            da3 = f32(B, n1)
            hip.call("oneprot_sgemm", dy, lin4.weight, da3, B, n1, D, 0, 1, 1.0, 0)       # da = dy W
            dg1, dgam3, dbet3 = f32(B, n1), f32(n1), f32(n1)
            hip.call("oneprot_layernorm_bwd", da3, 1, None, 0, st["g1"], 0, ln3.weight, st["m3"], st["r3"], None, dg1, None, dgam3, dbet3,
                     _ws(hip.query("oneprot_layernorm_bwd_workspace", n1), dev), B, n1, 0)
            dy1 = f32(B, n1)
            hip.call("oneprot_gelu_bwd_f32", st["y1"], dg1, dy1, B * n1)
            tail = [dgam3, dbet3, dW4]
        else:
            dy1 = dy
            tail = []
        dW1 = f32(n1, d)
        hip.call("oneprot_sgemm", dy1, st["a0"], dW1, n1, d, B, 1, 1, 1.0, 0)
        da0 = f32(B, d)
        hip.call("oneprot_sgemm", dy1, lin1.weight, da0, B, d, n1, 0, 1, 1.0, 0)
        dpooled, dgam0, dbet0 = f32(B, d), f32(d), f32(d)
        hip.call("oneprot_layernorm_bwd", da0, 1, None, 0, pooled, 0, ln0.weight, st["m0"], st["r0"], None, dpooled, None, dgam0, dbet0,
                 _ws(hip.query("oneprot_layernorm_bwd_workspace", d), dev), B, d, 0)
        grads = [dgam0, dbet0, dW1] + tail
        return dpooled, grads


class _L2NormFn(torch.autograd.Function):
    """stand-alone Normalize (only used when someone calls encoder.norm directly)"""

    @staticmethod
    def forward(ctx, x, scale):
        if not x.is_cuda:
            raise hip.HipKernelError("OneProt HIP path needs CUDA(ROCm) tensors; there is no CPU fallback")
        shape = x.shape
        x = x.contiguous().float().view(-1, shape[-1])
        B, D = x.shape
        y, inv = torch.empty_like(x), torch.empty(B, device=x.device)
        hip.call("oneprot_l2norm_fwd", x, y, inv, B, D, scale)
        ctx.save_for_backward(y, inv)
        ctx.scale = scale
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        dx = torch.empty_like(y)
        hip.call("oneprot_l2norm_bwd", y, dy.contiguous().float().view(y.shape), inv, dx, y.shape[0], y.shape[1], ctx.scale, 0.0)
        return dx.view(dy.shape), None


# ------------------------------------------------------------------------------------------------- whole-encoder node
class _EncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, ids, flat, n_extra, *params):
        tr = enc.transformer
        # (grad mode is off inside Function.forward; ctx.needs_input_grad already folds in torch.no_grad())
        need_tr_grad = bool(ctx.needs_input_grad[2])
        need_any = need_tr_grad or any(ctx.needs_input_grad[4:])
        if need_tr_grad:
            tr._live_apps = getattr(tr, "_live_apps", 0) + 1       # applications awaiting their backward (see backward: overlap guard)
        x, saved = tr.run_layers(ids, save=need_tr_grad)
        B, L = ids.shape
        d = tr.d
        dev = ids.device
        ids = ids.contiguous()
        pooled = torch.empty(B, d, device=dev)
        mode = enc.pooling.mode
        final_ln = getattr(tr, "final_layer_norm", True)
        mean = rstd = wrow = hidden = attn = None
        if need_tr_grad:
            mean, rstd, wrow = (torch.empty(B * L, device=dev) for _ in range(3))
        if mode == 2:                                   # attention1d: needs the normalised hidden state itself
            pw = enc.pooling.layer.weight
            if pw.numel() != d:
                raise RuntimeError(f"Attention1dPooling was built for hidden size {pw.numel()} but the encoder width is {d} "
                                   "(the reference hard-codes 1280: base_encoder.py:180)")
            if final_ln:
                hidden = torch.empty(B, L, d, device=dev)
                hip.call("oneprot_lnpool_fwd", x, ids, tr.config.pad_token_id, tr.view("encoder.emb_layer_norm_after.weight"),
                         tr.view("encoder.emb_layer_norm_after.bias"), pooled, mean, rstd, wrow, None, hidden, B, L, d, tr.config.layer_norm_eps, 0)
            else:
                hidden = x.view(B, L, d)
            attn = torch.empty(B, L, device=dev)
            hip.call("oneprot_attnpool_fwd", hidden, ids, tr.config.pad_token_id, pw, enc.pooling.layer.bias, pooled, attn, B, L, d)
        elif final_ln:
            hip.call("oneprot_lnpool_fwd", x, ids, tr.config.pad_token_id, tr.view("encoder.emb_layer_norm_after.weight"),
                     tr.view("encoder.emb_layer_norm_after.bias"), pooled, mean, rstd, wrow, None, None, B, L, d, tr.config.layer_norm_eps, mode)
        else:       # BERT: the last layer's output is already post-LN
            hip.call("oneprot_pool_fwd", x, ids, tr.config.pad_token_id, pooled, B, L, d, mode)
        learn = len(enc.norm) > 1 and enc.norm[1].learnable
        scale_t, dscale_t = enc.norm[1].scale_device() if learn else (None, None)
        scale = 1.0 if learn else enc.logit_scale_value()
        feat, hst = _Head.forward(pooled, enc.proj, scale, need_any, scale_t)
        ctx.enc, ctx.saved, ctx.hst, ctx.scale = enc, saved, hst, scale
        ctx.scale_t, ctx.dscale_t = scale_t, dscale_t
        ctx.fin = (mean, rstd, wrow)
        ctx.pool = (hidden, attn) if (mode == 2 and need_any) else None
        ctx.need_tr_grad = need_tr_grad
        ctx.n_extra, ctx.n_params = n_extra, len(params)
        ctx.ids = ids
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        enc, tr = ctx.enc, ctx.enc.transformer
        dev = dfeat.device
        dfeat = dfeat.contiguous()
        dpooled, hgrads = _Head.backward(dfeat, enc.proj, ctx.scale, ctx.hst, ctx.scale_t)
        extra_grads = []
        dhidden = None
        mode = enc.pooling.mode
        if mode == 2:
            hidden, attn = ctx.pool
            B, L, d = hidden.shape
            dw, db = torch.empty(d, device=dev), torch.empty(1, device=dev)
            if ctx.need_tr_grad:
                dhidden = torch.empty(B * L, d, device=dev)
            hip.call("oneprot_attnpool_bwd", hidden, attn, enc.pooling.layer.weight, dpooled, dw, db, dhidden,
                     _ws(hip.query("oneprot_attnpool_bwd_workspace", B, d), dev), B, L, d)
            extra_grads += [dw.view_as(enc.pooling.layer.weight), db]
        if len(enc.norm) > 1 and enc.norm[1].learnable:
            # d/d(log s) [clip(e^l) * xhat] = (dfeat . xhat) * d clip(e^l)/dl, the last factor e^l or 0 (clip active): all on the device
            g = torch.zeros(1, device=dev)
            hip.call("oneprot_sgemm", dfeat.view(1, -1), ctx.hst["feat"].view(1, -1), g, 1, 1, dfeat.numel(), 0, 0, 1.0, 0)
            hip.call("oneprot_scale_by_device_scalar", g, 1, ctx.dscale_t)
            extra_grads.append(g.reshape(()))
        gflat = None
        if ctx.need_tr_grad:
            saved = ctx.saved
            B, L, d = saved["B"], saved["L"], tr.d
            # the persistent buffer only for the encoder's single application of a step: with several (seqsim) the first gradient waits inside the
            # autograd engine, invisible as .grad, until the last one has been produced -- each application then gets a tensor of its own
            lone = getattr(tr, "_live_apps", 1) == 1 and not getattr(tr, "_multi_app_step", False)
            gflat = _arena_grad_buffer(tr, dev) if lone else torch.zeros(tr._total, device=dev)
            mean, rstd, wrow = ctx.fin
            g = torch.empty(B * L, d, device=dev)
            g16 = torch.empty(B * L, d, dtype=torch.bfloat16, device=dev)
            final_ln = getattr(tr, "final_layer_norm", True)
            if final_ln:
                lnw, lnb = tr.view("encoder.emb_layer_norm_after.weight", gflat), tr.view("encoder.emb_layer_norm_after.bias", gflat)
                ws = _ws(hip.query("oneprot_layernorm_bwd_workspace", d), dev)
            if not final_ln:        # BERT: pooling reads the last layer's output directly
                if mode == 2:
                    g.copy_(dhidden.view(B * L, d))
                else:
                    hip.call("oneprot_pool_bwd", dpooled, ctx.ids, tr.config.pad_token_id, g, g16, B, L, d, mode)
            elif mode == 2:
                hip.call("oneprot_layernorm_bwd", dhidden, 1, None, 0, saved["x_final"], 0, tr.view("encoder.emb_layer_norm_after.weight"), mean, rstd, None, g, g16,
                         lnw, lnb, ws, B * L, d, 0)
            else:
                hip.call("oneprot_layernorm_bwd", dpooled, 2, wrow, L, saved["x_final"], 0, tr.view("encoder.emb_layer_norm_after.weight"), mean, rstd, None, g, g16,
                         lnw, lnb, ws, B * L, d, 0)
            saved["x_final"] = None
            # Overlapped data-parallel reduction (oneprot_amd.distributed.GradOverlap): only when this is the encoder's single application in
            # the step and nothing has been accumulated yet -- then the arena gradient is installed as .grad here (autograd gets None for it)
            # and finished ranges are all-reduced while the lower layers are still in their backward.
            sync = getattr(tr, "_grad_overlap", None)
            single = getattr(tr, "_live_apps", 1) == 1 and not getattr(tr, "_multi_app_step", False)
            if getattr(tr, "_live_apps", 1) > 1:
                tr._multi_app_step = True
            tr._live_apps = max(getattr(tr, "_live_apps", 1) - 1, 0)
            if tr._live_apps == 0 and not single:
                tr._multi_app_step = False
            if sync is not None and single and tr.flat.grad is None and not tr._lora:
                tr.backward_layers(saved, g, g16, gflat, on_ready=lambda lo, hi: sync.reduce_range(tr.flat, gflat, lo, hi))
                tr.flat.grad = gflat
                gflat = None
            else:
                tr.backward_layers(saved, g, g16, gflat)
                if tr._lora:
                    dA, dB, gflat = tr.lora_backward(gflat)
                    extra_grads += [dA, dB]
            ctx.saved = None
        n_head = ctx.n_params - ctx.n_extra
        hg = list(hgrads) + [None] * (n_head - len(hgrads))
        eg = list(extra_grads) + [None] * (ctx.n_extra - len(extra_grads))
        return (None, None, gflat, None) + tuple(eg) + tuple(hg)


def _storage_users(t):
    """number of holders of t's storage (tensors, views, slices, .grad slots ...), or None when this torch build does not expose the count"""
    f = getattr(torch._C, "_storage_Use_Count", None)
    return f(t.untyped_storage()._cdata) if f is not None else None


def _arena_grad_buffer(tr, dev):
    """The fp32 arena-gradient tensor of a backward pass.  ONE persistent buffer per encoder (592 MB at ESM-2-150M), zero-filled per use: the
    gradient ranges handed to RCCL (distributed.GradOverlap) then sit at the same addresses in every step instead of wherever the caching allocator
    put a fresh tensor.  Autograd receives a fresh VIEW of it (a view object nobody else holds is adopted as .grad without a copy).

    ALIASING CONTRACT: the gradient of step n lives in this buffer only until the backward of step n+1 zero-fills it -- and it is zero-filled only
    when NOBODY else holds the buffer's storage: not `flat.grad` (a second application of the encoder in the same step, e.g. seqsim; accumulation
    without zero_grad), not a tensor a caller kept (`g = p.grad` across `zero_grad(set_to_none=True)`), not a slice or view derived from it (a hook's
    per-parameter views, a logger's list).  The storage's holder count says so; while there is a holder, a separate tensor is returned (and, where
    autograd accumulates, added)."""
    buf = getattr(tr, "_gflat_buf", None)
    if buf is not None and (buf.device != dev or buf.numel() != tr._total):
        buf = None
    if buf is not None:
        g = tr.flat.grad
        users = _storage_users(buf)
        held = users > tr._gflat_base_users if users is not None else (g is not None and g.data_ptr() == buf.data_ptr())
        if held:
            return torch.zeros(tr._total, device=dev)
        buf.zero_()
    else:
        buf = tr._gflat_buf = torch.zeros(tr._total, device=dev)
        tr._gflat_base_users = _storage_users(buf)          # the buffer itself (+ the temporary storage handle of the query)
    return buf.view(-1)


class _ProjNormFn(torch.autograd.Function):
    """proj -> L2-normalise [-> logit scale] of already pooled rows [B, d] on the HIP kernels, differentiable w.r.t. the rows, the head parameters
    and a learnable logit scale: what BaseEncoder.forward (ref base_encoder.py:190-194) and StructEncoder.forward (ref struct_graph_encoder.py:36-42)
    apply after pooling / after an opaque encoder."""

    @staticmethod
    def forward(ctx, enc, pooled, n_extra, *params):
        if not pooled.is_cuda:
            raise hip.HipKernelError("OneProt HIP path needs CUDA(ROCm) tensors; there is no CPU fallback")
        pooled = pooled.contiguous().float()
        learn = len(enc.norm) > 1 and enc.norm[1].learnable
        scale_t, dscale_t = enc.norm[1].scale_device() if learn else (None, None)
        scale = 1.0 if learn else enc.logit_scale_value()
        need = any(ctx.needs_input_grad)
        feat, hst = _Head.forward(pooled, enc.proj, scale, need, scale_t)
        ctx.enc, ctx.hst, ctx.scale, ctx.scale_t, ctx.dscale_t = enc, hst, scale, scale_t, dscale_t
        ctx.n_extra, ctx.n_params = n_extra, len(params)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        enc = ctx.enc
        dfeat = dfeat.contiguous()
        dpooled, hgrads = _Head.backward(dfeat, enc.proj, ctx.scale, ctx.hst, ctx.scale_t)
        extra = []
        if ctx.n_extra:           # learnable logit scale (the only extra parameter of a head-only application)
            g = torch.zeros(1, device=dfeat.device)
            hip.call("oneprot_sgemm", dfeat.view(1, -1), ctx.hst["feat"].view(1, -1), g, 1, 1, dfeat.numel(), 0, 0, 1.0, 0)
            hip.call("oneprot_scale_by_device_scalar", g, 1, ctx.dscale_t)
            extra.append(g.reshape(()))
        n_head = ctx.n_params - ctx.n_extra
        hg = list(hgrads) + [None] * (n_head - len(hgrads))
        return (None, dpooled, None) + tuple(extra) + tuple(hg)


# ------------------------------------------------------------------------------------------------- BaseEncoder
class BaseEncoder(nn.Module):
    """ref base_encoder.py:129-194 (same members; forward(x, input_mask) applies pool -> proj -> norm to features)."""

    def __init__(self, d_model: int, output_dim: int, proj_type: str = None, use_logit_scale: bool = False, learnable_logit_scale: bool = False,
                 pooling_type: str = 'mean'):
        super().__init__()
        self.d_model = d_model
        self.output_dim = output_dim
        self.pooling_type = pooling_type
        self.proj = self._create_projection(proj_type)
        self.norm = self._create_normalization(use_logit_scale, learnable_logit_scale)
        self.pooling = self._create_pooling(pooling_type)

    def _create_projection(self, proj_type):
        if proj_type == 'linear':
            return nn.Sequential(nn.LayerNorm(self.d_model), nn.Linear(self.d_model, self.output_dim, bias=False))
        if proj_type == 'mlp':
            hidden_size = (self.d_model + self.output_dim) // 2
            return nn.Sequential(nn.LayerNorm(self.d_model), nn.Linear(self.d_model, hidden_size, bias=False), nn.GELU(), nn.LayerNorm(hidden_size),
                                 nn.Linear(hidden_size, self.output_dim, bias=False))
        return nn.Sequential(nn.Identity())

    def _create_normalization(self, use_logit_scale, learnable_logit_scale=False):
        layers = [Normalize(dim=-1)]
        if use_logit_scale:
            layers.append(LearnableLogitScaling(learnable=bool(learnable_logit_scale)))
        return nn.Sequential(*layers)

    def _create_pooling(self, pooling_type, hidden_size=1280):
        if pooling_type == 'mean':
            return MeanPooling()
        if pooling_type == 'cls':
            return CLSTokenPooling()
        if pooling_type == 'attention1d':
            return Attention1dPooling(hidden_size)
        return _IdentityPooling()               # ref base_encoder.py:187-188: anything else is nn.Identity (StructEncoder passes None)

    def apply_head(self, pooled):
        """proj -> norm of pooled rows [B, d] on the HIP kernels (autograd-aware)"""
        extra = [self.norm[1].log_logit_scale] if (len(self.norm) > 1 and self.norm[1].learnable) else []
        return _ProjNormFn.apply(self, pooled, len(extra), *extra, *list(self.proj.parameters()))

    def forward(self, x, input_mask=None):
        """ref base_encoder.py:190-194: pooling -> proj -> norm of a hidden state x [B, L, d] (the token encoders override this with the fused
        ids -> features path)."""
        return self.apply_head(self.pooling(x, input_mask))

    def logit_scale_value(self) -> float:
        return self.norm[1].scale_value() if len(self.norm) > 1 else 1.0

    def _extra_params(self):
        """parameters outside the transformer arena that the fused node differentiates: pooling conv (attention1d), learnable logit scale"""
        ps = list(self.pooling.parameters())
        if len(self.norm) > 1 and self.norm[1].learnable:
            ps.append(self.norm[1].log_logit_scale)
        if getattr(self.transformer, "_lora", None):
            ps += [self.transformer.lora_A, self.transformer.lora_B]
        return ps

    def encode(self, input_ids):
        extra = self._extra_params()
        head_params = [p for p in self.proj.parameters()]
        return _EncodeFn.apply(self, input_ids, self.transformer.flat, len(extra), *extra, *head_params)


class SequenceEncoder(BaseEncoder):
    def __init__(self, model_name_or_path: str, output_dim: int, pooling_type: str = "mean", proj_type: str = None, use_logit_scale: bool = False,
                 learnable_logit_scale: bool = False, pretrained: bool = True, use_lora: bool = True, lora_r: int = 8, lora_alpha: int = 16,
                 lora_dropout: float = 0.1, lora_target_modules: list = ["query", "key", "value"], frozen: bool = True):
        self.config, _ = resolve_config(model_name_or_path)
        super().__init__(d_model=self.config.hidden_size, output_dim=output_dim, proj_type=proj_type, use_logit_scale=use_logit_scale,
                         learnable_logit_scale=learnable_logit_scale, pooling_type=pooling_type)
        if pretrained:
            self.transformer = EsmTransformer.from_pretrained(model_name_or_path, add_pooling_layer=False)
        else:
            self.transformer = EsmTransformer(self.config, add_pooling_layer=False)
        self.config = self.transformer.config
        if frozen:
            for param in self.transformer.parameters():
                param.requires_grad = False
        if use_lora:                                    # ref sequence_encoder.py:61-74 (peft LoraConfig(..., bias="all"))
            self.transformer.enable_lora(lora_r, lora_alpha, lora_target_modules, lora_dropout)

    def forward(self, x):
        return self.encode(x)


class StructTokenEncoder(BaseEncoder):
    def __init__(self, model_name_or_path: str = "esm2_t12_35M_UR50D", output_dim: int = 768, pooling_type: str = "mean", proj_type: str = "linear",
                 use_logit_scale: bool = False, learnable_logit_scale: bool = False):
        self.config, _ = resolve_config(model_name_or_path)
        super().__init__(d_model=self.config.hidden_size, output_dim=output_dim, proj_type=proj_type, use_logit_scale=use_logit_scale,
                         learnable_logit_scale=learnable_logit_scale, pooling_type=pooling_type)
        self.transformer = EsmTransformer.from_pretrained(model_name_or_path, add_pooling_layer=True)
        base_vocab = self.transformer.config.vocab_size
        self.transformer.resize_token_embeddings(base_vocab + 21)    # 21 foldseek structure tokens (ref struct_token_encoder.py:27)
        # NB the reference keeps config.vocab_size at the base value after resizing (it is the same object as transformer.config in HF);
        # here transformer.config.vocab_size is the resized table height, self.config mirrors HF behaviour of reporting the resized size.
        self.config = self.transformer.config

    def forward(self, input_ids):
        return self.encode(input_ids)


class StructEncoder(BaseEncoder):
    """ref struct_graph_encoder.py:5-42 -- the pocket / struct_graph modality of cfg-5 (configs/model/components/{pocket,struct_graph}.yaml).
    `encoder` is an OPAQUE torch module (the reference plugs in `dig.threedgraph.method.ProNet`, an un-vendored third-party GNN that is not on
    the contrastive hot path built here): it runs as it is, under torch autograd; what this class owns -- dropout, the projection head, L2
    normalisation and the logit scale -- runs on the HIP kernels and hands the gradient of the encoder output back to torch."""

    def __init__(self, encoder: torch.nn.Module, output_dim: int, proj_type: str = None, use_logit_scale: bool = False,
                 learnable_logit_scale: bool = False, pooling_type: str = None, level: str = "backbone", euler_noise: bool = True,
                 data_augment_eachlayer: bool = True, dropout: float = 0.25):
        super().__init__(d_model=output_dim, output_dim=output_dim, proj_type=proj_type, use_logit_scale=use_logit_scale,
                         learnable_logit_scale=learnable_logit_scale, pooling_type=pooling_type)
        self.encoder = encoder
        self.level = level
        self.euler_noise = euler_noise
        self.data_augment_eachlayer = data_augment_eachlayer
        self.dropout = nn.Dropout(dropout)

    def forward(self, batch):
        encoded = self.encoder(batch)
        encoded = self.dropout(encoded)
        return self.apply_head(encoded)


class TextEncoder(BaseEncoder):
    """ref text_encoder.py:8-62: BERT text tower (frozen in every shipped config, text.yaml:12; `frozen=False` -- the signature default --
    trains it through the hand-written BERT backward of oneprot_amd/bert.py).  HF's train-mode dropout is active in train mode, as in the reference (frozen tower included); `transformer.train_dropout = False` / ONEPROT_BERT_DROPOUT=0 / `.eval()` run p = 0 (see bert.py)."""

    def __init__(self, model_name_or_path: str, output_dim: int, pooling_type: str = "mean", proj_type: str = "linear", use_logit_scale: bool = False,
                 learnable_logit_scale: bool = False, frozen: bool = False, use_lora: bool = False, lora_r: int = 8, lora_alpha: int = 16,
                 lora_dropout: float = 0.1, lora_target_modules=None):
        from .bert import BertTransformer
        from .bert import resolve_bert_config
        self.config, _ = resolve_bert_config(model_name_or_path)
        super().__init__(d_model=self.config.hidden_size, output_dim=output_dim, proj_type=proj_type, use_logit_scale=use_logit_scale,
                         learnable_logit_scale=learnable_logit_scale, pooling_type=pooling_type)
        self.transformer = BertTransformer.from_pretrained(model_name_or_path)
        self.config = self.transformer.config
        if frozen:                                      # ref text_encoder.py:35-37
            for param in self.transformer.parameters():
                param.requires_grad = False
        if use_lora:                                    # ref text_encoder.py:39-52
            self.transformer.enable_lora(lora_r, lora_alpha, lora_target_modules if lora_target_modules is not None else ["query", "key", "value"], lora_dropout)
        self.use_lora = use_lora
        self.frozen = frozen

    def forward(self, input_ids):
        return self.encode(input_ids)

    def extra_repr(self):
        return f"use_lora={self.use_lora}, frozen={self.frozen}"
