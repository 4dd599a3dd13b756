"""Fused Adam + global-norm clipping on the HIP kernels.

`FusedAdam` is a torch.optim.Optimizer with torch.optim.Adam's hyper-parameters (ref configs/model/default.yaml:2-6:
lr 1e-3, weight_decay 0) whose step() is one `oneprot_adam_step` launch per parameter tensor -- the encoder arena is a
single tensor, so a 148 M-parameter encoder is ONE launch (28 B/param of HBM traffic).  `clip_grad_norm_` mirrors
torch.nn.utils.clip_grad_norm_ (ref oneprot_module.py:106 via Lightning) but never synchronises with the host:
the coefficient stays on the device and is consumed by the Adam kernel (`set_grad_scale`)."""
import torch

from . import hip


def _bump_version(t):
    try:
        torch.autograd.graph.increment_version(t)
    except Exception:        # pragma: no cover  (older torch)
        t.add_(0)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._grad_scale = None

    def set_grad_scale(self, coef_tensor):
        """device scalar multiplied into every gradient at the next step (the clip coefficient); consumed once."""
        self._grad_scale = coef_tensor

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        gs = self._grad_scale
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise hip.HipKernelError("FusedAdam needs parameters on the GPU (no CPU fallback)")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad.contiguous()
                n = p.numel()
                if n % 4 == 0 and p.is_contiguous():
                    hip.call("oneprot_adam_step", p.data, g, st["exp_avg"], st["exp_avg_sq"], n, group["lr"], b1, b2, group["eps"], group["weight_decay"],
                             st["step"], gs)
                else:   # tiny odd-sized tensors (e.g. a scalar): pad through a 4-element staging copy
                    pad = (4 - n % 4) % 4
                    buf = [torch.cat([t.reshape(-1), torch.zeros(pad, device=p.device)]) for t in (p.data, g, st["exp_avg"], st["exp_avg_sq"])]
                    hip.call("oneprot_adam_step", buf[0], buf[1], buf[2], buf[3], n + pad, group["lr"], b1, b2, group["eps"], group["weight_decay"], st["step"], gs)
                    p.data.copy_(buf[0][:n].view_as(p)); st["exp_avg"].copy_(buf[2][:n].view_as(p)); st["exp_avg_sq"].copy_(buf[3][:n].view_as(p))
                _bump_version(p)
        self._grad_scale = None
        return loss


@torch.no_grad()
def clip_grad_norm_(parameters, max_norm, optimizer=None):
    """Global L2 norm over all .grad tensors (torch.nn.utils.clip_grad_norm_ semantics: coef = min(1, max/(norm+1e-6))).
    With a FusedAdam `optimizer` the scaling is deferred into the Adam kernel (no extra pass over the gradients);
    otherwise gradients are scaled in place.  Returns the total norm as a 0-d device tensor (no host sync)."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.zeros(())
    dev = grads[0].device
    ss = torch.zeros(1, device=dev)
    ws = torch.empty(hip.query("oneprot_sumsq_workspace"), dtype=torch.uint8, device=dev)
    for g in grads:
        g = g.contiguous()
        if g.data_ptr() % 16:
            g = g.clone()
        hip.call("oneprot_sumsq", g, g.numel(), ss, ws)
    coef, norm = torch.empty(1, device=dev), torch.empty(1, device=dev)
    # (the sched workspace's sticky error word -- a bounded wait of oneprot_gemm_bf16_nt_resid_ln8 ran out, NaN rows were written -- turns this step's norm and
    # coefficient into NaN on the device: no host synchronisation)
    hip.call("oneprot_clip_coef", ss, float(max_norm), coef, norm, hip.sched_ptr_or_none(dev))
    if isinstance(optimizer, FusedAdam):
        optimizer.set_grad_scale(coef)
    else:
        for g in grads:
            hip.call("oneprot_scale_by_device_scalar", g, g.numel(), coef)
    return norm.reshape(())
