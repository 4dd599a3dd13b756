"""OneProtLitModule on the HIP path (replaces ref src/models/oneprot_module.py:9-170).

Same constructor, attributes (`network`, `modalities`, `loss_fn`, `hparams.optimizer/.scheduler`) and hook names
(`forward`, `training_step`, `validation_step`, `test_step`, `configure_optimizers`).  When pytorch_lightning is
importable the class derives from LightningModule and is driven by `Trainer.fit` exactly like the reference; when it is
not (this environment), a minimal base supplies the handful of LightningModule services `training_step` uses
(`optimizers()`, `manual_backward`, `clip_gradients`, `log`, `global_step`) and `fit_steps()` drives the loop.

Order of operations inside one sub-step is the reference's (oneprot_module.py:92-107): forward sequence, forward modality,
zero_grad, loss (+0.01*L1), backward, clip-norm 1.0, optimizer step.  What differs is HOW: every op is a C-ABI HIP kernel,
gradients are all-reduced only for the active pair's parameters in large flat buckets over RCCL, nothing in the loop
synchronises with the host (losses and norms stay on the device until someone reads them).
"""
import os
from types import SimpleNamespace
from typing import Any, Dict

import torch
import torch.nn as nn

from . import distributed as D
from .loss import ClipLoss, SigLipLoss, l1_penalty
from .metrics import RetrievalMetric
from .optim import FusedAdam, clip_grad_norm_

try:                                                    # pragma: no cover  (not installed in the build image)
    from pytorch_lightning import LightningModule as _LightningBase
    HAVE_LIGHTNING = True
except Exception:
    _LightningBase = None
    HAVE_LIGHTNING = False


class _DeviceMean:
    """torchmetrics.MeanMetric stand-in that never synchronises: running sum / count on the device."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.total, self.count = None, 0

    def __call__(self, value):
        v = value.detach().float().reshape(())
        self.total = v.clone() if self.total is None else self.total + v
        self.count += 1

    update = __call__

    def compute(self):
        return self.total / max(self.count, 1) if self.total is not None else torch.tensor(float("nan"))


class _DeviceMin:
    def __init__(self):
        self.reset()

    def reset(self):
        self.value = None

    def __call__(self, value):
        v = value.detach().float().reshape(())
        self.value = v if self.value is None else torch.minimum(self.value, v)

    def compute(self):
        return self.value if self.value is not None else torch.tensor(float("inf"))


class _PlainServices:
    """The slice of LightningModule that OneProtLitModule.training_step relies on, without a Trainer: used as the whole base when Lightning is
    absent, and as the fallback of a real LightningModule that is driven without a Trainer (fit_steps / bench.py)."""

    def _plain_optimizers(self):
        if getattr(self, "_optimizer", None) is None:
            cfg = self.configure_optimizers()
            self._optimizer = cfg["optimizer"]
            self._lr_scheduler = cfg.get("lr_scheduler")            # {"scheduler", "monitor", "interval", "frequency"} or None
        return self._optimizer

    def _plain_lr_schedulers(self):
        self._plain_optimizers()
        return self._lr_scheduler["scheduler"] if self._lr_scheduler else None

    def _plain_clip_gradients(self, optimizer, gradient_clip_val=None, gradient_clip_algorithm="norm"):
        assert gradient_clip_algorithm == "norm"
        params = [p for g in optimizer.param_groups for p in g["params"]]
        self.last_grad_norm = clip_grad_norm_(params, gradient_clip_val, optimizer)

    def _plain_log(self, name, value, **kw):
        if not hasattr(self, "logged"):
            self.logged = {}
        self.logged[name] = value


class _PlainBase(nn.Module, _PlainServices):
    def __init__(self):
        super().__init__()
        self.automatic_optimization = True
        self.hparams = SimpleNamespace()
        self.global_step = 0
        self._optimizer = None
        self._lr_scheduler = None
        self.logged = {}

    def save_hyperparameters(self, logger=False, **kw):
        pass   # hparams are filled explicitly by the subclass

    optimizers = _PlainServices._plain_optimizers
    lr_schedulers = _PlainServices._plain_lr_schedulers
    clip_gradients = _PlainServices._plain_clip_gradients
    log = _PlainServices._plain_log

    def manual_backward(self, loss):
        loss.backward()


if HAVE_LIGHTNING:                                      # pragma: no cover  (not installed in the build image)
    class _Base(_LightningBase, _PlainServices):
        """LightningModule whose Trainer-backed services fall back to the plain ones while no Trainer is attached"""

        def _has_trainer(self):
            return getattr(self, "_trainer", None) is not None

        def optimizers(self, *a, **kw):
            return super().optimizers(*a, **kw) if self._has_trainer() else self._plain_optimizers()

        def lr_schedulers(self):
            return super().lr_schedulers() if self._has_trainer() else self._plain_lr_schedulers()

        def manual_backward(self, loss, *a, **kw):
            return super().manual_backward(loss, *a, **kw) if self._has_trainer() else loss.backward()

        def clip_gradients(self, optimizer, gradient_clip_val=None, gradient_clip_algorithm=None):
            if self._has_trainer():
                return super().clip_gradients(optimizer, gradient_clip_val=gradient_clip_val, gradient_clip_algorithm=gradient_clip_algorithm)
            return self._plain_clip_gradients(optimizer, gradient_clip_val, gradient_clip_algorithm or "norm")

        def log(self, name, value, *a, **kw):
            return super().log(name, value, *a, **kw) if self._has_trainer() else self._plain_log(name, value)
else:
    _Base = _PlainBase


class OneProtLitModule(_Base):
    def __init__(self, components: Dict[str, Any], optimizer: Any, train_on_all_modalities_after_step: int = 0, scheduler: Any = None,
                 use_seqsim: bool = False, loss_fn: str = 'CLIP', use_l1_regularization: bool = False, local_loss: bool = True,
                 gather_with_grad: bool = True):
        super().__init__()
        self.automatic_optimization = False
        if HAVE_LIGHTNING:                                # pragma: no cover
            self.save_hyperparameters(logger=False)
        else:
            self.hparams = SimpleNamespace(optimizer=optimizer, scheduler=scheduler)
        self.network = torch.nn.ModuleDict(components)
        self.modalities = list(components.keys())
        # dropout stream ids carry the tower: tie it to the MODALITY NAME (stable across processes, whatever order the towers were built in; two copies of one
        # tower under two names draw different masks) rather than to the construction order the tower numbered itself with
        import zlib
        for name, enc in self.network.items():
            tr = getattr(enc, "transformer", None)
            if hasattr(tr, "_rng_uid"):
                tr._rng_uid = zlib.crc32(name.encode()) & 0xFFFF
        self.train_on_all_modalities_after_step = train_on_all_modalities_after_step
        self.use_l1_regularization = use_l1_regularization
        self.loss_fn = self._create_loss_fn(loss_fn, local_loss, gather_with_grad)
        self.train_loss, self.val_loss, self.test_loss = _DeviceMean(), _DeviceMean(), _DeviceMean()
        self.val_loss_best = _DeviceMin()
        self.use_seqsim = use_seqsim
        self.metrics = {f"{split}_{modality}": RetrievalMetric() for split in ["val", "test"]
                        for modality in list(self.network.keys()) + ["seqsim"] if modality != 'sequence'}

    def l1_regularization(self, features):
        return l1_penalty(features, 1.0)

    def _create_loss_fn(self, loss_fn, local_loss, gather_with_grad):
        # rank / world size come from the environment, as in the reference (KeyError if unset: oneprot_module.py:54-55)
        if loss_fn == 'CLIP':
            return ClipLoss(local_loss=local_loss, gather_with_grad=gather_with_grad, cache_labels=True, rank=int(os.environ['RANK']),
                            world_size=int(os.environ['WORLD_SIZE']))
        elif loss_fn == 'SIGLIP':
            return SigLipLoss(cache_labels=True, rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
        else:
            raise ValueError(f"Unknown loss function: {loss_fn}")

    def forward(self, x, modality="sequence"):
        if modality in ["sequence", "seqsim"]:
            modality = "sequence"
        return self.network[modality](x)

    def on_train_start(self) -> None:
        for m in (self.train_loss, self.val_loss, self.test_loss, self.val_loss_best):
            m.reset()

    # ------------------------------------------------------------------------------------------- the hot loop
    def training_step(self, batch, batch_idx=None):
        opt = self.optimizers()
        self._attach_grad_overlap()
        current_step = self.global_step if not (HAVE_LIGHTNING and self._owns_gradient_sync()) else getattr(self, "_standalone_step", 0)
        if current_step < self.train_on_all_modalities_after_step:
            modalities_to_train = ["struct_token"]
        else:
            modalities_to_train = list(batch.keys())
            if not self.use_seqsim and "seqsim" in modalities_to_train:
                modalities_to_train.remove("seqsim")
        loss = None
        for modality in modalities_to_train:
            sequence_inputs, modality_inputs, _, _ = batch[modality]
            sequence_features = self.forward(sequence_inputs, "sequence")
            modality_features = self.forward(modality_inputs, modality)
            opt.zero_grad()
            loss = self.loss_fn(sequence_features, modality_features)          # argument order as in the reference (symmetric)
            if self.use_l1_regularization:
                loss = loss + l1_penalty(sequence_features, 0.01) + l1_penalty(modality_features, 0.01)
            self.train_loss(loss)
            self.manual_backward(loss)
            self._sync_gradients(opt)
            self.clip_gradients(opt, gradient_clip_val=1.0, gradient_clip_algorithm="norm")
            opt.step()
            self.log("train/loss", self.train_loss, on_step=True, on_epoch=True, prog_bar=True, sync_dist=True)
            if self._owns_gradient_sync():
                self._bump_global_step()
        return loss

    def _bump_global_step(self):
        if HAVE_LIGHTNING:                               # pragma: no cover  (global_step is a Trainer-backed property there)
            self._standalone_step = getattr(self, "_standalone_step", 0) + 1
        else:
            self.global_step += 1

    def _sync_gradients(self, opt):
        """What Lightning's DDP wrapper does implicitly in the reference (C6 in SURVEY.md section 2.2) -- here explicit, and only
        over parameters that received a gradient in this sub-step.  The encoder arenas are reduced in ranges from inside their backward
        (distributed.GradOverlap, attached on first use); this call waits for those and reduces the small head parameters."""
        if getattr(self.loss_fn, "world_size", 1) > 1 and self._owns_gradient_sync():
            D.allreduce_gradients([p for g in opt.param_groups for p in g["params"]])

    def _owns_gradient_sync(self):
        """True when nothing else reduces the gradients: no Lightning, or a LightningModule that is driven without a Trainer (fit_steps /
        bench.py).  Under a Trainer the DDP strategy wraps the module and does it (configs/trainer/ddp.yaml)."""
        return (not HAVE_LIGHTNING) or getattr(self, "_trainer", None) is None

    def _attach_grad_overlap(self):
        if getattr(self, "_overlap_attached", False) or not self._owns_gradient_sync() or getattr(self.loss_fn, "world_size", 1) <= 1:
            return
        if D.is_dist_avail_and_initialized():
            if not D.grad_overlap_enabled():         # ONEPROT_GRAD_OVERLAP=0: allreduce_gradients reduces the arenas in buckets after the backward
                self._overlap_attached = True
                return
            ov = D.GradOverlap()
            for enc in self.network.values():
                ov.attach(enc.transformer)
            self._overlap_attached = True

    def validation_step(self, batch, batch_idx=None, dataloader_idx=0):
        sequence_inputs, modality_inputs, modality, _ = batch
        with torch.no_grad():
            sequence_features = self.forward(sequence_inputs, "sequence")
            modality_features = self.forward(modality_inputs, modality)
            self.metrics["val_" + modality].update(sequence_features, modality_features)
            loss = self.loss_fn(sequence_features, modality_features)
        self.val_loss(loss)
        self.log("val/loss", self.val_loss, on_step=False, on_epoch=True, prog_bar=True, sync_dist=True)
        return loss

    def test_step(self, batch, batch_idx=None):
        out = {}
        for modality, (seq_inputs, mod_inputs, _, _) in batch.items():
            with torch.no_grad():
                seq_features = self(seq_inputs, "sequence")
                mod_features = self(mod_inputs, modality)
                # the reference passes the logit scale a second time here (oneprot_module.py:142) -- reproduced
                loss = self.loss_fn(seq_features, mod_features, self.network[modality].norm[1].log_logit_scale.exp())
            self.test_loss(loss)
            self.log(f"test/loss_{modality}", loss, on_step=False, on_epoch=True, prog_bar=True)
            self.metrics[f"test_{modality}"].update(seq_features, mod_features)
            out[modality] = loss
        return out

    @staticmethod
    def _check_kernel_waits():
        """The FFN-2 + LayerNorm launches wait, bounded, for neighbouring work-groups (oneprot_gemm_bf16_nt_resid_ln8).  A wait that runs out is loud on the
        device: the rows concerned are written as NaN and the step's gradient norm / clip coefficient become NaN (oneprot_clip_coef reads the sticky flag), so the
        loss of that very step is NaN.  This host-synchronous query turns the flag into an exception with the reason, where the host waits anyway (end of a
        validation epoch, end of fit_steps), and clears it so that a run restarted with ONEPROT_FFN2_LN=0 in the same process is not poisoned."""
        from . import hip
        if hip.sched_error() != 0:
            hip.sched_error_clear()
            raise hip.HipKernelError("oneprot_gemm_bf16_nt_resid_ln8: a wait for the neighbouring column tiles ran out and NaN rows were written (fewer CUs "
                                     "available than the form needs: another process on this GPU?); set ONEPROT_FFN2_LN=0 to keep the GEMM and the LayerNorm as "
                                     "separate launches")

    def on_validation_epoch_end(self):
        """ref oneprot_module.py:123-135"""
        self._check_kernel_waits()
        loss = self._epoch_val_loss()
        self.val_loss_best(loss)
        self.log("val/loss_best", self.val_loss_best.compute(), sync_dist=True, prog_bar=True)
        for name, metric in self.metrics.items():
            if name.startswith("val_") and metric.global_count() > 0:       # collective decision: every rank enters compute() or none does
                for key, value in metric.compute().items():
                    self.log(f"val/{key}/{name}", value, sync_dist=True, prog_bar=True)
                metric.reset()
        sched = getattr(self, "_lr_scheduler", None)
        if sched and self._owns_gradient_sync():        # what Lightning does for {"interval": "epoch", "monitor": "val/loss_best"} (ref oneprot_module.py:161-169)
            s = sched["scheduler"]
            if "metrics" in s.step.__code__.co_varnames:
                s.step(self.val_loss_best.compute())      # identical on every rank (_epoch_val_loss): the replicas take the same plateau decisions
            else:
                s.step()
        if self._owns_gradient_sync():
            self.val_loss.reset()                          # per-epoch mean, as torchmetrics resets a logged metric at epoch end under Lightning

    def _epoch_val_loss(self):
        """Mean validation loss of the epoch over ALL ranks.  Under a Trainer `self.log(..., sync_dist=True)` / torchmetrics reduce it; without one
        the running sum and count are all-reduced here -- each rank's local_loss value differs, and a rank-local value fed to ReduceLROnPlateau
        would let the data-parallel replicas drift apart through different learning rates."""
        m = self.val_loss
        if not (self._owns_gradient_sync() and D.is_dist_avail_and_initialized() and D.get_world_size() > 1):
            return m.compute()
        import torch.distributed as dist
        total = m.total if m.total is not None else torch.zeros(())
        dev = total.device if dist.get_backend() != "nccl" or total.is_cuda else torch.device("cuda", torch.cuda.current_device())
        buf = torch.stack([total.to(dev).double(), torch.tensor(float(m.count), dtype=torch.float64, device=dev)])
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        return (buf[0] / buf[1].clamp(min=1.0)).float()

    def on_test_epoch_end(self):
        """ref oneprot_module.py:148-154"""
        for name, metric in self.metrics.items():
            if name.startswith("test_") and metric.global_count() > 0:
                for key, value in metric.compute().items():
                    self.log(f"test/{key}/{name}", value, prog_bar=True)
                metric.reset()

    def configure_optimizers(self):
        optimizer = self.hparams.optimizer(params=self.parameters())
        if self.hparams.scheduler is not None:
            scheduler = self.hparams.scheduler(optimizer=optimizer)
            return {"optimizer": optimizer, "lr_scheduler": {"scheduler": scheduler, "monitor": "val/loss_best", "interval": "epoch", "frequency": 1}}
        return {"optimizer": optimizer}

    # ------------------------------------------------------------------------------------------- checkpoint hooks
    # (LightningModule hook names: a Trainer calls them around torch.save / ckpt_path=...; oneprot_amd.data.save_checkpoint / load_weights_only call
    # them too.)  The dropout streams of the towers (LoRA dropout, the BERT tower's train-mode dropout) are counter-based -- (seed, call counter)
    # -- and live outside the state dict, whose key set is the reference's on-disk contract: they travel in a checkpoint entry of their own, so that
    # a resumed run continues the mask sequence instead of replaying it from call 0.
    def on_save_checkpoint(self, checkpoint):
        checkpoint["oneprot_amd_dropout_rng"] = {m: enc.transformer.rng_state() for m, enc in self.network.items()
                                                 if hasattr(getattr(enc, "transformer", None), "rng_state")}

    def on_load_checkpoint(self, checkpoint):
        saved = checkpoint.get("oneprot_amd_dropout_rng") or {}
        for m, state in saved.items():
            tr = getattr(self.network[m], "transformer", None) if m in self.network else None
            if hasattr(tr, "set_rng_state"):
                tr.set_rng_state(state)
        if saved:      # a tower with dropout streams that the checkpoint knows nothing about would silently restart its mask sequence
            missing = [m for m, enc in self.network.items() if m not in saved and getattr(enc, "transformer", None) is not None
                       and hasattr(enc.transformer, "rng_state") and enc.transformer.rng_state()]
            if missing:
                import warnings
                warnings.warn(f"checkpoint carries no dropout stream state for {missing}: their mask sequences restart from call 0")

    # ------------------------------------------------------------------------------------------- minimal driver
    def fit_steps(self, batches):
        """Drive training_step over an iterable of CombinedLoader-style batches ({modality: (seq_ids, mod_ids, name, raw)})."""
        self.train()
        self.on_train_start()
        last = None
        for i, batch in enumerate(batches):
            last = self.training_step(batch, i)
        self._check_kernel_waits()
        return last
