"""RetrievalMetric on the device (replaces ref src/models/components/retrieval_metric.py for the validation / test hooks of
OneProtLitModule).  Same interface (`update(sequence_features, modality_features)`, `compute() -> dict`,
`reset()`, keys `{seq_to_mod,mod_to_seq}_{median_rank,R@k}`), but the N x N similarity matrix stays on the GPU and the rank of each
matching pair is COUNTED (entries beating the diagonal) by one kernel instead of argsorting every row on the CPU."""
import numpy as np
import torch

from . import hip


class RetrievalMetric:
    def __init__(self, k=(1, 10, 100)):
        self.k = list(k)
        self.reset()

    def reset(self):
        self.preds, self.target = [], []

    def update(self, preds, target):
        self.preds.append(preds.detach().float())
        self.target.append(target.detach().float())

    def _gathered(self):
        s, m = torch.cat(self.preds), torch.cat(self.target)
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:   # dist_reduce_fx="cat"
            W = torch.distributed.get_world_size()
            outs = [torch.empty_like(s) for _ in range(W)], [torch.empty_like(m) for _ in range(W)]
            torch.distributed.all_gather(outs[0], s.contiguous()); torch.distributed.all_gather(outs[1], m.contiguous())
            s, m = torch.cat(outs[0]), torch.cat(outs[1])
        return s.contiguous(), m.contiguous()

    def compute(self):
        s, m = self._gathered()
        N, D = s.shape
        logits = torch.empty(N, N, device=s.device)
        hip.call("oneprot_sgemm", s, m, logits, N, N, D, 0, 0, 1.0, 0)
        rr = torch.empty(N, dtype=torch.int32, device=s.device)
        rc = torch.empty(N, dtype=torch.int32, device=s.device)
        hip.call("oneprot_diag_rank", logits, rr, rc, N)
        out = {}
        for name, ranks in (("seq_to_mod", rr), ("mod_to_seq", rc)):
            r = ranks.cpu().numpy()
            out[f"{name}_median_rank"] = float(np.floor(np.median(r)) + 1)
            for k in self.k:
                out[f"{name}_R@{k}"] = float(np.mean(r < k))
        return out
