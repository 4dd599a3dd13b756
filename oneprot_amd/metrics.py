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

    @staticmethod
    def _dist():
        d = torch.distributed
        return d if (d.is_available() and d.is_initialized() and d.get_world_size() > 1) else None

    def global_count(self, device=None):
        """number of accumulated pairs over ALL ranks (collective when distributed): the decision to enter compute() must be the same on
        every rank, a rank-local `if metric.preds` would dead-lock the gather below when one rank saw no batch of a modality"""
        n = sum(int(p.shape[0]) for p in self.preds)
        d = self._dist()
        if d is None:
            return n
        dev = self.preds[0].device if self.preds else (device or (torch.device("cuda", torch.cuda.current_device()) if d.get_backend() == "nccl" else torch.device("cpu")))
        t = torch.tensor([n], dtype=torch.int64, device=dev)
        d.all_reduce(t)
        return int(t)

    def _gathered(self, device=None, width=None):
        """torchmetrics dist_reduce_fx="cat" semantics: rank-major concatenation, ranks may hold different numbers of rows (even none):
        the counts travel first, every rank pads to the maximum, the padding is trimmed after the gather"""
        d = self._dist()
        if self.preds:
            s, m = torch.cat(self.preds).contiguous(), torch.cat(self.target).contiguous()
        else:
            s = m = None
        if d is None:
            return s, m
        W = d.get_world_size()
        if s is None:       # this rank saw no batch: learn the feature width / device from the others
            dev = device or (torch.device("cuda", torch.cuda.current_device()) if d.get_backend() == "nccl" else torch.device("cpu"))
            meta = torch.zeros(2, dtype=torch.int64, device=dev)
        else:
            dev = s.device
            meta = torch.tensor([s.shape[0], s.shape[1]], dtype=torch.int64, device=dev)
        metas = [torch.zeros_like(meta) for _ in range(W)]
        d.all_gather(metas, meta)
        counts = [int(x[0]) for x in metas]
        D = max(int(x[1]) for x in metas)
        nmax = max(counts)
        def padded(t):
            buf = torch.zeros(nmax, D, device=dev)
            if t is not None:
                buf[:t.shape[0]] = t
            return buf
        outs_s, outs_m = [torch.empty(nmax, D, device=dev) for _ in range(W)], [torch.empty(nmax, D, device=dev) for _ in range(W)]
        d.all_gather(outs_s, padded(s)); d.all_gather(outs_m, padded(m))
        s = torch.cat([o[:c] for o, c in zip(outs_s, counts)])
        m = torch.cat([o[:c] for o, c in zip(outs_m, counts)])
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
