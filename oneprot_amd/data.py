"""The caller side of the hot path: the mixed-batch scheduler and synthetic batches, plus the weights-only
checkpoint wire format (section 8f #3).

* `CombinedLoader(iterables, mode)` -- the two modes the reference uses (ref oneprot_datamodule.py:75; Lightning's CombinedLoader):
  "min_size": one dict {modality: batch} per step until the SHORTEST iterable is exhausted (training: `training_step` then loops over the
  dict's modalities, one optimiser sub-step each, ref oneprot_module.py:84-92); "sequential": one modality batch at a time, yielding
  (batch, batch_idx, dataloader_idx) (validation / test).
* `SyntheticPairs` -- batches laid out as the reference's collate functions produce them (ref struct_token_dataset.py:87-90, text_dataset.py):
  `(sequence_ids[B,Ls] int64, modality_ids[B,Lm] int64, modality_name, raw)`, ids per BASELINE.md section 3 (cls 0 / eos 2 / pad 1, amino acids 4..23,
  foldseek letters 33..52; text: cls 2 / sep 3 / pad 0, body 5..vocab-1), seeded, optionally ragged with right padding.
* `save_checkpoint` / `load_weights_only` -- Lightning-style {"state_dict": {...}} files with the reference's key names
  (`network.<modality>.transformer....`), loaded exactly as ref src/train.py:73-82 does (optional 'model.' prefix, strict=True, weights only).
"""
import torch


class CombinedLoader:
    def __init__(self, iterables, mode="min_size"):
        if mode not in ("min_size", "sequential"):
            raise ValueError(f"unsupported CombinedLoader mode {mode!r} (the reference uses 'min_size' and 'sequential')")
        self.iterables, self.mode = dict(iterables), mode

    def __iter__(self):
        if self.mode == "min_size":
            its = {k: iter(v) for k, v in self.iterables.items()}
            while True:
                out = {}
                for k, it in its.items():
                    try:
                        out[k] = next(it)
                    except StopIteration:
                        return
                yield out
        else:
            for idx, (k, v) in enumerate(self.iterables.items()):
                for bi, batch in enumerate(v):
                    yield batch, bi, idx

    def __len__(self):
        lens = [len(v) for v in self.iterables.values()]
        return min(lens) if self.mode == "min_size" else sum(lens)


class SyntheticPairs:
    """Re-iterable synthetic (sequence, modality) batches for one modality."""

    def __init__(self, modality, batch_size, seq_len, mod_len=None, n_batches=1, seed=1881, device="cpu", ragged=False, text_vocab=30522):
        self.modality, self.B, self.Ls, self.Lm = modality, batch_size, seq_len, mod_len or seq_len
        self.n, self.seed, self.device, self.ragged, self.text_vocab = n_batches, seed, device, ragged, text_vocab

    def __len__(self):
        return self.n

    @staticmethod
    def _frame(gen, B, L, lo, hi, cls, eos, pad, ragged):
        ids = torch.randint(lo, hi + 1, (B, L), generator=gen)
        ids[:, 0] = cls
        lens = torch.randint(max(L // 4, 2), L + 1, (B,), generator=gen) if ragged else torch.full((B,), L)
        for b in range(B):
            n = int(lens[b])
            ids[b, n - 1] = eos
            ids[b, n:] = pad
        return ids

    def __iter__(self):
        gen = torch.Generator().manual_seed(self.seed)
        for _ in range(self.n):
            seq = self._frame(gen, self.B, self.Ls, 4, 23, 0, 2, 1, self.ragged)
            if self.modality == "text":
                mod = self._frame(gen, self.B, self.Lm, 5, self.text_vocab - 1, 2, 3, 0, self.ragged)
            else:       # struct_token (and seqsim, which re-uses the sequence vocabulary in the reference)
                lo, hi = (33, 52) if self.modality == "struct_token" else (4, 23)
                mod = self._frame(gen, self.B, self.Lm, lo, hi, 0, 2, 1, self.ragged)
            yield seq.to(self.device), mod.to(self.device), self.modality, None


class StandInGraphEncoder(torch.nn.Module):
    """NOT ProNet.  A stand-in for the opaque `encoder` argument of StructEncoder (pocket / struct_graph modality, ref struct_graph_encoder.py:5-42;
    the reference plugs in dig.threedgraph.method.ProNet, un-vendored third-party code that is not part of the hot path built here) so that the
    mixed-batch scheduler can be exercised with a 4th modality of cfg-5's shape: per-node MLP on [B, nodes, in_dim] descriptors, mean over the
    nodes, linear to `out_channels` -- plain torch modules under torch autograd, exactly how a user-supplied GNN would run."""

    def __init__(self, in_dim=16, hidden=256, out_channels=1024):
        super().__init__()
        self.node = torch.nn.Linear(in_dim, hidden)
        self.out = torch.nn.Linear(hidden, out_channels)

    def forward(self, batch):
        return self.out(torch.nn.functional.silu(self.node(batch)).mean(dim=1))


def save_checkpoint(module, path):
    checkpoint = {"state_dict": {k: v.detach().cpu() for k, v in module.state_dict().items()}, "global_step": getattr(module, "global_step", 0)}
    if hasattr(module, "on_save_checkpoint"):
        module.on_save_checkpoint(checkpoint)            # dropout stream positions (plain ints), outside the state dict
    torch.save(checkpoint, path)


def load_weights_only(module, path):
    """ref src/train.py:73-82: weights only, strict, optional 'model.' prefix (optimizer state is NOT restored, as in the reference)."""
    checkpoint = torch.load(path, map_location="cpu", weights_only=True)
    sd = checkpoint["state_dict"]
    if "model." in next(iter(sd.keys())):
        sd = {k.replace("model.", ""): v for k, v in sd.items() if k.startswith("model.")}
    result = module.load_state_dict(sd, strict=True)
    if hasattr(module, "on_load_checkpoint"):
        module.on_load_checkpoint(checkpoint)            # no-op for checkpoints written by the reference
    return result
