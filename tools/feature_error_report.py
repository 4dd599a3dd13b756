"""Per-fixture error report of the HIP encoder vs the reference goldens (embedding output, layer-0 output): shows that the bf16
error per feature is the same for head_dim 16 / 32 / 24(padded).  Run on a GPU box: python tools/feature_error_report.py"""
import os, sys, tempfile, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_step_parity_gpu as T
from pathlib import Path
gd = "/root/repo/tests/golden"
def rel(a, b): return float((a - b).norm() / b.norm())
for tag in ("hd16", "hd32", "hd24"):
    tmp = Path(tempfile.mkdtemp())
    g, module = T._build(gd, tag, tmp)
    tr = module.network["sequence"].transformer
    ids = g["seq_ids"].cuda()
    with torch.no_grad():
        x, saved = tr.run_layers(ids, save=True)
    B, L = ids.shape
    a = g["acts"]
    mask = (g["seq_ids"] != 1)
    emb = saved["layers"][0]["x_in"].view(B, L, -1).cpu()
    l0 = saved["layers"][1]["x_in"].view(B, L, -1).cpu()
    xmid0 = saved["layers"][0]["x_mid"].view(B, L, -1).cpu()
    print(tag, "emb", rel(emb[mask], a["seq.embeddings"][mask]), "layer0", rel(l0[mask], a["seq.layer0"][mask]),
          "norm l0", float(a["seq.layer0"][mask].norm()))
