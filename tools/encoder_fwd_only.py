#!/usr/bin/env python3
"""Runs the ESM-2-150M sequence-encoder forward (frozen tower: embedding + 30 layers, 256 x L=512) a few times -- the workload of bench.py's
`encoder_fwd` object -- for rocprofv3 --pmc passes (MFMA-busy of the north_star figure).  usage: encoder_fwd_only.py [iters]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(RANK="0", WORLD_SIZE="1", ONEPROT_ALLOW_RANDOM_INIT="1")
warnings.filterwarnings("ignore")
import torch
from src.models.components.sequence_encoder import SequenceEncoder
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
torch.manual_seed(1881)
enc = SequenceEncoder("facebook/esm2_t30_150M_UR50D", output_dim=1024, pooling_type="mean", proj_type="mlp", use_lora=False, frozen=True).cuda()
g = torch.Generator().manual_seed(1881)
ids = torch.randint(4, 24, (256, 512), generator=g); ids[:, 0] = 0; ids[:, -1] = 2
ids = ids.cuda()
with torch.no_grad():
    for _ in range(iters):
        enc.transformer.run_layers(ids, save=False)
torch.cuda.synchronize()
