"""One rank of a data-parallel sub-step with the real HIP kernels (tests/test_multirank_gpu.py).  transport "gloo" (default): both ranks on cuda:0,
gloo moves the collectives (RCCL refuses two ranks on one device) -- runs on the one-GPU test box.  transport "nccl": rank r on cuda:r, backend
"nccl" (= RCCL over xGMI): the WHOLE sub-step as the product runs it on a node -- overlapped arena-gradient all-reduce from inside the backward,
packed feature all-gather + reduce-scatter (CLIP) or the peer exchange + one reduce-scatter (SigLIP) -- on a box with at least two GPUs."""
import functools
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    rank, world, port, out_dir, golden = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    loss_name = sys.argv[6] if len(sys.argv) > 6 else "CLIP"
    transport = sys.argv[7] if len(sys.argv) > 7 else "gloo"
    dev = f"cuda:{rank}" if transport == "nccl" else "cuda:0"
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank if transport == "nccl" else 0), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                      ONEPROT_ALLOW_RANDOM_INIT="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    warnings.filterwarnings("ignore")
    from oneprot_amd import distributed as D
    from oneprot_amd.optim import FusedAdam
    from src.models.components.sequence_encoder import SequenceEncoder
    from src.models.components.struct_token_encoder import StructTokenEncoder
    from src.models.oneprot_module import OneProtLitModule
    if transport == "nccl":
        torch.cuda.set_device(dev)
    if world > 1:
        D.setup_process_group(backend=transport)
        from oneprot_amd.esm import EsmTransformer
        EsmTransformer.GRAD_CHUNK_LAYERS = 1          # 2-layer fixture: still two ranges through the overlapped all-reduce
    g = torch.load(golden, weights_only=False)
    cfg = g["cfg"]
    p = os.path.join(out_dir, f"cfg_rank{rank}")
    os.makedirs(p, exist_ok=True)
    with open(os.path.join(p, "config.json"), "w") as f:
        json.dump(dict(model_type="esm", vocab_size=cfg["vocab"], hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"],
                       intermediate_size=cfg["ffn"], pad_token_id=1, mask_token_id=32, layer_norm_eps=cfg["eps"]), f)
    seq = SequenceEncoder(p, output_dim=cfg["output_dim"], pooling_type="mean", proj_type="mlp", use_lora=False, frozen=False)
    st = StructTokenEncoder(p, output_dim=cfg["output_dim"], pooling_type="mean", proj_type="linear", use_logit_scale=True)
    seq.load_state_dict(g["sd_seq"]); st.load_state_dict(g["sd_st"])
    module = OneProtLitModule(components={"sequence": seq, "struct_token": st}, optimizer=functools.partial(FusedAdam, lr=1e-3), loss_fn=loss_name,
                              use_l1_regularization=False, local_loss=True, gather_with_grad=True).to(dev)
    B = g["seq_ids"].shape[0]
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    batch = {"struct_token": (g["seq_ids"][sl].to(dev), g["st_ids"][sl].to(dev), "struct_token", None)}
    loss = module.training_step(batch, 0)
    torch.cuda.synchronize()
    ov = getattr(module.network["struct_token"].transformer, "_grad_overlap", None)
    out = {"loss": float(loss.detach()), "gnorm": float(module.last_grad_norm), "overlap_calls": (ov.calls if ov is not None else 0),
           "w": module.network["struct_token"].state_dict()["transformer.encoder.layer.0.output.dense.weight"].cpu(),
           "emb": module.network["sequence"].state_dict()["transformer.embeddings.word_embeddings.weight"].cpu()}
    torch.save(out, os.path.join(out_dir, f"{loss_name}_{transport}_w{world}_rank{rank}.pt" if transport != "gloo" else f"{loss_name}_w{world}_rank{rank}.pt"))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
