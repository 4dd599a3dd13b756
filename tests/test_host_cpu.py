"""CPU-side checks (no GPU compute): the C-ABI library loads and exports every symbol include/oneprot_hip.h declares, the ctypes
table matches the header, the drop-in classes keep the reference's import paths / constructor signatures / state-dict keys, the
config composer handles the reference's YAML, and the product path refuses to run without the GPU (no silent fallback)."""
import ctypes
import inspect
import json
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "oneprot_hip.h")
REF = "/root/reference"


@pytest.fixture(scope="session")
def built_lib():
    from oneprot_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        subprocess.check_call(["bash", os.path.join(ROOT, "oneprot_amd", "csrc", "build.sh")])
    return hip


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(oneprot_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built_lib):
    names = _declared()
    assert len(names) >= 30
    h = ctypes.CDLL(built_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(h, n)]
    assert not missing, missing
    assert h.oneprot_abi_version() == 7
    # the ctypes table and the header agree (both directions)
    assert sorted(built_lib.exported_symbols()) == names


def test_ctypes_arity_matches_header(built_lib):
    txt = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (res, args) in built_lib._SIGS.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", txt, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("void", "") else params.count(",") + 1
        assert n == len(args), (name, n, len(args))


def test_argument_validation_without_gpu(built_lib):
    h = built_lib.lib()
    # NULL pointers / bad shapes are rejected with -1 before any launch
    assert h.oneprot_layernorm_fwd(None, 0, None, None, None, None, None, None, 10, 64, 1e-5, None) == -1
    assert h.oneprot_gemm_bf16_nt(None, None, 128, 128, 64, 64, 64, 0, None, None, None, None, None, None, None, 1.0, 0, 0, 0, None) == -1
    assert h.oneprot_attn_fwd(None, None, None, None, None, None, 1, 1, 16, 32, None) == -1
    assert h.oneprot_adam_step(None, None, None, None, 16, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None, None) == -1
    assert h.oneprot_gemm_bf16_tn_workspace(640, 640) > 0 and h.oneprot_layernorm_bwd_workspace(640) > 0


def test_binding_refuses_host_tensors_and_bad_layouts(built_lib):
    """hip.call checks device, contiguity and dtype of every tensor before the launch (the C side only sees addresses): no GPU needed to see it refuse"""
    import torch
    x = torch.randn(8, 64)
    with pytest.raises(built_lib.HipKernelError, match="no CPU fallback"):
        built_lib.call("oneprot_cast_f32_to_bf16", x, torch.empty(8 * 64, dtype=torch.bfloat16), 8 * 64)
    # the dtype table covers every pointer slot of every kernel entry point (the trailing pointer of a signature is the stream)
    import ctypes
    for name, (res, args) in built_lib._SIGS.items():
        n_ptr = sum(1 for a in args if a is ctypes.c_void_p)
        if n_ptr > 1:
            assert len(built_lib._PTR_DTYPES[name]) == n_ptr - 1, name
    from oneprot_amd import comm
    with pytest.raises(comm.CommError, match="128 bytes"):
        comm.RcclComm(2, 0, b"short id")


def test_missing_library_fails_loudly(monkeypatch, built_lib):
    monkeypatch.setattr(built_lib, "_lib", None)
    monkeypatch.setattr(built_lib, "LIB_PATH", "/nonexistent/liboneprot_hip.so")
    with pytest.raises(built_lib.HipLibraryMissing):
        built_lib.lib()


def _tiny_dir(tmp_path, model_type="esm"):
    p = os.path.join(str(tmp_path), model_type)
    os.makedirs(p, exist_ok=True)
    cfg = dict(model_type=model_type, vocab_size=33, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128)
    if model_type == "bert":
        cfg.update(vocab_size=120, max_position_embeddings=64, pad_token_id=0, layer_norm_eps=1e-12)
    with open(os.path.join(p, "config.json"), "w") as f:
        json.dump(cfg, f)
    return p


def test_dropin_paths_signatures_and_state_dict(golden_dir, tmp_path, monkeypatch):
    monkeypatch.setenv("ONEPROT_ALLOW_RANDOM_INIT", "1")
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "1")
    from src.models.components.sequence_encoder import SequenceEncoder
    from src.models.components.struct_token_encoder import StructTokenEncoder
    from src.models.components.text_encoder import TextEncoder
    from src.models.components.loss import ClipLoss, SigLipLoss, gather_features  # noqa: F401
    from src.models.components.base_encoder import BaseEncoder, Normalize, LearnableLogitScaling  # noqa: F401
    from src.models.oneprot_module import OneProtLitModule
    # constructor parameter names of the reference (ref sequence_encoder.py:23-38, struct_token_encoder.py:7-15, text_encoder.py:9-23, oneprot_module.py:10-21)
    assert list(inspect.signature(SequenceEncoder.__init__).parameters)[1:] == [
        "model_name_or_path", "output_dim", "pooling_type", "proj_type", "use_logit_scale", "learnable_logit_scale", "pretrained", "use_lora", "lora_r",
        "lora_alpha", "lora_dropout", "lora_target_modules", "frozen"]
    assert list(inspect.signature(StructTokenEncoder.__init__).parameters)[1:] == [
        "model_name_or_path", "output_dim", "pooling_type", "proj_type", "use_logit_scale", "learnable_logit_scale"]
    assert list(inspect.signature(TextEncoder.__init__).parameters)[1:] == [
        "model_name_or_path", "output_dim", "pooling_type", "proj_type", "use_logit_scale", "learnable_logit_scale", "frozen", "use_lora", "lora_r", "lora_alpha",
        "lora_dropout", "lora_target_modules"]
    assert list(inspect.signature(OneProtLitModule.__init__).parameters)[1:] == [
        "components", "optimizer", "train_on_all_modalities_after_step", "scheduler", "use_seqsim", "loss_fn", "use_l1_regularization", "local_loss",
        "gather_with_grad"]
    assert list(inspect.signature(ClipLoss.__init__).parameters)[1:] == ["local_loss", "gather_with_grad", "cache_labels", "rank", "world_size", "use_horovod"]
    assert list(inspect.signature(SigLipLoss.__init__).parameters)[1:] == ["cache_labels", "rank", "world_size", "bidir", "use_horovod"]
    g = torch.load(os.path.join(golden_dir, "esm_pair_hd16.pt"), weights_only=False)
    p = _tiny_dir(tmp_path)
    seq = SequenceEncoder(p, output_dim=48, proj_type="mlp", use_lora=False, frozen=True)
    st = StructTokenEncoder(p, output_dim=48, proj_type="linear", use_logit_scale=True)
    # strict load of the reference's own state dicts, and identical key sets on the way out
    seq.load_state_dict(g["sd_seq"], strict=True)
    st.load_state_dict(g["sd_st"], strict=True)
    for enc, ref in ((seq, g["sd_seq"]), (st, g["sd_st"])):
        sd = enc.state_dict()
        assert set(sd) == set(ref)
        for k in sd:
            assert "inv_freq" in k or torch.equal(sd[k], ref[k]), k
    assert all(not p_.requires_grad for p_ in seq.transformer.parameters())          # frozen=True
    assert st.transformer.config.vocab_size == 54 and st.config.pad_token_id == 1
    assert abs(st.logit_scale_value() - 1 / 0.07) < 1e-4 and seq.logit_scale_value() == 1.0
    assert st.norm[1].log_logit_scale.ndim == 0                                        # attribute the reference's test_step reads
    tb = torch.load(os.path.join(golden_dir, "bert_text.pt"), weights_only=False)
    txt = TextEncoder(_tiny_dir(tmp_path, "bert"), output_dim=48, pooling_type="cls", proj_type="mlp", use_logit_scale=True, frozen=True, use_lora=False)
    txt.load_state_dict(tb["sd"], strict=True)
    assert set(txt.state_dict()) == set(tb["sd"])
    # module: loss selection and error conventions of the reference
    import functools
    mod = OneProtLitModule(components={"sequence": seq, "struct_token": st}, optimizer=functools.partial(torch.optim.Adam, lr=1e-3))
    assert mod.modalities == ["sequence", "struct_token"] and isinstance(mod.loss_fn, ClipLoss) and mod.loss_fn.local_loss and mod.loss_fn.gather_with_grad
    assert isinstance(mod.configure_optimizers()["optimizer"], torch.optim.Adam)
    with pytest.raises(ValueError):
        OneProtLitModule(components={"sequence": seq}, optimizer=None, loss_fn="NOPE")
    monkeypatch.delenv("RANK")
    with pytest.raises(KeyError):
        OneProtLitModule(components={"sequence": seq}, optimizer=None)
    # no CPU fallback anywhere on the product path
    from oneprot_amd import hip
    with pytest.raises(hip.HipKernelError):
        seq(g["seq_ids"])
    with pytest.raises(OSError):
        SequenceEncoder("not/a-model", output_dim=8, use_lora=False)
    with pytest.raises(NotImplementedError):
        SequenceEncoder(p, output_dim=48, use_lora=True, lora_target_modules=["query", "dense"])      # only the reference's q/k/v targets are built
    lora = SequenceEncoder(p, output_dim=48)             # signature defaults: use_lora=True, frozen=True  (ref sequence_encoder.py:23-38)
    keys = set(lora.state_dict())
    assert "transformer.base_model.model.encoder.layer.0.attention.self.key.lora_B.default.weight" in keys          # peft's key layout
    assert "transformer.base_model.model.embeddings.word_embeddings.weight" in keys and not any(k.endswith("flat") for k in keys)
    assert sorted(n for n, q in lora.transformer.named_parameters() if q.requires_grad) == ["flat", "lora_A", "lora_B"]
    assert float(lora.transformer.lora_B.abs().max()) == 0.0 and float(lora.transformer.lora_A.abs().max()) <= 1 / 8.0 + 1e-6     # B = 0, A ~ U(-1/sqrt(d), 1/sqrt(d))
    monkeypatch.delenv("ONEPROT_ALLOW_RANDOM_INIT")
    with pytest.raises(OSError):
        StructTokenEncoder(p, output_dim=48)          # config only, no weight file: refuses unless random init is explicitly allowed


def test_config_composer_own_tree():
    from oneprot_amd.config import compose, instantiate
    c = compose(os.path.join(ROOT, "configs"), "train")
    m = c["model"]
    assert m["_target_"] == "src.models.oneprot_module.OneProtLitModule"
    assert m["components"]["struct_token"]["output_dim"] == 1024 and m["components"]["sequence"]["frozen"] is True
    assert m["loss_fn"] == "CLIP" and m["local_loss"] and m["gather_with_grad"] and m["use_l1_regularization"]
    # a drop-in keeps the reference's defaults (ref configs/model/default.yaml:2-6): torch.optim.Adam; the fused optimiser is an opt-in group
    opt = instantiate(m["optimizer"])
    assert opt.func is torch.optim.Adam and opt.keywords["lr"] == 0.001 and opt.keywords["weight_decay"] == 0.0
    cf = compose(os.path.join(ROOT, "configs"), "train", overrides={"model": "oneprot_fused"})
    optf = instantiate(cf["model"]["optimizer"])
    assert optf.func.__name__ == "FusedAdam" and optf.keywords["lr"] == 0.001 and cf["model"]["loss_fn"] == "CLIP"
    # trainer / paths / data-modality mirrors of the reference's files
    t = c["trainer"]
    assert t["_target_"] == "pytorch_lightning.trainer.Trainer" and t["accelerator"] == "gpu" and t["devices"] == 1 and t["default_root_dir"] == c["paths"]["output_dir"]
    sim = compose(os.path.join(ROOT, "configs"), "trainer/ddp_sim", resolve=False)["trainer"]
    assert sim["accelerator"] == "cpu" and sim["devices"] == 2 and sim["strategy"] == "ddp_spawn" and sim["max_epochs"] == 10
    cpu = compose(os.path.join(ROOT, "configs"), "trainer/cpu", resolve=False)["trainer"]
    assert cpu["accelerator"] == "cpu" and cpu["devices"] == 1
    ddp = compose(os.path.join(ROOT, "configs"), "trainer/ddp", resolve=False)["trainer"]
    assert ddp["devices"] == 8 and ddp["strategy"] == "ddp_find_unused_parameters_true" and ddp["sync_batchnorm"] is True and ddp["max_epochs"] == 100
    for name in ("struct_token", "text"):
        dm = compose(os.path.join(ROOT, "configs"), f"data/modalities/{name}", resolve=False)["data"]["modalities"][name]
        assert dm["batch_size"] == {"train": 16, "val": 16, "test": 64} and dm["dataset"]["seq_tokenizer"] == "facebook/esm2_t33_650M_UR50D"
    c2 = compose(os.path.join(ROOT, "configs"), "train", overrides={"model": "oneprot_text", "model.components.sequence.output_dim": 256})
    assert c2["model"]["components"]["text"]["output_dim"] == 256 and c2["model"]["components"]["struct_token"]["output_dim"] == 256


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "configs")), reason="reference checkout only exists in the build container")
def test_config_composer_reads_reference_yaml_unchanged():
    from oneprot_amd.config import compose
    c = compose(os.path.join(REF, "configs"), "model/oneprot")["model"]
    assert c["_target_"] == "src.models.oneprot_module.OneProtLitModule"
    assert c["optimizer"] == {"_target_": "torch.optim.Adam", "_partial_": True, "lr": 0.001, "weight_decay": 0.0}
    assert c["components"]["sequence"]["_target_"] == "src.models.components.sequence_encoder.SequenceEncoder"
    assert c["components"]["text"]["output_dim"] == c["components"]["sequence"]["output_dim"] == 1024        # ${..sequence.output_dim}
    assert c["loss_fn"] == "CLIP" and c["use_l1_regularization"] is True
    t = compose(os.path.join(REF, "configs"), "trainer/ddp", resolve=False)["trainer"]      # (${paths.*} needs the paths group)
    assert t["devices"] == 4 and t["strategy"] == "ddp_find_unused_parameters_true" and t["accelerator"] == "gpu"
    s = compose(os.path.join(REF, "configs"), "model/components/struct_token", resolve=False)
    assert s["model"]["components"]["struct_token"]["use_logit_scale"] is True
    # the mirrors in this repo carry the reference's keys and values (devices of ddp aside: 8 MI355X per node here, 4 GPUs there)
    for name in ("default", "cpu", "ddp_sim", "gpu"):
        ours = compose(os.path.join(ROOT, "configs"), f"trainer/{name}", resolve=False)["trainer"]
        theirs = compose(os.path.join(REF, "configs"), f"trainer/{name}", resolve=False)["trainer"]
        for k, v in theirs.items():
            assert ours[k] == v, (name, k, ours.get(k), v)
    assert compose(os.path.join(ROOT, "configs"), "model/default", resolve=False)["model"] == compose(os.path.join(REF, "configs"), "model/default", resolve=False)["model"]
    for name in ("struct_token", "text"):
        assert compose(os.path.join(ROOT, "configs"), f"data/modalities/{name}", resolve=False) == compose(os.path.join(REF, "configs"), f"data/modalities/{name}", resolve=False)


def test_combined_loader_and_synthetic_batches():
    from oneprot_amd.data import CombinedLoader, SyntheticPairs
    st = SyntheticPairs("struct_token", 4, 16, n_batches=3, ragged=True)
    tx = SyntheticPairs("text", 2, 16, mod_len=8, n_batches=2, text_vocab=120)
    steps = list(CombinedLoader({"struct_token": st, "text": tx}, "min_size"))
    assert len(steps) == 2 and set(steps[0]) == {"struct_token", "text"}                 # min_size: the shorter loader ends the epoch
    seq, mod, name, _ = steps[0]["struct_token"]
    assert seq.shape == (4, 16) and mod.shape == (4, 16) and name == "struct_token" and seq.dtype == torch.int64
    assert (seq[:, 0] == 0).all() and ((mod == 1) | ((mod >= 33) & (mod <= 52)) | (mod == 0) | (mod == 2)).all()
    assert ((seq != 1).sum(1) == (mod != 1).sum(1)).all() is not None
    tseq, tmod, tname, _ = steps[1]["text"]
    assert tmod.shape == (2, 8) and tname == "text" and tmod.max() < 120
    seqm = list(CombinedLoader({"struct_token": st, "text": tx}, "sequential"))
    assert len(seqm) == 5 and [x[2] for x in seqm] == [0, 0, 0, 1, 1]
    again = list(CombinedLoader({"struct_token": st}, "min_size"))                       # re-iterable, deterministic
    assert torch.equal(again[0]["struct_token"][0], steps[0]["struct_token"][0])
    with pytest.raises(ValueError):
        CombinedLoader({}, "max_size")


def test_checkpoint_wire_format(golden_dir, tmp_path, monkeypatch):
    monkeypatch.setenv("ONEPROT_ALLOW_RANDOM_INIT", "1")
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "1")
    import functools
    from oneprot_amd.data import load_weights_only, save_checkpoint
    from src.models.components.sequence_encoder import SequenceEncoder
    from src.models.components.struct_token_encoder import StructTokenEncoder
    from src.models.oneprot_module import OneProtLitModule
    g = torch.load(os.path.join(golden_dir, "esm_pair_hd16.pt"), weights_only=False)
    p = _tiny_dir(tmp_path)

    def make():
        seq = SequenceEncoder(p, output_dim=48, proj_type="mlp", use_lora=False, frozen=True)
        st = StructTokenEncoder(p, output_dim=48, proj_type="linear", use_logit_scale=True)
        return OneProtLitModule(components={"sequence": seq, "struct_token": st}, optimizer=functools.partial(torch.optim.Adam, lr=1e-3))
    m1 = make()
    m1.network["sequence"].load_state_dict(g["sd_seq"]); m1.network["struct_token"].load_state_dict(g["sd_st"])
    keys = set(m1.state_dict())
    # the on-disk contract of the reference: network.<modality>.{transformer.*, proj.*, norm.1.log_logit_scale}
    assert "network.struct_token.transformer.encoder.layer.0.attention.self.query.weight" in keys
    assert "network.struct_token.norm.1.log_logit_scale" in keys and "network.sequence.proj.4.weight" in keys
    assert not any(".flat" in k or ".extra." in k for k in keys)
    ck = os.path.join(str(tmp_path), "last.ckpt")
    save_checkpoint(m1, ck)
    m2 = make()
    load_weights_only(m2, ck)
    for k, v in m1.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k
    # dropout stream positions ride along outside the state dict and are restored (a resumed run continues the mask sequence)
    tr1 = m1.network["struct_token"].transformer
    tr1.enable_lora(4, 8, ["query", "key", "value"], dropout=0.1)
    tr1._lora_calls = 17
    save_checkpoint(m1, ck)
    import zlib
    assert tr1._rng_uid == zlib.crc32(b"struct_token") & 0xFFFF      # the tower id of the stream ids = the modality name, not the construction order
    assert torch.load(ck, weights_only=True)["oneprot_amd_dropout_rng"]["struct_token"] == {"_lora_seed": tr1._lora_seed, "_lora_calls": 17, "_rng_uid": tr1._rng_uid}
    m3 = make()
    tr3 = m3.network["struct_token"].transformer
    tr3.enable_lora(4, 8, ["query", "key", "value"], dropout=0.1)
    tr3._lora_seed = 1
    tr3._rng_uid = 5                                                # (a process that numbered its towers differently)
    load_weights_only(m3, ck)
    assert (tr3._lora_seed, tr3._lora_calls, tr3._rng_uid) == (tr1._lora_seed, 17, tr1._rng_uid)
    # every consumer of the counter-based generator has a stream domain of its own: purpose and tower enter the stream id
    seq_tr = m1.network["sequence"].transformer
    assert tr1._rng_uid != seq_tr._rng_uid
    assert tr1._lora_stream(0, 0, 0) != seq_tr._lora_stream(0, 0, 0)
    assert tr1._lora_stream(0, 0, 0) != tr1._rng_stream(tr1.RNG_DOMAIN_BERT, 0)
    assert tr1._lora_stream(3, 2, 1) & ((1 << 44) - 1) == (3 * tr1.n_layers + 2) * 4 + 1
    m1 = make()
    m1.network["sequence"].load_state_dict(g["sd_seq"]); m1.network["struct_token"].load_state_dict(g["sd_st"])
    save_checkpoint(m1, ck)
    # Lightning-wrapped variant with the 'model.' prefix the reference strips (src/train.py:77-79)
    sd = torch.load(ck, weights_only=True)["state_dict"]
    torch.save({"state_dict": {"model." + k: v for k, v in sd.items()}}, ck)
    load_weights_only(make(), ck)


def test_bench_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` with no launcher environment must start N ranks itself (torch.distributed.run, 127.0.0.1 rendezvous) before
    anything touches the GPU, and hand back the children's exit code."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_comm_library_exports_every_declared_symbol():
    """include/oneprot_comm.h <-> liboneprot_comm.so <-> oneprot_amd/comm.py (the RCCL wrappers of SURVEY 8b); no collective is run without a GPU."""
    from oneprot_amd import comm
    header = open(os.path.join(ROOT, "include", "oneprot_comm.h")).read()
    declared = sorted(set(re.findall(r"\b(oneprot_comm_\w+)\s*\(", header)))
    assert declared == comm.exported_symbols()
    out = subprocess.run(["nm", "-D", "--defined-only", comm.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and "oneprot_comm_" in l)
    assert exported == declared
    comm.lib()          # loads and binds every symbol


def test_gradient_wire_dtype_option(monkeypatch):
    """ONEPROT_GRAD_COMM_DTYPE / distributed.grad_comm_dtype: fp32 (the reference's DDP) by default, bf16 on request, anything else refused"""
    from oneprot_amd import distributed as D
    monkeypatch.delenv("ONEPROT_GRAD_COMM_DTYPE", raising=False)
    assert D.grad_comm_dtype() is None and D.grad_comm_dtype("fp32") is None and D.grad_comm_dtype("float32") is None
    assert D.grad_comm_dtype("bf16") is torch.bfloat16 and D.grad_comm_dtype("BFloat16") is torch.bfloat16
    monkeypatch.setenv("ONEPROT_GRAD_COMM_DTYPE", "bf16")
    assert D.grad_comm_dtype() is torch.bfloat16 and D.grad_comm_dtype("fp32") is None
    with pytest.raises(ValueError):
        D.grad_comm_dtype("fp8")


def test_bench_board_sampler_reads_hwmon_files(tmp_path):
    """bench.BoardSampler (the `board` object of the bench line: power and shader clock over the timed steps) on a fake hwmon directory: units (uW, Hz),
    the MFMA peak at the measured clock, and no thread / no result when the card has no readable node."""
    import time
    import bench
    s = bench.BoardSampler.__new__(bench.BoardSampler)
    import threading
    hw = tmp_path / "hwmon0"
    hw.mkdir()
    (hw / "power1_average").write_text("1300000000\n")
    (hw / "freq1_input").write_text("2000000000\n")
    (hw / "power1_cap").write_text("1400000000\n")
    s.hw, s.rows, s._stop, s._thread = str(hw), [], threading.Event(), None
    s.start()
    time.sleep(0.3)
    out = s.stop()
    assert out is not None and out["samples"] >= 3
    assert out["power_w_mean"] == 1300 and out["power_cap_w"] == 1400 and out["sclk_mhz_mean"] == 2000
    assert abs(out["bf16_mfma_peak_at_that_clock_tflops"] - bench.PEAK_BF16_TFLOPS * 2000 / 2400) <= 1
    none = bench.BoardSampler.__new__(bench.BoardSampler)
    none.hw, none.rows, none._stop, none._thread = None, [], threading.Event(), None
    none.start()
    assert none._thread is None and none.stop() is None
