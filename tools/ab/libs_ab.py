#!/usr/bin/env python3
"""Development tool (GPU): SEVERAL builds of liboneprot_hip.so against each other in ONE process (boxes differ by +-4 %, builds are only ever
compared inside one process): interleaved rounds, medians, outputs checked against the first library's.
usage: libs_ab.py name=path.so [name=path.so ...]     env AB_CASES=nt,nt64,gln,attn,attn64,tn (default nt,gln,attn,tn)  AB_ROUNDS=5  AB_ONLY=<substring of a case name>
       AB_FORCE_SHAPE=<oneprot_gemm_force_shape id, e.g. 40 / 41>  AB_TUNE=<first argument of oneprot_gemm_tune, e.g. 19968 = 256 * 78: main loop only>
The first library is the reference column; `name=product` stands for oneprot_amd/liboneprot_hip.so."""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from oneprot_amd import hip

libs = {}
for spec in sys.argv[1:]:
    name, _, path = spec.partition("=")
    path = hip.LIB_PATH if path in ("", "product") else os.path.abspath(path)
    h = ctypes.CDLL(path)
    for sym, (res, args) in hip._SIGS.items():
        fn = getattr(h, sym, None)
        if fn is not None:
            fn.restype, fn.argtypes = res, args
    libs[name] = h
    if os.environ.get("AB_FORCE_SHAPE"):
        h.oneprot_gemm_force_shape(int(os.environ["AB_FORCE_SHAPE"]))
    if os.environ.get("AB_TUNE"):
        h.oneprot_gemm_tune(int(os.environ["AB_TUNE"]), 0)
    if "@gln" in name and hasattr(h, "oneprot_gemm_ln_form"):      # e.g. old@gln0=product: the same library with the eight-wave fused GEMM + LN kernel
        h.oneprot_gemm_ln_form(int(name.split("@gln")[1]))          # (form | start delay of the second half of the grid in us << 8)
    if "@bwd" in name:                                             # e.g. split@bwd0=product: the attention backward path forced (0 split, 1 fused 16 waves, 2 fused 8 waves)
        h.oneprot_attn_force_bwd_path(int(name.split("@bwd")[1]))
    if "@dyn" in name:                                             # e.g. dyn@dyn=tools/ab/lib_copy.so: tiles of the persistent GEMMs drawn from the work queues of a sched workspace
        wsb = h.oneprot_sched_workspace_bytes(131072)
        out = ctypes.c_void_p()
        assert h.oneprot_alloc_uncached(ctypes.byref(out), wsb) == 0
        assert h.oneprot_sched_workspace_init(out.value, wsb, None) == 0
        torch.cuda.synchronize()
        h.oneprot_dynamic_tiles(out.value, wsb)
    if "@fwd" in name:                                             # e.g. chunked@fwd2=product: the same library with the attention forward path forced
        h.oneprot_attn_force_fwd_path(int(name.split("@fwd")[1]))
if not libs:
    sys.exit(__doc__)
groups = os.environ.get("AB_CASES", "nt,gln,attn,tn").split(",")
rounds = int(os.environ.get("AB_ROUNDS", "5"))
only = os.environ.get("AB_ONLY", "")


def call(h, sym, *args):
    rc = getattr(h, sym)(*[hip.ptr(a) if isinstance(a, torch.Tensor) or a is None else a for a in args], hip.stream())
    if rc != 0:
        raise RuntimeError(f"{sym} returned {rc}")


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


B, L, H, hd = 256, 512, 20, 32
d, f, T = 640, 2560, 256 * 512
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
cases = []      # (name, flops, make(h) -> fn, outputs: list of tensors to compare)


def nt_case(name, N, K, epi, two=True):
    A = rnd(T, K).to(torch.bfloat16); W = (rnd(N, K) * 0.05).to(torch.bfloat16); bias = rnd(N)
    cos = torch.rand(L, hd // 2, device="cuda"); sin = torch.rand(L, hd // 2, device="cuda")
    if epi == hip.EPI_QKV_ROPE:
        o = [torch.empty(B, H, L, hd, dtype=torch.bfloat16, device="cuda") for _ in range(3)]
        mk = lambda h: (lambda: call(h, "oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias, o[0], o[1], o[2], None, cos, sin, hd ** -0.5, L, H, hd))
        outs = o
    elif epi == hip.EPI_BIAS_RESID:
        res = rnd(T, N); o0 = torch.empty_like(res)
        mk = lambda h: (lambda: call(h, "oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias, o0, None, None, res, None, None, 1.0, 0, 0, 0))
        outs = [o0]
    elif epi == hip.EPI_BIAS_GELU:
        o0 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda"); o1 = torch.empty(T, N, dtype=torch.uint8, device="cuda") if two else None
        mk = lambda h: (lambda: call(h, "oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, bias, o0, o1, None, None, None, None, 1.0, 0, 0, 0))
        outs = [o0] + ([o1] if two else [])
    elif epi == hip.EPI_GELU_BWD:
        o0 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda"); aux = torch.randint(0, 256, (T, N), dtype=torch.uint8, device="cuda", generator=g)
        mk = lambda h: (lambda: call(h, "oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, None, o0, None, None, aux, None, None, 1.0, 0, 0, 0))
        outs = [o0]
    else:
        o0 = torch.empty(T, N, dtype=torch.bfloat16, device="cuda")
        mk = lambda h: (lambda: call(h, "oneprot_gemm_bf16_nt", A, W, T, N, K, K, K, epi, None, o0, None, None, None, None, None, 1.0, 0, 0, 0))
        outs = [o0]
    cases.append((name, 2.0 * T * N * K, mk, outs))


if "nt" in groups:
    nt_case("nt qkv_fwd    N1920 K640  rope", 3 * d, d, hip.EPI_QKV_ROPE)
    nt_case("nt ffn1_fwd   N2560 K640  gelu only", f, d, hip.EPI_BIAS_GELU, two=False)
    nt_case("nt ffn1_fwd   N2560 K640  gelu+gelu'", f, d, hip.EPI_BIAS_GELU)
    nt_case("nt ffn2_fwd   N640  K2560 resid", d, f, hip.EPI_BIAS_RESID)
    nt_case("nt out_fwd    N640  K640  resid", d, d, hip.EPI_BIAS_RESID)
    nt_case("nt plain      N2560 K640  bf16", f, d, hip.EPI_BF16)
    nt_case("nt ffn2_dgrad N2560 K640  gelu'", f, d, hip.EPI_GELU_BWD)
    nt_case("nt ffn1_dgrad N640  K2560 bf16", d, f, hip.EPI_BF16)
    nt_case("nt out_dgrad  N640  K640  bf16", d, d, hip.EPI_BF16)
    nt_case("nt qkv_dgrad  N640  K1920 bf16", d, 3 * d, hip.EPI_BF16)

if "nt64" in groups:      # head_dim 64 QKV launches: BERT-base (T = 65536, d = 768) and ESM-2-650M at 128 pairs (T = 65536, d = 1280)
    for nm, B_, L_, H_, hd_ in (("nt64 BERT-base QKV d768", 256, 256, 12, 64), ("nt64 ESM-2-650M QKV d1280", 128, 512, 20, 64)):
        d_, T_ = H_ * hd_, B_ * L_
        A = rnd(T_, d_).to(torch.bfloat16); W = (rnd(3 * d_, d_) * 0.05).to(torch.bfloat16); bias = rnd(3 * d_)
        cos = torch.rand(L_, hd_ // 2, device="cuda"); sin = torch.rand(L_, hd_ // 2, device="cuda")
        o = [torch.empty(B_, H_, L_, hd_, dtype=torch.bfloat16, device="cuda") for _ in range(3)]
        mk = lambda h, A=A, W=W, bias=bias, cos=cos, sin=sin, o=o, T_=T_, d_=d_, L_=L_, H_=H_, hd_=hd_: (lambda: call(h, "oneprot_gemm_bf16_nt", A, W, T_, 3 * d_, d_, d_, d_, hip.EPI_QKV_ROPE, bias, o[0], o[1], o[2], None, cos, sin, hd_ ** -0.5, L_, H_, hd_))
        cases.append((nm, 2.0 * T_ * 3 * d_ * d_, mk, o))

if "gln" in groups:
    for nm, K in (("gln out-proj K=640", 640), ("gln FFN-2   K=2560", 2560)):
        A = rnd(T, K).to(torch.bfloat16); W = (rnd(d, K) * 0.05).to(torch.bfloat16); bias = rnd(d); gamma = rnd(d); beta = rnd(d)
        x = rnd(T, d); xo = torch.empty_like(x); hh = torch.empty(T, d, dtype=torch.bfloat16, device="cuda"); mean = torch.empty(T, device="cuda"); rstd = torch.empty(T, device="cuda")
        packed = {}

        def mk(h, A=A, W=W, K=K, bias=bias, gamma=gamma, beta=beta, x=x, xo=xo, hh=hh, mean=mean, rstd=rstd, packed=packed):
            Wp = torch.empty(d * K, dtype=torch.bfloat16, device="cuda")
            call(h, "oneprot_gemm_ln_pack_weight", W, Wp, d, K)      # every library packs for its own kernel
            packed[id(h)] = Wp
            return lambda: call(h, "oneprot_gemm_bf16_nt_resid_ln", A, Wp, T, d, K, K, bias, x, xo, gamma, beta, 1e-5, hh, mean, rstd)
        cases.append((nm, 2.0 * T * d * K, mk, [xo, hh, mean, rstd]))

if "attn" in groups:
    mkq = lambda: (rnd(B, H, L, hd) * 0.7).to(torch.bfloat16)
    q, k, v = mkq(), mkq(), mkq()
    lens = torch.randint(L // 2, L + 1, (B,), device="cuda", generator=g)
    ragged = torch.where(torch.arange(L, device="cuda")[None, :] < lens[:, None], 0.0, float("-inf")).float().contiguous()
    nopad = torch.zeros_like(ragged)
    ctx = torch.empty(B * L, H * hd, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, L, device="cuda")
    dctx = (rnd(B * L, H * hd) * 0.1).to(torch.bfloat16)
    dqkv = torch.empty(B * L, 3 * H * hd, dtype=torch.bfloat16, device="cuda")
    cos = torch.rand(L, hd // 2, device="cuda"); sin = torch.rand(L, hd // 2, device="cuda")
    fl = 4.0 * B * H * L * L * hd
    for nm, kb in (("no padding", nopad), ("ragged", ragged)):
        cases.append((f"attn fwd {nm}", fl, (lambda h, kb=kb: (lambda: call(h, "oneprot_attn_fwd", q, k, v, kb, ctx, lse, B, H, L, hd))), [ctx, lse]))

        def mkb(h, kb=kb):
            ws = torch.empty(h.oneprot_attn_bwd_workspace(B, H, L), dtype=torch.uint8, device="cuda")
            call(h, "oneprot_attn_fwd", q, k, v, kb, ctx, lse, B, H, L, hd)
            return lambda: call(h, "oneprot_attn_bwd", q, k, v, kb, ctx, dctx, lse, cos, sin, hd ** -0.5, dqkv, ws, B, H, L, hd)
        cases.append((f"attn bwd {nm}", 2.5 * fl, mkb, [dqkv]))

if "attn64" in groups:      # head_dim 64 forwards: ESM-2-650M at 128 pairs (L = 512) and BERT-base (L = 256)
    for nm, B_, H_, L_ in (("attn64 fwd 650M  B128 H20 L512", 128, 20, 512), ("attn64 fwd BERT  B256 H12 L256", 256, 12, 256), ("attn64 fwd BERT  B256 H12 L512", 256, 12, 512)):
        q6, k6, v6 = [(rnd(B_, H_, L_, 64) * 0.7).to(torch.bfloat16) for _ in range(3)]
        lens6 = torch.randint(L_ // 2, L_ + 1, (B_,), device="cuda", generator=g)
        kb6 = torch.where(torch.arange(L_, device="cuda")[None, :] < lens6[:, None], 0.0, float("-inf")).float().contiguous()
        ctx6 = torch.empty(B_ * L_, H_ * 64, dtype=torch.bfloat16, device="cuda"); lse6 = torch.empty(B_, H_, L_, device="cuda")
        for tag, kbx in (("no padding", torch.zeros_like(kb6)), ("ragged", kb6)):
            cases.append((f"{nm} {tag}", 4.0 * B_ * H_ * L_ * L_ * 64,
                          (lambda h, q6=q6, k6=k6, v6=v6, kbx=kbx, ctx6=ctx6, lse6=lse6, B_=B_, H_=H_, L_=L_: (lambda: call(h, "oneprot_attn_fwd", q6, k6, v6, kbx, ctx6, lse6, B_, H_, L_, 64))),
                          [ctx6, lse6]))

if "ln8" in groups:      # FFN-2 + residual + the next LayerNorm: the pair of launches against oneprot_gemm_bf16_nt_resid_ln8 (150M and 650M shapes)
    for nm, T_, N_, K_ in (("ln8 150M FFN-2 N640 K2560", T, 640, 2560), ("ln8 650M FFN-2 N1280 K5120", 65536, 1280, 5120), ("ln8 650M out   N1280 K1280", 65536, 1280, 1280), ("ln8 150M out   N640  K640 ", T, 640, 640)):
        A = rnd(T_, K_).to(torch.bfloat16); W = (rnd(N_, K_) * 0.05).to(torch.bfloat16); bias = rnd(N_); gamma = rnd(N_); beta = rnd(N_)
        res = rnd(T_, N_); xo = torch.empty_like(res); hh = torch.empty(T_, N_, dtype=torch.bfloat16, device="cuda"); st = torch.empty(2, T_, device="cuda")

        def mk_pair(h, A=A, W=W, bias=bias, gamma=gamma, beta=beta, res=res, xo=xo, hh=hh, st=st, T_=T_, N_=N_, K_=K_):
            def f():
                call(h, "oneprot_gemm_bf16_nt", A, W, T_, N_, K_, K_, K_, hip.EPI_BIAS_RESID, bias, xo, None, None, res, None, None, 1.0, 0, 0, 0)
                call(h, "oneprot_layernorm_fwd", xo, 0, gamma, beta, hh, None, st[0], st[1], T_, N_, 1e-5)
            return f
        mk_fused = lambda h, A=A, W=W, bias=bias, gamma=gamma, beta=beta, res=res, xo=xo, hh=hh, st=st, T_=T_, N_=N_, K_=K_: (
            lambda: call(h, "oneprot_gemm_bf16_nt_resid_ln8", A, W, T_, N_, K_, K_, K_, bias, res, xo, gamma, beta, 1e-5, hh, st))
        cases.append((nm + " pair ", 2.0 * T_ * N_ * K_, mk_pair, [xo, hh, st]))
        cases.append((nm + " fused", 2.0 * T_ * N_ * K_, mk_fused, [xo, hh, st]))

if "tn" in groups:
    for nm, N, K in (("tn qkv  dW[1920,640]", 3 * d, d), ("tn out  dW[640,640]", d, d), ("tn ffn1 dW[2560,640]", f, d), ("tn ffn2 dW[640,2560]", d, f)):
        dY = rnd(T, N).to(torch.bfloat16); X = rnd(T, K).to(torch.bfloat16)
        dW, db = torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda")

        def mk(h, dY=dY, X=X, N=N, K=K, dW=dW, db=db):
            ws = torch.empty(h.oneprot_gemm_bf16_tn_workspace(N, K), dtype=torch.uint8, device="cuda")
            return lambda: call(h, "oneprot_gemm_bf16_tn", dY, X, T, N, K, N, K, dW, db, ws, ws.numel(), 0)
        cases.append((nm, 2.0 * T * N * K, mk, [dW, db]))

print(f"median us per launch over {rounds} interleaved rounds (TFLOP/s) | max abs difference of the outputs from `{next(iter(libs))}`", flush=True)
for name, fl, mk, outs in cases:
    if only and only not in name:
        continue
    fns = {n: mk(h) for n, h in libs.items()}
    res = {n: [] for n in libs}
    for r in range(rounds):
        for n in (list(libs) if r % 2 == 0 else list(libs)[::-1]):
            res[n].append(timeit(fns[n]))
    ref, diffs = None, {}
    for n in libs:
        fns[n](); torch.cuda.synchronize()
        cur = [o.float().clone() for o in outs]
        if ref is None:
            ref = cur
        else:
            diffs[n] = max(float((a - b).abs().nan_to_num(nan=1e30).max()) for a, b in zip(cur, ref))
    row = "  ".join(f"{n}:{statistics.median(t):7.1f} ({fl / statistics.median(t) / 1e6:4.0f})" for n, t in res.items())
    print(f"{name:36s} {row}   | " + " ".join(f"{n}:{v:.2e}" for n, v in diffs.items()), flush=True)
