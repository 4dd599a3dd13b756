"""Per-kernel parity: every C-ABI entry point vs a plain fp32 torch restatement of the same op (and the oracle's
functions where the op is composite).  Runs on the MI355X only (-m gpu); all calls go through the C ABI.

Tolerances: fp32 kernels 1e-5-ish; bf16-operand MFMA kernels are compared against an fp32 reference computed from the
SAME bf16-rounded inputs, so the only difference is accumulation order / output rounding (bf16 outputs: 2^-8 rel)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oneprot_amd import hip  # noqa: E402
from oracle import oneprot_oracle as O  # noqa: E402

DEV = "cuda"


def bf(x):
    return x.to(torch.bfloat16)


def rel_err(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-20)).item()


def assert_close(a, b, rtol, atol, msg=""):
    a, b = a.float(), b.float()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = (err > tol)
    assert not bad.any(), f"{msg}: {int(bad.sum())}/{bad.numel()} off, max err {err.max().item():.3e} (ref max {b.abs().max().item():.3e})"


def ws(nbytes):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=DEV)


# ------------------------------------------------------------------------------------------------------
def test_library_loads():
    assert hip.query("oneprot_abi_version") == hip.ABI_VERSION == 7


@pytest.mark.parametrize("B,L,d,vocab", [(3, 17, 64, 33), (4, 130, 640, 54)])
def test_embed_fwd_bwd(B, L, d, vocab):
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(4, vocab - 1, (B, L), generator=g)
    ids[0, L - 5:] = 1
    ids[1, 2] = 32
    ids[1, 4] = 32
    ids[:, 0] = 0
    W = torch.randn(vocab, d, generator=g)
    Wd, idd = W.to(DEV), ids.to(DEV)
    x = torch.empty(B, L, d, device=DEV)
    rs = torch.empty(B, device=DEV)
    hip.call("oneprot_esm_embed_fwd", idd, Wd, x, rs, B, L, d, vocab, 1, 32, 1)
    Wr = W.clone().requires_grad_(True)
    ref = O.esm_embeddings(ids, (ids != 1).long(), Wr, 32)
    assert_close(x.cpu(), ref.detach(), 1e-6, 1e-6, "embed fwd")
    dx = torch.randn(B, L, d, generator=g)
    ref.backward(dx)
    dW = torch.full((vocab, d), 7.0, device=DEV)
    w = ws(hip.query("oneprot_esm_embed_bwd_workspace", B * L, d, vocab))
    hip.call("oneprot_esm_embed_bwd", idd, dx.to(DEV), rs, dW, w, B, L, d, vocab, 1, 32, 1, 0)
    assert_close(dW.cpu(), Wr.grad, 1e-5, 1e-5, "embed bwd")


@pytest.mark.parametrize("T,d", [(37, 64), (1000, 640), (130, 1280), (64, 48)])
def test_layernorm_fwd_bwd(T, d):
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(T, d, generator=g) * 2 + 0.5).to(DEV)
    gamma = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(d, generator=g)).to(DEV)
    yb = torch.empty(T, d, dtype=torch.bfloat16, device=DEV)
    yf = torch.empty(T, d, device=DEV)
    mean = torch.empty(T, device=DEV)
    rstd = torch.empty(T, device=DEV)
    hip.call("oneprot_layernorm_fwd", x, 0, gamma, beta, yb, yf, mean, rstd, T, d, 1e-5)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (d,), gr, br, 1e-5)
    assert_close(yf, ref.detach(), 1e-5, 1e-5, "ln fwd f32")
    assert_close(yb, ref.detach(), 2 ** -8, 1e-6, "ln fwd bf16")
    # backward, fp32 dy, accumulate into an existing residual gradient
    dy = torch.randn(T, d, generator=g).to(DEV)
    add = torch.randn(T, d, generator=g).to(DEV)
    ref.backward(dy)
    dx = torch.empty(T, d, device=DEV)
    dg = torch.zeros(d, device=DEV)
    db = torch.zeros(d, device=DEV)
    w = ws(hip.query("oneprot_layernorm_bwd_workspace", d))
    hip.call("oneprot_layernorm_bwd", dy, 1, None, 0, x, 0, gamma, mean, rstd, add, dx, None, dg, db, w, T, d, 0)
    assert_close(dx, xr.grad + add, 1e-4, 1e-4, "ln bwd dx")
    assert_close(dg, gr.grad, 1e-4, 1e-3, "ln bwd dgamma")
    assert_close(db, br.grad, 1e-4, 1e-3, "ln bwd dbeta")
    # bf16 dy, in place on the residual gradient, accumulate param grads
    dyb = bf(dy)
    dx2 = add.clone()
    dx2b = torch.empty(T, d, dtype=torch.bfloat16, device=DEV)
    hip.call("oneprot_layernorm_bwd", dyb, 0, None, 0, x, 0, gamma, mean, rstd, dx2, dx2, dx2b, dg, db, w, T, d, 1)
    assert torch.equal(dx2b, dx2.to(torch.bfloat16))
    xr2 = x.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr2, (d,), gamma, beta, 1e-5).backward(dyb.float())
    assert_close(dx2, xr2.grad + add, 1e-4, 1e-4, "ln bwd dx (bf16 dy, in place)")


@pytest.mark.parametrize("mode", [0, 1])
def test_lnpool_fwd_and_pooled_bwd(mode):
    B, L, d = 5, 37, 128
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(4, 24, (B, L), generator=g)
    for b, n in enumerate([37, 20, 5, 37, 1]):
        ids[b, n:] = 1
    x = torch.randn(B, L, d, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    xd, idd = x.to(DEV), ids.to(DEV)
    pooled = torch.empty(B, d, device=DEV)
    mean, rstd, wrow = (torch.empty(B * L, device=DEV) for _ in range(3))
    hid = torch.empty(B, L, d, device=DEV)
    hip.call("oneprot_lnpool_fwd", xd, idd, 1, gamma.to(DEV), beta.to(DEV), pooled, mean, rstd, wrow, None, hid, B, L, d, 1e-5, mode)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    h = O.layer_norm(xr, gr, br, 1e-5)
    ref = O.mean_pool(h, (ids != 1).long()) if mode == 0 else O.cls_pool(h)
    assert_close(hid.cpu(), h.detach(), 1e-5, 1e-5, "hidden")
    assert_close(pooled.cpu(), ref.detach(), 1e-5, 1e-5, "pooled")
    dp = torch.randn(B, d, generator=g)
    ref.backward(dp)
    dx = torch.empty(B * L, d, device=DEV)
    dg, db = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    w = ws(hip.query("oneprot_layernorm_bwd_workspace", d))
    hip.call("oneprot_layernorm_bwd", dp.to(DEV), 2, wrow, L, xd, 0, gamma.to(DEV), mean, rstd, None, dx, None, dg, db, w, B * L, d, 0)
    assert_close(dx.cpu().view(B, L, d), xr.grad, 1e-4, 1e-5, "pooled bwd dx")
    assert_close(dg.cpu(), gr.grad, 1e-4, 1e-4, "pooled bwd dgamma")
    assert_close(db.cpu(), br.grad, 1e-4, 1e-4, "pooled bwd dbeta")


# ------------------------------------------------------------------------------------------------------
def _gemm_inputs(M, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    W = bf(torch.randn(N, K, generator=g) * 0.1).to(DEV)
    bias = (torch.randn(N, generator=g) * 0.5).to(DEV)
    return A, W, bias


@pytest.fixture(params=[-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 17, 19, 20, 32, 40, 41, 42], ids=["auto", "128x128", "256x128", "256x256", "128x128bk64", "256x256bk64", "256x256nopipe", "4w128x128bk64",
                                                                                     "4w128x128bk32", "regstaged256x256", "direct256x128", "direct128x128bk64", "direct256x256bk64", "pingpong",
                                                                                     "8phase256x256", "8phase256x320", "8phase256x256merged"])
def gemm_shape(request):
    hip.query("oneprot_gemm_force_shape", request.param)
    yield request.param
    hip.query("oneprot_gemm_force_shape", -1)


@pytest.mark.parametrize("M,N,K", [(300, 136, 72), (128, 128, 64), (1024, 640, 640), (257, 2560, 640), (515, 640, 2560), (144, 64, 160), (2304, 1920, 640), (8192, 512, 160),
                                   (1024, 1280, 640), (512, 768, 2560), (8192, 640, 640), (4096, 2560, 128), (16384, 128, 192),
                                   (4096, 256, 512), (2048, 384, 576), (256, 128, 1024), (768, 256, 640), (16384, 2560, 640), (2048, 1920, 640), (66304, 640, 256), (256, 320, 128)])
# (1024, 1280, 640) onwards are whole tiles in every block shape (direct-store and ping-pong forms).  For the persistent ping-pong kernel (K >= 512):
# several tiles per work-group, K = 512 (16 K-steps: every slot carries an epilogue chunk) and 576, a single tile, and XCDs without any tile.
# For the persistent 8-phase kernels (whole 256 x 256 / 256 x 320 tiles, K % 128 == 0): (16384, 2560, 640) gives every work-group two or three tiles
# (the operand stream crosses tile boundaries), (66304, 640, 256) = 259 row panels x 2 leaves some work-groups with three tiles and others with two,
# (256, 320, 128) is a single tile of two K-tiles (the shortest stream); shapes that are not whole tiles fall through to the per-tile kernels.
def test_gemm_nt_epilogues(M, N, K, gemm_shape):
    A, W, bias = _gemm_inputs(M, N, K, 3)
    ref = A.float() @ W.float().t()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    hip.call("oneprot_gemm_bf16_nt", A, W, M, N, K, K, K, hip.EPI_BF16, bias, out, None, None, None, None, None, 1.0, 0, 0, 0)
    assert_close(out, ref + bias, 2 ** -7, 2e-2, "EPI_BF16")
    outf = torch.empty(M, N, device=DEV)
    hip.call("oneprot_gemm_bf16_nt", A, W, M, N, K, K, K, hip.EPI_F32, None, outf, None, None, None, None, None, 1.0, 0, 0, 0)
    assert_close(outf, ref, 1e-4, 1e-3 * math.sqrt(K / 64), "EPI_F32")
    # bias + gelu (+z)
    u = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    z = torch.empty(M, N, dtype=torch.uint8, device=DEV)                 # gelu'(z) as one-byte codes c = rint(192 g' + 25) (include/oneprot_hip.h)
    hip.call("oneprot_gemm_bf16_nt", A, W, M, N, K, K, K, hip.EPI_BIAS_GELU, bias, u, z, None, None, None, None, 1.0, 0, 0, 0)
    zr = (ref + bias).requires_grad_(True)
    gz = torch.nn.functional.gelu(zr)
    gz.sum().backward()
    assert_close(u, gz.detach(), 2 ** -7, 2e-2, "gelu(z)")
    zdec = (z.float() - 25.0) / 192.0
    # half a code step (1/384) + the slope of gelu' (<= 0.8) times the bf16-operand error of z itself; exact at the saturated ends
    assert float((zdec - zr.grad).abs().max()) < 1 / 384 + 2e-3, "gelu'(z) codes"
    assert float((zdec - zr.grad).abs().mean()) < 1.6e-3
    sat = zr.detach().abs() > 6.0
    if bool(sat.any()):
        assert torch.equal(zdec[sat], (zr.detach()[sat] > 0).float())
    u2 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)            # forward-only form (frozen tower): no derivative output
    hip.call("oneprot_gemm_bf16_nt", A, W, M, N, K, K, K, hip.EPI_BIAS_GELU, bias, u2, None, None, None, None, None, 1.0, 0, 0, 0)
    assert_close(u2, gz.detach(), 2 ** -7, 2e-2, "gelu(z), no derivative")
    # bias + fp32 residual, in place
    g = torch.Generator().manual_seed(4)
    resid = torch.randn(M, N, generator=g).to(DEV)
    x = resid.clone()
    hip.call("oneprot_gemm_bf16_nt", A, W, M, N, K, K, K, hip.EPI_BIAS_RESID, bias, x, None, None, x, None, None, 1.0, 0, 0, 0)
    assert_close(x, ref + bias + resid, 1e-4, 1e-3 * math.sqrt(K / 64), "bias+resid")
    x2, x2h = torch.empty(M, N, device=DEV), torch.empty(M, N, dtype=torch.bfloat16, device=DEV)      # out of place, with the bf16 copy
    hip.call("oneprot_gemm_bf16_nt", A, W, M, N, K, K, K, hip.EPI_BIAS_RESID, bias, x2, x2h, None, resid, None, None, 1.0, 0, 0, 0)
    assert_close(x2, ref + bias + resid, 1e-4, 1e-3 * math.sqrt(K / 64), "bias+resid out of place")
    assert_close(x2h, ref + bias + resid, 2 ** -7, 2e-2, "bias+resid bf16 copy")
    # gelu backward epilogue
    dz = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    hip.call("oneprot_gemm_bf16_nt", A, W, M, N, K, K, K, hip.EPI_GELU_BWD, None, dz, None, None, z, None, None, 1.0, 0, 0, 0)
    assert_close(dz, ref * zdec, 2 ** -6, 3e-2, "gelu bwd")


@pytest.mark.parametrize("form", [1, 0])
@pytest.mark.parametrize("M,K,inplace", [(128, 64, False), (256, 640, True), (1024, 2560, False), (33280, 640, True), (65536 + 128, 128, False), (98304 + 384, 192, True)])
def test_gemm_resid_layernorm_fused(M, K, inplace, form):
    """Full-row N = 640 GEMM with bias + residual AND the following LayerNorm in the epilogue (oneprot_gemm_bf16_nt_resid_ln) against fp32 torch:
    x (fp32), h = LN(x) (bf16), mean, rstd, both kernel forms (1: four waves on 64-row tiles, two work-groups per CU -- the default; 0: eight waves on
    128-row tiles).  33280 rows = 260 / 520 tiles (work-groups with one and with two tiles: the operand stream crosses a tile boundary behind an
    epilogue), 65664 rows = 513 / 1026 tiles (two and three), 98688 rows = 771 / 1542 (three and four), K = 64 is a single K-tile, K = 128 / 192 streams
    of a few weight units only (requests run out inside the first tiles), in-place residual as the frozen tower runs it."""
    prev = hip.query("oneprot_gemm_ln_form_get")
    hip.query("oneprot_gemm_ln_form", form)
    try:
        _gemm_resid_layernorm_fused(M, K, inplace)
    finally:
        hip.query("oneprot_gemm_ln_form", prev)      # (0, the eight-wave form, is the library default)


@pytest.mark.parametrize("M,N,K,has_bias,inplace,stats", [(24576, 640, 256, True, False, True), (24832, 640, 2560, True, True, False), (49152, 320, 128, False, False, True),
                                                          (12288, 1280, 384, True, True, True), (131072, 640, 2560, True, True, True), (33024, 640, 640, True, False, True), (2048, 640, 2560, True, False, True),
                                                          (256, 1280, 256, True, False, True), (768, 320, 128, False, True, True)])
def test_gemm_resid_layernorm_across_work_groups(M, N, K, has_bias, inplace, stats):
    """oneprot_gemm_bf16_nt_resid_ln8 -- the 8-phase GEMM whose epilogue finishes the row statistics across the work-groups of a row panel (FFN-2 + the next
    layer's LayerNorm in one launch) -- against the pair it replaces, oneprot_gemm_bf16_nt(BIAS_RESID) + oneprot_layernorm_fwd: x bit for bit (same
    arithmetic), h / mean / rstd to the rounding of another summation order.  N = 320 / 640 / 1280 = one, two and four column tiles per row panel (2 / 4 / 8
    partial statistics per row); 24832 / 33024 rows = 97 / 129 panels (work-groups with different tile counts; the neighbours still meet); the cfg-2 shape;
    twice in a row (the arrival counters of consecutive launches live in different sets) and the bounded waits never ran out."""
    big = (M // 256) * (N // 320) >= 192
    assert hip.query("oneprot_gemm_resid_ln8_eligible", M, N, K) == (2 if big else 1)
    assert hip.query("oneprot_gemm_resid_ln8_eligible", M + 64, N, K) == 0 and hip.query("oneprot_gemm_resid_ln8_eligible", M, 960, K) == 0
    g = torch.Generator().manual_seed(77 + N + K)
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    W = bf(torch.randn(N, K, generator=g) * 0.1).to(DEV)
    bias = (torch.randn(N, generator=g) * 0.5).to(DEV) if has_bias else None
    gamma, beta = (1 + 0.2 * torch.randn(N, generator=g)).to(DEV), (0.3 * torch.randn(N, generator=g)).to(DEV)
    resid = (torch.randn(M, N, generator=g) * 2.0 + 0.3).to(DEV)
    resid[: M // 2] += 40.0                                       # rows with a mean far from zero: what a one-pass variance would lose
    x_ref = torch.empty(M, N, device=DEV)
    hip.call("oneprot_gemm_bf16_nt", A, W, M, N, K, K, K, hip.EPI_BIAS_RESID, bias, x_ref, None, None, resid, None, None, 1.0, 0, 0, 0)
    h_ref = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    m_ref, r_ref = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    hip.call("oneprot_layernorm_fwd", x_ref, 0, gamma, beta, h_ref, None, m_ref, r_ref, M, N, 1e-5)
    for rep in range(2):
        x = resid.clone() if inplace else torch.full((M, N), float("nan"), device=DEV)
        h = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        st = torch.full((2, M), float("nan"), device=DEV) if stats else None
        hip.call("oneprot_gemm_bf16_nt_resid_ln8", A, W, M, N, K, K, K, bias, x if inplace else resid, x, gamma, beta, 1e-5, h, st, *hip.sched_workspace(M))
        if big:
            assert torch.equal(x, x_ref), f"x_out, launch {rep}"
        else:                                                    # (fewer than 192 tiles: the unfused GEMM runs on another kernel with another summation order)
            assert_close(x, x_ref, 2e-5, 2e-4, f"x_out, launch {rep}")
        if stats:
            assert_close(st[0], m_ref, 1e-5, 1e-5 if big else 1e-4, "mean")
            assert_close(st[1], r_ref, 2e-5 if big else 2e-4, 0.0, "rstd")
        assert torch.isfinite(h.float()).all()
        assert_close(h.float(), h_ref.float(), 2 ** -7, 2e-3, f"h, launch {rep}")
        assert (h.float() - h_ref.float()).abs().gt(1e-6).float().mean() < 0.02      # a bf16 ulp here and there, not a different function
        # ... and directly against fp32 torch: two-pass LayerNorm of A W^T + b + r (hf modeling_esm.py:442-463, then :429)
        if M <= 33024:
            xr = A.float() @ W.float().t() + resid + (bias if has_bias else 0.0)
            mu = xr.mean(-1, keepdim=True)
            var = ((xr - mu) ** 2).mean(-1, keepdim=True)
            hr = (xr - mu) * torch.rsqrt(var + 1e-5) * gamma + beta
            assert_close(x, xr, 2e-5, 3e-4, "x_out vs fp32 torch")
            assert_close(h.float(), hr, 2 ** -7, 4e-3, "h vs fp32 torch two-pass LayerNorm")
            if stats:
                assert_close(st[0], mu[:, 0], 1e-5, 1e-4, "mean vs torch")
                assert_close(st[1], torch.rsqrt(var + 1e-5)[:, 0], 1e-4, 0.0, "rstd vs torch")
    assert hip.sched_error() == 0


@pytest.mark.parametrize("N,K", [(640, 2560), (1280, 256)])
def test_gemm_resid_layernorm_wait_that_runs_out_is_loud(N, K):
    """Failure semantics of oneprot_gemm_bf16_nt_resid_ln8 (include/oneprot_hip.h): with the poll bound at zero every wait that does not find its partners'
    partial statistics at the FIRST look gives up.  Then (i) the sticky error word of the sched workspace is set, (ii) every row of h / mean / rstd is either
    right or NaN -- never a stale or half-combined statistic --, (iii) x_out (which does not depend on the exchange) is right everywhere, (iv)
    oneprot_clip_coef, handed the workspace, turns the step's gradient norm and clip coefficient into NaN on the device, (v) after
    oneprot_gemm_resid_ln8_error_clear a launch with the normal bound is clean again."""
    M = 32768
    g = torch.Generator().manual_seed(5 + N)
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    W = bf(torch.randn(N, K, generator=g) * 0.1).to(DEV)
    bias = (torch.randn(N, generator=g) * 0.5).to(DEV)
    gamma, beta = (1 + 0.2 * torch.randn(N, generator=g)).to(DEV), (0.3 * torch.randn(N, generator=g)).to(DEV)
    resid = (torch.randn(M, N, generator=g) * 2.0 + 0.3).to(DEV)
    ws = hip.sched_workspace(M)
    assert hip.sched_error() == 0

    def run():
        x = torch.full((M, N), float("nan"), device=DEV)
        h = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        st = torch.full((2, M), float("nan"), device=DEV)
        hip.call("oneprot_gemm_bf16_nt_resid_ln8", A, W, M, N, K, K, K, bias, resid, x, gamma, beta, 1e-5, h, st, *ws)
        torch.cuda.synchronize()
        return x, h, st
    x0, h0, st0 = run()
    assert hip.sched_error() == 0 and torch.isfinite(h0.float()).all()
    hip.query("oneprot_gemm_resid_ln8_poll_bound", 0)
    try:
        hit = False
        for _ in range(8):                                       # (whether a first look misses is a matter of timing: a few launches always produce one)
            x1, h1, st1 = run()
            if hip.sched_error() != 0:
                hit = True
                break
        assert hit, "no wait ran out with the poll bound at zero"
    finally:
        hip.query("oneprot_gemm_resid_ln8_poll_bound", -1)
    assert torch.equal(x1, x0)
    bad = torch.isnan(h1.float())
    assert bad.any()                                              # NaN rows exist ...
    ok = ~bad
    assert torch.equal(h1[ok], h0[ok])                           # everything that is not NaN is what the clean launch wrote
    sbad = torch.isnan(st1)
    assert torch.equal(st1[~sbad], st0[~sbad])
    # the step's gradient norm on the device
    ss = torch.full((1,), 4.0, device=DEV)
    coef, nrm = torch.empty(1, device=DEV), torch.empty(1, device=DEV)
    hip.call("oneprot_clip_coef", ss, 1.0, coef, nrm, ws[0])
    assert torch.isnan(coef).all() and torch.isnan(nrm).all()
    hip.call("oneprot_clip_coef", ss, 1.0, coef, nrm, None)
    assert abs(nrm.item() - 2.0) < 1e-6 and abs(coef.item() - 0.5) < 1e-5
    hip.sched_error_clear()
    x2, h2, st2 = run()
    assert hip.sched_error() == 0 and torch.equal(h2, h0) and torch.equal(st2, st0)
    hip.call("oneprot_clip_coef", ss, 1.0, coef, nrm, ws[0])
    assert abs(nrm.item() - 2.0) < 1e-6


def test_persistent_gemm_tiles_from_the_work_queue_bit_identical():
    """oneprot_dynamic_tiles: the 8-phase GEMMs draw their tiles from the per-XCD queue heads of the sched workspace instead of walking static lists.  Which CU
    computes a tile changes no summation order: every epilogue bit for bit the same as with static lists, launch after launch (the last work-group to leave
    resets the queues on the device), also when the launches are captured into a graph and replayed (nothing on the host advances)."""
    ws = hip.sched_workspace(131072)
    g = torch.Generator().manual_seed(99)
    M, d, f = 98560, 640, 2560                              # 385 row panels: three to seven tiles per work-group, an odd share per XCD
    A = bf(torch.randn(M, d, generator=g)).to(DEV)
    W1 = bf(torch.randn(f, d, generator=g) * 0.1).to(DEV)
    b1 = (torch.randn(f, generator=g) * 0.5).to(DEV)
    U = bf(torch.randn(M, f, generator=g)).to(DEV)
    W2 = bf(torch.randn(d, f, generator=g) * 0.05).to(DEV)
    b2 = (torch.randn(d, generator=g) * 0.5).to(DEV)
    resid = torch.randn(M, d, generator=g).to(DEV)
    gamma, beta = (1 + 0.2 * torch.randn(d, generator=g)).to(DEV), (0.3 * torch.randn(d, generator=g)).to(DEV)

    def run_all():
        u = torch.empty(M, f, dtype=torch.bfloat16, device=DEV)
        z = torch.empty(M, f, dtype=torch.uint8, device=DEV)
        hip.call("oneprot_gemm_bf16_nt", A, W1, M, f, d, d, d, hip.EPI_BIAS_GELU, b1, u, z, None, None, None, None, 1.0, 0, 0, 0)
        x = torch.empty(M, d, device=DEV)
        hip.call("oneprot_gemm_bf16_nt", U, W2, M, d, f, f, f, hip.EPI_BIAS_RESID, b2, x, None, None, resid, None, None, 1.0, 0, 0, 0)
        x8 = torch.empty(M, d, device=DEV)
        h8 = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
        st = torch.empty(2, M, device=DEV)
        hip.call("oneprot_gemm_bf16_nt_resid_ln8", U, W2, M, d, f, f, f, b2, resid, x8, gamma, beta, 1e-5, h8, st, *ws)
        dg = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
        hip.call("oneprot_gemm_bf16_nt", U, W2, M, d, f, f, f, hip.EPI_BF16, None, dg, None, None, None, None, None, 1.0, 0, 0, 0)
        return u, z, x, x8, h8, st, dg
    try:
        hip.query("oneprot_dynamic_tiles", None, 0)
        ref = run_all()
        hip.query("oneprot_dynamic_tiles", ws[0], ws[1])
        for rep in range(3):
            got = run_all()
            for a, b, name in zip(got, ref, ("gelu", "gelu'", "resid", "x ln8", "h ln8", "stats", "bf16")):
                assert torch.equal(a, b), (name, rep)
    finally:
        hip.query("oneprot_dynamic_tiles", ws[0] if hip.dynamic_tiles_wanted() else None, ws[1])
    assert hip.sched_error() == 0


def test_work_queue_that_does_not_start_at_zero_is_loud():
    """A queue head that is not zero when a launch starts (a previous launch on the workspace that never finished, a second stream sharing it) would leave output
    tiles silently unwritten.  The last work-group to leave compares the tiles the launch computed with the tiles it was given: the sticky error word becomes 2,
    oneprot_clip_coef poisons the step, the heads are reset and the next launch is clean."""
    ws = hip.sched_workspace(131072)
    g = torch.Generator().manual_seed(43)
    M, N, K = 65536, 640, 640
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    W = bf(torch.randn(N, K, generator=g) * 0.1).to(DEV)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    run = lambda: hip.call("oneprot_gemm_bf16_nt", A, W, M, N, K, K, K, hip.EPI_BF16, None, out, None, None, None, None, None, 1.0, 0, 0, 0)
    try:
        hip.query("oneprot_dynamic_tiles", ws[0], ws[1])
        run(); torch.cuda.synchronize()
        ref = out.clone()
        assert hip.sched_error() == 0
        class _Raw:                                              # a zero-copy torch view of the workspace's first words (test only: the product never touches them from the host)
            __cuda_array_interface__ = {"shape": (8,), "typestr": "<i4", "data": (ws[0], False), "version": 2}
        torch.as_tensor(_Raw(), device=DEV)[0] = 3               # queue head of XCD 0 := 3
        torch.cuda.synchronize()
        out.fill_(float("nan"))
        run(); torch.cuda.synchronize()
        assert hip.sched_error() == 2
        assert torch.isnan(out.float()).any()                    # three tiles of XCD 0's share were never handed out
        ss = torch.full((1,), 4.0, device=DEV)
        coef, nrm = torch.empty(1, device=DEV), torch.empty(1, device=DEV)
        hip.call("oneprot_clip_coef", ss, 1.0, coef, nrm, ws[0])
        assert torch.isnan(nrm).all() and torch.isnan(coef).all()
        hip.sched_error_clear()
        out.fill_(float("nan"))
        run(); torch.cuda.synchronize()
        assert hip.sched_error() == 0 and torch.equal(out, ref)  # the failed launch reset the queues itself
    finally:
        hip.sched_error_clear()
        hip.query("oneprot_dynamic_tiles", ws[0] if hip.dynamic_tiles_wanted() else None, ws[1])


def test_sched_workspace_launches_replay_from_a_graph():
    """include/oneprot_hip.h: "every function that takes a stream only enqueues work on it ... graph-capturable".  The launches that keep bookkeeping between
    launches -- the FFN-2 + LayerNorm GEMM (tagged partial statistics) and the GEMMs that draw their tiles from the work queues -- are captured ONCE into a HIP graph
    and replayed: nothing on the host advances between replays, so the tags and the queue heads must be advanced by the kernels themselves (the last work-group
    to leave).  Every replay reproduces the eager results bit for bit on fresh inputs' outputs, no wait runs out, and the device-side epoch has moved by one per launch."""
    ws = hip.sched_workspace(131072)
    g = torch.Generator().manual_seed(41)
    M, d, f = 33024, 640, 2560
    U = bf(torch.randn(M, f, generator=g)).to(DEV)
    W2 = bf(torch.randn(d, f, generator=g) * 0.05).to(DEV)
    b2 = (torch.randn(d, generator=g) * 0.5).to(DEV)
    resid = torch.randn(M, d, generator=g).to(DEV)
    gamma, beta = (1 + 0.2 * torch.randn(d, generator=g)).to(DEV), (0.3 * torch.randn(d, generator=g)).to(DEV)
    x8 = torch.empty(M, d, device=DEV); h8 = torch.empty(M, d, dtype=torch.bfloat16, device=DEV); st = torch.empty(2, M, device=DEV)
    xg = torch.empty(M, d, device=DEV)

    def launches():
        hip.call("oneprot_gemm_bf16_nt_resid_ln8", U, W2, M, d, f, f, f, b2, resid, x8, gamma, beta, 1e-5, h8, st, *ws)
        hip.call("oneprot_gemm_bf16_nt", U, W2, M, d, f, f, f, hip.EPI_BIAS_RESID, b2, xg, None, None, resid, None, None, 1.0, 0, 0, 0)

    try:
        hip.query("oneprot_dynamic_tiles", ws[0], ws[1])
        launches()
        torch.cuda.synchronize()
        ref = [t.clone() for t in (x8, h8, st, xg)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            launches()                                           # warm-up on the capture stream
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=side):
                launches()
        torch.cuda.synchronize()
        e0 = hip.query("oneprot_sched_epoch", ws[0])
        for rep in range(4):
            for t in (x8, h8, st, xg):
                t.fill_(float("nan"))
            graph.replay()
            torch.cuda.synchronize()
            for a_, b_, name in zip((x8, h8, st, xg), ref, ("x ln8", "h ln8", "stats", "x plain")):
                assert torch.equal(a_, b_), (name, rep)
        assert hip.query("oneprot_sched_epoch", ws[0]) - e0 == 4 * 2          # two launches per replay went through the workspace: the device counted them
        assert hip.sched_error() == 0
    finally:
        hip.query("oneprot_dynamic_tiles", ws[0] if hip.dynamic_tiles_wanted() else None, ws[1])


def test_persistent_kernels_beside_a_kernel_that_holds_compute_units():
    import os
    """Co-residency (DESIGN section 5): a side-stream kernel holds k CUs (tests/cu_spin.hip: 100 KB of LDS per work-group, so no persistent work-group fits
    beside one) in windows that start and end INSIDE the product's launches -- what the RCCL channels of an overlapped gradient all-reduce do to the backward.
    Work-groups of the persistent kernels then start late, in the middle of a launch or after the others have finished.  With tiles / slabs drawn from the
    work queues every output stays bit for bit what the undisturbed launch wrote (every tile computed exactly once, no ticket lost or handed out twice, the
    queues reset by the last work-group), the FFN-2 + LayerNorm launch (static list: its work-groups wait for each other) finishes without a wait running
    out (a ticket draw found late is reported as a warning: it costs time, not correctness)."""
    import ctypes
    import time
    lib_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcu_spin.so")
    if not os.path.exists(lib_path):
        pytest.skip("tests/libcu_spin.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    spin = ctypes.CDLL(lib_path)
    spin.cu_spin_launch.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
    ws = hip.sched_workspace(131072)
    g = torch.Generator().manual_seed(31)
    M, d, f, B, H, L, hd = 65536, 640, 2560, 128, 20, 512, 32
    A = bf(torch.randn(M, d, generator=g)).to(DEV)
    W1 = bf(torch.randn(f, d, generator=g) * 0.1).to(DEV)
    b1 = (torch.randn(f, generator=g) * 0.5).to(DEV)
    U = bf(torch.randn(M, f, generator=g)).to(DEV)
    W2 = bf(torch.randn(d, f, generator=g) * 0.05).to(DEV)
    b2 = (torch.randn(d, generator=g) * 0.5).to(DEV)
    resid = torch.randn(M, d, generator=g).to(DEV)
    gamma, beta = (1 + 0.2 * torch.randn(d, generator=g)).to(DEV), (0.3 * torch.randn(d, generator=g)).to(DEV)
    q, k, v = [bf(torch.randn(B, H, L, hd, generator=g) * 0.7).to(DEV) for _ in range(3)]
    kb = torch.zeros(B, L, device=DEV)
    sink = torch.zeros(4, dtype=torch.int32, device=DEV)
    side = torch.cuda.Stream()

    def run_all():
        u = torch.empty(M, f, dtype=torch.bfloat16, device=DEV)
        hip.call("oneprot_gemm_bf16_nt", A, W1, M, f, d, d, d, hip.EPI_BIAS_GELU, b1, u, None, None, None, None, None, 1.0, 0, 0, 0)
        x = torch.empty(M, d, device=DEV)
        hip.call("oneprot_gemm_bf16_nt", U, W2, M, d, f, f, f, hip.EPI_BIAS_RESID, b2, x, None, None, resid, None, None, 1.0, 0, 0, 0)
        x8 = torch.empty(M, d, device=DEV)
        h8 = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
        hip.call("oneprot_gemm_bf16_nt_resid_ln8", U, W2, M, d, f, f, f, b2, resid, x8, gamma, beta, 1e-5, h8, None, *ws)
        dg = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
        hip.call("oneprot_gemm_bf16_nt", U, W2, M, d, f, f, f, hip.EPI_BF16, None, dg, None, None, None, None, None, 1.0, 0, 0, 0)
        ctx = torch.empty(B * L, H * hd, dtype=torch.bfloat16, device=DEV)
        lse = torch.empty(B, H, L, device=DEV)
        hip.call("oneprot_attn_fwd", q, k, v, kb, ctx, lse, B, H, L, hd)
        return u, x, x8, h8, dg, ctx, lse
    hip.query("oneprot_dynamic_tiles", None, 0)
    ref = run_all()                                              # static lists, nothing else on the GPU
    hip.query("oneprot_dynamic_tiles", ws[0], ws[1])             # (the multi-rank default)
    torch.cuda.synchronize()
    late0 = hip.sched_late_draws()
    for rep, (cus, win_us, gap_s) in enumerate([(8, 300, 0.0004), (24, 150, 0.0002), (4, 700, 0.0007), (32, 100, 0.0001), (8, 2000, 0.001)]):
        outs = []
        for _ in range(4):                                      # ~1 ms of launches per pass; the windows land at host-timed points inside them
            spin.cu_spin_launch(cus, win_us, sink.data_ptr(), side.cuda_stream)
            outs.append(run_all())
            time.sleep(gap_s)
            spin.cu_spin_launch(cus, win_us, sink.data_ptr(), side.cuda_stream)
        torch.cuda.synchronize()
        for got in outs:
            for a_, b_, name in zip(got, ref, ("gelu", "resid", "x ln8", "h ln8", "bf16", "ctx", "lse")):
                assert torch.equal(a_, b_), (name, rep, cus)
    hip.query("oneprot_dynamic_tiles", ws[0] if hip.dynamic_tiles_wanted() else None, ws[1])
    assert hip.sched_error() == 0
    if hip.sched_late_draws() != late0:      # not an error (the validated read waited and got the right ticket: outputs above are bit-identical), but worth knowing
        import warnings
        warnings.warn(f"{hip.sched_late_draws() - late0} ticket draws had not returned behind the counted wait that should cover them")


def test_gemm_resid_layernorm_forms_bit_identical():
    """the two kernel forms run the same arithmetic in the same order: equal bits"""
    M, K, N = 4096, 640, 640
    g = torch.Generator().manual_seed(12)
    A = bf(torch.randn(M, K, generator=g)).to(DEV); W = bf(torch.randn(N, K, generator=g) * 0.1).to(DEV)
    bias, gamma, beta = torch.randn(N, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    resid = torch.randn(M, N, generator=g).to(DEV)
    Wp = torch.empty(N * K, dtype=torch.bfloat16, device=DEV)
    hip.call("oneprot_gemm_ln_pack_weight", W, Wp, N, K)
    outs = []
    prev = hip.query("oneprot_gemm_ln_form_get")
    try:
        for form in (0, 1):
            hip.query("oneprot_gemm_ln_form", form)
            x, h = torch.empty(M, N, device=DEV), torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
            hip.call("oneprot_gemm_bf16_nt_resid_ln", A, Wp, M, N, K, K, bias, resid, x, gamma, beta, 1e-5, h, mean, rstd)
            outs.append((x, h, mean, rstd))
    finally:
        hip.query("oneprot_gemm_ln_form", prev)      # (0 is the library default: the rest of the process runs the shipped form)
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def _gemm_resid_layernorm_fused(M, K, inplace):
    N = 640
    g = torch.Generator().manual_seed(11)
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    W = bf(torch.randn(N, K, generator=g) * 0.1).to(DEV)
    bias = (torch.randn(N, generator=g) * 0.5).to(DEV)
    resid = (torch.randn(M, N, generator=g) * 2.0 + 0.3).to(DEV)
    gamma = (1.0 + 0.2 * torch.randn(N, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(N, generator=g)).to(DEV)
    Wp = torch.empty(N * K, dtype=torch.bfloat16, device=DEV)
    hip.call("oneprot_gemm_ln_pack_weight", W, Wp, N, K)
    xr = A.float() @ W.float().t() + bias + resid
    mu = xr.mean(-1, keepdim=True)
    var = ((xr - mu) ** 2).mean(-1, keepdim=True)
    hr = (xr - mu) * torch.rsqrt(var + 1e-5) * gamma + beta
    x = resid.clone() if inplace else torch.empty(M, N, device=DEV)
    h = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    hip.call("oneprot_gemm_bf16_nt_resid_ln", A, Wp, M, N, K, K, bias, x if inplace else resid, x, gamma, beta, 1e-5, h, mean, rstd)
    assert_close(x, xr, 1e-4, 1e-3 * math.sqrt(K / 64), "x = A W^T + bias + resid")
    assert_close(mean, mu.squeeze(-1), 1e-4, 1e-4 * math.sqrt(K / 64), "mean")
    assert_close(rstd, torch.rsqrt(var + 1e-5).squeeze(-1), 1e-3, 1e-5, "rstd")
    assert_close(h, hr, 2 ** -7, 2e-2, "h = LayerNorm(x)")
    # not built for this: the caller keeps the GEMM + LayerNorm pair
    with pytest.raises(hip.HipKernelError):
        hip.call("oneprot_gemm_bf16_nt_resid_ln", A[:100].contiguous(), Wp, 100, N, K, K, bias, resid, x, gamma, beta, 1e-5, h, mean, rstd)


@pytest.mark.parametrize("B,L,H,hd", [(3, 24, 4, 16), (2, 37, 2, 32), (2, 130, 20, 32), (2, 50, 2, 64), (5, 512, 20, 32), (16, 512, 8, 32), (4, 128, 4, 64), (8, 96, 8, 32),
                                      (16, 512, 12, 64), (24, 336, 4, 64)])      # head_dim 64 on the 8-phase 256 x 256 form (BERT-base width; rows that cross batch elements inside a wave block)
def test_gemm_qkv_rope_epilogue(B, L, H, hd, gemm_shape):
    d = H * hd
    M, N, K = B * L, 3 * d, d
    A, W, bias = _gemm_inputs(M, N, K, 5)
    cos, sin = O.rope_tables(L, hd)
    cosd, sind = cos[:, : hd // 2].contiguous().to(DEV), sin[:, : hd // 2].contiguous().to(DEV)
    q, k, v = (torch.empty(B, H, L, hd, dtype=torch.bfloat16, device=DEV) for _ in range(3))
    hip.call("oneprot_gemm_bf16_nt", A, W, M, N, K, K, K, hip.EPI_QKV_ROPE, bias, q, k, v, None, cosd, sind, hd ** -0.5, L, H, hd)
    y = (A.float() @ W.float().t() + bias).cpu().view(B, L, 3, H, hd).permute(2, 0, 3, 1, 4)
    qr = O.apply_rope(y[0] * hd ** -0.5, cos, sin)
    kr = O.apply_rope(y[1], cos, sin)
    assert_close(q.cpu(), qr, 2 ** -7, 2e-2, "q")
    assert_close(k.cpu(), kr, 2 ** -7, 2e-2, "k")
    assert_close(v.cpu(), y[2], 2 ** -7, 2e-2, "v")


@pytest.fixture(params=[-1, 0, 1, 2, 3], ids=["auto", "dma64x2", "dma32x3", "regstaged", "8phase"])
def tn_variant(request):
    hip.query("oneprot_gemm_tn_variant", request.param)
    yield request.param
    hip.query("oneprot_gemm_tn_variant", -1)


# (16384, 1920, 640), (8192, 640, 2560), (32768, 320, 128) and (4096, 640, 640) are whole 320 x 128 tiles with M a multiple of 64 x splits: they take the
# 8-phase form (auto and forced); the others fall through to the 128 x 128 kernels
@pytest.mark.parametrize("M,N,K", [(300, 136, 72), (4096, 640, 640), (5000, 1920, 640), (777, 160, 64), (1500, 2560, 640), (16384, 1920, 640), (8192, 640, 2560), (32768, 320, 128)])
def test_gemm_tn(M, N, K, tn_variant):
    g = torch.Generator().manual_seed(6)
    dY = bf(torch.randn(M, N, generator=g)).to(DEV)
    X = bf(torch.randn(M, K, generator=g)).to(DEV)
    ref = dY.float().t() @ X.float()
    dW = torch.full((N, K), 3.0, device=DEV)
    w = ws(hip.query("oneprot_gemm_bf16_tn_workspace", N, K))
    db = torch.full((N,), 5.0, device=DEV)
    hip.call("oneprot_gemm_bf16_tn", dY, X, M, N, K, N, K, dW, db, w, w.numel(), 0)
    assert_close(dW, ref, 1e-4, 2e-3 * math.sqrt(M / 64), "tn")
    assert_close(db, dY.float().sum(0), 1e-4, 1e-2, "tn fused bias grad")
    hip.call("oneprot_gemm_bf16_tn", dY, X, M, N, K, N, K, dW, None, w, w.numel(), 1)
    assert_close(dW, 2 * ref, 1e-4, 4e-3 * math.sqrt(M / 64), "tn accumulate")
    # strided views: columns [N0:N0+n) of a wider matrix
    cs = torch.full((N,), -1.0, device=DEV)
    w2 = ws(hip.query("oneprot_colsum_workspace", N))
    hip.call("oneprot_colsum_bf16", dY, cs, w2, M, N, 0)
    assert_close(cs, dY.float().sum(0), 1e-4, 1e-2, "colsum")


@pytest.mark.parametrize("M,N,K", [(8192, 1280, 1280), (8192, 1280, 5120), (8448, 480, 640), (6400, 1920, 480), (8320, 768, 768), (8320, 768, 3072)])
def test_gemm_tn_large_m_every_width(M, N, K):
    """Weight-gradient shapes of the ESM-2-650M, ESM-2-35M (padded heads) and BERT-base widths at token counts where the split count is no
    longer capped by M (a workspace sized for another (N, K) was overrun there).  The call must also refuse a workspace that is
    too small instead of writing past it."""
    g = torch.Generator().manual_seed(16)
    dY = bf(torch.randn(M, N, generator=g) * 0.5).to(DEV)
    X = bf(torch.randn(M, K, generator=g) * 0.5).to(DEV)
    need = hip.query("oneprot_gemm_bf16_tn_workspace", N, K)
    guard = 4096
    w = torch.zeros(need + guard, dtype=torch.uint8, device=DEV)
    w[need:] = 0xA5
    dW, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    hip.call("oneprot_gemm_bf16_tn", dY, X, M, N, K, N, K, dW, db, w, need, 0)
    assert bool((w[need:] == 0xA5).all()), "wrote past the workspace"
    ref = dY.float().t() @ X.float()
    assert_close(dW, ref, 1e-4, 2e-3 * math.sqrt(M / 64), "tn large M")
    assert_close(db, dY.float().sum(0), 1e-4, 2e-2, "tn large M bias")
    with pytest.raises(hip.HipKernelError):
        hip.call("oneprot_gemm_bf16_tn", dY, X, M, N, K, N, K, dW, db, w, need // 2, 0)


def test_binding_refuses_what_the_c_side_cannot_see():
    """a strided view, a wrong dtype or a host tensor used to reach the kernels as a bare data_ptr() and compute garbage."""
    x = torch.randn(64, 128, device=DEV)
    y = torch.empty(64, 64, dtype=torch.bfloat16, device=DEV)
    g, b = torch.ones(64, device=DEV), torch.zeros(64, device=DEV)
    with pytest.raises(hip.HipKernelError, match="not contiguous"):
        hip.call("oneprot_layernorm_fwd", x[:, ::2], 0, g, b, y, None, None, None, 64, 64, 1e-5)
    with pytest.raises(hip.HipKernelError, match="dtype"):
        hip.call("oneprot_layernorm_fwd", x[:, :64].contiguous(), 0, g.half(), b, y, None, None, None, 64, 64, 1e-5)
    with pytest.raises(hip.HipKernelError, match="dtype"):
        hip.call("oneprot_cast_f32_to_bf16", x.contiguous(), torch.empty(64 * 128, dtype=torch.float16, device=DEV), 64 * 128)
    with pytest.raises(hip.HipKernelError, match="no CPU fallback"):
        hip.call("oneprot_cast_f32_to_bf16", x.cpu(), torch.empty(64 * 128, dtype=torch.bfloat16, device=DEV), 64 * 128)
    ok = torch.empty(64, 64, dtype=torch.bfloat16, device=DEV)
    hip.call("oneprot_layernorm_fwd", x[:, :64].contiguous(), 0, g, b, ok, None, None, None, 64, 64, 1e-5)      # the dense form still runs
    assert torch.isfinite(ok.float()).all()


@pytest.mark.parametrize("tA,bkn", [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_sgemm(tA, bkn):
    M, N, K = 70, 130, 45
    g = torch.Generator().manual_seed(7)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(K, N, generator=g)
    Ad = (A.t().contiguous() if tA else A).to(DEV)
    Bd = (Bm if bkn else Bm.t().contiguous()).to(DEV)
    C = torch.ones(M, N, device=DEV)
    hip.call("oneprot_sgemm", Ad, Bd, C, M, N, K, tA, bkn, 0.5, 1)
    assert_close(C.cpu(), 1 + 0.5 * (A @ Bm), 1e-5, 1e-5, "sgemm")


# ------------------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, bias):
    s = q.float() @ k.float().transpose(-1, -2)
    if bias is not None:
        s = s + bias[:, None, None, :]
    p = torch.softmax(s, -1)
    return p @ v.float(), torch.logsumexp(s, -1)


@pytest.fixture(params=[0, 1, 2], ids=["bwd_split", "bwd_fused", "bwd_fused_64keys"])
def attn_bwd_path(request):
    hip.query("oneprot_attn_force_bwd_path", request.param)
    yield request.param
    hip.query("oneprot_attn_force_bwd_path", -1)


@pytest.fixture(params=[0, 1, 2], ids=["fwd_rowmax", "fwd_nomax", "fwd_nomax_chunked"])
def attn_fwd_path(request):
    hip.query("oneprot_attn_force_fwd_path", request.param)
    yield request.param
    hip.query("oneprot_attn_force_fwd_path", -1)


@pytest.mark.parametrize("B,H,L,hd", [(2, 3, 37, 32), (1, 2, 128, 16), (2, 2, 300, 32), (1, 2, 70, 64), (1, 1, 513, 32), (2, 2, 512, 32), (1, 3, 256, 32),
                                      (2, 1, 257, 16), (1, 2, 31, 16), (3, 2, 480, 32), (1, 2, 300, 64), (1, 1, 1100, 32), (1, 1, 600, 64)])
def test_attention_fwd_bwd(B, H, L, hd, attn_bwd_path, attn_fwd_path):
    g = torch.Generator().manual_seed(8)
    q = bf(torch.randn(B, H, L, hd, generator=g) * 0.7 * hip.LOG2E).to(DEV)          # the kernels take q x log2(e) (scores in log2 units)
    k = bf(torch.randn(B, H, L, hd, generator=g)).to(DEV)
    v = bf(torch.randn(B, H, L, hd, generator=g)).to(DEV)
    bias = torch.zeros(B, L)
    bias[0, L - L // 3:] = torch.finfo(torch.float32).min
    bias = bias.to(DEV)
    ctx = torch.empty(B * L, H * hd, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, H, L, device=DEV)
    hip.call("oneprot_attn_fwd", q, k, v, bias, ctx, lse, B, H, L, hd)
    qr, kr, vr = ((q.float() / hip.LOG2E).requires_grad_(True), k.float().requires_grad_(True), v.float().requires_grad_(True))
    o_ref, lse_ref = _attn_ref(qr, kr, vr, bias)
    o_tok = o_ref.permute(0, 2, 1, 3).reshape(B * L, H * hd)
    assert_close(ctx, o_tok.detach(), 2 ** -7, 1e-2, "attn fwd")
    # the row sum is taken by an all-ones MFMA over the bf16-rounded probabilities (the same values the context numerator uses): a row
    # dominated by one key carries that key's 2^-9 rounding into log(l)
    assert_close(lse, lse_ref.detach(), 1e-4, 5e-3, "lse")
    # backward (no rope: cos/sin NULL => dqkv is the raw gradient * q_scale)
    dctx = bf(torch.randn(B * L, H * hd, generator=g)).to(DEV)
    o_tok.backward(dctx.float())
    dqkv = torch.zeros(B * L, 3 * H * hd, dtype=torch.bfloat16, device=DEV)
    w = ws(hip.query("oneprot_attn_bwd_workspace", B, H, L))
    hip.call("oneprot_attn_bwd", q, k, v, bias, ctx, dctx, lse, None, None, 1.0, dqkv, w, B, H, L, hd)
    got = dqkv.float().view(B, L, 3, H, hd).permute(2, 0, 3, 1, 4)
    for name, gt, rf in (("dq", got[0], qr.grad), ("dk", got[1], kr.grad), ("dv", got[2], vr.grad)):
        assert rel_err(gt, rf) < 2e-2, f"{name} rel err {rel_err(gt, rf)}"
        assert_close(gt, rf, 5e-2, 5e-2 * rf.abs().max().item(), name)


@pytest.mark.parametrize("L", [1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257, 288, 479, 511, 512, 513, 777])
@pytest.mark.parametrize("hd", [16, 32, 64])
def test_attention_fwd_nomax_equals_rowmax_at_block_edges(L, hd):
    """k_attn_fwd2 (no per-tile maximum, whole (b, h) key range staged once, masked tiles skipped) against k_attn_fwd at every tile / wave / chunk
    boundary, with ragged key padding that ends inside a tile, on a tile edge, and leaves whole tiles masked.  Both kernels round P to bf16 after the
    same fp32 arithmetic up to the (here unused) maximum shift, so context and LSE agree to the bf16 rounding of the output."""
    B, H = 3, 2
    g = torch.Generator().manual_seed(2000 + L + hd)
    q = bf(torch.randn(B, H, L, hd, generator=g) * 0.7 * hip.LOG2E).to(DEV)
    k = bf(torch.randn(B, H, L, hd, generator=g)).to(DEV)
    v = bf(torch.randn(B, H, L, hd, generator=g)).to(DEV)
    bias = torch.zeros(B, L)
    if L > 2:
        bias[1, L - L // 4:] = torch.finfo(torch.float32).min       # ends inside a tile
        bias[2, max(1, (L // 2) & ~31):] = float("-inf")            # ends on a tile edge (when L >= 64): the rest are whole masked tiles
    bias = bias.to(DEV)
    outs = []
    for path in (0, 1, 2):           # 1: the persistent LDS-DMA kernel where eligible (hd <= 32, L <= 512); 2: the chunked kernel everywhere
        hip.query("oneprot_attn_force_fwd_path", path)
        try:
            for rep in range(2):     # twice: the persistent kernel alternates LDS halves, a stale half would show on the repeat
                ctx = torch.full((B * L, H * hd), float("nan"), dtype=torch.bfloat16, device=DEV)
                lse = torch.full((B, H, L), float("nan"), device=DEV)
                hip.call("oneprot_attn_fwd", q, k, v, bias, ctx, lse, B, H, L, hd)
            outs.append((ctx.float(), lse.clone()))
        finally:
            hip.query("oneprot_attn_force_fwd_path", -1)
    (c0, l0) = outs[0]
    o_ref, lse_ref = _attn_ref(q.float() / hip.LOG2E, k.float(), v.float(), bias)
    for name, (c1, l1) in zip(("fwd3/fwd2", "fwd2"), outs[1:]):
        assert torch.isfinite(c1).all() and torch.isfinite(l1).all(), name
        assert_close(c1, c0, 2 ** -7, 2 ** -8 * float(c0.abs().max()), f"ctx {name}")
        # the row sum is a sum of bf16-rounded probabilities in both kernels; with the maximum subtracted the largest one is exactly 1, without it
        # it carries its own 2^-9 rounding: log(l) of a row dominated by few keys differs by up to 2 x 2e-3
        assert_close(l1, l0, 1e-5, 8e-3, f"lse {name}")
        assert_close(c1, o_ref.permute(0, 2, 1, 3).reshape(B * L, H * hd), 2 ** -7, 1e-2, f"ctx vs fp32 {name}")
        assert_close(l1, lse_ref, 1e-4, 5e-3, f"lse vs fp32 {name}")


@pytest.mark.parametrize("L,hd", [(512, 32), (300, 16), (700, 32), (400, 64)])
def test_attention_fwd_nomax_overflow_underflow_net(L, hd):
    """Scores far outside the range a maximum-free exp2 can hold: every key is u + noise and query row i is alpha_i * u, so row i's scores all sit
    near alpha_i |u|^2 (log2 units): -400 / -200 (every exponential underflows or is tiny: the row-sum check at the end sends the work-group to the
    exact pass), 0, +45 / +70 (sums pass 2^40: rescaled in the fast pass, the wave continues with the bookkeeping k-step), +200 / +400 (inf
    in one tile: exact pass).  Rows of different kinds share waves and work-groups; one batch element keeps ordinary scores throughout (its
    work-groups must stay in the fast pass and still be right), one mixes padding in."""
    B, H = 3, 2
    g = torch.Generator().manual_seed(3000 + L + hd)
    u = torch.randn(hd, generator=g)
    u = u / u.norm()
    k = u[None, None, None, :] * 4.0 + 0.05 * torch.randn(B, H, L, hd, generator=g)
    kinds = torch.tensor([-400.0, -200.0, 0.0, 45.0, 70.0, 200.0, 400.0])
    alpha = kinds[torch.randint(0, len(kinds), (B, H, L), generator=g)] / 4.0            # k ~ 4 u, so q = alpha u gives q.k ~ 4 alpha = the kind
    alpha[0] = 0.0                                                                         # batch element 0: ordinary rows only
    q = alpha[..., None] * u[None, None, None, :] + 0.3 * torch.randn(B, H, L, hd, generator=g)
    q, k = bf(q).to(DEV), bf(k).to(DEV)
    v = bf(torch.randn(B, H, L, hd, generator=g)).to(DEV)
    bias = torch.zeros(B, L)
    bias[2, L - L // 3:] = float("-inf")
    bias = bias.to(DEV)
    sc = (q.float() @ k.float().transpose(-1, -2))                  # already log2 units
    assert float(sc.max()) > 300 and float(sc.min()) < -300 and float(sc[0].abs().max()) < 40
    s = sc * 0.6931471805599453 + bias[:, None, None, :]
    p = torch.softmax(s, -1)
    o_ref = (p @ v.float()).permute(0, 2, 1, 3).reshape(B * L, H * hd)
    lse_ref = torch.logsumexp(s, -1)
    for path in (0, 1, 2):          # (0: the round-1 kernel with the per-tile maximum must hold the same inputs)
        hip.query("oneprot_attn_force_fwd_path", path)
        try:
            ctx = torch.full((B * L, H * hd), float("nan"), dtype=torch.bfloat16, device=DEV)
            lse = torch.full((B, H, L), float("nan"), device=DEV)
            hip.call("oneprot_attn_fwd", q, k, v, bias, ctx, lse, B, H, L, hd)
        finally:
            hip.query("oneprot_attn_force_fwd_path", -1)
        assert torch.isfinite(ctx.float()).all() and torch.isfinite(lse).all(), path
        assert_close(ctx, o_ref, 2 ** -6, 1.5e-2, f"ctx (extreme scores, path {path})")
        assert_close(lse, lse_ref, 2e-4, 2e-2, f"lse (extreme scores, path {path})")


@pytest.mark.parametrize("L", [1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257, 288, 479, 511, 512])
@pytest.mark.parametrize("hd", [16, 32])
def test_attention_bwd_fused_equals_split_at_block_edges(L, hd):
    """The fused short-sequence backward against the split kernels at every block / group boundary of its walk (32-query blocks, 8-block
    groups, 16 key waves): same inputs, both paths.  The sums run in different orders (dK / dV: each wave starts its walk over the query blocks at a
    different block; dQ: ticket order over key blocks vs one wave's key loop), so the results agree up to the bf16 rounding of the stored gradient."""
    B, H = 2, 3
    g = torch.Generator().manual_seed(1000 + L + hd)
    q = bf(torch.randn(B, H, L, hd, generator=g) * 0.7 * hip.LOG2E).to(DEV)
    k = bf(torch.randn(B, H, L, hd, generator=g)).to(DEV)
    v = bf(torch.randn(B, H, L, hd, generator=g)).to(DEV)
    bias = torch.zeros(B, L)
    if L > 2:
        bias[1, L - L // 4:] = torch.finfo(torch.float32).min
    bias = bias.to(DEV)
    ctx = torch.empty(B * L, H * hd, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, H, L, device=DEV)
    hip.call("oneprot_attn_fwd", q, k, v, bias, ctx, lse, B, H, L, hd)
    dctx = bf(torch.randn(B * L, H * hd, generator=g)).to(DEV)
    cos, sin = O.rope_tables(L, hd)
    cosd, sind = cos[:, : hd // 2].contiguous().to(DEV), sin[:, : hd // 2].contiguous().to(DEV)
    outs = []
    for path in (0, 1, 1, 2, 2):     # the fused kernels twice: their dQ sums run in ticket order, so a repeat is bit-identical
        hip.query("oneprot_attn_force_bwd_path", path)
        try:
            dqkv = torch.full((B * L, 3 * H * hd), float("nan"), dtype=torch.bfloat16, device=DEV)
            w = ws(hip.query("oneprot_attn_bwd_workspace", B, H, L))
            hip.call("oneprot_attn_bwd", q, k, v, bias, ctx, dctx, lse, cosd, sind, hd ** -0.5, dqkv, w, B, H, L, hd)
            outs.append(dqkv.float().view(B * L, 3, H * hd))
        finally:
            hip.query("oneprot_attn_force_bwd_path", -1)
    split, fused, fused2, f64, f64b = outs
    assert torch.isfinite(fused).all() and torch.isfinite(f64).all()
    assert torch.equal(fused, fused2)
    assert torch.equal(f64, f64b)
    if L > 2:      # masked keys (probability exactly 0) receive exactly zero dK / dV -- also from the waves that skip their all-padding key blocks
        assert float(f64[L + L - L // 4: 2 * L, 1:].abs().max()) == 0.0 and float(fused[L + L - L // 4: 2 * L, 1:].abs().max()) == 0.0
    # two key blocks per wave: the walks start at other query blocks and dQ adds its two key blocks first -- the same sums in yet another order
    for part, name in enumerate(("dQ", "dK", "dV")):
        assert_close(f64[:, part], split[:, part], 2 ** -7, 2 ** -7 * float(split[:, part].abs().max()), name + " (64 keys per wave)")
        assert rel_err(f64[:, part], split[:, part]) < 3e-3, name
    for part, name in enumerate(("dQ", "dK", "dV")):
        assert_close(fused[:, part], split[:, part], 2 ** -7, 2 ** -7 * float(split[:, part].abs().max()), name)
        assert rel_err(fused[:, part], split[:, part]) < 3e-3, name


@pytest.mark.parametrize("B,H,L,hd", [(2, 3, 70, 64), (1, 2, 300, 64), (2, 2, 96, 32)])
def test_attention_probability_dropout_fwd_bwd(B, H, L, hd):
    """oneprot_attn_fwd_dropout / oneprot_attn_bwd_dropout (hf BertSelfAttention in train mode: softmax -> dropout -> @ V): forward and the gradients
    w.r.t. q, k, v against torch autograd on the same bf16 inputs with the mask the kernels drew (exported by oneprot_attn_dropout_keep); keep rate,
    independence across heads / streams, undropped log-sum-exp."""
    g = torch.Generator().manual_seed(11 + L)
    q = bf(torch.randn(B, H, L, hd, generator=g) * hd ** -0.5)
    k, v = bf(torch.randn(B, H, L, hd, generator=g)), bf(torch.randn(B, H, L, hd, generator=g))
    bias = torch.zeros(B, L)
    bias[-1, L - L // 5:] = -3.0e38
    p_, seed, stream = 0.1, 0xABCDEF0123, 5
    keep = torch.empty(B, H, L, L, dtype=torch.uint8, device=DEV)
    hip.call("oneprot_attn_dropout_keep", keep, B, H, L, p_, seed, stream)
    keep2 = torch.empty_like(keep)
    hip.call("oneprot_attn_dropout_keep", keep2, B, H, L, p_, seed, stream + 1)
    kf = keep.float().cpu()
    thr = int(p_ * 65536 + 0.5)
    kp = 1 - thr / 65536
    assert abs(float(kf.mean()) - kp) < 0.01
    assert abs(float((keep == keep2).float().mean()) - (kp ** 2 + (1 - kp) ** 2)) < 0.01          # another stream: an independent mask
    assert not torch.equal(keep[0, 0], keep[0, 1])                                               # heads differ
    m = kf[0, 0] - kf[0, 0].mean()
    assert abs(float((m[1:] * m[:-1]).mean() / m.var())) < 0.03 and abs(float((m[:, 1:] * m[:, :-1]).mean() / m.var())) < 0.03      # neighbours along q and along k: uncorrelated
    qd = (q * hip.LOG2E).to(torch.bfloat16).to(DEV)
    ctx = torch.empty(B * L, H * hd, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, H, L, device=DEV)
    hip.call("oneprot_attn_fwd_dropout", qd, k.to(DEV), v.to(DEV), bias.to(DEV), ctx, lse, B, H, L, hd, p_, seed, stream)
    ql = (qd.float().cpu() / hip.LOG2E).requires_grad_(True)
    kr, vr = k.float().requires_grad_(True), v.float().requires_grad_(True)
    s_ = ql @ kr.transpose(-1, -2) + bias[:, None, None, :]
    pr = torch.softmax(s_, -1)
    ref = ((pr * kf / kp) @ vr).permute(0, 2, 1, 3).reshape(B * L, H * hd)
    assert_close(ctx.float().cpu(), ref.detach(), 2 ** -6, 2e-2, "ctx with probability dropout")
    assert_close(lse.cpu(), torch.logsumexp(s_, -1).detach(), 2e-3, 2e-2, "lse (undropped)")
    nodrop = (pr @ vr).permute(0, 2, 1, 3).reshape(B * L, H * hd).detach()
    assert rel_err(ctx.float().cpu(), nodrop) > 5 * rel_err(ctx.float().cpu(), ref.detach())          # the mask matters
    dctx = bf(torch.randn(B * L, H * hd, generator=g))
    ref.backward(dctx.float())
    dqkv = torch.zeros(B * L, 3 * H * hd, dtype=torch.bfloat16, device=DEV)
    w = ws(hip.query("oneprot_attn_bwd_workspace", B, H, L))
    hip.call("oneprot_attn_bwd_dropout", qd, k.to(DEV), v.to(DEV), bias.to(DEV), ctx, dctx.to(DEV), lse, None, None, 1.0, dqkv, w, B, H, L, hd, p_, seed, stream)
    got = dqkv.float().cpu().view(B, L, 3, H, hd).permute(2, 0, 3, 1, 4)
    # q was stored x log2(e): the kernel returns d/d(unscaled q) with q_scale = 1 applied to ... the gradient w.r.t. the stored q / log2(e)
    for i, (name, r_) in enumerate((("dq", ql.grad), ("dk", kr.grad), ("dv", vr.grad))):
        assert rel_err(got[i], r_) < 2e-2, f"{name}: {rel_err(got[i], r_)}"
    # and the masked backward is not the plain one
    dq0 = torch.zeros_like(dqkv)
    hip.query("oneprot_attn_force_bwd_path", 0)
    try:
        hip.call("oneprot_attn_bwd", qd, k.to(DEV), v.to(DEV), bias.to(DEV), ctx, dctx.to(DEV), lse, None, None, 1.0, dq0, w, B, H, L, hd)
    finally:
        hip.query("oneprot_attn_force_bwd_path", -1)
    assert rel_err(dq0.float().cpu().view(B, L, 3, H, hd).permute(2, 0, 3, 1, 4)[2], vr.grad) > 5 * rel_err(got[2], vr.grad)


@pytest.mark.parametrize("L,expected", [(256, 0), (288, 0), (320, 1), (416, 1), (448, 2), (512, 2)])
def test_attention_bwd_automatic_path_by_length(L, expected):
    """The launcher picks the backward kernel by sequence length (split pair up to 288, 16-wave fused up to 416, 8-wave fused up to 512: measured
    crossovers, attention.hip launch_bwd): the automatic result is bit-identical to the forced path it should have taken."""
    B, H, hd = 2, 2, 32
    g = torch.Generator().manual_seed(77 + L)
    q = bf(torch.randn(B, H, L, hd, generator=g) * 0.7 * hip.LOG2E).to(DEV)
    k = bf(torch.randn(B, H, L, hd, generator=g)).to(DEV)
    v = bf(torch.randn(B, H, L, hd, generator=g)).to(DEV)
    ctx = torch.empty(B * L, H * hd, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, H, L, device=DEV)
    hip.call("oneprot_attn_fwd", q, k, v, None, ctx, lse, B, H, L, hd)
    dctx = bf(torch.randn(B * L, H * hd, generator=g)).to(DEV)
    outs = {}
    for path in (-1, 0, 1, 2):
        hip.query("oneprot_attn_force_bwd_path", path)
        try:
            dqkv = torch.full((B * L, 3 * H * hd), float("nan"), dtype=torch.bfloat16, device=DEV)
            w = ws(hip.query("oneprot_attn_bwd_workspace", B, H, L))
            hip.call("oneprot_attn_bwd", q, k, v, None, ctx, dctx, lse, None, None, hd ** -0.5, dqkv, w, B, H, L, hd)
            outs[path] = dqkv.float()
        finally:
            hip.query("oneprot_attn_force_bwd_path", -1)
    assert torch.equal(outs[-1], outs[expected])
    others = [p for p in (0, 1, 2) if p != expected]
    assert any(not torch.equal(outs[-1], outs[p]) for p in others)      # (the paths do differ in their rounding: the equality above is not vacuous)


@pytest.mark.parametrize("L,hd", [(45, 32), (300, 32), (77, 16)])
def test_attention_bwd_rope_chain(L, hd, attn_bwd_path):
    """dqkv must be the gradient w.r.t. the un-rotated, un-scaled projections (transpose of q-scale + RoPE)."""
    B, H = 2, 2
    g = torch.Generator().manual_seed(9)
    ylin = (torch.randn(3, B, H, L, hd, generator=g)).requires_grad_(True)
    cos, sin = O.rope_tables(L, hd)
    qs = O.apply_rope(ylin[0] * hd ** -0.5, cos, sin)
    ks = O.apply_rope(ylin[1], cos, sin)
    qb, kb, vb = bf(qs.detach() * hip.LOG2E).to(DEV), bf(ks.detach()).to(DEV), bf(ylin[2].detach()).to(DEV)
    ctx = torch.empty(B * L, H * hd, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, H, L, device=DEV)
    hip.call("oneprot_attn_fwd", qb, kb, vb, None, ctx, lse, B, H, L, hd)
    o, _ = _attn_ref(qs, ks, ylin[2], None)
    dctx = bf(torch.randn(B * L, H * hd, generator=g))
    o.permute(0, 2, 1, 3).reshape(B * L, H * hd).backward(dctx.float())
    dqkv = torch.zeros(B * L, 3 * H * hd, dtype=torch.bfloat16, device=DEV)
    w = ws(hip.query("oneprot_attn_bwd_workspace", B, H, L))
    cosd, sind = cos[:, : hd // 2].contiguous().to(DEV), sin[:, : hd // 2].contiguous().to(DEV)
    hip.call("oneprot_attn_bwd", qb, kb, vb, None, ctx, dctx.to(DEV), lse, cosd, sind, hd ** -0.5, dqkv, w, B, H, L, hd)
    got = dqkv.float().cpu().view(B, L, 3, H, hd).permute(2, 0, 3, 1, 4)
    for i, name in enumerate(("dq", "dk", "dv")):
        assert rel_err(got[i], ylin.grad[i]) < 2e-2, f"{name}: {rel_err(got[i], ylin.grad[i])}"


# ------------------------------------------------------------------------------------------------------
def test_feature_ops():
    R, D = 37, 100
    g = torch.Generator().manual_seed(10)
    x = torch.randn(R, D, generator=g)
    xd = x.to(DEV)
    y = torch.empty(R, D, device=DEV)
    inv = torch.empty(R, device=DEV)
    scale = 1 / 0.07
    hip.call("oneprot_l2norm_fwd", xd, y, inv, R, D, scale)
    xr = x.clone().requires_grad_(True)
    yr = O.l2_normalize(xr) * scale
    assert_close(y.cpu(), yr.detach(), 1e-5, 1e-5, "l2norm fwd")
    dy = torch.randn(R, D, generator=g)
    l1c = 0.01 / (R * D)
    (yr * dy).sum().backward(retain_graph=True)
    g0 = xr.grad.clone()
    xr.grad = None
    ((yr * dy).sum() + 0.01 * yr.abs().mean()).backward()
    dx = torch.empty(R, D, device=DEV)
    hip.call("oneprot_l2norm_bwd", y, dy.to(DEV), inv, dx, R, D, scale, 0.0)
    assert_close(dx.cpu(), g0, 1e-4, 1e-5, "l2norm bwd")
    hip.call("oneprot_l2norm_bwd", y, dy.to(DEV), inv, dx, R, D, scale, l1c)
    assert_close(dx.cpu(), xr.grad, 1e-4, 1e-5, "l2norm bwd + L1")
    # gelu
    gy = torch.empty(R, D, device=DEV)
    hip.call("oneprot_gelu_f32", xd, gy, R * D)
    assert_close(gy.cpu(), O.gelu_erf(x), 1e-5, 1e-6, "gelu")
    x2 = x.clone().requires_grad_(True)
    O.gelu_erf(x2).backward(dy)
    hip.call("oneprot_gelu_bwd_f32", xd, dy.to(DEV), gy, R * D)
    assert_close(gy.cpu(), x2.grad, 1e-4, 1e-5, "gelu bwd")
    # abs sum
    acc = torch.ones(1, device=DEV)
    w = ws(hip.query("oneprot_sumsq_workspace"))
    hip.call("oneprot_abs_sum", xd, acc, w, R * D, 0.5)
    assert abs(acc.item() - (1 + 0.5 * x.abs().sum().item())) < 1e-2


@pytest.mark.parametrize("R,C,off", [(6, 6, 0), (5, 15, 5), (256, 2048, 512)])
def test_cross_entropy(R, C, off):
    g = torch.Generator().manual_seed(11)
    logits = torch.randn(R, C, generator=g) * 8
    lr = logits.clone().requires_grad_(True)
    labels = torch.arange(R) + off
    loss = torch.nn.functional.cross_entropy(lr, labels) / 2
    loss.backward()
    ld = logits.to(DEV)
    out = torch.zeros(1, device=DEV)
    rw = torch.empty(R, device=DEV)
    hip.call("oneprot_ce_fwd_bwd", ld, out, rw, R, C, off, 0.5 / R)
    assert abs(out.item() - loss.item()) < 1e-5 * max(1, abs(loss.item()))
    assert_close(ld.cpu(), lr.grad, 1e-4, 1e-7, "dlogits")


def test_optimizer_kernels():
    n = 4 * 1000 + 8
    g = torch.Generator().manual_seed(12)
    p = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * 3 for _ in range(3)]
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3)
    pd, m, v = p.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    w = ws(hip.query("oneprot_sumsq_workspace"))
    for step, gr in enumerate(grads, 1):
        pr.grad = gr.clone()
        tn = torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        gd = gr.to(DEV)
        ss = torch.zeros(1, device=DEV)
        coef, nrm = torch.empty(1, device=DEV), torch.empty(1, device=DEV)
        hip.call("oneprot_sumsq", gd, n, ss, w)
        hip.call("oneprot_clip_coef", ss, 1.0, coef, nrm, None)
        assert abs(nrm.item() - tn.item()) < 1e-3 * tn.item()
        hip.call("oneprot_adam_step", pd, gd, m, v, n, 1e-3, 0.9, 0.999, 1e-8, 0.0, step, coef)
        assert_close(pd.cpu(), pr.detach(), 1e-5, 2e-6, f"adam step {step}")


def test_casts():
    g = torch.Generator().manual_seed(13)
    x = torch.randn(70, 132, generator=g).to(DEV)
    y = torch.empty(70, 132, dtype=torch.bfloat16, device=DEV)
    hip.call("oneprot_cast_f32_to_bf16", x, y, x.numel())
    assert torch.equal(y, x.to(torch.bfloat16))
    yt = torch.empty(132, 70, dtype=torch.bfloat16, device=DEV)
    hip.call("oneprot_transpose_cast_f32_to_bf16", x, yt, 70, 132)
    assert torch.equal(yt, x.t().contiguous().to(torch.bfloat16))
    # the same weight of every layer in one launch: matrices at a constant pitch inside a larger arena
    arena = torch.randn(5 * 10000 + 64, device=DEV)
    out = torch.full((5, 132, 70), 7.0, dtype=torch.bfloat16, device=DEV)
    hip.call("oneprot_transpose_cast_f32_to_bf16_batched", arena[64:], out, 70, 132, 10000, 132 * 70, 5)
    for z in range(5):
        assert torch.equal(out[z], arena[64 + z * 10000: 64 + z * 10000 + 70 * 132].view(70, 132).t().contiguous().to(torch.bfloat16))
    with pytest.raises(hip.HipKernelError):
        hip.call("oneprot_transpose_cast_f32_to_bf16_batched", arena[64:], out, 70, 132, 10000, 100, 5)      # overlapping outputs


def test_dropout_streams_of_different_consumers_and_towers_are_independent():
    """The BERT tower's hidden dropout and the LoRA dropout draw from the same generator with the same seed (torch.initial_seed): purpose and tower
    are part of the stream id (ArenaModule._rng_stream), so the embedding mask of call 0 is NOT the layer-0 query-adapter mask, and two towers'
    LoRA masks differ."""
    from oneprot_amd.esm import ArenaModule
    towers = []
    for uid in (0, 1):
        t = ArenaModule.__new__(ArenaModule)
        t._rng_uid = uid
        towers.append(t)
    n, p_, seed = 64 * 768, 0.1, 0x5EED
    ones = torch.ones(n, dtype=torch.bfloat16, device=DEV)
    def mask(stream):
        out = torch.empty_like(ones)
        hip.call("oneprot_dropout_bf16", ones, out, n, p_, seed, stream)
        return out != 0
    bert_emb = mask(towers[0]._rng_stream(ArenaModule.RNG_DOMAIN_BERT, 0))
    lora_q0 = mask(towers[0]._rng_stream(ArenaModule.RNG_DOMAIN_LORA, 0))
    lora_q0_other = mask(towers[1]._rng_stream(ArenaModule.RNG_DOMAIN_LORA, 0))
    for a, b in ((bert_emb, lora_q0), (lora_q0, lora_q0_other)):
        agree = float((a == b).float().mean())
        assert abs(agree - (0.9 * 0.9 + 0.1 * 0.1)) < 0.01, agree


def test_dropout_f32_and_its_residual_form():
    """oneprot_dropout_f32 (the BERT tower's hidden dropout) and oneprot_dropout_add_f32 = resid + dropout(x) in one pass (hf BertSelfOutput / BertOutput:
    dropout of the dense output, then the residual add): same mask for the same (seed, stream), exact values, aliasing allowed."""
    g = torch.Generator().manual_seed(9)
    n = 64 * 768
    x = torch.randn(n, generator=g).to(DEV)
    resid = torch.randn(n, generator=g).to(DEV)
    p_, seed, stream = 0.1, 0xABCDEF, (1 << 60) | 77
    d = torch.empty_like(x)
    hip.call("oneprot_dropout_f32", x, d, n, p_, seed, stream)
    keep = d != 0
    thr = int(p_ * 65536 + 0.5)
    assert abs(float(keep.float().mean()) - (1 - thr / 65536)) < 0.01
    assert torch.equal(d[keep], x[keep] * (65536.0 / (65536 - thr)))
    y = torch.empty_like(x)
    hip.call("oneprot_dropout_add_f32", x, resid, y, n, p_, seed, stream)
    assert torch.equal(y, resid + d)
    y2 = resid.clone()
    hip.call("oneprot_dropout_add_f32", x, y2, y2, n, p_, seed, stream)          # in place on the residual stream
    assert torch.equal(y2, y)
    with pytest.raises(hip.HipKernelError):
        hip.call("oneprot_dropout_add_f32", x, None, y, n, p_, seed, stream)


@pytest.mark.parametrize("T,d", [(67, 768), (9, 1024), (130, 264), (5, 2048)])
def test_dropout_add_layernorm_fused_equals_the_two_kernels(T, d):
    """oneprot_dropout_add_layernorm_fwd (hf BertSelfOutput / BertOutput: dense -> dropout -> LayerNorm(. + input) in one pass over the rows) against
    oneprot_dropout_add_f32 followed by oneprot_layernorm_fwd: the sum bit for bit (same mask, same arithmetic), the LayerNorm outputs to the
    rounding of a different summation order (eight elements per lane and step instead of four); without the optional sum; in place on the residual."""
    g = torch.Generator().manual_seed(31 + d)
    x = torch.randn(T, d, generator=g).to(DEV)
    resid = torch.randn(T, d, generator=g).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV), (0.1 * torch.randn(d, generator=g)).to(DEV)
    p_, seed, stream = 0.1, 0x5EED, (1 << 60) | (3 << 44) | 9
    s_ref = torch.empty_like(x)
    hip.call("oneprot_dropout_add_f32", x, resid, s_ref, T * d, p_, seed, stream)
    y16_ref, y_ref = torch.empty(T, d, dtype=torch.bfloat16, device=DEV), torch.empty_like(x)
    m_ref, r_ref = torch.empty(T, device=DEV), torch.empty(T, device=DEV)
    hip.call("oneprot_layernorm_fwd", s_ref, 0, gamma, beta, y16_ref, y_ref, m_ref, r_ref, T, d, 1e-12)
    s, y16, y, m, r = torch.empty_like(x), torch.empty_like(y16_ref), torch.empty_like(x), torch.empty(T, device=DEV), torch.empty(T, device=DEV)
    hip.call("oneprot_dropout_add_layernorm_fwd", x, resid, s, gamma, beta, y16, y, m, r, T, d, 1e-12, p_, seed, stream)
    assert torch.equal(s, s_ref)
    assert_close(m, m_ref, 1e-6, 1e-6, "mean")
    assert_close(r, r_ref, 1e-5, 0.0, "rstd")
    assert_close(y, y_ref, 1e-5, 2e-5, "LayerNorm fp32")
    assert_close(y16.float(), y16_ref.float(), 2 ** -7, 1e-5, "LayerNorm bf16")
    # a frozen tower: no sum, fp32 output in place on the residual stream
    r2, y16b = resid.clone(), torch.empty_like(y16_ref)
    hip.call("oneprot_dropout_add_layernorm_fwd", x, r2, None, gamma, beta, y16b, r2, None, None, T, d, 1e-12, p_, seed, stream)
    assert torch.equal(r2, y) and torch.equal(y16b, y16)
    with pytest.raises(hip.HipKernelError):
        hip.call("oneprot_dropout_add_layernorm_fwd", x, resid, None, gamma, beta, y16, y, None, None, T, 12, 1e-12, p_, seed, stream)


def test_dropout_bf16_mask_is_a_function_of_seed_stream_and_element():
    """oneprot_dropout_bf16 (peft's lora_dropout on the adapter branch's input): keep probability, exact values, determinism, independent streams,
    and the two backward forms regenerate the forward's mask."""
    g = torch.Generator().manual_seed(5)
    n = 96 * 320
    x = torch.randn(n, generator=g).to(torch.bfloat16).to(DEV)
    p_, seed = 0.1, 0x1234ABCD5678
    thr = int(p_ * 65536 + 0.5)
    scale = 65536.0 / (65536 - thr)
    y, y2, y3 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    hip.call("oneprot_dropout_bf16", x, y, n, p_, seed, 7)
    hip.call("oneprot_dropout_bf16", x, y2, n, p_, seed, 7)
    hip.call("oneprot_dropout_bf16", x, y3, n, p_, seed, 8)
    assert torch.equal(y, y2) and not torch.equal(y, y3)
    keep = y != 0
    kept_or_zero_input = keep | (x == 0)
    assert abs(float(kept_or_zero_input.float().mean()) - (1 - thr / 65536)) < 4 * (0.1 * 0.9 / n) ** 0.5 + 1e-3
    assert torch.equal(y[keep], (x[keep].float() * scale).to(torch.bfloat16))
    k3 = y3 != 0
    agree = float((keep == k3).float().mean())
    assert abs(agree - (0.9 * 0.9 + 0.1 * 0.1)) < 0.01, agree            # two streams: independent masks
    # lag-1 correlation of the mask along the element index is that of independent draws
    m = keep.float() - keep.float().mean()
    assert abs(float((m[1:] * m[:-1]).mean() / m.var())) < 0.02
    dy = torch.randn(n, generator=g).to(torch.bfloat16).to(DEV)
    base = torch.randn(n, generator=g)
    d16 = base.to(torch.bfloat16).to(DEV)
    hip.call("oneprot_dropout_bwd_add_bf16", dy, d16, n, p_, seed, 7)
    ones = torch.ones(n, dtype=torch.bfloat16, device=DEV)
    mk = torch.empty_like(ones)
    hip.call("oneprot_dropout_bf16", ones, mk, n, p_, seed, 7)
    mask = (mk > 0).float()
    ref16 = (base.to(torch.bfloat16).float().to(DEV) + mask * scale * dy.float()).to(torch.bfloat16)
    assert torch.equal(d16, ref16)
    d32 = base.clone().to(DEV)
    hip.call("oneprot_dropout_bwd_add_f32", dy, d32, n, p_, seed, 7)
    assert_close(d32.cpu(), (base.to(DEV) + mask * scale * dy.float()).cpu(), 1e-6, 1e-6, "dropout bwd add f32")
    with pytest.raises(hip.HipKernelError):
        hip.call("oneprot_dropout_bf16", x, y, n - 4, p_, seed, 7)            # whole 16-byte pieces only
    with pytest.raises(hip.HipKernelError):
        hip.call("oneprot_dropout_bf16", x, y, n, 1.0, seed, 7)


@pytest.mark.parametrize("world,local_loss", [(2, True), (3, True), (2, False), (4, True), (4, False), (8, True), (8, False)])
def test_clip_loss_node_multirank_semantics(golden_dir, world, local_loss):
    """The fused CLIP node (SGEMM logits + fused CE fwd/bwd) on gathered features, per rank, vs the losses/gradients the reference
    produced on real ranks (gather_with_grad=False variant: the local gradient is exactly d loss_rank / d local features).  World 8 = the node size of
    cfg-3..5: label offsets rank * B_loc up to 7 * B_loc with B_glob = 8 * B_loc inside the fused CE kernel (ref loss.py:72-83)."""
    import os
    from oneprot_amd.loss import _ClipLossFn
    g = torch.load(os.path.join(golden_dir, f"loss_world{world}.pt"), weights_only=False)
    M, S = g["m"], g["s"]
    B, D = M.shape[1:]
    for rank in range(world):
        m = M[rank].clone().to(DEV).requires_grad_(True)
        s = S[rank].clone().to(DEV).requires_grad_(True)
        if local_loss:
            all_m = M.reshape(-1, D).to(DEV)
            all_s = S.reshape(-1, D).to(DEV)
            loss = _ClipLossFn.apply(m, all_s, s, all_m, 1.0, B * rank)
        else:
            am = torch.cat([m if r == rank else M[r].to(DEV) for r in range(world)])
            as_ = torch.cat([s if r == rank else S[r].to(DEV) for r in range(world)])
            loss = _ClipLossFn.apply(am, as_, as_, am, 1.0, 0)
        loss.backward()
        rl, rgm, rgs = g["per_rank"][rank][f"clip_ll{int(local_loss)}_gwg0"]
        assert abs(loss.item() - rl.item()) / abs(rl.item()) < 1e-5
        assert_close(m.grad.cpu(), rgm, 1e-4, 1e-6, "dm")
        assert_close(s.grad.cpu(), rgs, 1e-4, 1e-6, "ds")


@pytest.mark.parametrize("negative_only", [False, True])
def test_siglip_block(negative_only):
    from oneprot_amd.loss import _SigLipBlockFn
    g = torch.Generator().manual_seed(14)
    B, D = 37, 48
    m = (torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=-1) * (1 / 0.07))
    s = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=-1)
    mr, sr = m.clone().requires_grad_(True), s.clone().requires_grad_(True)
    ref = O.siglip_block(mr, sr, 1.0, -2.5, negative_only)
    (ref * 0.7).backward()
    md, sd = m.to(DEV).requires_grad_(True), s.to(DEV).requires_grad_(True)
    from oneprot_amd.loss import _siglip_block_hip
    loss = _SigLipBlockFn.apply(md, sd, 1.0, -2.5, negative_only, _siglip_block_hip)
    (loss * 0.7).backward()
    assert abs(loss.item() - ref.item()) < 1e-4 * abs(ref.item())
    assert_close(md.grad.cpu(), mr.grad, 1e-4, 1e-6, "siglip dm")
    assert_close(sd.grad.cpu(), sr.grad, 1e-4, 1e-6, "siglip ds")


@pytest.mark.parametrize("scale_on", ["cuda", "cpu"])
def test_siglip_tensor_scale_and_bias_stay_on_device_and_get_gradients(scale_on, golden_dir):
    """ref loss.py:241-245: `logit_scale * m @ s.T` and `logits += logit_bias` with (learnable) tensors.  SigLipLoss takes them without a host
    synchronisation and returns their gradients (on the caller's device) -- against the oracle's autograd at world size 1; the multi-rank
    sums are pinned by the reference's own numbers in tests/test_distributed_cpu.py (`siglip_bidir*_tensor`)."""
    from oneprot_amd.loss import SigLipLoss
    g = torch.Generator().manual_seed(21)
    B, D = 37, 48
    m = (torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=-1) * (1 / 0.07))
    s = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=-1)
    mr, sr = m.clone().requires_grad_(True), s.clone().requires_grad_(True)
    sc_r, bi_r = torch.tensor(1.3, requires_grad=True), torch.tensor(-0.7, requires_grad=True)
    ref = O.siglip_block(mr, sr, sc_r, bi_r, False)
    (ref * 0.7).backward()
    md, sd = m.to(DEV).requires_grad_(True), s.to(DEV).requires_grad_(True)
    sc = torch.tensor(1.3, device=scale_on, requires_grad=True)
    bi = torch.tensor(-0.7, device=scale_on, requires_grad=True)
    loss = SigLipLoss(rank=0, world_size=1)(md, sd, logit_scale=sc, logit_bias=bi)
    (loss * 0.7).backward()
    assert abs(loss.item() - ref.item()) < 1e-4 * abs(ref.item())
    assert_close(md.grad.cpu(), mr.grad, 1e-4, 1e-6, "siglip dm")
    assert_close(sd.grad.cpu(), sr.grad, 1e-4, 1e-6, "siglip ds")
    assert sc.grad.device.type == scale_on and sc.grad.shape == sc.shape
    assert abs(sc.grad.item() - sc_r.grad.item()) < 1e-4 * abs(sc_r.grad.item()) + 1e-6, (sc.grad, sc_r.grad)
    assert abs(bi.grad.item() - bi_r.grad.item()) < 1e-4 * abs(bi_r.grad.item()) + 1e-6, (bi.grad, bi_r.grad)


def test_retrieval_metric_vs_reference_golden(golden_dir):
    """device-side RetrievalMetric (SGEMM + rank-counting kernel) vs the numbers the reference's RetrievalMetric.compute produced
    (tests/golden/retrieval.pt: exact fp32 similarities, no ties with the diagonal): equal, not close; updates arrive in several batches."""
    import os
    from oneprot_amd.metrics import RetrievalMetric
    cases = torch.load(os.path.join(golden_dir, "retrieval.pt"), weights_only=False)
    for name, c in cases.items():
        met = RetrievalMetric()
        o = 0
        for n in c["cuts"]:
            met.update(c["s"][o:o + n].to(DEV), c["m"][o:o + n].to(DEV))
            o += n
        assert met.global_count() == c["s"].shape[0]
        got = met.compute()
        assert set(got) == set(c["expected"]), name
        for k, v in c["expected"].items():
            assert got[k] == v, (name, k, got[k], v)


def test_retrieval_metric_counts_ranks():
    from oneprot_amd.metrics import RetrievalMetric
    g = torch.Generator().manual_seed(15)
    N, D = 300, 32
    s = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=-1)
    m = torch.nn.functional.normalize(s + 0.6 * torch.randn(N, D, generator=g), dim=-1) * (1 / 0.07)
    met = RetrievalMetric()
    for i in range(0, N, 100):
        met.update(s[i:i + 100].to(DEV), m[i:i + 100].to(DEV))
    got = met.compute()
    ref = O.retrieval_metrics(s, m)
    assert set(got) == set(ref)
    for k in ref:
        assert abs(got[k] - ref[k]) <= (1.0 if "median" in k else 0.011), (k, got[k], ref[k])


def test_clip_loss_tensor_logit_scale_stays_on_device_and_gets_a_gradient():
    """ClipLoss.forward(m, s, logit_scale=<tensor>) (what ref oneprot_module.py:142 passes): same value as the python-number path and as the
    oracle, no host read of the scale, and d loss / d scale when the scale requires grad (a CLIP-style learnable temperature handed in)."""
    from src.models.components.loss import ClipLoss
    g = torch.Generator().manual_seed(21)
    B, D = 9, 40
    m = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=-1)
    s = torch.nn.functional.normalize(m + 0.5 * torch.randn(B, D, generator=g), dim=-1)
    mr, sr = m.clone().requires_grad_(True), s.clone().requires_grad_(True)
    log_scale_ref = torch.tensor(1.7, requires_grad=True)
    ref = O.clip_loss(mr, sr, 1.0) * 0 + torch.nn.functional.cross_entropy(log_scale_ref.exp() * mr @ sr.t(), torch.arange(B)) / 2 \
        + torch.nn.functional.cross_entropy(log_scale_ref.exp() * sr @ mr.t(), torch.arange(B)) / 2
    ref.backward()
    md, sd = m.to(DEV).requires_grad_(True), s.to(DEV).requires_grad_(True)
    log_scale = torch.tensor(1.7, device=DEV, requires_grad=True)
    fn = ClipLoss()
    loss = fn(md, sd, log_scale.exp())
    loss.backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item())
    assert_close(md.grad.cpu(), mr.grad, 1e-4, 1e-6, "dm (tensor scale)")
    assert_close(sd.grad.cpu(), sr.grad, 1e-4, 1e-6, "ds (tensor scale)")
    assert abs(log_scale.grad.item() - log_scale_ref.grad.item()) < 1e-4 * abs(log_scale_ref.grad.item()) + 1e-6
    with torch.no_grad():
        assert abs(fn(md, sd, log_scale.exp()).item() - fn(md, sd, float(log_scale.exp())).item()) < 1e-6
    # a scale parameter that lives on the host (a bare tensor the caller never moved): its gradient comes back on the host, same value
    host_scale = torch.tensor(1.7, requires_grad=True)
    md2, sd2 = m.to(DEV).requires_grad_(True), s.to(DEV).requires_grad_(True)
    fn(md2, sd2, host_scale.exp()).backward()
    assert host_scale.grad.device.type == "cpu"
    assert abs(host_scale.grad.item() - log_scale_ref.grad.item()) < 1e-4 * abs(log_scale_ref.grad.item()) + 1e-6
    # at a batch where the reduction spans many work-groups (R*C = 262144 products per logits matrix)
    B2 = 512
    m2 = torch.nn.functional.normalize(torch.randn(B2, D, generator=g), dim=-1)
    s2 = torch.nn.functional.normalize(m2 + 0.5 * torch.randn(B2, D, generator=g), dim=-1)
    lr = torch.tensor(2.3, requires_grad=True)
    r2 = torch.nn.functional.cross_entropy(lr.exp() * m2 @ s2.t(), torch.arange(B2)) / 2 + torch.nn.functional.cross_entropy(lr.exp() * s2 @ m2.t(), torch.arange(B2)) / 2
    r2.backward()
    ld = torch.tensor(2.3, device=DEV, requires_grad=True)
    l2 = fn(m2.to(DEV), s2.to(DEV), ld.exp())
    l2.backward()
    assert abs(l2.item() - r2.item()) < 1e-5 * abs(r2.item())
    assert abs(ld.grad.item() - lr.grad.item()) < 1e-4 * abs(lr.grad.item()) + 1e-6


@pytest.mark.parametrize("B,L,d,with_dx", [(3, 19, 1280, True), (2, 512, 1280, False), (5, 77, 640, True), (2, 1026, 320, True), (3, 130, 64, True),
                                            (2, 33, 2304, False), (1, 1, 1280, True), (2, 40, 4352, True)])
def test_attention1d_pooling_kernels_vs_fp64_torch(B, L, d, with_dx):
    """oneprot_attnpool_fwd / _bwd (ref base_encoder.py:88-103) against the same arithmetic in fp64 torch: widths with 1..17 chunks of 64 float4 columns
    (row groups side by side or not), lengths that are no multiple of the 16 waves / of the 4-row unroll, padding, with and without dx."""
    g = torch.Generator().manual_seed(B * 1000 + L + d)
    x = torch.randn(B, L, d, generator=g)
    w, bias = torch.randn(d, generator=g) * 0.05, torch.tensor([0.2])
    ids = torch.randint(4, 24, (B, L), generator=g)
    if L > 3:
        ids[-1, L // 2:] = 1                                      # padding id 1
    dp = torch.randn(B, d, generator=g)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), bias.double().requires_grad_(True)
    s = (xr * wr).sum(-1) + br
    s = s.masked_fill(ids == 1, float("-inf"))
    a = torch.softmax(s, dim=-1)
    ref = (a[..., None] * xr).sum(1)
    (ref * dp.double()).sum().backward()
    xd, wd, bd, idd, dpd = x.to(DEV), w.to(DEV), bias.to(DEV), ids.to(DEV), dp.to(DEV)
    pooled, attn = torch.empty(B, d, device=DEV), torch.empty(B, L, device=DEV)
    hip.call("oneprot_attnpool_fwd", xd, idd, 1, wd, bd, pooled, attn, B, L, d)
    assert_close(pooled.cpu().double(), ref.detach(), 2e-5, 2e-6, "attnpool pooled")
    assert_close(attn.cpu().double(), a.detach(), 2e-5, 1e-7, "attnpool weights")
    dw, db = torch.empty(d, device=DEV), torch.empty(1, device=DEV)
    dx = torch.empty(B, L, d, device=DEV) if with_dx else None
    ws = torch.empty(hip.query("oneprot_attnpool_bwd_workspace", B, d), dtype=torch.uint8, device=DEV)
    hip.call("oneprot_attnpool_bwd", xd, attn, wd, dpd, dw, db, dx, ws, B, L, d)
    assert_close(dw.cpu().double(), wr.grad, 1e-4, 1e-5 * float(wr.grad.abs().max()) + 1e-7, "attnpool dw")
    assert_close(db.cpu().double(), br.grad, 1e-4, 1e-5 * float(wr.grad.abs().max()) + 1e-6, "attnpool db")
    if with_dx:
        assert_close(dx.cpu().double(), xr.grad, 1e-4, 1e-6, "attnpool dx")


def test_public_pooling_and_normalize_modules():
    """`encoder.pooling` / `encoder.norm` are attributes other code reaches into: their forward must work stand-alone, with autograd
    (inside the encoders the same arithmetic is fused into the final-LayerNorm kernel).  ref base_encoder.py:6-12, 88-126."""
    from src.models.components.base_encoder import MeanPooling, Attention1dPooling, Normalize, CLSTokenPooling
    g = torch.Generator().manual_seed(22)
    B, L, d = 3, 19, 1280
    x = torch.randn(B, L, d, generator=g)
    mask = torch.ones(B, L)
    mask[1, 11:] = 0
    mask[2, 3:] = 0
    tgt = torch.randn(B, d, generator=g)
    # mean
    xr = x.clone().requires_grad_(True)
    ref = O.mean_pool(xr, mask)
    (ref * tgt).sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    got = MeanPooling()(xd, mask.to(DEV))
    (got * tgt.to(DEV)).sum().backward()
    assert_close(got.detach().cpu(), ref.detach(), 1e-5, 1e-6, "mean pool")
    assert_close(xd.grad.cpu(), xr.grad, 1e-5, 1e-7, "mean pool dx")
    assert_close(MeanPooling()(xd.detach()).cpu(), x.mean(1), 1e-5, 1e-6, "mean pool without mask")
    assert torch.equal(CLSTokenPooling()(xd.detach()), xd.detach()[:, 0])
    # attention1d
    pool = Attention1dPooling(d)
    with torch.no_grad():
        pool.layer.weight.normal_(0, 0.05, generator=g)
        pool.layer.bias.fill_(0.2)
    w_ref, b_ref = pool.layer.weight.detach().clone().requires_grad_(True), pool.layer.bias.detach().clone().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    ref = O.attention1d_pool(xr, w_ref, b_ref, mask)
    (ref * tgt).sum().backward()
    pool = pool.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    got = pool(xd, mask.to(DEV))
    (got * tgt.to(DEV)).sum().backward()
    assert_close(got.detach().cpu(), ref.detach(), 1e-4, 1e-5, "attention1d pool")
    assert_close(xd.grad.cpu(), xr.grad, 1e-3, 1e-5, "attention1d dx")
    assert_close(pool.layer.weight.grad.cpu(), w_ref.grad, 1e-3, 1e-4, "attention1d dw")
    # Normalize over the last and over another dimension
    y = torch.randn(5, 7, generator=g)
    assert_close(Normalize(dim=-1)(y.to(DEV)).cpu(), torch.nn.functional.normalize(y, dim=-1), 1e-5, 1e-6, "normalize -1")
    assert_close(Normalize(dim=0)(y.to(DEV)).cpu(), torch.nn.functional.normalize(y, dim=0), 1e-5, 1e-6, "normalize 0")
    with pytest.raises(hip.HipKernelError):
        MeanPooling()(x, mask)                       # CPU tensors: no fallback
