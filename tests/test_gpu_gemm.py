"""GPU: fp32 MFMA GEMM through the C ABI vs a torch fp32/fp64 reference of the same contraction."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _gemm(A, B, a_k, b_k, M, N, K, bias=None, act=None, alpha=1.0, R=None, want_z=False, DZ=None, dact=None,
          batch=None):
    from tavsr import _lib as L
    d = L.GemmDesc()
    d.M, d.N, d.K, d.a_kmajor, d.b_kmajor = M, N, K, int(a_k), int(b_k)
    nb = 1 if batch is None else batch
    Cout = torch.empty((nb, M, N), device="cuda")
    Z = torch.empty_like(Cout) if want_z else None
    d.A, d.lda = A.data_ptr(), A.stride(-2)
    d.B, d.ldb = B.data_ptr(), B.stride(-2)
    d.C, d.ldc = Cout.data_ptr(), N
    d.nb1, d.nb2 = nb, 1
    d.sA1 = A.stride(0) if batch else 0
    d.sB1 = B.stride(0) if batch else 0
    d.sC1 = M * N
    d.bias = None if bias is None else bias.data_ptr()
    d.act, d.alpha = L.ACT[act], alpha
    d.Z = None if Z is None else Z.data_ptr()
    if R is not None:
        d.R, d.ldr, d.sR1 = R.data_ptr(), N, M * N
    if DZ is not None:
        d.DZ, d.dact = DZ.data_ptr(), L.ACT[dact]
    L.check(L.lib().tavsr_gemm(C.byref(d), L.stream()), "tavsr_gemm")
    return Cout, Z


def _ref_act(z, act):
    if act == "relu":
        return torch.relu(z)
    if act == "swish":
        return z * torch.sigmoid(z)
    if act == "gelu":
        return torch.nn.functional.gelu(z)
    return z


@pytest.mark.parametrize("M,N,K", [(3168, 256, 2048), (3168, 2048, 256), (99, 41, 256), (197, 256, 256),
                                    (64, 64, 32), (1, 1, 1), (130, 70, 41), (3168, 768, 256)])
@pytest.mark.parametrize("mode", ["NT", "NN", "TN"])
def test_gemm_layouts(M, N, K, mode):
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda")
    b = torch.randn(K, N, device="cuda")
    ref = (a.double() @ b.double())
    A = a.t().contiguous() if mode == "TN" else a            # [K,M] kmajor
    B = b if mode in ("NN", "TN") else b.t().contiguous()    # NT: W[N,K]
    out, _ = _gemm(A, B, mode == "TN", mode != "NT", M, N, K)
    err = (out[0].double() - ref).abs().max() / ref.abs().max()
    assert err < 2e-6, (mode, float(err))


def test_gemm_asymmetric_identity():
    """A = I with an asymmetric B catches a transposed C write (cdna guide 3)."""
    n = 96
    A = torch.eye(n, device="cuda")
    B = torch.arange(n * n, device="cuda", dtype=torch.float32).reshape(n, n)  # B[k][n]
    out, _ = _gemm(A, B, False, True, n, n, n)
    assert torch.equal(out[0], B)


@pytest.mark.parametrize("act", [None, "relu", "swish", "gelu"])
def test_gemm_epilogue(act):
    torch.manual_seed(0)
    M, N, K = 300, 200, 128
    a, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / 8
    bias, R = torch.randn(N, device="cuda"), torch.randn(1, M, N, device="cuda")
    out, z = _gemm(a, w, False, False, M, N, K, bias=bias, act=act, alpha=0.5, R=R, want_z=True)
    zr = a.double() @ w.double().t() + bias.double()
    ref = R[0].double() + 0.5 * _ref_act(zr, act)
    assert (z[0].double() - zr).abs().max() < 1e-4
    assert (out[0].double() - ref).abs().max() < 1e-4
    # backward-style epilogue: acc * act'(z)
    zs = torch.randn(1, M, N, device="cuda")
    out2, _ = _gemm(a, w, False, False, M, N, K, DZ=zs, dact=act)
    zz = zs[0].double().requires_grad_(True)
    _ref_act(zz, act).sum().backward()
    ref2 = (a.double() @ w.double().t()) * zz.grad
    assert (out2[0].double() - ref2).abs().max() < 1e-4


def test_gemm_batched_strided():
    """heads addressed in place inside a [B*T, 3*D] QKV buffer (no transposes)."""
    from tavsr import _lib as L
    torch.manual_seed(1)
    Bn, H, T, dk = 3, 4, 37, 64
    D = H * dk
    qkv = torch.randn(Bn, T, 3 * D, device="cuda")
    out = torch.empty(H, Bn, T, T, device="cuda")
    d = L.GemmDesc()
    d.M, d.N, d.K, d.a_kmajor, d.b_kmajor = T, T, dk, 0, 0
    d.A, d.lda = qkv.data_ptr(), 3 * D
    d.B, d.ldb = qkv.data_ptr() + 4 * D, 3 * D
    d.C, d.ldc = out.data_ptr(), T
    d.nb1, d.nb2 = Bn, H
    d.sA1, d.sA2, d.sB1, d.sB2 = T * 3 * D, dk, T * 3 * D, dk
    d.sC1, d.sC2 = T * T, Bn * T * T
    d.alpha = 1.0
    L.check(L.lib().tavsr_gemm(C.byref(d), L.stream()), "tavsr_gemm")
    q = qkv[..., :D].reshape(Bn, T, H, dk).permute(2, 0, 1, 3).double()
    k = qkv[..., D:2 * D].reshape(Bn, T, H, dk).permute(2, 0, 1, 3).double()
    ref = q @ k.transpose(-1, -2)
    assert (out.double() - ref).abs().max() / ref.abs().max() < 2e-6


def test_gemm_rejects_bad_descriptor():
    from tavsr import _lib as L
    d = L.GemmDesc()
    d.M, d.N, d.K = 4, 4, 4
    assert L.lib().tavsr_gemm(C.byref(d), L.stream()) == -1
    assert b"null operand" in L.lib().tavsr_last_error_string()


@pytest.mark.parametrize("M,N,K", [(256, 256, 3168), (2048, 256, 3168), (197, 64, 3168), (41, 256, 1312), (3168, 256, 2048)])
def test_gemm_splitk_paths(M, N, K):
    """few-tile / long-K problems go through the split-K slabs + fused reduce epilogue (ops.gemm gives the workspace)."""
    from tavsr import ops
    torch.manual_seed(3)
    dy, x = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
    got = ops.linear_dw(dy, x, alpha=0.5)
    ref = 0.5 * dy.double().t() @ x.double()
    assert (got.double() - ref).abs().max() / ref.abs().max() < 2e-6
    a, w, b, r = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / 30, torch.randn(N, device="cuda"), torch.randn(M, N, device="cuda")
    y, z = ops.linear(a, w, b, act="swish", alpha=0.5, res=r, save_z=True)
    zr = a.double() @ w.double().t() + b.double()
    assert (z.double() - zr).abs().max() < 1e-4
    assert (y.double() - (r.double() + 0.5 * zr * torch.sigmoid(zr))).abs().max() < 1e-4


def test_gemm_unaligned_tail_vector_path():
    """K (or M) not a multiple of 4 with 16-byte aligned rows: vector loads + predicated tail."""
    from tavsr import ops
    torch.manual_seed(4)
    T, W, dk = 99, 197, 64
    sk = torch.randn(T, 200, device="cuda")          # padded rows, W valid columns
    p = torch.randn(W, dk, device="cuda")
    out = ops.empty(T, dk, like=sk)
    ops.gemm(T, dk, W, sk, 200, p, dk, out, dk, b_kmajor=True)
    ref = sk[:, :W].double() @ p.double()
    assert (out.double() - ref).abs().max() / ref.abs().max() < 2e-6
    out2 = ops.empty(W, dk, like=sk)                  # TN with M = W = 197 (row direction tail)
    q = torch.randn(T, dk, device="cuda")
    ops.gemm(W, dk, T, sk, 200, q, dk, out2, dk, a_kmajor=True, b_kmajor=True)
    ref2 = sk[:, :W].double().t() @ q.double()
    assert (out2.double() - ref2).abs().max() / ref2.abs().max() < 2e-6


@pytest.mark.parametrize("M,N,K", [(256, 256, 3168), (2048, 256, 3168), (768, 256, 3168), (41, 256, 1312), (256, 4864, 3168),
                                    (256, 2304, 6016)])
def test_gemm_fused_bias_grad(M, N, K):
    """dW = alpha dY^T X and db = alpha dY.sum(0) from ONE launch (a_rowsum), split-K and plain paths."""
    from tavsr import ops
    torch.manual_seed(5)
    dy, x = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
    gw, gb = ops.linear_dw(dy, x, alpha=0.5, bias_grad=True)
    ref_w = 0.5 * dy.double().t() @ x.double()
    ref_b = 0.5 * dy.double().sum(0)
    # fp32 accumulation over K = 3168..6016 terms (any order): a few 1e-6 of the largest output
    assert (gw.double() - ref_w).abs().max() / ref_w.abs().max() < 5e-6
    assert (gb.double() - ref_b).abs().max() / ref_b.abs().max() < 5e-6


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize("mode", ["NT", "NN", "TN"])
def test_gemm_every_tile_config(cfg, mode):
    """each tile configuration of the planner's table, forced, incl. ragged edges and a forced K split"""
    from tavsr import ops
    torch.manual_seed(cfg)
    M, N, K = 300, 200, 416   # N % 4 == 0, K % 32 == 0: the LDS-DMA kernels take it; cfg 9 = predicated fallback
    a, b = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
    A = a.t().contiguous() if mode == "TN" else a
    B = b if mode != "NT" else b.t().contiguous()
    ref = a.double() @ b.double()
    for ns in (1, 3):
        out = torch.zeros(M, N, device="cuda")
        ops.gemm(M, N, K, A, A.stride(0), B, B.stride(0), out, N, a_kmajor=mode == "TN", b_kmajor=mode != "NT",
                 force=(cfg, ns))
        assert (out.double() - ref).abs().max() / ref.abs().max() < 2e-6, (cfg, mode, ns)


@pytest.mark.parametrize("M,N,K,ws_cap", [(99, 64, 99, None), (64, 99 + 1, 100, None), (200, 132, 77, None), (99, 64, 197, None),
                                           (3168, 256, 99, None),
                                           # K tail AND a K split (few tiles, K >= 512): last slice ends at K, slab epilogue
                                           (128, 256, 2052, None), (256, 256, 4099, None), (256, 256, 3170, None),
                                           # ... and the planner's fallback when the caller's workspace cannot hold the slabs
                                           (128, 256, 2052, 4096)])
@pytest.mark.parametrize("mode", ["NT", "NN", "TN", "TT"])
def test_gemm_k_tail_on_the_lds_dma_kernel(M, N, K, ws_cap, mode):
    """K % 32 != 0 with 16-byte aligned rows takes the tail variant of the fast kernel: chunks past K come from a zero
    page and the 1..3 elements between K and the next multiple of 4 of a k-contiguous operand (here NaN) are zeroed in
    LDS - the result equals the fp64 product over exactly K terms."""
    from tavsr import ops
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda")
    b = torch.randn(K, N, device="cuda")
    ref = a.double() @ b.double()
    a_km, b_km = mode[0] == "T", mode[1] == "N"
    k4, m4, n4 = (K + 3) // 4 * 4, (M + 3) // 4 * 4, (N + 3) // 4 * 4
    if a_km:                                            # A given as [K, ld >= M]
        A = torch.full((K, m4), float("nan"), device="cuda")
        A[:, :M] = a.t()
        lda = m4
    else:                                               # A as [M, ld >= roundup4(K)], NaN in the row padding
        A = torch.full((M, k4), float("nan"), device="cuda")
        A[:, :K] = a
        lda = k4
    if b_km:
        B = torch.full((K, n4), float("nan"), device="cuda")
        B[:, :N] = b
        ldb = n4
    else:
        B = torch.full((N, k4), float("nan"), device="cuda")
        B[:, :K] = b.t()
        ldb = k4
    if (a_km and M % 4) or (b_km and N % 4):
        pytest.skip("k-major operands of the vector kernels need M / N % 4 == 0")
    out = torch.empty(M, N, device="cuda")
    ops.gemm(M, N, K, A, lda, B, ldb, out, N, a_kmajor=a_km, b_kmajor=b_km, ws_cap=ws_cap)
    assert bool(torch.isfinite(out).all())
    err = (out.double() - ref).abs().max() / ref.abs().max()
    assert err < (2e-6 if K < 512 else 5e-6), (mode, float(err))


@pytest.mark.parametrize("M,N,K,split", [(3168, 2048, 256, False), (3168, 256, 2048, True), (100, 64, 96, False), (777, 256, 1056, True)])
def test_gemm_epilogue_dropout_equals_the_standalone_mask(M, N, K, split):
    """tavsr_gemm with drop_p: the same mask (bit for bit) as tavsr_dropout on the contiguous [M, N] result with the same
    token - in the in-kernel epilogue and in the split-K epilogue kernel; forward (bias, activation, residual) and backward
    (mask * act'(z)) forms."""
    from tavsr import ops
    torch.manual_seed(M + N)
    x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / K ** 0.5, torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda")
    ops.manual_seed(17)
    out, z, tok = ops.linear_drop(x, w, b, 0.3, act="swish", alpha=0.5, res=res, save_z=True)
    plain, z2 = ops.linear(x, w, b, act="swish", save_z=True)
    assert torch.equal(z, z2)
    want = res + 0.5 * ops.dropout(plain, 0.3, token=tok)[0]
    assert float((out - want).abs().max()) <= 1e-6 * float(want.abs().max())
    kept = float(((out - res) != 0).float().mean())
    assert abs(kept - 0.7) < 0.01
    # backward form: dx = mask * (dy @ w2) * act'(z), w2 [N2, K2] -> [M, K2]
    dy = torch.randn(M, N, device="cuda")
    w2 = torch.randn(N, K, device="cuda") / N ** 0.5
    zz = torch.randn(M, K, device="cuda")
    tok2 = ops._new_token(0.2, M * K, dy.device)
    got = ops.linear_dx_drop(dy, w2, tok2, alpha=0.5, DZ=zz, dact="swish")
    ref = ops.dropout_act_bwd(ops.linear_dx(dy, w2, alpha=0.5), zz, "swish", tok2)
    assert float((got - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


@pytest.mark.parametrize("M,N,K,act,res", [(640, 512, 2048, None, True), (640, 256, 2048, None, True), (640, 512, 512, None, True),
                                           (640, 256, 256, None, True), (77, 256, 2048, "relu", False), (640, 1024, 512, None, True),
                                           (35, 128, 2048, None, True)])
def test_gemm_ln_result_and_its_layernorm_from_the_launch_that_finishes_the_rows(M, N, K, act, res):
    """tavsr_gemm_ln: y = res + act(x W^T + b) and LayerNorm(y) (eps 1e-12) - where K is split over workgroups the slab sum, the
    epilogue and the LayerNorm are one launch, else the LayerNorm is taken by one launch behind the GEMM - against fp64, and y bit-equal
    to the plain tavsr_gemm of the same problem (same split plan, same slab order, same epilogue arithmetic)."""
    from tavsr import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x, w, b = r(M, K), r(N, K) / K ** 0.5, r(N)
    rs = r(M, N) if res else None
    gam, bet = r(N).abs() + 0.5, r(N)
    y, n = ops.linear(x, w, b, act=act, res=rs, ln=(gam, bet, 1e-12))
    y0 = ops.linear(x, w, b, act=act, res=rs)
    assert torch.equal(y, y0)
    ref = x.double() @ w.double().t() + b.double()
    if act == "relu":
        ref = ref.relu()
    if res:
        ref = ref + rs.double()
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    nref = torch.nn.functional.layer_norm(ref, (N,), gam.double(), bet.double(), 1e-12)
    assert float((n.double() - nref).abs().max() / nref.abs().max()) < 5e-6
    n0 = ops.layernorm_fwd(y0, gam, bet, 1e-12, save=False)[0]
    assert float((n - n0).abs().max() / n0.abs().max()) < 2e-6
