"""GPU: tavsr_rowlin (csrc/decode.hip) - LayerNorm + Linear + bias + activation + residual of a one-token scorer step in
one launch - against the torch fp64 expression of espnet's decoder_layer / encoder_layer cache branch
(x + linear(norm(x)) pieces of TransformerDecoder.forward_one_step and TransformerLM.batch_score)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(x, w, b, ln, act, res):
    x = x.double()
    if ln is not None:
        x = torch.nn.functional.layer_norm(x, (x.shape[1],), ln[0].double(), ln[1].double(), ln[2])
    y = x @ w.double().t()
    if b is not None:
        y = y + b.double()
    if act == "relu":
        y = torch.relu(y)
    elif act == "swish":
        y = y * torch.sigmoid(y)
    if res is not None:
        y = y + res.double()
    return y


@pytest.mark.parametrize("N,K,Nout,ln,act,res,bias", [
    (10, 512, 1536, True, None, False, True),        # LM: LayerNorm + q/k/v projection, batch 1 x beam 10 (16-row tiles)
    (10, 512, 512, False, None, True, True),         # attention output + residual
    (10, 512, 2048, True, "relu", False, True),      # feed-forward w_1
    (10, 2048, 512, False, None, True, True),        # feed-forward w_2 (16 waves x 128 k)
    (30, 256, 768, True, None, False, True),         # decoder, batch 3 x beam 10: 32-row tiles
    (30, 2048, 256, False, None, True, True),
    (20, 256, 41, True, None, False, True),          # output layer: a partial column tile
    (16, 1024, 50, True, "relu", True, True),        # the largest LayerNorm prologue
    (1, 128, 512, False, "swish", False, False),     # K = 128: two waves; a single row
    (32, 64, 70, True, "relu", True, False),         # one wave
])
def test_rowlin_matches_fp64(N, K, Nout, ln, act, res, bias):
    from tavsr import ops
    g = torch.Generator(device="cuda").manual_seed(N * 7 + K)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x, w = r(N, K) * 2 + 0.3, r(Nout, K) / K ** 0.5
    b = r(Nout) if bias else None
    lnp = (r(K).abs() + 0.5, r(K), 1e-12) if ln else None
    rs = r(N, Nout) if res else None
    assert ops.rowlin_ok(x, w)
    y = ops.rowlin(x, w, b, ln=lnp, act=act, res=rs)
    ref = _ref(x, w, b, lnp, act, rs)
    err = (y.double() - ref).abs().max() / ref.abs().max()
    assert float(err) < 2e-6, float(err)


def test_rowlin_gathers_embedding_rows_and_may_write_over_its_residual():
    from tavsr import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    table = torch.randn(41, 128, device="cuda", generator=g)
    w, b = torch.randn(512, 128, device="cuda", generator=g) / 11, torch.randn(512, device="cuda", generator=g)
    tok = torch.randint(0, 41, (30,), device="cuda", generator=g)
    y = ops.rowlin(table, w, b, gather=tok)
    ref = _ref(table[tok], w, b, None, None, None)
    assert float((y.double() - ref).abs().max()) < 1e-5
    # strided windows of one buffer as input, residual aliasing the output
    buf = torch.randn(30, 3 * 128, device="cuda", generator=g)
    res = torch.randn(30, 512, device="cuda", generator=g)
    ref = _ref(buf[:, 128:256], w, b, None, None, res)
    out = ops.rowlin(buf[:, 128:256], w, b, res=res, out=res)
    assert out.data_ptr() == res.data_ptr()
    assert float((out.double() - ref).abs().max()) < 1e-5


@pytest.mark.parametrize("N,D,ks", [(10, 512, 4), (10, 256, 4), (16, 512, 2), (3, 256, 8)])
def test_rowlin_k_split_leaves_partial_tensors_the_next_launches_add_while_loading(N, D, ks):
    """tavsr_rowlin_parts: the closing projection of a feed-forward block (2048 -> D) as ``ks`` K slices, then the two consumers of
    its result in a scorer step - LayerNorm + Linear on the rows, and a Linear that takes them as its residual - against fp64"""
    from tavsr import ops
    g = torch.Generator(device="cuda").manual_seed(N + D + ks)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    t, w2, b2, x = r(N, 2048).relu(), r(D, 2048) / 45, r(D), r(N, D)
    parts = ops.rowlin(t, w2, b2, res=x, ksplit=ks)
    assert isinstance(parts, ops.RowParts) and parts.t.shape == (ks, N, D)
    ref = _ref(t, w2, b2, None, None, x)
    assert float((parts.t.double().sum(0) - ref).abs().max() / ref.abs().max()) < 2e-6
    assert float((ops.rowlin(t, w2, b2, res=x).double() - ref).abs().max() / ref.abs().max()) < 2e-6
    # consumer 1: LayerNorm + Linear of the summed rows
    w, b, lnp = r(3 * D, D) / D ** 0.5, r(3 * D), (r(D).abs() + 0.5, r(D), 1e-12)
    y = ops.rowlin(parts, w, b, ln=lnp)
    ref1 = _ref(ref.float(), w, b, lnp, None, None)
    assert float((y.double() - ref1).abs().max() / ref1.abs().max()) < 5e-6
    # consumer 2: the summed rows as the residual of another Linear
    a, wo, bo = r(N, D), r(D, D) / D ** 0.5, r(D)
    y2 = ops.rowlin(a, wo, bo, res=parts)
    ref2 = _ref(a, wo, bo, None, None, None) + ref
    assert float((y2.double() - ref2).abs().max() / ref2.abs().max()) < 2e-6
    with pytest.raises(RuntimeError):       # an activation is not additive over K slices
        ops.rowlin(t, w2, b2, act="relu", ksplit=ks)


def test_rowlin_rejects_what_it_cannot_do():
    from tavsr import ops
    w = torch.randn(8, 96, device="cuda")
    assert not ops.rowlin_ok(torch.randn(4, 96, device="cuda"), w)              # K not a supported size
    assert not ops.rowlin_ok(torch.randn(33, 128, device="cuda"), torch.randn(8, 128, device="cuda"))   # more than 32 rows
    with pytest.raises(RuntimeError):
        ops.rowlin(torch.randn(4, 96, device="cuda"), w)
    with pytest.raises(RuntimeError):
        ops.rowlin(torch.randn(33, 128, device="cuda"), torch.randn(8, 128, device="cuda"))


@pytest.mark.parametrize("N,H,dk,nkeys", [(10, 8, 64, 99), (7, 4, 64, 150), (5, 3, 96, 40), (3, 2, 16, 1)])
def test_tree_attention_step_matches_fp64(N, H, dk, nkeys):
    """softmax(q . K[anc] / sqrt(dk)) . V[anc] per (hypothesis, head) with random ancestor rows (several key passes, the
    unrolled and the tail part of the value loop, both head-dimension lane layouts)."""
    from tavsr import ops
    g = torch.Generator(device="cuda").manual_seed(nkeys)
    D = H * dk
    pool_rows = nkeys * N + 5
    kpool, vpool = (torch.randn(pool_rows, D, device="cuda", generator=g) for _ in range(2))
    q = torch.randn(N, 3 * D, device="cuda", generator=g)[:, :D]
    anc = torch.randint(0, pool_rows, (N, nkeys + 3), device="cuda", generator=g).to(torch.int32)
    out = ops.tree_attn_step(q, kpool, vpool, anc, nkeys, H, dk)
    idx = anc[:, :nkeys].long()
    K = kpool.double()[idx].view(N, nkeys, H, dk)
    V = vpool.double()[idx].view(N, nkeys, H, dk)
    sc = torch.einsum("nhd,njhd->nhj", q.double().reshape(N, H, dk), K) / dk ** 0.5
    ref = torch.einsum("nhj,njhd->nhd", torch.softmax(sc, -1), V).reshape(N, D)
    assert float((out.double() - ref).abs().max()) < 1e-5


@pytest.mark.parametrize("U,K,V,Cn,step", [(3, 5, 41, 7, 0), (2, 10, 41, 15, 4), (1, 4, 300, 64, 2), (2, 3, 41, 41, 1)])
def test_prebeam_inside_ctc_prefix_step_equals_topk_then_prefix_step(U, K, V, Cn, step):
    """tavsr_ctc_prefix_step_topk = torch.topk(full, C) (descending; no ties in random scores) followed by
    tavsr_ctc_prefix_step on those candidates: same candidate lists, bit-identical forward variables and scores."""
    from tavsr import ops
    g = torch.Generator(device="cuda").manual_seed(V + Cn)
    N, T = U * K, 37
    logp = torch.log_softmax(torch.randn(U, T, V, device="cuda", generator=g), -1)
    lens = torch.tensor([T, T - 9, T - 20][:U], device="cuda")
    full = torch.randn(N, V, device="cuda", generator=g)
    r_prev = -torch.rand(N, T, 2, device="cuda", generator=g) * 5
    s_prev = -torch.rand(N, device="cuda", generator=g) * 3
    tok = torch.randint(1, V, (N,), device="cuda", generator=g)
    cand0 = torch.topk(full, Cn, dim=-1)[1]
    want = ops.ctc_prefix_step(logp, lens, r_prev, s_prev, tok, cand0, K, step)
    got = ops.ctc_prefix_step_topk(logp, lens, r_prev, s_prev, tok, full, Cn, K, step)
    assert torch.equal(got[0], cand0)
    for a, b in zip(got[1:], want):
        assert torch.equal(a, b)
