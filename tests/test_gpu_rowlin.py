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
    (10, 2048, 512, False, None, True, True),        # feed-forward w_2 un-split (8 waves x 256 k)
    (30, 256, 768, True, None, False, True),         # decoder, batch 3 x beam 10: 32-row tiles
    (30, 1024, 256, False, None, True, True),        # 32 rows, 8 waves x 128 k
    (30, 512, 2048, True, "relu", False, True),      # LM feed-forward w_1, 32-row tiles
    (20, 256, 41, True, None, False, True),          # output layer: a partial column tile
    (16, 1024, 50, True, "relu", True, True),        # the largest LayerNorm prologue (16 rows)
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
    assert ops.rowlin_ok(x, w, ln=ln)
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


@pytest.mark.parametrize("N,D,ks", [(10, 512, 4), (10, 256, 4), (16, 512, 2), (3, 256, 8), (30, 512, 4), (20, 256, 2)])
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
    if N <= 16:          # (K = 2048 with more than 16 rows exists in slices only)
        assert float((ops.rowlin(t, w2, b2, res=x).double() - ref).abs().max() / ref.abs().max()) < 2e-6
    else:
        assert not ops.rowlin_ok(t, w2) and ops.rowlin_ok(t, w2, ksplit=ks)
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
    # the plans that would not fit their registers do not exist (every variant of the kernel is spill-free): the host is told so
    x32, w1k = torch.randn(30, 1024, device="cuda"), torch.randn(64, 1024, device="cuda")
    assert ops.rowlin_ok(x32, w1k) and not ops.rowlin_ok(x32, w1k, ln=True) and ops.rowlin_ok(x32[:16], w1k, ln=True)
    x2k, w2k = torch.randn(30, 2048, device="cuda"), torch.randn(64, 2048, device="cuda")
    assert not ops.rowlin_ok(x2k, w2k) and ops.rowlin_ok(x2k[:16], w2k) and ops.rowlin_ok(x2k, w2k, ksplit=2)
    with pytest.raises(RuntimeError):
        ops.rowlin(x2k, w2k)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("N,H,dk,nkeys", [(10, 8, 64, 99), (7, 4, 64, 150), (5, 3, 96, 40), (3, 2, 16, 1), (4, 2, 64, 700), (10, 8, 64, 3),
                                          (30, 8, 64, 257)])
def test_tree_attention_step_matches_fp64(N, H, dk, nkeys, mode):
    """softmax(q . K[anc] / sqrt(dk)) . V[anc] per (hypothesis, head) with random ancestor rows, on each launch plan of a small step
    (tavsr_tree_attn_tune - 0: four waves per item with the keys dealt to them, incl. runs of more than 64 keys, empty runs and the
    appended row; 1 / 2: one wave per item: several key passes, the unrolled and the tail part of the value loop), both head-dimension
    lane layouts; with and without this step's own key / value row handed in separately."""
    from tavsr import ops
    from tavsr._lib import lib
    g = torch.Generator(device="cuda").manual_seed(nkeys)
    D = H * dk
    pool_rows = nkeys * N + 5
    kpool, vpool = (torch.randn(pool_rows, D, device="cuda", generator=g) for _ in range(2))
    qkv = torch.randn(N, 3 * D, device="cuda", generator=g)
    q = qkv[:, :D]
    anc = torch.randint(0, max(1, (nkeys - 1) * N), (N, nkeys + 3), device="cuda", generator=g).to(torch.int32)   # (never a last-step row)
    idx = anc[:, :nkeys].long()
    K = kpool.double()[idx].view(N, nkeys, H, dk)
    V = vpool.double()[idx].view(N, nkeys, H, dk)
    sc = torch.einsum("nhd,njhd->nhj", q.double().reshape(N, H, dk), K) / dk ** 0.5
    ref = torch.einsum("nhj,njhd->nhd", torch.softmax(sc, -1), V).reshape(N, D)
    try:
        assert lib().tavsr_tree_attn_tune(mode) == 0
        out = ops.tree_attn_step(q, kpool, vpool, anc, nkeys, H, dk)
        assert float((out.double() - ref).abs().max()) < 1e-5
        # the last key of every hypothesis handed in as this step's own row (and appended to the pools by the launch)
        kp2, vp2 = kpool.clone(), vpool.clone()
        last = (nkeys - 1) * N + torch.arange(N, device="cuda")
        anc2 = anc.clone()
        anc2[:, nkeys - 1] = last.to(torch.int32)
        kp2[last], vp2[last] = float("nan"), float("nan")              # must not be read from the pool
        out2 = ops.tree_attn_step(q, kp2, vp2, anc2, nkeys, H, dk, k_new=qkv[:, D:2 * D], v_new=qkv[:, 2 * D:])
        idx2 = anc2[:, :nkeys].long()
        kf, vf = kpool.clone(), vpool.clone()
        kf[last], vf[last] = qkv[:, D:2 * D], qkv[:, 2 * D:]
        K2 = kf.double()[idx2].view(N, nkeys, H, dk)
        V2 = vf.double()[idx2].view(N, nkeys, H, dk)
        sc2 = torch.einsum("nhd,njhd->nhj", q.double().reshape(N, H, dk), K2) / dk ** 0.5
        ref2 = torch.einsum("nhj,njhd->nhd", torch.softmax(sc2, -1), V2).reshape(N, D)
        assert float((out2.double() - ref2).abs().max()) < 1e-5
        assert torch.equal(kp2[last], qkv[:, D:2 * D]) and torch.equal(vp2[last], qkv[:, 2 * D:])
    finally:
        lib().tavsr_tree_attn_tune(3)


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


@pytest.mark.parametrize("U,K,V,Cn,step,lm", [(1, 10, 41, 15, 3, True), (3, 10, 41, 15, 0, True), (2, 5, 41, 7, 2, False), (2, 16, 64, 24, 1, True),
                                             (1, 4, 29, 29, 2, True)])
def test_one_launch_beam_update_equals_the_launches_it_replaces(U, K, V, Cn, step, lm):
    """tavsr_beam_select_topk on the CTC prefix scores of EVERY token (computed beside the scorers in a captured step) against the
    sequence it replaces - tavsr_log_softmax_rows (accumulate), tavsr_ctc_prefix_step_topk, tavsr_beam_combine_topk - bit for bit: full
    scores, pre-beam candidates, weighted scores, the top-k; and the re-ordered CTC state through tavsr_beam_reorder on either layout."""
    from tavsr import ops
    g = torch.Generator(device="cuda").manual_seed(V + Cn + K)
    N, T, eos = U * K, 37, V - 1
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    logp = torch.log_softmax(r(U, T, V), -1)
    lens = torch.tensor([T, T - 9, T - 20][:U], device="cuda")
    dec = torch.log_softmax(r(N, V) * 2, -1) * 0.9
    z_lm = r(N, V) * 3 if lm else None
    r_prev, s_prev = -torch.rand(N, T, 2, device="cuda", generator=g) * 5, -torch.rand(N, device="cuda", generator=g) * 3
    tok = torch.randint(1, V, (N,), device="cuda", generator=g)
    score = -torch.rand(N, device="cuda", generator=g) * 30
    score[N - 1] = -float("inf")                                       # a slot that has left the beam
    w_lm, w_len, w_ctc = 0.6, 0.5, 0.1
    # the launches of rounds 1-4
    full = dec.clone()
    if lm:
        ops.log_softmax_rows(z_lm, out=full, alpha=w_lm, add=w_len, accumulate=True)
    cand, r_new, psi, psi_abs, eos_s, eos_abs = ops.ctc_prefix_step_topk(logp, lens, r_prev, s_prev, tok, full, Cn, K, step)
    ts, ti, weighted = ops.beam_combine_topk(full, cand, psi, psi_abs, eos_s, eos_abs, s_prev, score, eos, w_ctc, K, keep_weighted=True)
    # every token scored, then one launch
    cand_all = torch.arange(V, device="cuda").repeat(N, 1)
    r_all, psi_all, psi_abs_all, eos_s2, eos_abs2 = ops.ctc_prefix_step(logp, lens, r_prev, s_prev, tok, cand_all, K, step)
    assert torch.equal(eos_s2, eos_s) and torch.equal(eos_abs2, eos_abs)
    ts2, ti2, full2, weighted2, cand2 = ops.beam_select_topk(dec, z_lm, w_lm, w_len if lm else 0.0, psi_all, psi_abs_all, eos_s2, eos_abs2,
                                                             s_prev, score, eos, w_ctc, K, Cn, keep=True)
    assert torch.equal(full2, full) and torch.equal(cand2, cand)
    assert torch.equal(weighted2, weighted)
    assert torch.equal(ti2, ti) and torch.equal(ts2, ts)
    assert torch.equal(psi_abs_all[:, eos], eos_abs)
    # the same hypotheses come out of the re-ordering whichever layout the CTC state has
    steps = 6
    yseq = torch.randint(0, V, (N, steps + 2), device="cuda", generator=g)
    anc = torch.randint(0, 99, (N, steps), device="cuda", generator=g).to(torch.int32)
    step_dev = torch.tensor([step], dtype=torch.int32, device="cuda")
    outs = []
    for cnd, rn, pa in ((cand, r_new, psi_abs), (cand_all, r_all, psi_abs_all)):
        o = (torch.empty(N, T, 2, device="cuda"), torch.empty(N, device="cuda"), torch.empty_like(yseq), torch.empty_like(anc),
             torch.empty(N, dtype=torch.int64, device="cuda"), torch.empty(N, device="cuda"))
        ops.beam_reorder(ti, ts, cnd, rn, pa, yseq, anc, o, K, V, step_dev)
        outs.append(o)
    live = torch.isfinite(ts.view(N))                                  # (slots filled from dead rows carry junk on either path)
    for a, b in zip(*outs):
        assert torch.equal(a[live], b[live])
