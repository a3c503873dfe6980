"""Hybrid CTC/attention beam search + LM.  The oracle restates espnet's BatchBeamSearch and scorers
(parity unpinned: espnet is not installed and the reference holds no decode fixtures); its CTC prefix scorer is pinned
here against torch's CTC loss, and the HIP search is compared with it hypothesis by hypothesis."""
import argparse

import numpy as np
import pytest
import torch

from helpers import TOKENS_EN, asr_conf, rel_err
from oracle import beam_search as BS
from oracle.model import build_asr_oracle, fill_parameters_, synth

LM_KW = dict(pos_enc=None, embed_unit=32, att_unit=64, head=4, unit=128, layer=2, dropout_rate=0.0)


def _oracle_models(seed=5):
    m = build_asr_oracle(asr_conf(num_blocks=2, dec_blocks=2), TOKENS_EN).eval()
    fill_parameters_(m, seed=seed)
    lm = BS.TransformerLMOracle(len(TOKENS_EN), **LM_KW).eval()
    fill_parameters_(lm, seed=seed + 1)
    return m, lm


def test_ctc_prefix_score_of_a_complete_hypothesis_is_the_ctc_log_likelihood():
    """sum of the prefix scorer's increments along y + <eos> == log p_ctc(y | x) == -ctc_loss (torch)."""
    m, _ = _oracle_models()
    x = synth((1, 96, 80), seed=3)
    with torch.no_grad():
        enc, _ = m.encode(x, torch.tensor([96]))
        sc = BS.CTCPrefixScorer(m.ctc, m.eos)
        sc.batch_init_state(enc[0])
        y = [5, 9, 9, 14, 3]
        yseq, state, total = torch.tensor([[m.sos]]), [None], 0.0
        for tok in y + [m.eos]:
            ids = torch.tensor([[tok, 1, 2]]) if tok not in (1, 2) else torch.tensor([[tok, 3, 4]])
            s, st = sc.batch_score_partial(yseq, ids, state, enc[0])
            total += float(s[0, tok])
            state = [sc.select_state(st, 0, tok)]
            yseq = torch.cat([yseq, torch.tensor([[tok]])], dim=1)
        logp = m.ctc.log_softmax(enc).transpose(0, 1)
        ref = -torch.nn.functional.ctc_loss(logp, torch.tensor([y]), torch.tensor([logp.size(0)]), torch.tensor([len(y)]),
                                            blank=0, reduction="sum")
    assert abs(total - float(ref)) < 1e-3 * abs(float(ref))


def test_oracle_beam_search_is_deterministic_and_sorted():
    m, lm = _oracle_models()
    x = synth((1, 120, 80), seed=7)
    with torch.no_grad():
        enc, _ = m.encode(x, torch.tensor([120]))
        bs = BS.build_beam_search(m, lm, beam_size=5, ctc_weight=0.3, lm_weight=0.6, penalty=0.5)
        a, b = bs.forward(enc[0]), bs.forward(enc[0])
    assert [h.yseq.tolist() for h in a] == [h.yseq.tolist() for h in b]
    assert all(a[i].score >= a[i + 1].score for i in range(len(a) - 1))
    assert all(int(h.yseq[0]) == m.sos and int(h.yseq[-1]) == m.eos for h in a)
    # a hypothesis' score is the weighted sum of its scorers' accumulated scores
    w = dict(decoder=0.7, ctc=0.3, lm=0.6, length_bonus=0.5)
    for h in a[:3]:
        assert abs(sum(w[k] * v for k, v in h.scores.items()) - h.score) < 1e-3 * abs(h.score)


@pytest.mark.gpu
@pytest.mark.parametrize("beam,ctc_w,lm_w,pen", [(5, 0.3, 0.6, 0.5), (10, 0.1, 0.6, 0.5), (4, 0.5, 0.0, 0.0),
                                                  (5, 0.0, 0.6, 0.5), (5, 1.0, 0.6, 0.5)])     # pure attention / pure CTC + LM
def test_hip_beam_search_matches_oracle(beam, ctc_w, lm_w, pen):
    from tavsr.inference.beam_search import BatchBeamSearch
    from tavsr.lm.transformer_lm import TransformerLM
    from tavsr.tasks.asr import ASRTask
    m, lm = _oracle_models()
    conf = asr_conf(num_blocks=2, dec_blocks=2)
    conf["token_list"] = TOKENS_EN
    pm = ASRTask.build_model(argparse.Namespace(**conf)).eval()
    fill_parameters_(pm, seed=5)
    plm = TransformerLM(len(TOKENS_EN), **LM_KW).eval()
    fill_parameters_(plm, seed=6)
    assert sorted(plm.state_dict().keys()) == sorted(lm.state_dict().keys())
    pm, plm = pm.cuda(), plm.cuda()
    x = synth((3, 160, 80), seed=7)
    lens = torch.tensor([160, 120, 88])
    with torch.no_grad():
        enc, olens = m.encode(x, lens)
        ref = []
        for u in range(3):
            bs = BS.build_beam_search(m, lm if lm_w else None, beam, ctc_w, lm_w, pen)
            ref.append(bs.forward(enc[u, : int(olens[u])]))
        # LM logits of the product in teacher-forced form == oracle
        toks = synth((2, 7), seed=1, kind="int", lo=1, hi=40)
        assert rel_err(plm(toks.cuda())[0].cpu(), lm(toks)[0]) < 1e-4
        hip = BatchBeamSearch(pm, plm if lm_w else None, beam, ctc_w, lm_w, pen).decode(enc.cuda(), olens.cuda())
    for u in range(3):
        assert len(hip[u]) > 0 and len(ref[u]) > 0
        # best hypothesis: same tokens, same score
        assert hip[u][0][0] == ref[u][0].yseq.tolist(), (u, hip[u][0], ref[u][0].yseq.tolist())
        assert abs(hip[u][0][1] - ref[u][0].score) < 2e-4 * abs(ref[u][0].score)
        # the n-best lists agree (ties aside: compare as score-sorted sets of the top 3)
        top_h = {tuple(h[0]) for h in hip[u][:3]}
        top_r = {tuple(h.yseq.tolist()) for h in ref[u][:3]}
        assert len(top_h & top_r) >= 2, (u, top_h, top_r)


@pytest.mark.gpu
def test_hip_beam_search_with_more_than_32_rows_matches_oracle(monkeypatch):
    """7 utterances x beam 5 = 35 hypothesis rows: every Linear of the step runs on the GEMM kernels (the one-launch small-step
    Linear takes up to 32 rows), the K-split ones finish their rows together with the LayerNorm that follows (tavsr_gemm_ln) - against
    the oracle's one-by-one search, and with the LayerNorm launches kept separate (same best hypotheses, scores within 1e-5)."""
    from tavsr.inference import beam_search as PBS
    from tavsr.lm.transformer_lm import TransformerLM
    from tavsr.tasks.asr import ASRTask
    m, lm = _oracle_models()
    conf = asr_conf(num_blocks=2, dec_blocks=2)
    conf["token_list"] = TOKENS_EN
    pm = ASRTask.build_model(argparse.Namespace(**conf)).eval()
    fill_parameters_(pm, seed=5)
    plm = TransformerLM(len(TOKENS_EN), **LM_KW).eval()
    fill_parameters_(plm, seed=6)
    pm, plm = pm.cuda(), plm.cuda()
    U = 7
    x = synth((U, 128, 80), seed=27)
    lens = torch.tensor([128, 128, 120, 112, 96, 88, 64])
    with torch.no_grad():
        enc, olens = m.encode(x, lens)
        ref = [BS.build_beam_search(m, lm, 5, 0.3, 0.6, 0.5).forward(enc[u, : int(olens[u])]) for u in range(U)]
        outs = []
        for flag in (True, False):
            monkeypatch.setattr(PBS, "LN_IN_EPILOGUE", flag)
            outs.append(PBS.BatchBeamSearch(pm, plm, 5, 0.3, 0.6, 0.5).decode(enc.cuda(), olens.cuda()))
    for u in range(U):
        for hip in outs:
            assert hip[u][0][0] == ref[u][0].yseq.tolist(), (u, hip[u][0], ref[u][0].yseq.tolist())
            assert abs(hip[u][0][1] - ref[u][0].score) < 2e-4 * abs(ref[u][0].score)
        assert outs[0][u][0][0] == outs[1][u][0][0] and abs(outs[0][u][0][1] - outs[1][u][0][1]) < 1e-5 * abs(outs[1][u][0][1])


@pytest.mark.gpu
def test_captured_step_is_reused_for_the_next_batch_of_the_same_shape():
    """a stream of equally long clips: the second ``decode`` refills the first one's buffers and replays its hipGraph - same
    hypotheses and scores as a search object that has never seen another batch; a different shape captures anew"""
    from tavsr.inference.beam_search import BatchBeamSearch
    from tavsr.lm.transformer_lm import TransformerLM
    from tavsr.tasks.asr import ASRTask
    conf = asr_conf(num_blocks=2, dec_blocks=2)
    conf["token_list"] = TOKENS_EN
    pm = ASRTask.build_model(argparse.Namespace(**conf)).eval()
    fill_parameters_(pm, seed=5)
    plm = TransformerLM(len(TOKENS_EN), **LM_KW).eval()
    fill_parameters_(plm, seed=6)
    pm, plm = pm.cuda(), plm.cuda()
    with torch.no_grad():
        batches = []
        for seed, lens in ((7, [160, 120]), (8, [160, 97]), (9, [160, 160]), (10, [120, 120])):
            x = synth((2, max(lens), 80), seed=seed).cuda()
            batches.append(pm.encode(x, torch.tensor(lens).cuda()))
        search = BatchBeamSearch(pm, plm, 10, 0.1, 0.6, 0.5)
        graphs = []
        for enc, olens in batches:
            got = search.decode(enc, olens)
            graphs.append(search._captured["graph"])
            want = BatchBeamSearch(pm, plm, 10, 0.1, 0.6, 0.5).decode(enc, olens)
            for u in range(2):
                assert [h[0] for h in got[u]] == [h[0] for h in want[u]], u
                assert [h[1] for h in got[u]] == [h[1] for h in want[u]], u
    assert graphs[0] is graphs[1] is graphs[2], "same shape: the captured step must be re-used"
    assert graphs[3] is not graphs[2], "another shape: a new capture"


@pytest.mark.gpu
def test_reused_captured_step_follows_a_change_of_the_weights():
    """between two decodes of one shape the parameters change in place (an optimizer step, load_state_dict: periodic
    validation while training).  The re-used captured step must score with the NEW weights - its derived copies (q / k / v weights
    side by side, the LM's input table) are re-derived when a source parameter's version counter has moved - i.e. give what a search
    object built after the change gives."""
    from tavsr.inference.beam_search import BatchBeamSearch
    from tavsr.lm.transformer_lm import TransformerLM
    from tavsr.tasks.asr import ASRTask
    conf = asr_conf(num_blocks=2, dec_blocks=2)
    conf["token_list"] = TOKENS_EN
    pm = ASRTask.build_model(argparse.Namespace(**conf)).eval()
    fill_parameters_(pm, seed=5)
    plm = TransformerLM(len(TOKENS_EN), **LM_KW).eval()
    fill_parameters_(plm, seed=6)
    pm, plm = pm.cuda(), plm.cuda()
    with torch.no_grad():
        x = synth((2, 160, 80), seed=7).cuda()
        enc, olens = pm.encode(x, torch.tensor([160, 120]).cuda())
        search = BatchBeamSearch(pm, plm, 10, 0.1, 0.6, 0.5)
        before = search.decode(enc, olens)
        graph = search._captured["graph"]
        other = TransformerLM(len(TOKENS_EN), **LM_KW)
        fill_parameters_(other, seed=66)
        plm.load_state_dict(other.state_dict())                      # copy_ in place: same storages, new versions
        for l in pm.decoder.decoders:
            l.self_attn.linear_q.weight.mul_(0.5)
        got = search.decode(enc, olens)
        assert search._captured["graph"] is graph
        want = BatchBeamSearch(pm, plm, 10, 0.1, 0.6, 0.5).decode(enc, olens)
    assert [h[0] for h in got[0]] != [h[0] for h in before[0]] or [h[1] for h in got[0]] != [h[1] for h in before[0]]
    for u in range(2):
        assert [h[0] for h in got[u]] == [h[0] for h in want[u]], u
        assert [h[1] for h in got[u]] == [h[1] for h in want[u]], u


@pytest.mark.gpu
def test_captured_encode_equals_the_eager_encode():
    """inference.CapturedEncode: the replayed hipGraph of ``model.encode`` gives what the eager launches give (same kernels: 1e-6),
    for a second batch of the same shape with other lengths, and after a change of shape"""
    from tavsr.inference.beam_search import CapturedEncode
    from tavsr.tasks.asr import ASRTask
    conf = asr_conf(num_blocks=2, dec_blocks=1)
    conf["token_list"] = TOKENS_EN
    pm = ASRTask.build_model(argparse.Namespace(**conf)).eval()
    fill_parameters_(pm, seed=5)
    pm = pm.cuda()
    enc_g = CapturedEncode(pm)
    graphs = []
    with torch.no_grad():
        for seed, T, lens in ((7, 160, [160, 120]), (8, 160, [160, 97]), (9, 120, [120, 64])):
            x, l = synth((2, T, 80), seed=seed).cuda(), torch.tensor(lens).cuda()
            got, gl = enc_g(x, l)
            graphs.append(enc_g._cap["graph"])
            want, wl = pm.encode(x, l)
            assert torch.equal(gl, wl)
            assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())
    assert graphs[0] is graphs[1] and graphs[2] is not graphs[1]
    pm.train()
    with torch.no_grad():
        enc_g(x, l)                           # a model in training mode is encoded eagerly (dropout, batch statistics)
    assert enc_g._cap["graph"] is graphs[2]


@pytest.mark.gpu
@pytest.mark.parametrize("ratio", [0.25, -9, 0.8])
def test_hip_beam_search_length_ratio_matches_oracle(ratio):
    """maxlenratio != 0 (espnet BeamSearch.forward: max(1, int(ratio * T)) tokens, or -ratio tokens, no end detection; the
    last iteration closes every running hypothesis) - per utterance of the batch, against the oracle's one-by-one search."""
    from tavsr.inference.beam_search import BatchBeamSearch
    from tavsr.lm.transformer_lm import TransformerLM
    from tavsr.tasks.asr import ASRTask
    m, lm = _oracle_models()
    conf = asr_conf(num_blocks=2, dec_blocks=2)
    conf["token_list"] = TOKENS_EN
    pm = ASRTask.build_model(argparse.Namespace(**conf)).eval()
    fill_parameters_(pm, seed=5)
    plm = TransformerLM(len(TOKENS_EN), **LM_KW).eval()
    fill_parameters_(plm, seed=6)
    pm, plm = pm.cuda(), plm.cuda()
    x = synth((3, 100, 80), seed=17)
    lens = torch.tensor([100, 72, 48])
    with torch.no_grad():
        enc, olens = m.encode(x, lens)
        ref = [BS.build_beam_search(m, lm, 5, 0.3, 0.6, 0.5).forward(enc[u, : int(olens[u])], maxlenratio=ratio, minlenratio=0.1)
               for u in range(3)]
        hip = BatchBeamSearch(pm, plm, 5, 0.3, 0.6, 0.5, maxlenratio=ratio, minlenratio=0.1).decode(enc.cuda(), olens.cuda())
    for u in range(3):
        T = int(olens[u])
        maxlen = -int(ratio) if ratio < 0 else max(1, int(ratio * T))
        assert len(hip[u]) > 0 and len(ref[u]) > 0
        assert max(len(h[0]) for h in hip[u]) <= maxlen + 2
        assert hip[u][0][0] == ref[u][0].yseq.tolist(), (u, hip[u][0], ref[u][0].yseq.tolist())
        assert abs(hip[u][0][1] - ref[u][0].score) < 2e-4 * abs(ref[u][0].score)
    with pytest.raises(ValueError):            # more tokens than frames: espnet's CTC prefix scorer raises IndexError there
        BatchBeamSearch(pm, plm, 5, 0.3, 0.6, 0.5, maxlenratio=1.5).decode(enc.cuda(), olens.cuda())


@pytest.mark.gpu
def test_graph_replayed_scorer_step_equals_eager_launches(monkeypatch):
    """The captured scorer step (device-side step counter, kv_append, in-place token / ancestor buffers) yields the
    same hypotheses, bit for bit in the scores, as the eager launches with host-side step arguments."""
    from tavsr.inference import beam_search as PBS
    from tavsr.lm.transformer_lm import TransformerLM
    from tavsr.tasks.asr import ASRTask
    conf = asr_conf(num_blocks=2, dec_blocks=2)
    conf["token_list"] = TOKENS_EN
    pm = ASRTask.build_model(argparse.Namespace(**conf)).eval()
    fill_parameters_(pm, seed=5)
    plm = TransformerLM(len(TOKENS_EN), **LM_KW).eval()
    fill_parameters_(plm, seed=6)
    pm, plm = pm.cuda(), plm.cuda()
    x = synth((4, 120, 80), seed=9).cuda()
    lens = torch.tensor([120, 120, 96, 64]).cuda()
    with torch.no_grad():
        enc, olens = pm.encode(x, lens)
    outs = []
    for flag in (True, False):
        monkeypatch.setattr(PBS, "GRAPH_STEP", flag)
        outs.append(PBS.BatchBeamSearch(pm, plm, 6, 0.2, 0.5, 0.3).decode(enc, olens))
    assert outs[0] == outs[1]


@pytest.mark.gpu
def test_kv_append_and_device_step_tree_attention():
    from tavsr import ops
    N, H, dk, steps = 6, 4, 16, 5
    D = H * dk
    kpool, vpool = torch.zeros(steps * N, D).cuda(), torch.zeros(steps * N, D).cuda()
    qkv = [synth((N, 3 * D), seed=30 + i).cuda() for i in range(steps)]
    anc = (torch.arange(N).view(N, 1) + torch.arange(steps).view(1, steps) * N).to(torch.int32).cuda()
    step_dev = torch.zeros(1, dtype=torch.int32).cuda()
    for i in range(steps):
        step_dev.fill_(i)
        ops.kv_append(qkv[i][:, D:2 * D], qkv[i][:, 2 * D:], kpool, vpool, N, steps, step_dev)
        got = ops.tree_attn_step(qkv[i][:, :D], kpool, vpool, anc, steps, H, dk, step_dev=step_dev)
        want = ops.tree_attn_step(qkv[i][:, :D], kpool, vpool, anc, i + 1, H, dk)
        assert torch.equal(got, want)
    assert torch.equal(kpool, torch.cat([q[:, D:2 * D] for q in qkv]))
    assert torch.equal(vpool, torch.cat([q[:, 2 * D:] for q in qkv]))
    # the same steps with the append fused into the attention launch (the step's own row is read from k_new / v_new)
    kp2, vp2 = torch.zeros_like(kpool), torch.zeros_like(vpool)
    for i in range(steps):
        step_dev.fill_(i)
        want = ops.tree_attn_step(qkv[i][:, :D], kpool, vpool, anc, i + 1, H, dk)
        got = ops.tree_attn_step(qkv[i][:, :D], kp2, vp2, anc, steps, H, dk, step_dev=step_dev, k_new=qkv[i][:, D:2 * D],
                                 v_new=qkv[i][:, 2 * D:])
        assert torch.equal(got, want)
    assert torch.equal(kp2, kpool) and torch.equal(vp2, vpool)
    step_dev.fill_(steps)                                    # a step past the pool is dropped, not written
    before = kpool.clone()
    ops.kv_append(qkv[0][:, D:2 * D], qkv[0][:, 2 * D:], kpool, vpool, N, steps, step_dev)
    assert torch.equal(kpool, before)


@pytest.mark.gpu
def test_speech2text_from_waveforms_end_to_end():
    """waveform batch -> log-mel -> encoder -> beam search + LM -> (text, tokens, ids, hyp) as the reference's
    Speech2Text returns them; the best hypothesis equals the oracle's on the oracle's own encoder output."""
    from tavsr.inference.beam_search import Speech2Text
    from tavsr.lm.transformer_lm import TransformerLM
    from tavsr.tasks.asr import ASRTask
    conf = asr_conf(num_blocks=2, dec_blocks=2)
    conf["input_size"] = None
    om = build_asr_oracle(conf, TOKENS_EN).eval()
    fill_parameters_(om, seed=5)
    olm = BS.TransformerLMOracle(len(TOKENS_EN), **LM_KW).eval()
    fill_parameters_(olm, seed=6)
    pconf = asr_conf(num_blocks=2, dec_blocks=2)
    pconf["input_size"] = None
    pconf["token_list"] = TOKENS_EN
    pm = ASRTask.build_model(argparse.Namespace(**pconf)).eval()
    fill_parameters_(pm, seed=5)
    plm = TransformerLM(len(TOKENS_EN), **LM_KW).eval()
    fill_parameters_(plm, seed=6)
    wav = 0.1 * synth((2, 24000), seed=21, kind="uniform")
    lens = torch.tensor([24000, 17600])
    wav[1, 17600:] = 0
    s2t = Speech2Text(pm.cuda(), plm.cuda(), beam_size=5, ctc_weight=0.3, lm_weight=0.6, penalty=0.5, nbest=2)
    res = s2t(wav.cuda(), lens.cuda())
    assert len(res) == 2 and all(1 <= len(r) <= 2 for r in res)
    with torch.no_grad():
        enc, olens = om.encode(wav, lens)
    for u in range(2):
        text, token, token_int, (ys, sc) = res[u][0]
        ref = BS.build_beam_search(om, olm, 5, 0.3, 0.6, 0.5).forward(enc[u, : int(olens[u])])
        assert ys == ref[0].yseq.tolist()
        assert abs(sc - ref[0].score) < 5e-4 * abs(ref[0].score)
        assert token_int == [t for t in ys[1:-1] if t != 0] and len(token) == len(token_int)
        assert text == "".join(token).replace("<space>", " ")


@pytest.mark.gpu
def test_beam_update_kernels_equal_the_torch_expressions():
    """tavsr_beam_combine / tavsr_beam_reorder / tavsr_multi_copy against the elementwise + gather torch code of the eager
    path, bit for bit (separately rounded multiply / adds, first matching candidate column, <eos> candidates)."""
    from tavsr import ops
    torch.manual_seed(0)
    N, V, C, K, T, eos, steps = 24, 41, 9, 6, 15, 40, 12
    U = N // K
    full = (torch.randn(N, V) * 3 - 5).cuda()
    cand = torch.stack([torch.randperm(V)[:C] for _ in range(N)]).cuda()
    cand[5, 2] = eos
    psi, psi_abs = (torch.randn(N, C) * 4 - 20).cuda(), (torch.randn(N, C) * 4 - 40).cuda()
    eos_s, eos_abs, s_prev = (torch.randn(N) * 3 - 10).cuda(), (torch.randn(N) * 3 - 30).cuda(), (torch.randn(N) * 3 - 20).cuda()
    score = (torch.randn(N) * 5 - 30).cuda()
    score[3] = -float("inf")
    is_eos_c = cand == eos
    psi_t = torch.where(is_eos_c, eos_s.unsqueeze(1), psi)
    psi_abs_t = torch.where(is_eos_c, eos_abs.unsqueeze(1), psi_abs)
    ctc_full = torch.full((N, V), -10000000000.0, device="cuda") - s_prev.unsqueeze(1)
    ctc_full[:, eos] = eos_s
    ctc_full.scatter_(1, cand, psi_t)
    want = full + 0.2 * ctc_full + score.unsqueeze(1)
    pa = psi_abs.clone()
    got = ops.beam_combine(full, cand, psi, pa, eos_s, eos_abs, s_prev, score, eos, 0.2)
    assert torch.equal(got, want) and torch.equal(pa, psi_abs_t)
    # ... and the same with the top-k in the launch (distinct random scores: the order is unique)
    pa2 = psi_abs.clone()
    ts, ti, w2 = ops.beam_combine_topk(full, cand, psi, pa2, eos_s, eos_abs, s_prev, score, eos, 0.2, K, keep_weighted=True)
    top_s, top_i = torch.topk(want.view(U, K * V), K, dim=-1)
    assert torch.equal(w2, want) and torch.equal(pa2, psi_abs_t) and torch.equal(ts, top_s) and torch.equal(ti, top_i)
    tie = full.clone()
    tie[0:K] = tie[0:1]                        # six identical rows ... with identical scores: equal candidates, lower index first
    sc2 = score.clone(); sc2[0:K] = sc2[0]; sp2 = s_prev.clone(); sp2[0:K] = sp2[0]; es2 = eos_s.clone(); es2[0:K] = es2[0]
    cd2 = cand.clone(); cd2[0:K] = cd2[0:1]; ps2 = psi.clone(); ps2[0:K] = ps2[0:1]
    ts2, ti2, w3 = ops.beam_combine_topk(tie, cd2, ps2, psi_abs.clone(), es2, eos_abs, sp2, sc2, eos, 0.2, K, keep_weighted=True)
    flat = w3.view(U, K * V)[0]
    order = sorted(range(K * V), key=lambda i: (-float(flat[i]), i))[:K]
    assert ti2[0].tolist() == order and torch.equal(ts2[0], flat[order])
    # an utterance with fewer than K comparable scores (NaN rows of a diverged model) - the indices stay in range
    bad = full.clone()
    bad[0:K] = float("nan")
    bad[0, 3] = -1.0                                                  # one finite score in utterance 0
    ts3, ti3, _ = ops.beam_combine_topk(bad, cand, psi, psi_abs.clone(), eos_s, eos_abs, s_prev, score, eos, 0.2, K, keep_weighted=True)
    assert int(ti3.min()) >= 0 and int(ti3.max()) < K * V
    assert int(ti3[0, 0]) == 3 and bool(torch.isinf(ts3[0, 1:]).all()) and bool((ts3[0, 1:] < 0).all())
    assert torch.equal(ti3[1:], ti[1:]) and torch.equal(ts3[1:], ts[1:])
    # re-ordering
    r_new = torch.randn(N, T, 2, C).cuda()
    yseq = torch.randint(0, V, (N, steps + 2)).cuda()
    anc = torch.randint(0, 1000, (N, steps), dtype=torch.int32).cuda()
    i = 4
    ctr = torch.tensor([i, i + 1], dtype=torch.int64).cuda()
    outs = (torch.empty(N, T, 2).cuda(), torch.empty(N).cuda(), torch.empty_like(yseq), torch.empty_like(anc),
            torch.empty(N, dtype=torch.int64).cuda(), torch.empty(N).cuda())
    ops.beam_reorder(top_i, top_s, cand, r_new, pa, yseq, anc, outs, K, V, ctr.view(torch.int32)[0:1])
    prev = (top_i // V + (torch.arange(U).cuda() * K).view(U, 1)).view(N)
    new_tok = (top_i % V).view(N)
    cidx = (cand[prev] == new_tok.unsqueeze(1)).float().argmax(dim=1)
    y_want = yseq[prev]
    y_want[:, i + 1] = new_tok
    for g, w in zip(outs, (r_new[prev, :, :, cidx], pa[prev, cidx], y_want, anc[prev], new_tok, top_s.view(N))):
        assert torch.equal(g, w)
    dst = [torch.zeros_like(o) for o in outs]
    ops.multi_copy_(dst, list(outs))
    assert all(torch.equal(d, o) for d, o in zip(dst, outs))
    # the same with the head of the next step in the launch, and the commit launch counting the step
    outs2 = tuple(torch.empty_like(o) for o in outs)
    maxl = torch.tensor([20, i + 1, 20, 20], dtype=torch.int32).cuda()          # utterance 1 has used its budget after this token
    ops.beam_reorder(top_i, top_s, cand, r_new, pa, yseq, anc, outs2, K, V, ctr.view(torch.int32)[0:1], maxlen=maxl, eos=eos)
    kill = (new_tok == eos) | (torch.arange(N).cuda() // K == 1)
    a_want = anc[prev].clone()
    a_want[:, i + 1] = torch.arange(N, dtype=torch.int32).cuda() + (i + 1) * N
    s_want = torch.where(kill, torch.full_like(top_s.view(N), -float("inf")), top_s.view(N))
    for g, w in zip(outs2, (r_new[prev, :, :, cidx], pa[prev, cidx], y_want, a_want, new_tok, s_want)):
        assert torch.equal(g, w)
    ops.multi_copy_(dst, list(outs2), inc=ctr)
    assert all(torch.equal(d, o) for d, o in zip(dst, outs2)) and ctr.tolist() == [i + 1, i + 2]


@pytest.mark.gpu
def test_config5_full_av_model_beam10_lm16x512_matches_oracle():
    """BASELINE configs[4] at its own sizes: the full tailored AV-Branchformer (12 layers, Conv3d + ResNet-18 frontend,
    fusion, 6-layer decoder) with beam 10, ctc_weight 0.1 and the 16-layer x 512 x 8-head Transformer LM
    (configs/LM/lm-english.yaml) at lm_weight 0.6, length bonus 0.5 - two 4 s / 2.4 s utterances, HIP search against the
    oracle's (about a minute of host time): best hypothesis and score, n-best overlap."""
    from helpers import AVSR_YAML, avsr_conf
    from oracle.av import build_avsr_oracle
    from tavsr.inference.beam_search import BatchBeamSearch
    from tavsr.lm.transformer_lm import TransformerLM
    from tavsr.tasks.avsr import AVSRTask
    lm_kw = dict(pos_enc=None, embed_unit=128, att_unit=512, head=8, unit=2048, layer=16, dropout_rate=0.0)
    conf = avsr_conf(AVSR_YAML, num_blocks=12, dec_blocks=6)
    m = build_avsr_oracle(conf, TOKENS_EN).eval()
    fill_parameters_(m, seed=77)
    lm = BS.TransformerLMOracle(len(TOKENS_EN), **lm_kw).eval()
    fill_parameters_(lm, seed=78)
    pconf = avsr_conf(AVSR_YAML, num_blocks=12, dec_blocks=6)
    pconf["token_list"] = TOKENS_EN
    pm = AVSRTask.build_model(argparse.Namespace(**pconf)).eval()
    pm.load_state_dict(m.state_dict())
    plm = TransformerLM(len(TOKENS_EN), **lm_kw).eval()
    plm.load_state_dict(lm.state_dict())
    pm, plm = pm.cuda(), plm.cuda()
    audio, video = synth((2, 400, 80), seed=79), synth((2, 100, 88, 88), seed=80)
    alens, vlens = torch.tensor([400, 240]), torch.tensor([100, 60])
    with torch.no_grad():
        enc, olens = m.encode(audio, alens, video, vlens)
        genc, golens = pm.encode(audio.cuda(), alens.cuda(), video.cuda(), vlens.cuda())
        assert torch.equal(golens.cpu(), olens)
        assert float((genc.cpu() - enc).abs().max() / enc.abs().max()) < 1e-4
        nthreads = torch.get_num_threads()
        torch.set_num_threads(min(16, nthreads))     # one-token steps on [10, 512] rows: 128 host threads only add wake-ups
        try:
            ref = [BS.build_beam_search(m, lm, 10, 0.1, 0.6, 0.5).forward(enc[u, : int(olens[u])]) for u in range(2)]
        finally:
            torch.set_num_threads(nthreads)
        hip = BatchBeamSearch(pm, plm, 10, 0.1, 0.6, 0.5).decode(genc, golens)
    for u in range(2):
        assert len(hip[u]) > 0 and len(ref[u]) > 0
        assert hip[u][0][0] == ref[u][0].yseq.tolist(), (u, hip[u][0], ref[u][0].yseq.tolist())
        assert abs(hip[u][0][1] - ref[u][0].score) < 5e-4 * abs(ref[u][0].score)
        top_h = {tuple(h[0]) for h in hip[u][:5]}
        top_r = {tuple(h.yseq.tolist()) for h in ref[u][:5]}
        assert len(top_h & top_r) >= 3, (u, top_h, top_r)
