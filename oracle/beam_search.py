"""CPU restatement of the reference's decode path: hybrid CTC/attention beam search with LM scoring.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/ and bench.py's cpu_baseline leg, never by the
product.

The reference composes espnet classes (src/inference/avsr_inference.py:141-304: scorers {decoder, CTCPrefixScorer,
LengthBonus, lm}; weights :249-255; ``BeamSearch(..., pre_beam_score_key="full")`` class-swapped to ``BatchBeamSearch``
:277-304; called at :449-518 with ``maxlenratio``/``minlenratio``).  espnet==202402 is not vendored and not installed
here, so the classes below restate its published source (espnet/nets/batch_beam_search.py, beam_search.py,
scorers/{ctc,length_bonus}.py, ctc_prefix_score.py:CTCPrefixScoreTH, e2e_asr_common.py:end_detect,
espnet2/asr/decoder/transformer_decoder.py:forward_one_step/batch_score, espnet2/lm/transformer_lm.py,
transformer/{encoder,encoder_layer}.py).  PARITY UNPINNED for these leaves: the reference holds no decode fixtures and
the espnet package cannot be imported here; the product is compared against this restatement.
"""
from __future__ import annotations

import math
from typing import Dict, List, NamedTuple, Optional

import numpy as np
import torch
import torch.nn as nn

from . import leaves as L


# --------------------------------------------------------------------------------------------------------------------
# espnet2/lm/transformer_lm.py + transformer/encoder.py (input_layer="linear") + transformer/encoder_layer.py
# configured by configs/LM/lm-english.yaml (pos_enc: null, embed 128, att 512, 8 heads, 2048 units, 16 layers)
# --------------------------------------------------------------------------------------------------------------------
class EncoderLayer(nn.Module):
    def __init__(self, size, self_attn, feed_forward, dropout_rate):
        super().__init__()
        self.self_attn, self.feed_forward = self_attn, feed_forward
        self.norm1, self.norm2 = L.LayerNorm(size), L.LayerNorm(size)
        self.dropout = nn.Dropout(dropout_rate)
        self.size = size

    def forward(self, x, mask, cache=None):
        residual = x
        x = self.norm1(x)
        if cache is None:
            x_q = x
        else:
            assert cache.shape == (x.shape[0], x.shape[1] - 1, self.size)
            x_q, residual = x[:, -1:, :], residual[:, -1:, :]
            mask = None if mask is None else mask[:, -1:, :]
        x = residual + self.dropout(self.self_attn(x_q, x, x, mask))
        residual = x
        x = residual + self.dropout(self.feed_forward(self.norm2(x)))
        if cache is not None:
            x = torch.cat([cache, x], dim=1)
        return x, mask


class _LMEncoder(nn.Module):
    def __init__(self, idim, attention_dim, attention_heads, linear_units, num_blocks, dropout_rate, pos_enc: Optional[str]):
        super().__init__()
        pos = nn.Sequential() if pos_enc is None else L.PositionalEncoding(attention_dim, dropout_rate)
        self.embed = nn.Sequential(nn.Linear(idim, attention_dim), L.LayerNorm(attention_dim), nn.Dropout(dropout_rate),
                                   nn.ReLU(), pos)
        self.encoders = nn.ModuleList([
            EncoderLayer(attention_dim, L.MultiHeadedAttention(attention_heads, attention_dim, 0.0),
                         L.PositionwiseFeedForward(attention_dim, linear_units, dropout_rate), dropout_rate)
            for _ in range(num_blocks)])
        self.after_norm = L.LayerNorm(attention_dim)

    def forward(self, xs, masks):
        xs = self.embed(xs)
        for e in self.encoders:
            xs, masks = e(xs, masks)
        return self.after_norm(xs), masks

    def forward_one_step(self, xs, masks, cache=None):
        xs = self.embed(xs)
        if cache is None:
            cache = [None] * len(self.encoders)
        new_cache = []
        for c, e in zip(cache, self.encoders):
            xs, masks = e(xs, masks, cache=c)
            new_cache.append(xs)
        return self.after_norm(xs), masks, new_cache


class TransformerLMOracle(nn.Module):
    def __init__(self, vocab_size, pos_enc=None, embed_unit=128, att_unit=256, head=2, unit=1024, layer=4, dropout_rate=0.5):
        super().__init__()
        assert pos_enc in (None, "sinusoidal")
        self.embed = nn.Embedding(vocab_size, embed_unit)
        self.encoder = _LMEncoder(embed_unit, att_unit, head, unit, layer, dropout_rate, pos_enc)
        self.decoder = nn.Linear(att_unit, vocab_size)

    def _target_mask(self, ys_in_pad):
        ys_mask = ys_in_pad != 0
        m = L.subsequent_mask(ys_mask.size(-1), device=ys_mask.device).unsqueeze(0)
        return ys_mask.unsqueeze(-2) & m

    def forward(self, input, hidden=None):
        h, _ = self.encoder(self.embed(input), self._target_mask(input))
        return self.decoder(h), None

    def batch_init_state(self, x):
        return None

    def batch_score(self, ys, states, xs):
        n_batch, n_layers = len(ys), len(self.encoder.encoders)
        batch_state = None if states[0] is None else [torch.stack([states[b][i] for b in range(n_batch)]) for i in range(n_layers)]
        h, _, states = self.encoder.forward_one_step(self.embed(ys), self._target_mask(ys), cache=batch_state)
        logp = self.decoder(h[:, -1]).log_softmax(dim=-1)
        return logp, [[states[i][b] for i in range(n_layers)] for b in range(n_batch)]

    def select_state(self, state, i, new_id=None):
        return None if state is None else state[i]


# --------------------------------------------------------------------------------------------------------------------
# scorers
# --------------------------------------------------------------------------------------------------------------------
class DecoderScorer:
    """espnet2 TransformerDecoder.forward_one_step / batch_score on the oracle decoder (leaves.TransformerDecoder)."""

    def __init__(self, decoder: L.TransformerDecoder):
        self.d = decoder

    def batch_init_state(self, x):
        return None

    def forward_one_step(self, tgt, tgt_mask, memory, cache=None):
        d = self.d
        x = d.embed(tgt)
        if cache is None:
            cache = [None] * len(d.decoders)
        new_cache = []
        for c, layer in zip(cache, d.decoders):
            x, tgt_mask, memory, _ = layer(x, tgt_mask, memory, None, cache=c)
            new_cache.append(x)
        y = d.after_norm(x[:, -1]) if d.normalize_before else x[:, -1]
        return torch.log_softmax(d.output_layer(y), dim=-1), new_cache

    def batch_score(self, ys, states, xs):
        n_batch, n_layers = len(ys), len(self.d.decoders)
        batch_state = None if states[0] is None else [torch.stack([states[b][i] for b in range(n_batch)]) for i in range(n_layers)]
        ys_mask = L.subsequent_mask(ys.size(-1), device=xs.device).unsqueeze(0)
        logp, states = self.forward_one_step(ys, ys_mask, xs, cache=batch_state)
        return logp, [[states[i][b] for i in range(n_layers)] for b in range(n_batch)]

    def select_state(self, state, i, new_id=None):
        return None if state is None else state[i]


class LengthBonus:
    def __init__(self, n_vocab):
        self.n = n_vocab

    def batch_init_state(self, x):
        return None

    def batch_score(self, ys, states, xs):
        return torch.tensor([1.0], device=xs.device, dtype=xs.dtype).expand(ys.shape[0], self.n), None

    def select_state(self, state, i, new_id=None):
        return None


class CTCPrefixScoreTH:
    """espnet/nets/ctc_prefix_score.py:CTCPrefixScoreTH without attention windowing (margin 0)."""

    def __init__(self, x, xlens, blank, eos):
        self.logzero = -10000000000.0
        self.blank, self.eos = blank, eos
        self.batch, self.input_length, self.odim = x.size(0), x.size(1), x.size(2)
        self.dtype, self.device = x.dtype, x.device
        for i, l in enumerate(xlens):
            if l < self.input_length:
                x[i, l:, :] = self.logzero
                x[i, l:, blank] = 0
        xn = x.transpose(0, 1)                                   # (T, B, O)
        xb = xn[:, :, self.blank].unsqueeze(2).expand(-1, -1, self.odim)
        self.x = torch.stack([xn, xb])                           # (2, T, B, O)
        self.end_frames = torch.as_tensor(xlens) - 1
        self.idx_bo = (torch.arange(self.batch) * self.odim).unsqueeze(1)

    def __call__(self, y, state, scoring_ids):
        output_length = len(y[0]) - 1
        last_ids = [int(yi[-1]) for yi in y]
        n_bh = len(last_ids)
        n_hyps = n_bh // self.batch
        snum = scoring_ids.size(-1)
        if state is None:
            r_prev = torch.full((self.input_length, 2, self.batch, n_hyps), self.logzero, dtype=self.dtype)
            r_prev[:, 1] = torch.cumsum(self.x[0, :, :, self.blank], 0).unsqueeze(2)
            r_prev = r_prev.view(-1, 2, n_bh)
            s_prev = 0.0
        else:
            r_prev, s_prev = state[0], state[1]
        scoring_idmap = torch.full((n_bh, self.odim), -1, dtype=torch.long)
        scoring_idmap[torch.arange(n_bh).view(-1, 1), scoring_ids] = torch.arange(snum)
        scoring_idx = (scoring_ids + self.idx_bo.repeat(1, n_hyps).view(-1, 1)).view(-1)
        x_ = torch.index_select(self.x.view(2, -1, self.batch * self.odim), 2, scoring_idx).view(2, -1, n_bh, snum)
        r = torch.full((self.input_length, 2, n_bh, snum), self.logzero, dtype=self.dtype)
        if output_length == 0:
            r[0, 0] = x_[0, 0]
        r_sum = torch.logsumexp(r_prev, 1)
        log_phi = r_sum.unsqueeze(2).repeat(1, 1, snum)
        for idx in range(n_bh):
            pos = scoring_idmap[idx, last_ids[idx]]
            if pos >= 0:
                log_phi[:, idx, pos] = r_prev[:, 1, idx]
        start, end = max(output_length, 1), self.input_length
        for t in range(start, end):
            rp = r[t - 1]
            rr = torch.stack([rp[0], log_phi[t - 1], rp[0], rp[1]]).view(2, 2, n_bh, snum)
            r[t] = torch.logsumexp(rr, 1) + x_[:, t]
        log_phi_x = torch.cat((log_phi[0].unsqueeze(0), log_phi[:-1]), dim=0) + x_[0]
        log_psi = torch.full((n_bh, self.odim), self.logzero, dtype=self.dtype)
        log_psi_ = torch.logsumexp(torch.cat((log_phi_x[start:end], r[start - 1, 0].unsqueeze(0)), dim=0), dim=0)
        for si in range(n_bh):
            log_psi[si, scoring_ids[si]] = log_psi_[si]
        for si in range(n_bh):
            log_psi[si, self.eos] = r_sum[self.end_frames[si // n_hyps], si]
        log_psi[:, self.blank] = self.logzero
        return (log_psi - s_prev), (r, log_psi, 0, 0, scoring_idmap)


class CTCPrefixScorer:
    """espnet/nets/scorers/ctc.py (batch path)."""

    def __init__(self, ctc, eos):
        self.ctc, self.eos, self.impl = ctc, eos, None

    def batch_init_state(self, x):
        logp = self.ctc.log_softmax(x.unsqueeze(0))
        self.impl = CTCPrefixScoreTH(logp.detach().clone(), torch.tensor([logp.size(1)]), 0, self.eos)
        return None

    def batch_score_partial(self, y, ids, state, x):
        batch_state = None
        if state[0] is not None:
            batch_state = (torch.stack([s[0] for s in state], dim=2), torch.stack([s[1] for s in state]))
        return self.impl(y, batch_state, ids)

    def select_state(self, state, i, new_id=None):
        r, log_psi, _, _, scoring_idmap = state
        s = log_psi[i, new_id].expand(log_psi.size(1))
        return r[:, :, i, scoring_idmap[i, new_id]], s


# --------------------------------------------------------------------------------------------------------------------
# espnet/nets/beam_search.py + batch_beam_search.py
# --------------------------------------------------------------------------------------------------------------------
class Hypothesis(NamedTuple):
    yseq: torch.Tensor
    score: float = 0.0
    scores: Dict[str, float] = {}
    states: Dict[str, object] = {}


def end_detect(ended_hyps: List[Hypothesis], i: int, M: int = 3, D_end: float = math.log(1 * math.exp(-10))) -> bool:
    """espnet/nets/e2e_asr_common.py:end_detect"""
    if len(ended_hyps) == 0:
        return False
    count = 0
    best = max(float(h.score) for h in ended_hyps)
    for m in range(M):
        same = [float(h.score) for h in ended_hyps if len(h.yseq) == i - m]
        if len(same) > 0 and max(same) - best < D_end:
            count += 1
    return count == M


class BatchBeamSearch:
    def __init__(self, scorers: Dict[str, object], weights: Dict[str, float], beam_size: int, vocab_size: int, sos: int,
                 eos: int, pre_beam_ratio: float = 1.5, pre_beam_score_key: Optional[str] = "full"):
        self.weights, self.full, self.part = {}, {}, {}
        for k, v in scorers.items():
            w = weights.get(k, 0)
            if w == 0 or v is None:
                continue
            (self.part if isinstance(v, CTCPrefixScorer) else self.full)[k] = v
            self.weights[k] = w
        self.sos, self.eos, self.n_vocab, self.beam_size = sos, eos, vocab_size, beam_size
        self.pre_beam_size = int(pre_beam_ratio * beam_size)
        self.pre_beam_score_key = pre_beam_score_key
        self.do_pre_beam = (pre_beam_score_key is not None and self.pre_beam_size < vocab_size and len(self.part) > 0)

    def init_hyp(self, x):
        states = {k: d.batch_init_state(x) for k, d in {**self.full, **self.part}.items()}
        scores = {k: 0.0 for k in states}
        return [Hypothesis(yseq=torch.tensor([self.sos]), score=0.0, scores=scores, states=states)]

    def search(self, hyps: List[Hypothesis], x):
        n = len(hyps)
        yseq = torch.stack([h.yseq for h in hyps])              # all running hyps have the same length
        weighted = torch.zeros(n, self.n_vocab, dtype=x.dtype)
        scores, states = {}, {}
        for k, d in self.full.items():
            scores[k], states[k] = d.batch_score(yseq, [h.states[k] for h in hyps], x.expand(n, *x.shape))
            weighted += self.weights[k] * scores[k]
        part_ids = None
        if self.do_pre_beam:
            pre = weighted if self.pre_beam_score_key == "full" else scores[self.pre_beam_score_key]
            part_ids = torch.topk(pre, self.pre_beam_size, dim=-1)[1]
        part_scores, part_states = {}, {}
        for k, d in self.part.items():
            ids = part_ids if part_ids is not None else torch.arange(self.n_vocab).expand(n, -1)
            part_scores[k], part_states[k] = d.batch_score_partial(yseq, ids, [h.states[k] for h in hyps], x)
            weighted += self.weights[k] * part_scores[k]
        weighted += torch.tensor([float(h.score) for h in hyps], dtype=x.dtype).unsqueeze(1)
        top = weighted.view(-1).topk(min(self.beam_size, weighted.numel()))[1]
        best = []
        for t in top:
            hid, tok = int(t) // self.n_vocab, int(t) % self.n_vocab
            prev = hyps[hid]
            new_scores = {k: prev.scores[k] + float(v[hid, tok]) for k, v in scores.items()}
            new_scores.update({k: prev.scores[k] + float(v[hid, tok]) for k, v in part_scores.items()})
            new_states = {k: self.full[k].select_state(v, hid) for k, v in states.items()}
            new_states.update({k: self.part[k].select_state(v, hid, tok) for k, v in part_states.items()})
            best.append(Hypothesis(yseq=torch.cat([prev.yseq, torch.tensor([tok])]), score=float(weighted[hid, tok]),
                                   scores=new_scores, states=new_states))
        return best

    def post_process(self, i, maxlen, running, ended):
        if i == maxlen - 1:
            running = [h._replace(yseq=torch.cat([h.yseq, torch.tensor([self.eos])])) for h in running]
        remained = []
        for h in running:
            (ended if int(h.yseq[-1]) == self.eos else remained).append(h)
        return remained

    def forward(self, x: torch.Tensor, maxlenratio: float = 0.0, minlenratio: float = 0.0) -> List[Hypothesis]:
        if maxlenratio == 0:
            maxlen = x.shape[0]
        elif maxlenratio < 0:
            maxlen = -1 * int(maxlenratio)
        else:
            maxlen = max(1, int(maxlenratio * x.size(0)))
        running = self.init_hyp(x)
        ended: List[Hypothesis] = []
        for i in range(maxlen):
            best = self.search(running, x)
            running = self.post_process(i, maxlen, best, ended)
            if maxlenratio == 0.0 and end_detect(ended, i):
                break
            if len(running) == 0:
                break
        nbest = sorted(ended, key=lambda h: h.score, reverse=True)
        if len(nbest) == 0:
            return [] if minlenratio < 0.1 else self.forward(x, maxlenratio, max(0.0, minlenratio - 0.1))
        return nbest


def build_beam_search(model, lm: Optional[TransformerLMOracle], beam_size: int, ctc_weight: float, lm_weight: float,
                      penalty: float) -> BatchBeamSearch:
    """scorers and weights of src/inference/avsr_inference.py:141-153,249-255,277-286 on an oracle ASR/AVSR model."""
    vocab = len(model.token_list)
    scorers = dict(decoder=DecoderScorer(model.decoder), ctc=CTCPrefixScorer(ctc=model.ctc, eos=model.eos),
                   length_bonus=LengthBonus(vocab), lm=lm)
    weights = dict(decoder=1.0 - ctc_weight, ctc=ctc_weight, lm=lm_weight, length_bonus=penalty)
    return BatchBeamSearch(scorers, weights, beam_size, vocab, model.sos, model.eos,
                           pre_beam_score_key=None if ctc_weight == 1.0 else "full")


def results(nbest: List[Hypothesis], n: int = 1):
    """token ids as Speech2Text._decode_single_sample returns them (:492-505): sos/eos stripped, blanks removed."""
    out = []
    for h in nbest[:n]:
        tok = [int(t) for t in h.yseq[1:-1].tolist() if int(t) != 0]
        out.append((tok, float(h.score)))
    return out
