"""Batched hybrid CTC/attention beam search with LM scoring on the MI355X (BASELINE config 5).

Same search as the reference's decode path - espnet ``BatchBeamSearch`` with scorers {decoder, CTCPrefixScorer,
LengthBonus, lm}, weights {decoder: 1 - ctc_weight, ctc: ctc_weight, lm: lm_weight, length_bonus: penalty}, pre-beam of
``int(1.5 * beam)`` candidates on the weighted full scores, end detection of ``maxlenratio == 0``
(src/inference/avsr_inference.py:141-153, 249-255, 277-304, 449-518) - but the hypotheses of ALL utterances of a batch
advance together (the reference decodes one utterance at a time on the CPU):

* N = U x beam slots.  One step = one token for every slot: decoder step (6 layers) and LM step (16 layers) as
  [N, d] GEMMs on the MFMA kernel, self-attention over each hypothesis' own history by ``tavsr_tree_attn_step`` (keys
  and values stay where they were written; a hypothesis is an int32 list of ancestor rows), source attention against
  the per-utterance memory keys/values projected once, CTC prefix scores of the pre-beam candidates by
  ``tavsr_ctc_prefix_step``.
* The per-step bookkeeping (top-k, ancestor-list gathers, ended-hypothesis lists, end detection) is torch / host code:
  index plumbing on [N]-sized tensors.

``decode(enc, enc_lens)`` takes encoder outputs (``model.encode``), returns per utterance the ended hypotheses sorted
by score as (token ids incl. <sos>/<eos>, score).  ``Speech2Text`` wraps model + search like the reference's class.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch

from .. import ops

EPS = 1e-12
LOGZERO = -10000000000.0
GRAPH_STEP = os.environ.get("TAVSR_DECODE_GRAPH", "1") != "0"      # capture the scorer step into one hipGraph


def _cat(ws):
    return torch.cat([w.detach() for w in ws], dim=0).contiguous()


class _DerivedWeights:
    """The step objects keep a few DERIVED copies of parameters (q / k / v weights side by side for one launch, the LM's input
    table).  Everything else is read through the parameters' own storage.  ``sync()`` re-derives the copies IN PLACE (captured
    graphs keep reading the same buffers) when a source parameter's version counter has moved since they were made - an
    optimizer step or ``load_state_dict`` between two decodes (validation while training)."""

    def __init__(self):
        self._derived = []          # (destination, source parameters)
        self._seen = None

    def cat(self, ws):
        dst = _cat(ws)
        self._derived.append((dst, list(ws)))
        return dst

    def _versions(self):
        return tuple(w._version for _, ws in self._derived for w in ws)

    def sync(self) -> bool:
        now = self._versions()
        if self._seen is None:
            self._seen = now
        if now == self._seen:
            return False
        for dst, ws in self._derived:
            torch.cat([w.detach() for w in ws], dim=0, out=dst)
        self._seen = now
        return True


# One-token scorer steps are chains of small launches: the fused feed-forward block (LayerNorm + both GEMMs, csrc/ffn.hip)
# and the fused source attention cut the chain.  TAVSR_DECODE_FUSED=0 keeps the GEMM launches (A/B switch).
FUSED_STEP = True
SCORERS_PARALLEL = True   # decoder || LM on two streams (TAVSR_SINGLE_STREAM=1 disables every fork)
TREE_GROUP = 1     # beams of an utterance side by side in the tree attention
PREBEAM_FUSED = True   # pre-beam top-k inside the CTC prefix launch
# Vocabularies of up to 64 tokens (character models): the CTC prefix scores of EVERY token are computed beside the LM's chain, in front
# of the decoder's - the recursion over the frames (20 us) leaves the critical path of a captured step - and the whole beam update
# behind the scorers (LM log-softmax, pre-beam, weighted scores, top-k) is one launch (tavsr_beam_select_topk).
# (module constants, flipped in-process by tests/test_beam_search.py and scripts/decode_chain_probe.py - not environment switches)
# per-token records on a queue of their own (TAVSR_RECORD_QUEUE=1) or on the search queue between two replays (default).  A queue of their
# own needs an event behind every replay that ANOTHER queue waits on - and that wait costs the NEXT replay ~1.5 us per launch it holds
# (scripts/graph_pair_probe.py: 60 launches +90 us, 190 launches +275 us; an event record alone, an eager launch or the 128-byte copy on
# the replaying queue itself: +5 us).  In the search: 633 -> 563 us per token at batch 1, 2359 -> 2188 at batch 64 (same process, alternating).
RECORD_QUEUE = os.environ.get("TAVSR_RECORD_QUEUE", "0") == "1"
LN_IN_EPILOGUE = True      # batched steps: tavsr_gemm_ln at the x + f(x) -> norm(x) seams
CTC_BESIDE_SCORERS = True


class _Normed:
    """rows of the residual stream together with their LayerNorm under one particular (gamma, beta): what the launch that finished
    the rows leaves when it was told which norm follows (``_linear_res(..., ln_next=)``, GEMM path: tavsr_gemm_ln)"""
    __slots__ = ("x", "n", "gamma")

    def __init__(self, x, n, gamma):
        self.x, self.n, self.gamma = x, n, gamma

    @property
    def shape(self):
        return self.x.shape


def _rows(x):
    """the plain [N, d] tensor of a residual stream held as ``ops.RowParts`` / ``_Normed``"""
    if isinstance(x, _Normed):
        return x.x
    if isinstance(x, ops.RowParts):
        return x.t.sum(0)
    return x


def _ln_linear(x, norm, w, b, act=None):
    """act(W LN(x) + b): one launch (ops.rowlin) when the shapes allow, else LayerNorm + GEMM launches (no LayerNorm launch when
    the rows arrive with this norm already taken)."""
    if isinstance(x, _Normed):
        if x.gamma is norm[0]:
            return ops.linear(x.n, w, b, act=act)
        x = x.x
    if ops.rowlin_ok(x, w, ln=True):
        return ops.rowlin(x, w, b, ln=(norm[0], norm[1], EPS), act=act)
    n = ops.layernorm_fwd(_rows(x), *norm, EPS, save=False)[0]
    return ops.linear(n, w, b, act=act)


def _linear_res(x, w, b, res, ksplit=1, ln_next=None):
    """res + W x + b (``ksplit`` > 1: K dealt to that many blocks per column tile, the result an ``ops.RowParts``; a K the one-launch
    kernel only takes in slices - 2048 with more than 16 rows - is dealt to ``ops.ROWLIN_KSPLIT`` blocks whatever the caller asked).
    ``ln_next`` = (gamma, beta) of the LayerNorm the result goes through next: on the GEMM path (more than 32 rows) the launch that
    finishes the rows - the sum of the K-split slabs - also normalises them (``_Normed``)."""
    if isinstance(res, _Normed):
        res = res.x
    if ksplit == 1 and not ops.rowlin_ok(x, w) and ops.rowlin_ok(x, w, ksplit=max(2, ops.ROWLIN_KSPLIT)):
        ksplit = max(2, ops.ROWLIN_KSPLIT)
    if ksplit > 1 and ops.rowlin_ok(x, w, ksplit=ksplit):
        return ops.rowlin(x, w, b, res=res, ksplit=ksplit)
    if ops.rowlin_ok(x, w):
        return ops.rowlin(x, w, b, res=res)
    res = _rows(res)
    if ln_next is not None and LN_IN_EPILOGUE and w.shape[0] % 4 == 0 and w.shape[0] <= 2048:
        y, n = ops.linear(x, w, b, res=res, ln=(ln_next[0], ln_next[1], EPS))
        return _Normed(y, n, ln_next[0])
    return ops.linear(x, w, b, res=res)


def _ffn_step(x, norm, L, ln_next=None):
    """x + W2 relu(W1 LN(x) + b1) + b2 of one decoder / LM layer for the current token rows.  Hidden size 2048: the closing
    projection deals its K to ops.ROWLIN_KSPLIT blocks per column tile and the result stays a sum of that many tensors
    (``ops.RowParts``) until the next launches of the chain add them while they load their operands."""
    if not isinstance(x, _Normed) and ops.rowlin_ok(x, L["w1"], ln=True):
        t = ops.rowlin(x, L["w1"], L["b1"], ln=(norm[0], norm[1], EPS), act="relu")
        return _linear_res(t, L["w2"], L["b2"], x, ops.ROWLIN_KSPLIT if t.shape[1] == 2048 else 1)
    t = _ln_linear(x, norm, L["w1"], L["b1"], act="relu")
    return _linear_res(t, L["w2"], L["b2"], x, ln_next=ln_next)


class _DecoderStep:
    """espnet2 TransformerDecoder.forward_one_step with K/V kept per token instead of per-layer output caches."""

    def __init__(self, dec):
        self.dec = dec
        self.weights = _DerivedWeights()
        _cat = self.weights.cat
        self.D = dec.after_norm.weight.numel()
        self.H = dec.heads
        self.dk = self.D // self.H
        self.layers = []
        for l in dec.decoders:
            sa, ca, ff = l.self_attn, l.src_attn, l.feed_forward
            self.layers.append(dict(
                n1=(l.norm1.weight, l.norm1.bias), n2=(l.norm2.weight, l.norm2.bias), n3=(l.norm3.weight, l.norm3.bias),
                wqkv=_cat([sa.linear_q.weight, sa.linear_k.weight, sa.linear_v.weight]),
                bqkv=_cat([sa.linear_q.bias, sa.linear_k.bias, sa.linear_v.bias]),
                wo=sa.linear_out.weight, bo=sa.linear_out.bias,
                wq2=ca.linear_q.weight, bq2=ca.linear_q.bias,
                wkv2=_cat([ca.linear_k.weight, ca.linear_v.weight]), bkv2=_cat([ca.linear_k.bias, ca.linear_v.bias]),
                wo2=ca.linear_out.weight, bo2=ca.linear_out.bias,
                w1=ff.w_1.weight, b1=ff.w_1.bias, w2=ff.w_2.weight, b2=ff.w_2.bias))

    def start(self, enc, enc_lens, N, K, max_steps):
        U, T, D = enc.shape
        self.U, self.T, self.N, self.K = U, T, N, K
        self.enc_lens = enc_lens
        mem2 = enc.reshape(U * T, D)
        self.memkv = [ops.linear(mem2, L["wkv2"], L["bkv2"]) for L in self.layers]          # [U*T, 2D] per layer
        self.kpool = [ops.empty(max_steps * N, D, like=enc) for _ in self.layers]
        self.vpool = [ops.empty(max_steps * N, D, like=enc) for _ in self.layers]
        self.pe = self.dec.embed[1].table(max_steps, enc.device)
        self.emb = self.dec.embed[0].weight
        self.xscale = math.sqrt(D)

    def refill(self, enc):
        """a new batch of the same shape into the buffers a captured step reads (the source-attention keys / values)"""
        U, T, D = enc.shape
        mem2 = enc.reshape(U * T, D)
        for L, kv in zip(self.layers, self.memkv):
            ops.linear(mem2, L["wkv2"], L["bkv2"], out=kv, ldc=kv.stride(0))

    def step(self, i, tok, anc, dyn=None, **score):
        """tok [N] last tokens, anc int32 [N, >= i+1] (column i already points at this step's rows) -> logp [N, V].
        ``dyn`` = (step_dev,): the step index - and with it the positional row - is read from a device buffer (``i`` is then the
        pool capacity in steps), so the launches can be captured once and replayed for every step."""
        N, D, H, dk, U, T, K = self.N, self.D, self.H, self.dk, self.U, self.T, self.K
        if dyn is None:
            x = ops.embed_pe(tok.view(N, 1), self.emb, self.pe[i:i + 1].contiguous(), self.xscale).view(N, D)
        else:      # the positional row of the device-side step counter: read by the embedding launch itself
            x = ops.embed_pe(tok.view(N, 1), self.emb, self.pe, self.xscale, step_dev=dyn[0]).view(N, D)
        after = (self.dec.after_norm.weight, self.dec.after_norm.bias)
        for li, L in enumerate(self.layers):
            qkv = _ln_linear(x, L["n1"], L["wqkv"], L["bqkv"])
            # this step's keys / values are the last key of every hypothesis: the attention launch appends them to the pools
            a = ops.tree_attn_step(qkv[:, :D], self.kpool[li], self.vpool[li], anc, i + 1 if dyn is None else i, H, dk,
                                   step_dev=None if dyn is None else dyn[0], k_new=qkv[:, D:2 * D], v_new=qkv[:, 2 * D:],
                                   group=TREE_GROUP * self.K)
            x = _linear_res(a, L["wo"], L["bo"], x, ln_next=L["n2"])
            q2 = _ln_linear(x, L["n2"], L["wq2"], L["bq2"])
            # source attention: the K slots of an utterance are K query rows against that utterance's memory
            kv = self.memkv[li]
            if FUSED_STEP and dk == 64:      # scores, mask, softmax and context in one launch (csrc/attn_fused.hip)
                c2 = ops.attn_fwd(q2, 0, kv, 0, kv, D, U, K, T, H, dk, klens=self.enc_lens)[0]
            else:
                S = ops.pad4(T)
                sc = ops.empty(H, U, K, S, like=x)
                ops.gemm(K, T, dk, q2, D, kv, 2 * D, sc, S, nb1=U, nb2=H, sA=(K * D, dk), sB=(T * 2 * D, dk),
                         sC=(K * S, U * K * S))
                att = ops.softmax_fwd(sc, None, self.enc_lens, 1.0 / math.sqrt(dk), T2=T)
                c2 = ops.empty(N, D, like=x)
                ops.gemm(K, dk, T, att, S, kv, 2 * D, c2, D, b_off=D, b_kmajor=True, nb1=U, nb2=H, sA=(K * S, U * K * S),
                         sB=(T * 2 * D, dk), sC=(K * D, dk))
            x = _linear_res(c2, L["wo2"], L["bo2"], x, ln_next=L["n3"])
            x = _ffn_step(x, L["n3"], L, ln_next=self.layers[li + 1]["n1"] if li + 1 < len(self.layers) else after)
        z = _ln_linear(x, after, self.dec.output_layer.weight, self.dec.output_layer.bias)
        return ops.log_softmax_rows(z, **score)


class _LMStep:
    """espnet2 TransformerLM.batch_score (encoder.forward_one_step) with per-token K/V."""

    def __init__(self, lm):
        self.lm = lm
        self.weights = _DerivedWeights()
        _cat = self.weights.cat
        self.D, self.H = lm.att_unit, lm.heads
        self.dk = self.D // self.H
        self.layers = []
        for l in lm.encoder.encoders:
            a, ff = l.self_attn, l.feed_forward
            self.layers.append(dict(
                n1=(l.norm1.weight, l.norm1.bias), n2=(l.norm2.weight, l.norm2.bias),
                wqkv=_cat([a.linear_q.weight, a.linear_k.weight, a.linear_v.weight]),
                bqkv=_cat([a.linear_q.bias, a.linear_k.bias, a.linear_v.bias]),
                wo=a.linear_out.weight, bo=a.linear_out.bias,
                w1=ff.w_1.weight, b1=ff.w_1.bias, w2=ff.w_2.weight, b2=ff.w_2.bias))

    def start(self, like, N, max_steps, K=1):
        self.N, self.K = N, K
        # the LM's input layer (embedding -> Linear -> LayerNorm -> ReLU; espnet TransformerLM with pos_enc: null) depends on the token
        # alone: one [V, d] table per search instead of three launches per token
        self.in_table = self._input_table()
        emb = self.lm.encoder.embed
        self._table_ver = tuple(w._version for w in (self.lm.embed.weight, emb[0].weight, emb[0].bias, emb[1].weight, emb[1].bias))
        self.kpool = [ops.empty(max_steps * N, self.D, like=like) for _ in self.layers]
        self.vpool = [ops.empty(max_steps * N, self.D, like=like) for _ in self.layers]

    def _input_table(self):
        lm = self.lm
        emb = lm.encoder.embed
        t = ops.linear(lm.embed.weight.contiguous(), emb[0].weight, emb[0].bias)
        t = ops.layernorm_fwd(t, emb[1].weight, emb[1].bias, EPS, save=False)[0]
        return ops.act_(t, "relu")

    def sync_weights(self):
        """derived copies after a change of the LM's parameters: the q / k / v concatenations and - into the buffer a captured step
        gathers from - the input table (one small GEMM, a LayerNorm, a ReLU)"""
        lm = self.lm
        emb = lm.encoder.embed
        src = (lm.embed.weight, emb[0].weight, emb[0].bias, emb[1].weight, emb[1].bias)
        ver = tuple(w._version for w in src)
        changed = self.weights.sync()
        if getattr(self, "in_table", None) is not None and (changed or ver != self._table_ver):
            self.in_table.copy_(self._input_table())
        self._table_ver = ver

    def step(self, i, tok, anc, dyn=None, logits_only=False, **score):
        N, D, H, dk = self.N, self.D, self.H, self.dk
        lm = self.lm
        # the token's row of the input table is gathered by the first layer's launches themselves where the one-launch Linear
        # runs (its ``gather`` / ``res_gather`` operands): no gather launch at the head of the LM's chain
        after = (lm.encoder.after_norm.weight, lm.encoder.after_norm.bias)
        fold = ops.rowlin_ok(self.in_table, self.layers[0]["wqkv"], n_rows=N, ln=True)
        h = None if fold else self.in_table.index_select(0, tok)
        for li, L in enumerate(self.layers):
            if li == 0 and fold:
                qkv = ops.rowlin(self.in_table, L["wqkv"], L["bqkv"], ln=(L["n1"][0], L["n1"][1], EPS), gather=tok)
            else:
                qkv = _ln_linear(h, L["n1"], L["wqkv"], L["bqkv"])
            # this step's keys / values are the last key of every hypothesis: the attention launch appends them to the pools
            a = ops.tree_attn_step(qkv[:, :D], self.kpool[li], self.vpool[li], anc, i + 1 if dyn is None else i, H, dk,
                                   step_dev=None if dyn is None else dyn[0], k_new=qkv[:, D:2 * D], v_new=qkv[:, 2 * D:],
                                   group=TREE_GROUP * self.K)
            if li == 0 and fold:
                h = ops.rowlin(a, L["wo"], L["bo"], res=self.in_table, res_gather=tok)
            else:
                h = _linear_res(a, L["wo"], L["bo"], h, ln_next=L["n2"])
            h = _ffn_step(h, L["n2"], L, ln_next=self.layers[li + 1]["n1"] if li + 1 < len(self.layers) else after)
        z = _ln_linear(h, after, lm.decoder.weight, lm.decoder.bias)
        return z if logits_only else ops.log_softmax_rows(z, **score)


class BatchBeamSearch:
    def __init__(self, model, lm=None, beam_size: int = 10, ctc_weight: float = 0.1, lm_weight: float = 0.6,
                 penalty: float = 0.5, maxlenratio: float = 0.0, minlenratio: float = 0.0):
        if model.decoder is None or model.ctc is None:
            raise ValueError("the hybrid search needs both the attention decoder and the CTC head (0 < ctc_weight < 1)")
        if not (0.0 <= ctc_weight <= 1.0):
            raise ValueError(f"ctc_weight must lie in [0, 1]: {ctc_weight}")
        # ctc_weight 0 / 1: espnet skips a scorer whose weight is 0; here its launches still run and enter the sums with
        # weight 0 (finite log-probabilities), which selects the same hypotheses
        # espnet BeamSearch.forward: maxlenratio == 0 -> up to T tokens with end detection; > 0 -> max(1, int(ratio * T)) tokens,
        # < 0 -> -int(ratio) tokens, both without end detection; minlenratio is accepted and (as in espnet 202402) only logged
        self.maxlenratio, self.minlenratio = float(maxlenratio), float(minlenratio)
        self.model, self.lm = model, lm
        self.K = beam_size
        self.V = len(model.token_list)
        self.sos, self.eos = model.sos, model.eos
        self.w_dec, self.w_ctc = 1.0 - ctc_weight, ctc_weight
        self.w_lm = lm_weight if lm is not None else 0.0
        self.w_len = penalty
        # avsr_inference.py:298: pre_beam_score_key = None when ctc_weight == 1 -> the CTC prefix scorer sees every token
        self.C = self.V if ctc_weight == 1.0 else min(int(1.5 * beam_size), self.V)
        self._pinned = None
        self._copy_q = None        # the queue the per-token records leave on (created with the first captured search)
        self._captured = None      # the captured step of the last batch shape (buffers + hipGraph), re-used while the shape repeats
        self.dec_step = _DecoderStep(model.decoder)
        self.lm_step = _LMStep(lm) if (lm is not None and lm_weight != 0.0) else None

    @torch.no_grad()
    def decode(self, enc: torch.Tensor, enc_lens: torch.Tensor, nbest: Optional[int] = None):
        """enc [U, T, D] encoder outputs, enc_lens [U] -> per utterance [(yseq list incl. sos/eos, score), ...] sorted."""
        U, T, D = enc.shape
        K, V, C = self.K, self.V, self.C
        N = U * K
        dev = enc.device
        enc = enc.contiguous().float()
        enc_lens = enc_lens.to(dev).to(torch.int64)
        lens_h = [int(v) for v in enc_lens.cpu()]
        if self.maxlenratio == 0.0:                                # at most T tokens per utterance, end detection below
            maxl_h = list(lens_h)
        elif self.maxlenratio < 0:
            maxl_h = [-1 * int(self.maxlenratio)] * U
        else:
            maxl_h = [max(1, int(self.maxlenratio * t)) for t in lens_h]
        if any(m > t for m, t in zip(maxl_h, lens_h)):
            # espnet's CTCPrefixScoreTH indexes its forward variables by output length: a hypothesis longer than the number
            # of frames raises IndexError there; refused up front here
            raise ValueError(f"maxlenratio {self.maxlenratio} asks for more tokens than encoder frames {lens_h}: "
                             "the CTC prefix scorer cannot score a prefix longer than its input")
        steps = max(maxl_h)
        ctc = self.model.ctc
        # derived weight copies follow the parameters (an optimizer step / load_state_dict since the last call): ADVICE round 4
        self.dec_step.weights.sync()
        if self.lm_step is not None:
            self.lm_step.sync_weights()
        # A captured step is tied to its buffers, not to a batch: while the shape (utterances, frames, token budget) repeats -
        # a stream of 4 s clips - the buffers are refilled in place and the hipGraph of the previous call is replayed; capture
        # and warm-up were 8-10 ms of a 95 ms batch-1 search.
        key = (U, T, D, steps, str(dev), self.lm_step is not None)
        cap = self._captured if (GRAPH_STEP and self._captured is not None and self._captured["key"] == key) else None
        if cap is not None:
            logp_ctc, lens_buf = cap["logp_ctc"], cap["enc_lens"]
            lens_buf.copy_(enc_lens)
            enc_lens = lens_buf
            ops.log_softmax_rows(ops.linear(enc.reshape(U * T, D), ctc.ctc_lo.weight, ctc.ctc_lo.bias), out=logp_ctc.view(U * T, V))
            self.dec_step.refill(enc)
            tok, yseq, score, anc, slot_ids, r_prev, s_prev, utt_base = cap["bufs"]
        else:
            self._captured = None
            enc_lens = enc_lens.clone()
            logp_ctc = ops.log_softmax_rows(ops.linear(enc.reshape(U * T, D), ctc.ctc_lo.weight, ctc.ctc_lo.bias)).view(U, T, V)
            self.dec_step.start(enc, enc_lens, N, K, steps)
            if self.lm_step is not None:
                self.lm_step.start(enc, N, steps, K)
            # running state
            tok = torch.full((N,), self.sos, dtype=torch.int64, device=dev)
            yseq = torch.full((N, steps + 2), self.eos, dtype=torch.int64, device=dev)
            yseq[:, 0] = self.sos
            score = torch.full((U, K), -float("inf"), device=dev)
            score[:, 0] = 0.0                                          # one <sos> hypothesis per utterance
            score = score.view(N)
            anc = torch.zeros(N, steps, dtype=torch.int32, device=dev)
            slot_ids = torch.arange(N, dtype=torch.int32, device=dev)
            r_prev = torch.zeros(N, T, 2, device=dev)
            s_prev = torch.zeros(N, device=dev)
            utt_base = (torch.arange(U, device=dev) * K).view(U, 1)
        # host-side bookkeeping, vectorised over utterances ([U] / [N]-sized CPU tensors)
        lens_c = torch.tensor(maxl_h)                               # per-utterance maxlen: the last iteration closes every hypothesis
        active = torch.ones(U, dtype=torch.bool)
        best = torch.full((U,), -float("inf"))                      # best ended score per utterance
        best_len = torch.full((U, steps + 4), -float("inf"))        # best ended score per (utterance, hypothesis length)
        D_end = math.log(1 * math.exp(-10))
        NEG_INF = -float("inf")
        state = dict(r_prev=r_prev, s_prev=s_prev, yseq=yseq, score=score)

        def device_step(i, dyn):
            """everything the device does for one token: both scorers, the pre-beam, CTC prefix scores of the candidates,
            the beam update and the re-ordering of the running state.  ``dyn`` = None: eager launches with the host's
            step index i.  ``dyn`` = device counters: every buffer is updated IN PLACE and the step index is read from
            device memory, so the whole step is one captured hipGraph (i is then the pool capacity in steps)."""
            r_prev, s_prev, yseq, score = state["r_prev"], state["s_prev"], state["yseq"], state["score"]
            if dyn is None:
                anc[:, i] = slot_ids + i * N
                sdyn = None
            else:
                # (hypotheses that ended with the previous token - <eos>, or their utterance's last iteration - have left the beam
                # and column `step` of the ancestor lists is filled: the previous step's re-ordering launch did both, reset_state()
                # for step 0)
                sdyn = (dyn["step"],)
            # full = w_dec * decoder + w_lm * lm + w_len (LengthBonus: 1 per token), summed by the scorers' last launches
            # the two scorers are independent chains of small launches: the LM runs on the side stream next to the decoder
            has_lm = self.lm_step is not None
            spec = dyn is not None and CTC_BESIDE_SCORERS and ops.beam_select_topk_ok(K, V) and C <= V
            if has_lm:
                with ops.BranchScope(enabled=SCORERS_PARALLEL) as br:
                    z_lm = self.lm_step.step(i, tok, anc, sdyn, logits_only=True)
            if spec:
                # on the decoder's queue, in front of its chain: the decoder has ~100 us of slack behind the LM (a third queue for it
                # made the whole step 45 us SLOWER: profiles/r05_notes.md)
                r_new, psi, psi_abs, eos_s, eos_abs = ops.ctc_prefix_step(logp_ctc, enc_lens, r_prev, s_prev, tok, dyn["cand_all"], K, i,
                                                                          step_dev=dyn["step"])
            full = self.dec_step.step(i, tok, anc, sdyn, alpha=self.w_dec, add=0.0 if has_lm else self.w_len)
            if has_lm:
                br.join()
            if spec:
                top_s, top_i = ops.beam_select_topk(full, z_lm if has_lm else None, self.w_lm, self.w_len if has_lm else 0.0, psi, psi_abs,
                                                    eos_s, eos_abs, s_prev, score, self.eos, self.w_ctc, K, C)
                ops.beam_reorder(top_i, top_s, dyn["cand_all"], r_new, psi_abs, yseq, anc, dyn["shadow"], K, V, dyn["step"],
                                 hist=dyn["hist"], maxlen=dyn["maxl"], eos=self.eos)
                ops.multi_copy_([r_prev, s_prev, yseq, anc, tok, score], list(dyn["shadow"]), inc=dyn["ctr"])
                return anc, tok
            if has_lm:
                ops.log_softmax_rows(z_lm, out=full, alpha=self.w_lm, add=self.w_len, accumulate=True)
            if PREBEAM_FUSED and C <= 64 and V <= 4096:            # pre-beam on the weighted full scores inside the CTC launch
                cand, r_new, psi, psi_abs, eos_s, eos_abs = ops.ctc_prefix_step_topk(
                    logp_ctc, enc_lens, r_prev, s_prev, tok, full, C, K, i, step_dev=None if dyn is None else dyn["step"])
            else:
                cand = torch.topk(full, C, dim=-1)[1]
                r_new, psi, psi_abs, eos_s, eos_abs = ops.ctc_prefix_step(logp_ctc, enc_lens, r_prev, s_prev, tok, cand, K, i,
                                                                          step_dev=None if dyn is None else dyn["step"])
            if dyn is not None:
                # beam update in three launches: weighted scores, top-k, gather of the extended slots' state into the
                # shadow buffers + one multi-buffer commit (+ the counters); same arithmetic as the torch ops below
                if ops.beam_combine_topk_ok(K, V):
                    top_s, top_i = ops.beam_combine_topk(full, cand, psi, psi_abs, eos_s, eos_abs, s_prev, score, self.eos, self.w_ctc, K)
                else:
                    weighted = ops.beam_combine(full, cand, psi, psi_abs, eos_s, eos_abs, s_prev, score, self.eos, self.w_ctc)
                    top_s, top_i = torch.topk(weighted.view(U, K * V), K, dim=-1)
                ops.beam_reorder(top_i, top_s, cand, r_new, psi_abs, yseq, anc, dyn["shadow"], K, V, dyn["step"], hist=dyn["hist"],
                                 maxlen=dyn["maxl"], eos=self.eos)
                # commit + the step counters (step, step64, stepp1 are views of this one tensor) in one launch
                ops.multi_copy_([r_prev, s_prev, yseq, anc, tok, score], list(dyn["shadow"]), inc=dyn["ctr"])
                return anc, tok
            is_eos_c = cand == self.eos
            psi = torch.where(is_eos_c, eos_s.unsqueeze(1), psi)
            psi_abs = torch.where(is_eos_c, eos_abs.unsqueeze(1), psi_abs)
            ctc_full = torch.full((N, V), LOGZERO, device=dev) - s_prev.unsqueeze(1)
            ctc_full[:, self.eos] = eos_s
            ctc_full.scatter_(1, cand, psi)
            weighted = full + self.w_ctc * ctc_full + score.unsqueeze(1)
            top_s, top_i = torch.topk(weighted.view(U, K * V), K, dim=-1)
            prev = (top_i // V + utt_base).view(N)                  # slot the new hypothesis extends
            new_tok = (top_i % V).view(N)
            # CTC state of the chosen candidate
            cidx = (cand[prev] == new_tok.unsqueeze(1)).float().argmax(dim=1)
            state["r_prev"], state["s_prev"] = r_new[prev, :, :, cidx], psi_abs[prev, cidx]
            yseq = yseq[prev]
            yseq[:, i + 1] = new_tok
            state["yseq"], state["score"] = yseq, top_s.view(N)
            return anc[prev], new_tok

        def reset_state():
            tok.fill_(self.sos)
            yseq.fill_(self.eos)
            yseq[:, 0] = self.sos
            score.fill_(NEG_INF)
            score.view(U, K)[:, 0] = 0.0
            anc.zero_()
            anc[:, 0] = slot_ids                                      # step 0's own key / value rows
            r_prev.zero_()
            s_prev.zero_()

        graph = dyn = None
        if cap is not None:
            graph, dyn = cap["graph"], cap["dyn"]
            reset_state()
            dyn["ctr"].copy_(torch.tensor([0, 1], dtype=torch.int64))
            dyn["maxl"].copy_(torch.tensor(maxl_h, dtype=torch.int32))
        elif GRAPH_STEP:
            # One token costs ~300 scorer launches plus ~90 small ones for the beam update, all on [N, d]-sized operands.
            # The whole device side of a step is captured once per decode() - the step index, the positional row, tokens,
            # ancestor lists, CTC state and scores live in fixed device buffers updated in place - and replayed per token;
            # the device also retires ended hypotheses itself and records (token, back-pointer, score) per slot and token, so
            # the host neither uploads anything nor sits on the critical path: it reads the records one token behind the
            # device (end detection, collecting ended hypotheses by back-tracking) while the next step already runs.
            # The step is bound by its chain of dependent small kernels (5-6 us each), not by the host: what shortens it is
            # fewer launches - the K/V append rides in the attention launch, the scorer weights in the log-softmax
            # launches, the beam update is three launches (beam_combine, top-k, beam_reorder) plus one commit.
            ctr = torch.tensor([0, 1], dtype=torch.int64, device=dev)          # (i, i + 1); the kernels read i as int32
            dyn = dict(ctr=ctr, step64=ctr[0:1], stepp1=ctr[1:2], step=ctr.view(torch.int32)[0:1],     # (low word: little endian)
                       maxl=torch.tensor(maxl_h, dtype=torch.int32, device=dev),
                       hist=torch.zeros(steps, 3, N, dtype=torch.int32, device=dev),
                       cand_all=torch.arange(V, dtype=torch.int64, device=dev).repeat(N, 1),      # every token a "candidate" of the CTC scorer
                       shadow=(torch.empty_like(r_prev), torch.empty_like(s_prev), torch.empty_like(yseq), torch.empty_like(anc),
                               torch.empty_like(tok), torch.empty_like(score)))
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                device_step(steps, dyn)                             # warm-up outside the capture; the state is reset below
                reset_state()
                ctr.copy_(torch.tensor([0, 1], dtype=torch.int64))
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                device_step(steps, dyn)
            self._captured = dict(key=key, graph=graph, dyn=dyn, logp_ctc=logp_ctc, enc_lens=enc_lens,
                                  bufs=(tok, yseq, score, anc, slot_ids, r_prev, s_prev, utt_base))
        ended = [[] for _ in range(U)]

        def host_step(i, tok_h, score_h, rows_of):
            """espnet post_process / end_detect for token i of every utterance: collects the hypotheses that end here
            (``rows_of(slot indices)`` -> their token lists incl. <sos>), returns the slots to retire."""
            nonlocal active
            valid = torch.isfinite(score_h).view(U, K) & active.view(U, 1)
            last = (lens_c - 1 == i).view(U, 1)
            # espnet appends <eos> to EVERY hypothesis of the last iteration (also to the ones that just ended)
            take = valid & ((tok_h.view(U, K) == self.eos) | last)
            if bool(take.any()):
                idx = take.view(N).nonzero().view(-1)
                us = idx // K
                lns = torch.where(last.view(U)[us], torch.full_like(us, i + 3), torch.full_like(us, i + 2))
                scs = score_h[idx]
                for ys, u_, ln, sc in zip(rows_of(idx), us.tolist(), lns.tolist(), scs.tolist()):
                    ys = (ys + [self.eos] * 2)[:ln]
                    ys[ln - 1] = self.eos
                    ended[u_].append((ys, sc))
                best.index_reduce_(0, us, scs, "amax")
                flat = us * best_len.shape[1] + lns
                best_len.view(-1).index_reduce_(0, flat, scs, "amax")
            running = (valid & ~take).sum(dim=1)
            count = torch.zeros(U, dtype=torch.int64)
            for m in range(3):                                     # end_detect: M = 3 most recent lengths
                if i - m >= 0:
                    bl = best_len[:, i - m]
                    count += (torch.isfinite(bl) & (bl - best < D_end)).to(torch.int64)
            if self.maxlenratio != 0.0:                            # end_detect runs only for maxlenratio == 0
                count.zero_()
            stop = (count == 3) | (running == 0) | last.view(U)
            active = active & ~stop
            return (take | ~active.view(U, 1)).view(N)

        if graph is not None:
            hist = dyn["hist"]
            need = steps * 3 * N                                    # page-locked landing zone of the records: ONE buffer that
            if self._pinned is None or self._pinned.numel() < need:   # grows to the largest search seen
                self._pinned = torch.empty(max(need, 1 << 16), dtype=torch.int32, pin_memory=True)
            pin = self._pinned[:need].view(steps, 3, N)
            rec = pin.numpy()                                       # same memory: rec[token][0 / 1 / 2][slot]

            def rows_from_records(i):
                def rows_of(idx):
                    out_rows = []
                    for n in idx.tolist():
                        toks, cur = [], n
                        for t in range(i, -1, -1):                  # back-track: token of slot `cur`, then the slot it extended
                            toks.append(int(rec[t, 0, cur]))
                            cur = int(rec[t, 1, cur])
                        out_rows.append([self.sos] + toks[::-1])
                    return out_rows
                return rows_of

            events, done = [], 0
            timing = os.environ.get("TAVSR_DECODE_TIMING") == "1"   # host-side cost of a token: replay call / record processing
            t_replay = t_host = t_wait = 0.0
            import time as _time
            # the record of token i leaves behind step i on the search queue itself (see RECORD_QUEUE for the queue of their own the
            # records had, and what the event it needs costs the next replay)
            main_q = torch.cuda.current_stream()
            if self._copy_q is None:
                self._copy_q = torch.cuda.Stream() if RECORD_QUEUE else main_q
            copy_q = self._copy_q
            copy_q.wait_stream(main_q)
            for i in range(steps):
                t0 = _time.perf_counter()
                graph.replay()
                if copy_q is not main_q:
                    stepped = torch.cuda.Event()
                    stepped.record(main_q)
                    copy_q.wait_event(stepped)
                with torch.cuda.stream(copy_q):
                    pin[i].copy_(hist[i], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(copy_q)
                events.append(ev)
                t1 = _time.perf_counter()
                while done < i and bool(active.any()):              # the host works one token behind the device
                    events[done].synchronize()
                    t2 = _time.perf_counter()
                    host_step(done, pin[done, 0].to(torch.int64), pin[done, 2].view(torch.float32), rows_from_records(done))
                    done += 1
                    t_wait += t2 - t1
                    t_host += _time.perf_counter() - t2
                t_replay += t1 - t0
                if not bool(active.any()):
                    break
            if timing:
                n_ = max(1, len(events))
                print(f"decode timing (host, per token, N = {N}): replay + copy + event {1e6 * t_replay / n_:.0f} us, waiting for the "
                      f"device {1e6 * t_wait / n_:.0f} us, record processing {1e6 * t_host / n_:.0f} us", flush=True)
            while done < len(events) and bool(active.any()):
                events[done].synchronize()
                host_step(done, pin[done, 0].to(torch.int64), pin[done, 2].view(torch.float32), rows_from_records(done))
                done += 1
            main_q.wait_stream(copy_q)
            main_q.synchronize()                                    # a step the device ran ahead may still be in flight
        else:
            for i in range(steps):
                anc, tok = device_step(i, None)
                yseq, score = state["yseq"], state["score"]
                kill = host_step(i, tok.cpu(), score.cpu(), lambda idx: yseq[idx.to(dev)].cpu().tolist())
                if bool(kill.any()):
                    state["score"] = torch.where(kill.to(dev), torch.full_like(score, NEG_INF), score)
                if not bool(active.any()):
                    break
        out = []
        for u in range(U):
            hyps = sorted(ended[u], key=lambda h: h[1], reverse=True)
            out.append(hyps if nbest is None else hyps[:nbest])
        return out


class CapturedEncode:
    """``model.encode`` of an eval-mode model as a replayed hipGraph while the input shapes repeat (a stream of equally long clips):
    at batch 1 the encoder is ~600 launches of a few microseconds each, i.e. host-bound when enqueued one by one.  Inputs are
    copied into the capture's buffers, outputs are the capture's buffers (valid until the next call).  Lengths are device data the
    kernels read, so utterances of different lengths inside one padded shape replay the same graph; another shape captures anew.
    (The reference encodes eagerly, avsr_inference.py:449-470; results are the same launches' results.)"""

    def __init__(self, model, max_batch: int = 8):
        self.model, self.max_batch = model, max_batch      # (larger batches keep the GPU busy without a graph)
        self._cap = None

    @torch.no_grad()
    def __call__(self, *batch):
        if self.model.training or batch[0].shape[0] > self.max_batch or not all(t.is_cuda for t in batch):
            return self.model.encode(*batch)
        key = tuple((tuple(t.shape), t.dtype) for t in batch)
        cap = self._cap
        if cap is None or cap["key"] != key:
            static = [t.clone() for t in batch]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.model.encode(*static)                 # warm-up outside the capture (lazy buffers, allocator pools)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.model.encode(*static)
            cap = self._cap = dict(key=key, static=static, graph=graph, out=out)
        else:
            for s_, t in zip(cap["static"], batch):
                s_.copy_(t)
        cap["graph"].replay()
        return cap["out"]


class Speech2Text:
    """src/inference/avsr_inference.py:Speech2Text for models that are already built: encode + beam search, results as
    the reference returns them (:492-518): (text, tokens, token ids without sos/eos/blank, (yseq, score))."""

    def __init__(self, asr_model, lm=None, beam_size: int = 20, ctc_weight: float = 0.5, lm_weight: float = 1.0,
                 penalty: float = 0.0, nbest: int = 1, maxlenratio: float = 0.0, minlenratio: float = 0.0):
        self.asr_model = asr_model.eval()
        self.lm = None if lm is None else lm.eval()
        self.nbest = nbest
        self.beam_search = BatchBeamSearch(asr_model, lm, beam_size, ctc_weight, lm_weight, penalty, maxlenratio, minlenratio)
        self.encode = CapturedEncode(self.asr_model)

    @torch.no_grad()
    def __call__(self, *batch):
        """batch: the tensors of ``asr_model.encode`` (speech, lengths) or (audio, lengths, video, lengths)."""
        enc, enc_lens = self.encode(*batch)
        return self._results(enc, enc_lens)

    def _results(self, enc, enc_lens):
        if isinstance(enc, tuple):
            enc = enc[0]
        results = []
        for hyps in self.beam_search.decode(enc, enc_lens, nbest=self.nbest):
            res = []
            for ys, sc in hyps:
                token_int = [t for t in ys[1:-1] if t != 0]
                token = [self.asr_model.token_list[t] for t in token_int]
                text = "".join(token).replace("<space>", " ")
                res.append((text, token, token_int, (ys, sc)))
            results.append(res)
        return results
