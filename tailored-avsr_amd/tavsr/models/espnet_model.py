"""``ESPnetASRModel`` - drop-in for src/models/espnet_model.py:38-593 (hybrid CTC/attention branch).

``forward(speech, speech_lengths, text, text_lengths) -> (loss, stats, weight)`` with the reference's
stats keys; ``encode`` as used by inference.  Host code here is orchestration only (tiny integer
tensor manipulation such as add_sos_eos); every floating-point op runs in libtavsr_hip.so.
"""
from __future__ import annotations

import os
from itertools import groupby
from typing import Dict, List, Optional, Tuple, Union

import weakref

import torch

from .. import functional as F_
from .. import ops
from ..ctc.ctc import CTC


def add_sos_eos(ys_pad, ys_lens, sos, eos, ignore_id):
    """espnet add_sos_eos on padded int64 batches without a per-utterance Python loop."""
    B, L = ys_pad.shape
    idx = torch.arange(L + 1, device=ys_pad.device)[None, :]
    lens = ys_lens[:, None]
    body = torch.nn.functional.pad(ys_pad, (1, 0), value=sos)
    ys_in = torch.where(idx <= lens, body, torch.full_like(body, eos))
    ys_in[:, 0] = sos
    tail = torch.nn.functional.pad(ys_pad, (0, 1), value=ignore_id)
    ys_out = torch.where(idx < lens, tail, torch.full_like(tail, ignore_id))
    ys_out = torch.where(idx == lens, torch.full_like(tail, eos), ys_out)
    return ys_in, ys_out


def _levenshtein(a, b) -> int:
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, cb in enumerate(b, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb))
        prev = cur
    return prev[-1]


class ErrorCalculator:
    """CER/WER of espnet.nets.e2e_asr_common.ErrorCalculator on id sequences (host side, eval only)."""

    def __init__(self, char_list, sym_space, sym_blank, report_cer=False, report_wer=False):
        self.char_list, self.space, self.blank = char_list, sym_space, sym_blank
        self.report_cer, self.report_wer = report_cer, report_wer
        self.idx_blank = char_list.index(sym_blank)
        self.idx_space = char_list.index(sym_space) if sym_space in char_list else None

    def cer_ctc(self, ys_hat, ys_pad):
        errs, n = 0, 0
        for y, ref in zip(ys_hat.tolist(), ys_pad.tolist()):
            keep = lambda i: i != -1 and i != self.idx_blank and i != self.idx_space
            # the reference joins the token strings and measures the edit distance over CHARACTERS of the joined
            # text, so a multi-character token such as "<unk>" counts as five symbols (espnet ErrorCalculator.calculate_cer_ctc)
            hyp = "".join(self.char_list[i] for i, _ in groupby(y) if keep(i))
            tru = "".join(self.char_list[i] for i in ref if keep(i))
            if tru:
                errs += _levenshtein(hyp, tru)
                n += len(tru)
        return errs / n if n else None

    def __call__(self, ys_hat, ys_pad, is_ctc=False):
        if is_ctc:
            return self.cer_ctc(ys_hat, ys_pad)
        hyps, refs = [], []
        for y, ref in zip(ys_hat.tolist(), ys_pad.tolist()):
            ymax = ref.index(-1) if -1 in ref else len(ref)
            h = "".join(self.char_list[i] for i in y[:ymax]).replace(self.space, " ").replace(self.blank, "")
            r = "".join(self.char_list[i] for i in ref if i != -1).replace(self.space, " ")
            hyps.append(h)
            refs.append(r)
        cer = wer = None
        if self.report_cer:
            d = sum(_levenshtein(list(h.replace(" ", "")), list(r.replace(" ", ""))) for h, r in zip(hyps, refs))
            cer = d / sum(len(r.replace(" ", "")) for r in refs)
        if self.report_wer:
            d = sum(_levenshtein(h.split(), r.split()) for h, r in zip(hyps, refs))
            wer = d / sum(len(r.split()) for r in refs)
        return cer, wer


LOSS_BRANCH = True      # CTC branch beside the attention decoder (TAVSR_SINGLE_STREAM=1 disables every fork)


_MAX_SEEN = {}      # id(lengths tensor) -> (weak reference, tensor version, maximum)


def host_max(lengths: torch.Tensor) -> int:
    """``int(lengths.max())`` without stalling the launch queue where that can be avoided.  Reading a device scalar makes the
    host wait for everything enqueued before it - at the top of a training step that is the previous step's backward pass, so
    the host cannot run ahead across steps and the GPU starves at every step start (2.3 ms of a 17.7 ms audio-only step).
    The collate functions know the lengths on the host and attach the maximum (``_tavsr_max``, utils/avsr_dataloader.py);
    a tensor object that was already asked about (same object, same version counter: a fixed benchmark or validation batch)
    answers from a cache; anything else costs the one sync the reference pays too (espnet_model.py:372)."""
    m = getattr(lengths, "_tavsr_max", None)
    if m is not None:
        # (version counter at the time the maximum was attached, maximum): an in-place edit of the collated tensor (copy_, clamp_,
        # sub_) bumps the version and the hint is dropped for the synced path below
        if isinstance(m, tuple):
            if m[0] == lengths._version:
                return int(m[1])
        else:
            return int(m)
    if not lengths.is_cuda:
        return int(lengths.max())
    key = id(lengths)
    e = _MAX_SEEN.get(key)
    if e is not None and e[0]() is lengths and e[1] == lengths._version:
        return e[2]
    m = int(lengths.max())
    _MAX_SEEN[key] = (weakref.ref(lengths, lambda _r, k=key: _MAX_SEEN.pop(k, None)), lengths._version, m)
    return m


def cut_to_longest(x: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
    """``x[:, : lengths.max()]`` - the reference's "for data-parallel" cut of an over-padded batch
    (src/models/espnet_model.py:372 / avsr_espnet_model.py:499 ``_extract_feats``, ``forward`` text cut :247).  The
    maximum is a host value (``host_max``); while a hipGraph is being captured no sync is possible and the batch must
    already be tightly padded (a captured step has static shapes anyway)."""
    if x.is_cuda and torch.cuda.is_current_stream_capturing():
        return x
    m = host_max(lengths)
    return x if m >= x.shape[1] else x[:, :m].contiguous()


class UtteranceMVN(torch.nn.Module):
    """espnet2 UtteranceMVN(norm_means=True, norm_vars=False) on the HIP path."""

    def __init__(self, norm_means: bool = True, norm_vars: bool = False, eps: float = 1.0e-20):
        super().__init__()
        if not norm_means or norm_vars:
            raise ValueError("HIP path covers norm_means=true, norm_vars=false (all shipped configs)")

    def forward(self, x, ilens):
        return ops.utterance_mvn(x.contiguous(), ilens.to(torch.int64)), ilens


class ESPnetASRModel(torch.nn.Module):
    def __init__(self, vocab_size: int, token_list: Union[Tuple[str, ...], List[str]], frontend, specaug, normalize,
                 preencoder, encoder, postencoder, decoder, ctc: CTC, joint_network=None, aux_ctc: dict = None,
                 ctc_weight: float = 0.5, interctc_weight: float = 0.0, ignore_id: int = -1, lsm_weight: float = 0.0,
                 length_normalized_loss: bool = False, report_cer: bool = True, report_wer: bool = True,
                 sym_space: str = "<space>", sym_blank: str = "<blank>", transducer_multi_blank_durations: List = [],
                 transducer_multi_blank_sigma: float = 0.05, sym_sos: str = "<sos/eos>", sym_eos: str = "<sos/eos>",
                 extract_feats_in_collect_stats: bool = True, lang_token_id: int = -1):
        assert 0.0 <= ctc_weight <= 1.0, ctc_weight
        assert 0.0 <= interctc_weight < 1.0, interctc_weight
        super().__init__()
        if joint_network is not None or preencoder is not None or postencoder is not None:
            raise ValueError("transducer / pre- / post-encoder branches are out of the hot path (no shipped config)")
        self.blank_id = token_list.index(sym_blank) if sym_blank in token_list else 0
        self.sos = token_list.index(sym_sos) if sym_sos in token_list else vocab_size - 1
        self.eos = token_list.index(sym_eos) if sym_eos in token_list else vocab_size - 1
        self.vocab_size, self.ignore_id = vocab_size, ignore_id
        self.ctc_weight, self.interctc_weight, self.aux_ctc = ctc_weight, interctc_weight, aux_ctc
        self.token_list = list(token_list)
        self.frontend, self.specaug, self.normalize, self.encoder = frontend, specaug, normalize, encoder
        if not hasattr(self.encoder, "interctc_use_conditioning"):
            self.encoder.interctc_use_conditioning = False
        if self.encoder.interctc_use_conditioning:       # espnet_model.py:109-112
            self.encoder.conditioning_layer = torch.nn.Linear(vocab_size, self.encoder.output_size())
        self.decoder = decoder if ctc_weight < 1.0 else None
        self.lsm_weight, self.length_normalized_loss = lsm_weight, length_normalized_loss
        self.error_calculator = (ErrorCalculator(self.token_list, sym_space, sym_blank, report_cer, report_wer)
                                 if (report_cer or report_wer) else None)
        self.ctc = ctc if ctc_weight != 0.0 else None

    # ---------------------------------------------------------------- espnet_model.py:369-430
    def encode(self, speech: torch.Tensor, speech_lengths: torch.Tensor):
        # espnet_model.py:372: the batch is cut to its longest utterance (the STFT's reflect padding sees the tensor end)
        speech = cut_to_longest(speech, speech_lengths)
        if self.frontend is not None:
            feats, feats_lengths = self.frontend(speech, speech_lengths)
        else:
            feats, feats_lengths = speech, speech_lengths
        if self.specaug is not None and self.training:
            feats, feats_lengths = self.specaug(feats, feats_lengths)
        if self.normalize is not None:
            feats, feats_lengths = self.normalize(feats, feats_lengths)
        if self.encoder.interctc_use_conditioning:       # espnet_model.py:397-402
            encoder_out, encoder_out_lens, _ = self.encoder(feats, feats_lengths, ctc=self.ctc)
        else:
            encoder_out, encoder_out_lens, _ = self.encoder(feats, feats_lengths)
        return encoder_out, encoder_out_lens

    # ---------------------------------------------------------------- espnet_model.py:206-356
    def forward(self, speech, speech_lengths, text, text_lengths, **kwargs):
        assert text_lengths.dim() == 1, text_lengths.shape
        assert speech.shape[0] == speech_lengths.shape[0] == text.shape[0] == text_lengths.shape[0], (
            speech.shape, speech_lengths.shape, text.shape, text_lengths.shape)
        batch_size = speech.shape[0]
        ops.rng_step_begin(speech.device)    # fresh dropout masks for this step (a kernel: captured graphs replay it)
        text = cut_to_longest(text.to(torch.int64).masked_fill(text == -1, self.ignore_id), text_lengths)
        encoder_out, encoder_out_lens = self.encode(speech, speech_lengths)
        return self._hybrid_loss(encoder_out, encoder_out_lens, text, text_lengths, batch_size)

    # CTC + attention losses and the stats dict: espnet_model.py:258-356 == avsr_espnet_model.py:253-367
    def _hybrid_loss(self, encoder_out, encoder_out_lens, text, text_lengths, batch_size):
        intermediate_outs = None
        if isinstance(encoder_out, tuple):
            encoder_out, intermediate_outs = encoder_out
        stats: Dict[str, Optional[torch.Tensor]] = dict()
        loss_ctc = loss_att = None
        # training: the CTC branch (projection, loss, their backward) is independent of the attention decoder - it is enqueued on
        # the side queue and runs beside the decoder, forward and (autograd replays a node on its forward queue) backward
        br = ops.BranchScope(LOSS_BRANCH and self.training and encoder_out.is_cuda and self.ctc_weight not in (0.0, 1.0))
        with br:
            if self.ctc_weight != 0.0:
                loss_ctc = self.ctc(encoder_out, encoder_out_lens, text, text_lengths)
                cer_ctc = None
                if not self.training and self.error_calculator is not None:
                    ys_hat = self.ctc.argmax(encoder_out).data
                    cer_ctc = self.error_calculator(ys_hat.cpu(), text.cpu(), is_ctc=True)
                stats["loss_ctc"], stats["cer_ctc"] = loss_ctc.detach(), cer_ctc
            if self.interctc_weight != 0.0 and intermediate_outs is not None:      # espnet_model.py:260-304
                if self.aux_ctc is not None:
                    raise NotImplementedError("aux_ctc tasks are not used by any shipped recipe")
                loss_interctc = None
                for layer_idx, intermediate_out in intermediate_outs:
                    loss_ic = self.ctc(intermediate_out, encoder_out_lens, text, text_lengths)
                    cer_ic = None
                    if not self.training and self.error_calculator is not None:
                        ys_hat = self.ctc.argmax(intermediate_out).data
                        cer_ic = self.error_calculator(ys_hat.cpu(), text.cpu(), is_ctc=True)
                    loss_interctc = loss_ic if loss_interctc is None else F_.WeightedSumFn.apply(loss_interctc, loss_ic, 1.0, 1.0)
                    stats[f"loss_interctc_layer{layer_idx}"] = loss_ic.detach()
                    stats[f"cer_interctc_layer{layer_idx}"] = cer_ic
                n_ic = len(intermediate_outs)
                loss_ctc = F_.WeightedSumFn.apply(loss_ctc, loss_interctc, 1 - self.interctc_weight, self.interctc_weight / n_ic)
        acc_att = cer_att = wer_att = None
        if self.ctc_weight != 1.0:
            ys_in, ys_out = add_sos_eos(text, text_lengths, self.sos, self.eos, self.ignore_id)
            decoder_out, _ = self.decoder(encoder_out, encoder_out_lens, ys_in, text_lengths + 1)
            loss_att, correct = F_.LabelSmoothingLossFn.apply(decoder_out, ys_out, self.ignore_id, self.lsm_weight,
                                                             self.length_normalized_loss)
            acc_att = _Accuracy(correct)
            if not self.training and self.error_calculator is not None:
                ids, _, _ = ops.ctc_greedy(decoder_out.detach().contiguous(), None, -1, collapse=False)  # argmax(-1)
                cer_att, wer_att = self.error_calculator(ids.cpu(), text.cpu())
        br.join()
        if self.ctc_weight == 0.0:
            loss = loss_att
        elif self.ctc_weight == 1.0:
            loss = loss_ctc
        else:
            loss = F_.WeightedSumFn.apply(loss_ctc, loss_att, self.ctc_weight, 1 - self.ctc_weight)
        stats["loss_att"] = loss_att.detach() if loss_att is not None else None
        stats["acc"], stats["cer"], stats["wer"] = acc_att, cer_att, wer_att
        stats["loss"] = loss.detach()
        weight = torch.full((1,), batch_size, dtype=torch.long, device=loss.device)
        return loss.view(1), stats, weight

    @torch.no_grad()
    def ctc_greedy(self, speech, speech_lengths):
        """-> (ids (B,T), hyp (B,T) padded -1, hyp_len (B)): argmax + collapse, integer exact."""
        enc, olens = self.encode(speech, speech_lengths)
        return self.ctc.greedy(enc, olens, self.blank_id)


class _Accuracy:
    """Lazy th_accuracy: the reference returns a Python float (a device->host sync, nets_utils.th_accuracy);
    the counters stay on the device until somebody asks."""

    def __init__(self, correct):
        self.correct = correct

    def __float__(self):
        c = self.correct
        return float((c == 1).sum()) / float((c >= 0).sum())
