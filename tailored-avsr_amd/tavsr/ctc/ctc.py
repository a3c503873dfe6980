"""``CTC`` - drop-in for src/ctc/ctc.py:9-188 (``ctc_type="builtin"``).

Linear(eprojs -> odim) + CTC loss / argmax on the HIP path; ``greedy`` adds the collapse used as the
reference's CTC-greedy semantics (src/models/maskctc_model.py:289-291) with bit-exact integer output.
"""
from __future__ import annotations

import torch

from .. import functional as F_
from .. import ops


class CTC(torch.nn.Module):
    def __init__(self, odim: int, encoder_output_size: int, dropout_rate: float = 0.0, ctc_type: str = "builtin",
                 reduce: bool = True, ignore_nan_grad: bool = None, zero_infinity: bool = True):
        super().__init__()
        self.dropout_rate = dropout_rate
        self.ctc_lo = torch.nn.Linear(encoder_output_size, odim)
        self.ctc_type = ctc_type
        if ignore_nan_grad is not None:
            zero_infinity = ignore_nan_grad
        if ctc_type != "builtin":
            raise ValueError(f'ctc_type must be "builtin" on the HIP path: {ctc_type}')
        self.zero_infinity = zero_infinity
        self.reduce = reduce

    def forward(self, hs_pad, hlens, ys_pad, ys_lens):
        """hs_pad (B,Tmax,D), hlens (B), ys_pad (B,Lmax) padded with -1, ys_lens (B) -> scalar loss."""
        if self.dropout_rate > 0:
            # the reference applies F.dropout(hs_pad, p) with its default training=True: active in eval mode too
            # (ctc.py:143) - followed as it is
            hs_pad = F_.DropoutFn.apply(hs_pad, self.dropout_rate)
        return F_.CTCLossFn.apply(hs_pad, self.ctc_lo.weight, self.ctc_lo.bias, hlens.to(torch.int64),
                                  ys_pad.to(torch.int64).contiguous(), ys_lens.to(torch.int64), self.reduce,
                                  self.zero_infinity)

    def _logits(self, hs_pad):
        B, T, D = hs_pad.shape
        return ops.linear(hs_pad.reshape(B * T, D), self.ctc_lo.weight, self.ctc_lo.bias).view(B, T, -1)

    def argmax(self, hs_pad):
        """(B,Tmax,eprojs) -> int64 ids (B,Tmax); ties -> lowest index as torch.argmax."""
        with torch.no_grad():
            ids, _, _ = ops.ctc_greedy(self._logits(hs_pad), None, 0, collapse=False)
        return ids

    def greedy(self, hs_pad, hlens, blank: int = 0):
        """ids, collapsed hypotheses (padded with -1) and their lengths, all int64 on the device."""
        with torch.no_grad():
            return ops.ctc_greedy(self._logits(hs_pad), hlens.to(torch.int64), blank, collapse=True)

    def softmax(self, hs_pad):
        """(B,Tmax,eprojs) -> frame posteriors (B,Tmax,odim)  (src/ctc/ctc.py:160-168).  Forward only: the differentiable
        use of the posteriors (self-conditioned intermediate CTC) is ``functional.InterCTCConditionFn``."""
        B, T, D = hs_pad.shape
        with torch.no_grad():
            logits = self._logits(hs_pad)
            V, S = logits.shape[-1], ops.pad4(logits.shape[-1])
            if S != V:      # the softmax kernel reads 4-padded rows
                padded = ops.empty(B * T, S, like=logits)
                ops.copy2d(logits.view(B * T, V), padded[:, :V])
            else:
                padded = logits.view(B * T, S)
            vlen = torch.full((1,), V, dtype=torch.int64, device=hs_pad.device)
            prob = ops.softmax_fwd(padded.view(1, 1, B * T, S), None, vlen, 1.0, T2=V)
        return prob.view(B * T, S)[:, :V].reshape(B, T, V)

    def log_softmax(self, hs_pad):
        """(B,Tmax,eprojs) -> frame log-posteriors (B,Tmax,odim)  (src/ctc/ctc.py:170-178); forward only (beam search)."""
        B, T, D = hs_pad.shape
        with torch.no_grad():
            return ops.log_softmax_rows(self._logits(hs_pad).view(B * T, -1)).view(B, T, -1)
