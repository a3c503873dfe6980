"""WER / CER with bootstrap confidence intervals on the MI355X.

Drop-in for src/evaluation/bootstrap_wer.py:compute_bootstrap_wer, which runs the C programs of
src/evaluation/tasas/ (``tasas -f "#" [-s " "] -ie`` and ``tasasIntervalo``) on the ``reference#hypothesis`` file that
avsr_main.py:84-105 writes.  Same file format and symbol rules (tasas.c:644-754): a line splits at its first '#';
word mode splits on runs of blanks, character mode takes single BYTES (an accented letter counts as its UTF-8 bytes, as
in the C program).  The edit distances of all sentence pairs are computed in one launch (``tavsr_edit_distance``), the
bootstrap resamples in another (``tavsr_bootstrap_rates``); the rate is 100 * sum(distance) / sum(reference length)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

from .. import ops

BOOTSTRAP_ITERS = 1000          # tasasIntervalo.c:802
Z = 1.64                        # tasasIntervalo.c:977


def _symbols(raw: bytes, word_mode: bool) -> List[bytes]:
    if word_mode:
        return [w for w in raw.split(b" ") if w]
    return [raw[i:i + 1] for i in range(len(raw))]


def read_pairs(path: str, word_mode: bool) -> List[Tuple[List[bytes], List[bytes]]]:
    pairs = []
    with open(path, "rb") as f:
        for ln, line in enumerate(f.read().split(b"\n")):
            if not line:
                continue
            line = line[:2047]                                   # the C reader's line buffer
            k = line.find(b"#")
            if k < 0:
                raise ValueError(f"no '#' separator in line {ln + 1} of {path}")
            pairs.append((_symbols(line[:k], word_mode), _symbols(line[k + 1:], word_mode)))
    return pairs


def _pack(seqs: Sequence[Sequence[int]], device):
    off = [0]
    for s in seqs:
        off.append(off[-1] + len(s))
    flat = [t for s in seqs for t in s] or [0]
    return (torch.tensor(flat, dtype=torch.int32).to(device), torch.tensor(off, dtype=torch.int64).to(device))


def pair_distances(pairs, device="cuda"):
    """-> (dist int32 [n], reflen int32 [n]) on ``device`` for a list of (reference symbols, hypothesis symbols)."""
    ids = {}
    enc = lambda seq: [ids.setdefault(t, len(ids) + 1) for t in seq]          # noqa: E731 - symbol ids as the C dictionary
    refs = [enc(c) for c, _ in pairs]
    hyps = [enc(s) for _, s in pairs]
    ref, ref_off = _pack(refs, device)
    hyp, hyp_off = _pack(hyps, device)
    max_len = max([0] + [len(s) for s in refs] + [len(s) for s in hyps])
    dist = ops.edit_distance(ref, ref_off, hyp, hyp_off, len(pairs), max_len)
    reflen = (ref_off[1:] - ref_off[:-1]).to(torch.int32)
    return dist, reflen


def error_rate(path: str, word_mode: bool, iters: int = BOOTSTRAP_ITERS, seed: int = 0, device="cuda"):
    """-> (rate, bootstrap mean, 1.64 sigma): ``tasas -ie`` and ``tasasIntervalo -ie`` of one file."""
    pairs = read_pairs(path, word_mode)
    if not pairs:
        raise ValueError(f"{path}: no sentence pairs")
    dist, reflen = pair_distances(pairs, device)
    rate = 100.0 * float(dist.sum(dtype=torch.int64)) / float(reflen.sum(dtype=torch.int64))
    rates = ops.bootstrap_rates(dist, reflen, iters, seed)
    mean = float(rates.mean())
    var = float((rates * rates).mean()) - mean * mean
    return rate, mean, Z * max(var, 0.0) ** 0.5


def compute_bootstrap_wer(path: str, iters: int = BOOTSTRAP_ITERS, seed: int = 0):
    """(wer, cer, ci_wer, ci_cer) as src/evaluation/bootstrap_wer.py:3-16 returns them."""
    wer, _, ci_wer = error_rate(path, True, iters, seed)
    cer, _, ci_cer = error_rate(path, False, iters, seed + 1)
    return wer, cer, ci_wer, ci_cer
