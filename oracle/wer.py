"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's error-rate evaluator.

Follows src/evaluation/tasas/tasas.c as called by src/evaluation/bootstrap_wer.py:3-16 (``-f "#" [-s " "] -ie``):
  * ``read_pairs``   - lee_datos (tasas.c:696-754): one "reference#hypothesis" pair per line, split at the FIRST '#';
                       symbols are blank-separated words (cadena_con_separadores, :672-699: runs of separators collapse,
                       leading / trailing ones are skipped) or single BYTES (cadena_sin_separadores with size 1, :644-662).
  * ``gp_counts``    - gp() (tasas.c:336-389) with p = 1: unit costs, ties resolved substitution/match <= deletion,
                       insertion < deletion, backtrace from the end -> (substitutions, insertions, deletions, hits).
  * ``rate_ie``      - tasa_ie (tasas.c:471-475): 100 (ns + ni + nb) / (ns + nb + na).
  * ``tasas``        - Gp over all pairs + the rate (tasas.c:404-414, 882-887).
  * ``tasas_intervalo`` - tasasIntervalo.c:934-977: max_iter resamples of the sentence set with replacement, mean rate and
                       1.64 x standard deviation (the C program seeds rand() with time(0): only its distribution is defined).
Pinned against the reference's own programs compiled from /root/reference (oracle/Makefile -> oracle/_ref/) in
tests/test_wer.py, and by the fixture tests/golden/wer_cases.json those programs produced (oracle/gen_golden_wer.py)."""
from __future__ import annotations

import numpy as np


def _symbols(raw: bytes, word_mode: bool):
    if word_mode:
        return [w for w in raw.split(b" ") if w]
    return [raw[i:i + 1] for i in range(len(raw))]


def read_pairs(path, word_mode: bool):
    pairs = []
    with open(path, "rb") as f:
        for ln, line in enumerate(f.read().split(b"\n")):
            if line == b"" :
                continue
            line = line[:2047]                                   # fgets(linea, 2048, fp)
            k = line.find(b"#")
            if k < 0:
                raise ValueError(f"no '#' separator in line {ln + 1}")
            pairs.append((_symbols(line[:k], word_mode), _symbols(line[k + 1:], word_mode)))
    return pairs


def gp_counts(c, s):
    """(ns, ni, nb, na) of reference ``c`` against hypothesis ``s`` as tasas.c:gp() counts them (p = 1)."""
    n, m = len(c), len(s)
    d = np.zeros((n + 1, m + 1))
    va = np.zeros((n + 1, m + 1), dtype=np.int8)               # 0 end, 1 substitution/match, 2 insertion, 3 deletion
    for i in range(1, n + 1):
        d[i, 0] = d[i - 1, 0] + 1.0
        va[i, 0] = 3
    for j in range(1, m + 1):
        d[0, j] = d[0, j - 1] + 1.0
        va[0, j] = 2
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            dsa = d[i - 1, j - 1] + (0.0 if c[i - 1] == s[j - 1] else 1.0)
            di = d[i, j - 1] + 1.0
            db = d[i - 1, j] + 1.0
            if dsa <= di:
                if dsa <= db:
                    d[i, j], va[i, j] = dsa, 1
                else:
                    d[i, j], va[i, j] = db, 3
            elif di < db:
                d[i, j], va[i, j] = di, 2
            else:
                d[i, j], va[i, j] = db, 3
    ns = ni = nb = na = 0
    i, j = n, m
    while va[i, j] != 0:
        if va[i, j] == 1:
            if c[i - 1] == s[j - 1]:
                na += 1
            else:
                ns += 1
            i, j = i - 1, j - 1
        elif va[i, j] == 2:
            ni += 1
            j -= 1
        else:
            nb += 1
            i -= 1
    return ns, ni, nb, na


def rate_ie(ns, ni, nb, na):
    return 100.0 * (ns + ni + nb) / (ns + nb + na)


def tasas(path, word_mode: bool):
    tot = np.zeros(4, dtype=np.int64)
    for c, s in read_pairs(path, word_mode):
        tot += gp_counts(c, s)
    return rate_ie(*tot)


def tasas_intervalo(path, word_mode: bool, max_iter: int = 1000, seed: int = 0):
    counts = np.array([gp_counts(c, s) for c, s in read_pairs(path, word_mode)], dtype=np.int64)
    rng = np.random.default_rng(seed)
    rates = np.empty(max_iter)
    for it in range(max_iter):
        t = counts[rng.integers(0, len(counts), len(counts))].sum(0)
        rates[it] = rate_ie(*t)
    mean = rates.mean()
    return mean, 1.64 * np.sqrt(max((rates ** 2).mean() - mean * mean, 0.0))
