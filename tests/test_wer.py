"""WER / CER evaluator: the oracle against the reference's own programs (oracle/_ref, compiled from
/root/reference by oracle/Makefile) and the fixture they produced; the HIP path against the oracle and the fixture."""
import json
import os
import random
import subprocess

import numpy as np
import pytest
import torch

from helpers import ROOT

from oracle import wer as OW

CASES = json.load(open(os.path.join(ROOT, "tests", "golden", "wer_cases.json"), encoding="utf-8"))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "tasas")


def _write(tmp_path, case):
    p = tmp_path / f"case{case['seed']}.inf"
    p.write_text(case["text"], encoding="utf-8")
    return str(p)


@pytest.mark.parametrize("case", CASES, ids=[f"seed{c['seed']}" for c in CASES])
def test_oracle_equals_the_reference_programs_outputs(case, tmp_path):
    path = _write(tmp_path, case)
    for mode, wm in (("wer", True), ("cer", False)):
        tot = np.zeros(4, dtype=np.int64)
        for c, s in OW.read_pairs(path, wm):
            tot += OW.gp_counts(c, s)
        assert list(tot) == case[mode + "_counts"]                     # substitutions, insertions, deletions, hits
        assert abs(OW.rate_ie(*tot) - case[mode]) < 1e-5               # the program prints 6 decimals
        mean, ci = OW.tasas_intervalo(path, wm, 1000, seed=3)
        ref_mean, ref_ci = case[mode + "_interval"]
        assert abs(mean - ref_mean) < 0.35 * ref_ci                    # two bootstrap means: sigma / sqrt(1000) apart
        assert abs(ci - ref_ci) < 0.15 * ref_ci


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref not built (make -C oracle ref needs /root/reference)")
def test_oracle_equals_the_compiled_reference_on_fresh_files(tmp_path):
    rng = random.Random(9)
    words = ["a", "bb", "ccc", "dd", "e", "ñu", "zz"]
    lines = []
    for _ in range(60):
        r = [rng.choice(words) for _ in range(rng.randint(1, 9))]
        h = [rng.choice(words) for _ in range(rng.randint(0, 9))]
        lines.append(" ".join(r) + "#" + " ".join(h))
    p = tmp_path / "fresh.inf"
    p.write_text("\n".join(lines) + "\n", encoding="utf-8")
    for wm in (True, False):
        cmd = [REF_BIN, "-f", "#"] + (["-s", " "] if wm else []) + ["-ie", str(p)]
        want = float(subprocess.check_output(cmd).strip())
        assert abs(OW.tasas(str(p), wm) - want) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[f"seed{c['seed']}" for c in CASES])
def test_hip_error_rates_equal_fixture_and_oracle(case, tmp_path):
    from tavsr.evaluation.bootstrap_wer import compute_bootstrap_wer, error_rate, pair_distances, read_pairs
    path = _write(tmp_path, case)
    for mode, wm in (("wer", True), ("cer", False)):
        pairs = read_pairs(path, wm)
        assert pairs == OW.read_pairs(path, wm)
        dist, reflen = pair_distances(pairs)
        want = [OW.gp_counts(c, s) for c, s in pairs]
        assert dist.cpu().tolist() == [ns + ni + nb for ns, ni, nb, _ in want]          # bit-exact distances
        assert reflen.cpu().tolist() == [ns + nb + na for ns, _, nb, na in want]
        rate, mean, ci = error_rate(path, wm, 1000, seed=5)
        assert abs(rate - case[mode]) < 1e-5
        ref_mean, ref_ci = case[mode + "_interval"]
        assert abs(mean - ref_mean) < 0.35 * ref_ci and abs(ci - ref_ci) < 0.15 * ref_ci
    wer, cer, ci_wer, ci_cer = compute_bootstrap_wer(path)
    assert abs(wer - case["wer"]) < 1e-5 and abs(cer - case["cer"]) < 1e-5 and ci_wer > 0 and ci_cer > 0


@pytest.mark.gpu
def test_hip_edit_distance_edge_cases_and_long_sequences():
    from tavsr import ops
    rng = random.Random(2)
    seqs = [([], []), ([1], []), ([], [1, 2, 3]), ([5] * 7, [5] * 7), ([1, 2, 3], [3, 2, 1])]
    seqs += [([rng.randint(1, 6) for _ in range(rng.randint(0, 40))], [rng.randint(1, 6) for _ in range(rng.randint(0, 40))])
             for _ in range(40)]
    long_r = [rng.randint(1, 30) for _ in range(2047)]
    long_h = [t for t in long_r if rng.random() > 0.1] + [7] * 50
    seqs.append((long_r, long_h))

    def pack(ss):
        off = [0]
        for s in ss:
            off.append(off[-1] + len(s))
        return (torch.tensor([t for s in ss for t in s] or [0], dtype=torch.int32).cuda(), torch.tensor(off, dtype=torch.int64).cuda())
    ref, ro = pack([a for a, _ in seqs])
    hyp, ho = pack([b for _, b in seqs])
    dist = ops.edit_distance(ref, ro, hyp, ho, len(seqs), 2047 + 50).cpu().tolist()
    for (a, b), d in zip(seqs, dist):
        prev = list(range(len(b) + 1))                              # textbook two-row Levenshtein
        for i in range(1, len(a) + 1):
            cur = [i] + [0] * len(b)
            for j in range(1, len(b) + 1):
                cur[j] = min(prev[j - 1] + (a[i - 1] != b[j - 1]), prev[j] + 1, cur[j - 1] + 1)
            prev = cur
        assert d == prev[len(b)]
    with pytest.raises(Exception):
        ops.edit_distance(ref, ro, hyp, ho, len(seqs), 5000)


@pytest.mark.gpu
def test_bootstrap_rates_are_resamples_of_the_sentence_set():
    from tavsr import ops
    g = torch.Generator().manual_seed(1)
    n = 333
    reflen = torch.randint(1, 20, (n,), generator=g, dtype=torch.int32)
    dist = (reflen.float() * torch.rand(n, generator=g) * 0.5).to(torch.int32)
    rates = ops.bootstrap_rates(dist.cuda(), reflen.cuda(), 4000, 11).cpu()
    again = ops.bootstrap_rates(dist.cuda(), reflen.cuda(), 4000, 11).cpu()
    other = ops.bootstrap_rates(dist.cuda(), reflen.cuda(), 4000, 12).cpu()
    assert torch.equal(rates, again) and not torch.equal(rates, other)            # counter-based: seed -> same resamples
    full = 100.0 * float(dist.sum()) / float(reflen.sum())
    # bootstrap mean ~ the full-set rate; spread ~ the delta-method standard error of a ratio estimator
    r = dist.double() - full / 100.0 * reflen.double()
    se = 100.0 * float(r.std()) * n ** 0.5 / float(reflen.sum())
    assert abs(float(rates.mean()) - full) < 4 * se / 4000 ** 0.5 + 0.02 * se
    assert abs(float(rates.std()) - se) < 0.1 * se
