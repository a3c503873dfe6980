"""TEST INFRASTRUCTURE ONLY - writes tests/golden/wer_cases.json: synthetic "reference#hypothesis" files and what the
reference's OWN evaluator (oracle/_ref/tasas, tasasIntervalo = src/evaluation/tasas/*.c compiled by oracle/Makefile)
prints for them with the options of src/evaluation/bootstrap_wer.py:4-11.  Run here (needs /root/reference):
    make -C oracle ref && python -m oracle.gen_golden_wer"""
from __future__ import annotations

import json
import os
import random
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
VOCAB = ["hola", "que", "tal", "buenos", "días", "el", "la", "niño", "señor", "está", "aquí", "mañana", "por", "favor",
         "gracias", "a", "de", "y", "lluvia", "café", "uno", "dos", "tres", "x", "sí", "no"]


def corrupt(words, rng, rate):
    out = []
    for w in words:
        u = rng.random()
        if u < rate / 3:
            continue                                     # deletion
        if u < 2 * rate / 3:
            out.append(rng.choice(VOCAB))                # substitution
        else:
            out.append(w)
        if rng.random() < rate / 3:
            out.append(rng.choice(VOCAB))                # insertion
    return out


def make_case(seed, n, rate, quirks):
    rng = random.Random(seed)
    lines = []
    for i in range(n):
        ref = [rng.choice(VOCAB) for _ in range(rng.randint(1, 14))]
        hyp = corrupt(ref, rng, rate)
        r, h = " ".join(ref), " ".join(hyp)
        if quirks and i % 7 == 3:
            h = "  " + h.replace(" ", "   ", 1) + " "      # runs of blanks, leading / trailing blanks
        if quirks and i % 11 == 5:
            h = ""                                         # empty hypothesis
        if quirks and i % 13 == 6:
            h = h + "#" + h                                # a second '#': only the first one separates
        lines.append(r + "#" + h)
    return "\n".join(lines) + "\n"


def run(tool, path, word_mode, extra=()):
    cmd = [os.path.join(REF, tool), "-f", "#"] + (["-s", " "] if word_mode else []) + ["-ie", path] + list(extra)
    return subprocess.check_output(cmd).decode("utf-8", "replace")


def main():
    cases = []
    for seed, n, rate, quirks in [(1, 40, 0.3, False), (2, 200, 0.15, True), (3, 25, 0.6, True), (4, 600, 0.08, False)]:
        text = make_case(seed, n, rate, quirks)
        with tempfile.NamedTemporaryFile("w", suffix=".inf", delete=False, encoding="utf-8") as f:
            f.write(text)
            path = f.name
        case = {"seed": seed, "text": text}
        for mode, wm in (("wer", True), ("cer", False)):
            out = run("tasas", path, wm, ["-v"]).split("\n")
            case[mode] = float(out[0])
            case[mode + "_counts"] = [int(x.split("=")[1]) for x in out[1].split()]      # sust ins borr ac
            iv = [run("tasasIntervalo", path, wm) for _ in range(1)][0]
            case[mode + "_interval"] = [float(x) for x in iv.replace("+-", " ").split()]
        os.unlink(path)
        cases.append(case)
    dst = os.path.join(ROOT, "tests", "golden", "wer_cases.json")
    json.dump(cases, open(dst, "w"), ensure_ascii=False, indent=0)
    for c in cases:
        print(c["seed"], c["wer"], c["wer_counts"], c["wer_interval"], c["cer"], c["cer_counts"], c["cer_interval"])


if __name__ == "__main__":
    main()
