"""ring depth of the streaming feed-forward kernel at M = 3168 / 3200 / 6400: plans (wpb, NS) interleaved, three rounds"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from ffn2_bench import timed
from tavsr import ops
D, N1 = 256, 2048
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
ln_w, ln_b, w1, b1, w2, b2 = 1 + 0.1 * r(D), 0.1 * r(D), r(N1, D) / 16, 0.1 * r(N1), r(D, N1) / 45, 0.1 * r(D)
g2, c2 = 1 + 0.1 * r(D), 0.1 * r(D)
for M in (3168, 3200, 6400):
    x = r(M, D)
    wpb = max(1, 256 // ((M + 127) // 128))
    for rnd in range(3):
        row = []
        for mode, p, save in (("eval", 0.0, False), ("train", 0.1, True)):
            for ns in (3, 4, 5):
                os.environ["TAVSR_FFN2_CFG"] = f"{wpb},{ns}"
                us = timed(lambda: ops.ffn2_fwd(x, ln_w, ln_b, 1e-12, w1, b1, w2, b2, "swish", 0.5, p=p, save=save, ln2=((g2, c2),), ln2_stats=save))
                row.append(f"{mode} NS={ns} {us:6.1f}")
        print(f"M={M} wpb={wpb} round {rnd}: " + " | ".join(row), flush=True)
os.environ.pop("TAVSR_FFN2_CFG", None)
