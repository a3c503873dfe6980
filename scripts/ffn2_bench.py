"""Times the streaming feed-forward block (tavsr_ffn2_fwd) under several plans against the LayerNorm + GEMM + GEMM launches,
M = 3168 (encoder) and 6400 (both AV streams), hidden 2048, eval and train (save + dropout) forms.  One process, interleaved."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
from tavsr import functional as F_  # noqa: E402
from tavsr import ops  # noqa: E402


def timed(fn, iters=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(10):
            keep = fn()   # noqa: F841
    for _ in range(2):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters // 10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters // 10 * 10) * 1e3


def main():
    D, N1 = 256, 2048
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    ln_w, ln_b, w1, b1, w2, b2 = 1 + 0.1 * r(D), 0.1 * r(D), r(N1, D) / 16, 0.1 * r(N1), r(D, N1) / 45, 0.1 * r(D)
    g2, c2 = 1 + 0.1 * r(D), 0.1 * r(D)
    for M in (3168, 6400):
        x = r(M, D)
        gf = 2 * 2 * M * D * N1 / 1e9
        rows = []
        for mode, p, save in (("eval", 0.0, False), ("train", 0.1, True)):
            ops.FFN2 = False
            def base():
                y, sv = F_._FFN.fwd(x, ln_w, ln_b, w1, b1, w2, b2, "swish", 0.5, p=p, save=save)
                return ops.layernorm_fwd(y, g2, c2, 1e-12)
            us = timed(base)
            rows.append((f"{mode} LN+GEMM+GEMM(+LN)", us))
            ops.FFN2 = True
            for cfg in (None, "10,3", "10,5", "8,4"):
                if cfg is None:
                    os.environ.pop("TAVSR_FFN2_CFG", None)
                else:
                    os.environ["TAVSR_FFN2_CFG"] = cfg
                def fused():
                    return ops.ffn2_fwd(x, ln_w, ln_b, 1e-12, w1, b1, w2, b2, "swish", 0.5, p=p, save=save, ln2=((g2, c2),),
                                        ln2_stats=save)
                rows.append((f"{mode} ffn2 wpb,NS={cfg}", timed(fused)))
            os.environ.pop("TAVSR_FFN2_CFG", None)
        for name, us in rows:
            print(f"M={M} {name:34s} {us:8.1f} us  {gf / us * 1e3:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
