"""Times the backward of the feed-forward block w.r.t. its activations: tavsr_ffn2_bwd_dx against the two dgrad GEMM
launches, and the whole _FFN.bwd both ways.  M = 3168, hidden 2048, dropout 0.1.  One process, hipGraph replay."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from ffn2_bench import timed  # noqa: E402
from tavsr import functional as F_  # noqa: E402
from tavsr import ops  # noqa: E402


def main():
    D, N1 = 256, 2048
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    ln_w, ln_b, w1, b1, w2, b2 = 1 + 0.1 * r(D), 0.1 * r(D), r(N1, D) / 16, 0.1 * r(N1), r(D, N1) / 45, 0.1 * r(D)
    for M in (3168, 6400, 1312):
        x, dy = r(M, D), r(M, D)
        ops.manual_seed(1)
        y, saved = F_._FFN.fwd(x, ln_w, ln_b, w1, b1, w2, b2, "swish", 0.5, p=0.1)
        _, mean, rstd, n, z, h, t_in, t_out = saved
        dyd = F_._drop_bwd(dy, t_out)
        gf = 2 * 2 * M * D * N1 / 1e9
        rows = []
        def gemms():
            dz = ops.linear_dx_drop(dyd, w2, t_in, alpha=0.5, DZ=z, dact="swish")
            return dz, ops.linear_dx(dz, w1)
        rows.append(("dgrad GEMM + GEMM", timed(gemms)))
        for cfg in (None, "8,4", "13,4"):
            if cfg is None:
                os.environ.pop("TAVSR_FFN2_CFG", None)
            else:
                os.environ["TAVSR_FFN2_CFG"] = cfg
            rows.append((f"ffn2_bwd_dx wpb={cfg}", timed(lambda: ops.ffn2_bwd_dx(dyd, 0.5, w1, w2, z, "swish", t_in))))
        os.environ.pop("TAVSR_FFN2_CFG", None)
        os.environ["TAVSR_FFN2_DBG"] = "1"
        rows.append(("ffn2_bwd_dx same weight tile", timed(lambda: ops.ffn2_bwd_dx(dyd, 0.5, w1, w2, z, "swish", t_in))))
        os.environ.pop("TAVSR_FFN2_DBG", None)
        for v in (False, True):
            ops.FFN2_BWD = v
            rows.append((f"_FFN.bwd streaming={v}", timed(lambda: F_._FFN.bwd(dy, saved, ln_w, w1, w2, "swish", 0.5))))
        ops.FFN2_BWD = True
        for name, us in rows:
            print(f"M={M} {name:34s} {us:8.1f} us  {gf / us * 1e3:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
