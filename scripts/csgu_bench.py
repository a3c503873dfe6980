"""Times the CSGU between the channel projections of the cgMLP branch (M = 3168 rows, 2 x 1024 channels, 31 taps): the fused pass
(tavsr_csgu_fwd: statistics launch + normalise / convolve / gate / dropout launch) against LayerNorm + dwconv_gate (+ dropout)."""
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
    B, T, Cn, K = 32, 99, 1024, 31
    g = torch.randn(B * T, 2 * Cn, device="cuda")
    lw, lb = torch.ones(Cn, device="cuda"), torch.zeros(Cn, device="cuda")
    w, bias = torch.randn(Cn, K, device="cuda") / 5, torch.randn(Cn, device="cuda")
    for mode, p, save in (("eval", 0.0, False), ("train", 0.1, True)):
        def base():
            gn, m, r = ops.layernorm_fwd(g[:, Cn:], lw, lb, 1e-12, save=save)
            u, conv = ops.dwconv_gate_fwd(gn, g[:, :Cn], w, bias, B, T)
            return u, conv, F_._drop_(u, p)
        print(f"{mode:6s} LayerNorm + dwconv_gate (+ dropout) {timed(base):7.1f} us", flush=True)
        print(f"{mode:6s} tavsr_csgu_fwd                       {timed(lambda: ops.csgu_fwd(g, lw, lb, 1e-12, w, bias, B, T, p=p, save=save)):7.1f} us", flush=True)
        # with the LayerNorm statistics from channel_proj1's epilogue (the layer's form: no statistics launch)
        gg = g[:, Cn:].double().view(B * T, Cn // 64, 64)
        rst = torch.zeros(B * T, 2 * Cn // 64, 2, device="cuda")
        rst[:, Cn // 64:, 0], rst[:, Cn // 64:, 1] = gg.sum(-1).float(), (gg * gg).sum(-1).float()
        print(f"{mode:6s} tavsr_csgu_fwd, statistics given     {timed(lambda: ops.csgu_fwd(g, lw, lb, 1e-12, w, bias, B, T, p=p, save=save, rowstat=rst)):7.1f} us", flush=True)


if __name__ == "__main__":
    main()
