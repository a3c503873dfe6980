"""Times the streaming linear launch (tavsr_lin2_fwd) against the tiled GEMM launches at the encoder's shapes: the grouped query /
key / value projections (M = 3168, 3 x 256 columns) and cgMLP's channel_proj1 + GELU (2048 columns, eval and with the
pre-activations kept)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from ffn2_bench import timed  # noqa: E402
from tavsr import ops  # noqa: E402


def main():
    D = 256
    for M in (3168, 6400, 1312):
        x = torch.randn(M, D, device="cuda")
        ws = [(torch.randn(D, D, device="cuda") / 16, torch.randn(D, device="cuda"), j * D) for j in range(3)]
        w1, b1 = torch.randn(2048, D, device="cuda") / 16, torch.randn(2048, device="cuda")
        qkv = torch.empty(M, 3 * D, device="cuda")
        for on in (False, True):
            ops.LIN2 = on
            name = "streaming" if on else "tiled GEMM"
            t_qkv = timed(lambda: ops.linear_group(x, ws, qkv))
            t_p1 = timed(lambda: ops.linear(x, w1, b1, act="gelu"))
            t_p1z = timed(lambda: ops.linear(x, w1, b1, act="gelu", save_z=True))
            print(f"M={M} {name:11s} qkv {t_qkv:6.1f} us ({2 * M * 768 * D / t_qkv / 1e6:5.1f} TFLOP/s)   proj1+GELU {t_p1:6.1f} us "
                  f"({2 * M * 2048 * D / t_p1 / 1e6:5.1f})   with z {t_p1z:6.1f} us", flush=True)
        if M == 3168:
            for wpb in ("6", "8", "10"):
                os.environ["TAVSR_LIN2_WPB"] = wpb
                print(f"M={M} wpb={wpb}: qkv {timed(lambda: ops.linear_group(x, ws, qkv)):6.1f} us   proj1+GELU "
                      f"{timed(lambda: ops.linear(x, w1, b1, act='gelu')):6.1f} us", flush=True)
            os.environ.pop("TAVSR_LIN2_WPB", None)


if __name__ == "__main__":
    main()
