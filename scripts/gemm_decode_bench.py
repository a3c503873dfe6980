"""The GEMMs of a batched one-token search step (640 hypothesis rows = 64 utterances x beam 10: 80 - 320 tiles of 64 x 64 per launch, every
tile alone on its compute unit): the planner's choice against other tile configurations and K splits (tavsr_gemm_tune), us per call
in a captured chain."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from merge_bench import timed  # noqa: E402
from tavsr import ops  # noqa: E402


def main():
    torch.manual_seed(0)
    M = 640
    for N, K, what in ((1536, 512, "LM q/k/v"), (512, 512, "LM out"), (2048, 512, "LM w_1"), (512, 2048, "LM w_2"), (768, 256, "dec q/k/v"),
                       (256, 256, "dec out / q2"), (2048, 256, "dec w_1"), (256, 2048, "dec w_2")):
        x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.randn(N, device="cuda")
        res = [f"plan {timed(lambda: ops.linear(x, w, b)):6.1f}"]
        for cfg, ns in ((8, 1), (5, 1), (4, 1), (2, 1), (3, 1), (8, 2), (5, 2), (8, 4), (5, 4)):
            try:
                res.append(f"cfg {cfg} x{ns} {timed(lambda: ops.linear(x, w, b, force=(cfg, ns))):6.1f}")
            except Exception as e:       # noqa: BLE001
                res.append(f"cfg {cfg} x{ns} -")
        print(f"{what:12s} M={M} N={N} K={K} ({2.0 * M * N * K / 1e9:.2f} GFLOP): " + "  ".join(res) + "  us", flush=True)


if __name__ == "__main__":
    main()
