"""The K = 256 projections of the encoder / decoder layers (fewer 64x64 tiles than CUs: one workgroup per CU, so nothing hides a
K-step's load latency behind another workgroup's MFMAs): LDS-DMA ring depth 2 (the planner's choice so far), 3, 4 and 8 stages."""
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
    for M, N, K, kind in ((3168, 256, 256, "NT"), (3168, 256, 256, "NN"), (1312, 256, 256, "NT"), (3168, 768, 256, "NT"), (3168, 256, 768, "NN"),
                          (3168, 256, 1024, "NT"), (3168, 1024, 256, "NN"), (3168, 2048, 256, "NT"), (6400, 256, 256, "NT")):
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") if kind == "NT" else torch.randn(K, N, device="cuda")
        b = torch.randn(N, device="cuda")
        res = []
        for cfg in (None, 8, 4, 5, 10):
            force = None if cfg is None else (cfg, 1)
            if kind == "NT":
                fn = lambda: ops.linear(x, w, b, force=force)
            else:
                fn = lambda: ops.linear_dx(x, w, force=force)      # dy [M, K] @ w [K, N]
            try:
                res.append(f"{'plan' if cfg is None else 'cfg ' + str(cfg)} {timed(fn):6.1f}")
            except Exception as e:
                res.append(f"cfg {cfg} failed ({str(e)[:40]})")
        fl = 2.0 * M * N * K
        print(f"{kind} M={M} N={N} K={K} ({fl / 1e9:.2f} GFLOP): " + "  ".join(res) + "  us", flush=True)


if __name__ == "__main__":
    main()
