"""The layer's tail behind the branch join at the encoder's shape (B 32, T 99, D 256): row-parallel merge + merge_proj GEMM
launches against the one fused launch (tavsr_merge_proj_fwd), eval and train (dropout 0.1, mixed rows kept)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from tavsr import ops  # noqa: E402
from merge_bench import timed  # noqa: E402


def main():
    B, T, D = 32, 99, 256
    torch.manual_seed(0)
    lens = torch.randint(40, T + 1, (B,), device="cuda")
    x1, x2, res = (torch.randn(B * T, D, device="cuda") for _ in range(3))
    prm = [torch.randn(1, D, device="cuda") / 4, torch.randn(1, D, device="cuda") / 4, torch.randn(1, device="cuda"),
           torch.randn(1, device="cuda"), torch.randn(1, D, device="cuda") / 4, torch.randn(1, D, device="cuda") / 4,
           torch.randn(1, device="cuda"), torch.randn(1, device="cuda")]
    W, b = torch.randn(D, D, device="cuda") / 16, torch.randn(D, device="cuda")
    ops.manual_seed(1)
    for p, save in ((0.0, False), (0.1, True)):
        def launches():
            _, _, _, m = ops.merge_fwd(x1, x2, lens, prm, B, T)
            return ops.linear_drop(m, W, b, p, alpha=1.0, res=res)

        def fused():
            return ops.merge_proj_fwd(x1, x2, lens, prm, W, b, res, 1.0, p, B, T, save=save)

        print(f"p={p} save={int(save)}: merge_rows + GEMM launches {timed(launches):.1f} us, fused {timed(fused):.1f} us")


if __name__ == "__main__":
    main()
