"""A few launches of selected GEMM shapes (planner's choice and forced tiles) for rocprofv3 --pmc runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tailored-avsr_amd"), os.path.join(ROOT, "profiles")]
import torch
from gemm_sweep import run
for mode, M, N, K, cfg in [("NT", 3168, 2048, 256, 4), ("NT", 3168, 256, 2048, 4), ("NT", 60192, 256, 2304, 6), ("NT", 60192, 256, 2304, 4),
                           ("TN", 2048, 256, 3168, 4), ("NT", 3168, 2048, 256, 6)]:
    a = torch.randn(1, M, K, device="cuda"); b = torch.randn(1, K, N, device="cuda")
    A = a.transpose(1, 2).contiguous() if mode == "TN" else a
    B = b if mode != "NT" else b.transpose(1, 2).contiguous()
    C = torch.empty(1, M, N, device="cuda")
    t = run(mode, M, N, K, 1, (cfg, 1), A, B, C, 3)
    print(mode, M, N, K, cfg, f"{t:.1f} us")
