"""Timing-only ablation of the GEMM K-step (needs a build with -DTAVSR_GEMM_ABLATE): which part costs what."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tailored-avsr_amd")]
import torch
from tavsr import ops
from gemm_sweep import run

NAMES = {108: "64x64 full, reads pinned", 115: "64x64 pure loop, reads pinned", 118: "128x128 full, pinned", 125: "128x128 pure, pinned", 4: "64x64p2 full", 101: "64x64 no-gload", 102: "64x64 no-ldswrite", 103: "64x64 no-gload/ldswrite",
         107: "64x64 no-gload/ldswrite/barrier", 0: "128x128p1 full", 111: "128x128 no-gload",
         113: "128x128 no-gload/ldswrite", 117: "128x128 no-gload/ldswrite/barrier"}
for mode, M, N, K in [("NT", 3168, 256, 2048), ("NT", 3168, 2048, 256), ("TN", 2048, 256, 3168), ("NT", 60192, 256, 2304)]:
    a = torch.randn(1, M, K, device="cuda"); b = torch.randn(1, K, N, device="cuda")
    A = a.transpose(1, 2).contiguous() if mode == "TN" else a
    B = b if mode != "NT" else b.transpose(1, 2).contiguous()
    C = torch.empty(1, M, N, device="cuda")
    ideal = 2.0 * M * N * K / 157.3e12 * 1e6
    print(f"{mode} M={M} N={N} K={K}  ideal {ideal:.1f} us")
    for cfg, name in NAMES.items():
        t = run(mode, M, N, K, 1, (cfg, 1), A, B, C, 10)
        print(f"   {name:36s} {t:8.1f} us")
