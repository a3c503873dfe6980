"""Tile-configuration sweep of tavsr_gemm over the hot-path shapes (MI355X): for each shape, every tile config x
K-split is timed with HIP events around R back-to-back launches; the planner's own choice is timed beside them.
usage: python profiles/gemm_sweep.py [--quick] > gpurun_out/gemm_sweep.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tailored-avsr_amd")]
import torch  # noqa: E402

from tavsr import ops  # noqa: E402

CFG_NAMES = ["128x128s3", "128x128w8s3", "128x64s3", "64x128s3", "64x64s3", "64x64s4", "128x64s4", "64x64kw2", "64x64s2"]
# (mode, M, N, K, nb)
SHAPES = [
    ("NT", 3168, 2048, 256, 1), ("NT", 3168, 256, 2048, 1), ("NT", 3168, 256, 256, 1), ("NT", 3168, 768, 256, 1),
    ("NT", 3168, 256, 1024, 1), ("NT", 60192, 256, 2304, 1), ("NT", 3168, 256, 4864, 1), ("NT", 1312, 256, 256, 1),
    ("NT", 1312, 2048, 256, 1), ("NT", 1312, 256, 2048, 1),
    ("NN", 3168, 256, 2048, 1), ("NN", 3168, 2048, 256, 1), ("NN", 3168, 256, 256, 1), ("NN", 3168, 1024, 256, 1),
    ("NN", 60192, 2304, 256, 1), ("NN", 3168, 4864, 256, 1), ("NN", 3168, 256, 768, 1),
    ("TN", 2048, 256, 3168, 1), ("TN", 256, 2048, 3168, 1), ("TN", 256, 256, 3168, 1), ("TN", 256, 1024, 3168, 1),
    ("TN", 768, 256, 3168, 1), ("TN", 256, 2304, 60192, 1), ("TN", 256, 4864, 3168, 1), ("TN", 256, 256, 1312, 1),
    ("NT", 99, 99, 64, 128), ("NT", 99, 197, 64, 128), ("NN", 99, 64, 99, 128), ("TN", 99, 64, 99, 128),
]
SPLITS = [1, 2, 3, 4, 6]


def run(mode, M, N, K, nb, force, A, B, Cout, reps):
    kw = dict(a_kmajor=mode == "TN", b_kmajor=mode != "NT", nb1=nb,
              sA=(A.stride(0), 0), sB=(B.stride(0), 0), sC=(M * N, 0))
    lda = A.stride(1)
    ldb = B.stride(1)
    ops.gemm(M, N, K, A, lda, B, ldb, Cout, N, force=force, **kw)   # warm
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(M, N, K, A, lda, B, ldb, Cout, N, force=force, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps   # us


def main():
    quick = "--quick" in sys.argv
    torch.manual_seed(0)
    print(f"{'shape':34s} {'ideal_us':>8s} | planner us (TF/s) | best forced: cfg split us (TF/s) | all forced (us)")
    for mode, M, N, K, nb in SHAPES:
        a = torch.randn(nb, M, K, device="cuda")
        b = torch.randn(nb, K, N, device="cuda")
        A = a.transpose(1, 2).contiguous() if mode == "TN" else a          # [nb,K,M] k-major
        B = b if mode != "NT" else b.transpose(1, 2).contiguous()          # NT: W[N,K]
        Cout = torch.empty(nb, M, N, device="cuda")
        ref = (a.double() @ b.double())
        flops = 2.0 * M * N * K * nb
        reps = 5 if flops > 5e10 else 20
        t_plan = run(mode, M, N, K, nb, None, A, B, Cout, reps)
        err = float((Cout.double() - ref).abs().max() / ref.abs().max())
        res = []
        for cfg in range(len(CFG_NAMES)):
            for ns in SPLITS:
                if ns > 1 and (K // ns < 128 or nb > 1 or ns * M * N > 3e8):
                    continue
                if quick and ns not in (1, 4):
                    continue
                Cout.zero_()
                t = run(mode, M, N, K, nb, (cfg, ns), A, B, Cout, reps)
                e = float((Cout.double() - ref).abs().max() / ref.abs().max())
                assert e < 2e-5, (mode, M, N, K, cfg, ns, e)
                res.append((t, cfg, ns))
        res.sort()
        t, cfg, ns = res[0]
        ideal = flops / 157.3e12 * 1e6
        allr = " ".join(f"{CFG_NAMES[c]}/{s}:{tt:.1f}" for tt, c, s in sorted(res, key=lambda r: (r[1], r[2])))
        print(f"{mode} M={M} N={N} K={K} nb={nb:<4d}".ljust(34) + f" {ideal:8.1f} | {t_plan:7.1f} ({flops / t_plan / 1e6:5.1f}) err {err:.1e} | "
              f"{CFG_NAMES[cfg]} {ns:2d} {t:7.1f} ({flops / t / 1e6:5.1f}) | {allr}", flush=True)


if __name__ == "__main__":
    main()
