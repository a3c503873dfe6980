"""learned_ave merge at the encoder's shape (B 32, T 99, D 256): the one-workgroup-per-utterance launches against the
row-parallel ones, forward and backward (the old route with its two dropout launches, which the new one absorbs)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
from tavsr import ops  # noqa: E402


def timed(fn, n=50, reps=10):
    """per-call GPU time of ``fn`` replayed from a captured graph of n calls (no host enqueue time in the figure)"""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            g.replay()
        b.record()
        torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * reps) * 1e3


def main():
    B, T, D = 32, 99, 256
    torch.manual_seed(0)
    lens = torch.randint(40, T + 1, (B,), device="cuda")
    x1, x2, dm = (torch.randn(B * T, D, device="cuda") for _ in range(3))
    prm = [torch.randn(1, D, device="cuda") / 4, torch.randn(1, D, device="cuda") / 4, torch.randn(1, device="cuda"),
           torch.randn(1, device="cuda"), torch.randn(1, D, device="cuda") / 4, torch.randn(1, D, device="cuda") / 4,
           torch.randn(1, device="cuda"), torch.randn(1, device="cuda")]
    ops.manual_seed(1)
    t1, t2 = ops._new_token(0.1, B * T * D, dm.device), ops._new_token(0.1, B * T * D, dm.device)
    for rows in (False, True):
        ops.MERGE_ROWS = rows
        score, aux, w, m = ops.merge_fwd(x1, x2, lens, prm, B, T)
        f = timed(lambda: ops.merge_fwd(x1, x2, lens, prm, B, T))
        b = timed(lambda: ops.merge_bwd(dm, x1, x2, lens, prm, score, aux, w, B, T, drop1=t1, drop2=t2))
        print(f"rows={int(rows)}  forward {f:.1f} us  backward (+ both branch masks) {b:.1f} us")


if __name__ == "__main__":
    main()
