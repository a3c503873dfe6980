"""Where the host time of an eager (un-captured) audio-only training step goes: enqueue time of forward / backward without
waiting for the GPU, the GPU time of the same step, and a cProfile table of the Python side."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
import bench  # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="asr")
    a = ap.parse_args()
    bench.WORKLOAD = a.workload
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = bench.build_product_model().to(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    batch = bench.make_batch(bench.B_PER_GPU, 1234, dev)

    def fwd():
        for p in params:
            p.grad = None
        return model(*batch)[0]

    for _ in range(3):
        fwd().backward()
    torch.cuda.synchronize()
    for _ in range(3):
        t0 = time.perf_counter()
        loss = fwd()
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print(f"enqueue forward {1e3 * (t1 - t0):.2f} ms, backward {1e3 * (t2 - t1):.2f} ms, GPU drain {1e3 * (t3 - t2):.2f} ms, "
              f"step {1e3 * (t3 - t0):.2f} ms", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        fwd().backward()
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
    print(s.getvalue()[:9000])
    # the backward functions run on the autograd engine's thread, which the profile above does not see: profile the encoder
    # layer's backward from inside (12 calls per step)
    from tavsr import functional as F_
    prb = cProfile.Profile()
    orig = F_.BranchformerLayerFn.backward

    def wrapped(ctx, dy):
        prb.enable()
        try:
            return orig(ctx, dy)
        finally:
            prb.disable()

    F_.BranchformerLayerFn.backward = staticmethod(wrapped)
    for _ in range(3):
        fwd().backward()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(prb, stream=s).sort_stats("tottime").print_stats(40)
    print("# BranchformerLayerFn.backward, 36 calls (3 steps x 12 layers)")
    print(s.getvalue()[:9000])


if __name__ == "__main__":
    main()
