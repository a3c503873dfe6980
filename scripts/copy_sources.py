"""Which host calls put `__amd_rocclr_copyBuffer` (hipMemcpyAsync) launches into an eager training step: one step under the
torch profiler, aten::copy_ / aten::clone / aten::to events grouped by input shapes and by the innermost Python frames."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
import bench  # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="asr")
    ap.add_argument("--no-opt", action="store_true", help="fwd+bwd only, as bench.py steps (gradients land through AccumulateGrad)")
    a = ap.parse_args()
    bench.WORKLOAD = a.workload
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = bench.build_product_model().to(dev).train()
    from tavsr.train import FusedAdam
    opt = None if a.no_opt else FusedAdam(model.parameters(), lr=1e-4)
    params = [p for p in model.parameters() if p.requires_grad]
    batch = bench.make_batch(bench.B_PER_GPU, 1234, dev)

    def step():
        for p in params:
            p.grad = None
        model(*batch)[0].backward()
        if opt is not None:
            opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    by = collections.Counter()
    for e in prof.events():
        if e.name in ("aten::copy_", "aten::clone", "aten::_to_copy", "aten::contiguous", "aten::add_", "aten::add"):
            st = [s for s in (e.stack or []) if "tavsr" in s or "bench" in s or "scripts" in s][:2]
            by[(e.name, str(e.input_shapes)[:60], " <- ".join(s.split("/")[-1] for s in st))] += 1
    for (name, shp, st), n in sorted(by.items(), key=lambda kv: -kv[1])[:60]:
        print(f"{n:5d}  {name:16s} {shp:60s} {st}")


if __name__ == "__main__":
    main()
