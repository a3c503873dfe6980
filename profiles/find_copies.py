"""Which host-side calls produce device-to-device copies in one eager fwd+bwd step (torch profiler, stacks).
usage: python profiles/find_copies.py [asr|avsr] > gpurun_out/copies.txt"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tailored-avsr_amd")]
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402


def main():
    if len(sys.argv) > 1:
        bench.WORKLOAD = sys.argv[1]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = bench.build_product_model().to(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    batch = bench.make_batch(bench.B_PER_GPU, 1234, dev)

    def step():
        for p in params:
            p.grad = None
        loss = model(*batch)[0]
        loss.backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    cnt = collections.Counter()
    for ev in prof.events():
        if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::mul",
                       "aten::to", "aten::_to_copy", "aten::sum", "aten::eq", "aten::ne", "aten::index", "aten::masked_fill_"):
            stack = [s for s in ev.stack if "tavsr" in s or "bench" in s or "autograd" in s][:2]
            cnt[(ev.name, str(ev.input_shapes)[:60], " <- ".join(s.split("/")[-1][:70] for s in stack))] += 1
    for (name, shp, st), n in cnt.most_common(60):
        print(f"{n:5d} {name:18s} {shp:60s} {st}")
    print()
    # device-to-device copies by the op that issued them (autograd's AccumulateGrad clones views / non-stealable grads)
    by = collections.Counter()
    for ev in prof.events():
        if "copy" in ev.name.lower() or "clone" in ev.name.lower():
            par, chain = ev.cpu_parent, []
            while par is not None and len(chain) < 3:
                chain.append(par.name[:50])
                par = par.cpu_parent
            by[(ev.name[:40], str(ev.input_shapes)[:50], " <- ".join(chain))] += 1
    for (name, shp, chain), n in by.most_common(40):
        print(f"{n:5d} {name:40s} {shp:50s} {chain}")
    print()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))


if __name__ == "__main__":
    main()
