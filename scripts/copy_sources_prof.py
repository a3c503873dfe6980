"""Where do the ~130 device-to-device copies (`__amd_rocclr_copyBuffer`) of an audio-only step come from?  torch profiler with
Python stacks over one eager step: aten::copy_ / aten::clone / aten::contiguous call sites by count."""
import collections, os, sys, torch
ROOT = "/root/repo" if os.path.exists("/root/repo/scripts") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
import bench
bench.WORKLOAD = "asr"
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench.build_product_model().to(dev).train()
batch = bench.make_batch(bench.B_PER_GPU, 1234, dev)
params = [p for p in model.parameters() if p.requires_grad]
def step():
    for p in params:
        p.grad = None
    model(*batch)[0].backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::add_", "aten::zeros"):
        st = [s for s in (e.stack or []) if "tavsr" in s or "bench" in s or "autograd" in s][:2]
        cnt[(e.name, str(e.input_shapes)[:60], " <- ".join(s.split("/")[-1][:70] for s in st))] += 1
for (k, n) in cnt.most_common(40):
    print(n, k)
ker = collections.Counter()
for e in prof.events():
    if e.device_type is not None and str(e.device_type).endswith("CUDA") and ("copy" in e.name.lower() or "Memcpy" in e.name):
        ker[e.name[:80]] += 1
print(ker.most_common(10))
