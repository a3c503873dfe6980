"""Per-kernel device time of ONE eval forward of the 12 Branchformer layers at batch 32 (torch profiler, eager)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tailored-avsr_amd")]
import torch
from torch.profiler import ProfilerActivity, profile
import bench
bench.WORKLOAD = "asr"
model = bench.build_product_model().cuda().eval()
enc = model.encoder
from tavsr.layers import make_pad_mask
g = torch.Generator().manual_seed(1234)
speech = torch.randn(32, 400, 80, generator=g).cuda()
ilens = torch.full((32,), 400, dtype=torch.int64, device="cuda")
with torch.no_grad():
    masks = (~make_pad_mask(ilens, 400)[:, None, :]).cuda()
    (x0, pos), m0 = enc.embed(speech, masks)
    lens = m0.squeeze(1).sum(1).to(torch.int64)
    def layers():
        xs = (x0, pos)
        for layer in enc.encoders:
            xs, _ = layer(xs, m0, lens=lens)
        return enc.after_norm(xs[0])
    for _ in range(3): layers()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        layers(); torch.cuda.synchronize()
tot = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        t = tot[ev.name[:110]]; t[0] += 1; t[1] += ev.device_time
allus = sum(v[1] for v in tot.values())
print(f"# one eval forward of 12 layers: {allus/1e3:.3f} ms of kernel time, {sum(v[0] for v in tot.values())} launches")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{v[0]:5d} {v[1]/1e3:8.3f} ms {v[1]/v[0]:8.2f} us  {k}")
