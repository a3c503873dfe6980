"""Per-shape GEMM timing table (HIP events around every tavsr_gemm launch of one bench step).
usage: python profiles/gemm_shapes.py [--workload asr|avsr] [--fwd-only]   (fwd-only: the forward pass alone, no_grad off)"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tailored-avsr_amd")]
import torch
import bench
from tavsr import ops

ap = argparse.ArgumentParser()
ap.add_argument("--workload", choices=("asr", "avsr"), default="asr")
ap.add_argument("--fwd-only", action="store_true")
args = ap.parse_args()
bench.WORKLOAD = args.workload
model = bench.build_product_model().cuda().train()
batch = bench.make_batch(32, 1234, "cuda")
def step():
    for p in model.parameters(): p.grad = None
    loss = model(*batch)[0]
    if not args.fwd_only: loss.backward()
for _ in range(2): step()
prof = ops.GemmProfile(by_shape=True); ops.PROFILE = prof
n = 3
for _ in range(n): step()
ops.PROFILE = None
s = prof.summary()
tot = sum(v["seconds"] for v in s.values()) / n
print(f"# {args.workload}{' forward only' if args.fwd_only else ' fwd+bwd'}: all GEMM launches: {1e3*tot:.3f} ms/step, "
      f"{sum(v['calls'] for v in s.values())//n} launches/step, {sum(v['flops'] for v in s.values())/n/tot/1e12:.1f} TFLOP/s")
print(f"{'calls':>6} {'ms/step':>8} {'avg_us':>8} {'TF/s':>7}  shape")
for k, v in sorted(s.items(), key=lambda kv: -kv[1]["seconds"]):
    print(f"{v['calls']//n:6d} {1e3*v['seconds']/n:8.3f} {1e6*v['seconds']/v['calls']:8.2f} {v['flops']/v['seconds']/1e12:7.2f}  {k}")
