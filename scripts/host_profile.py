"""Host-side cost of one eager fwd+bwd step (cProfile): where the Python time of the launch sequence goes.
usage: python scripts/host_profile.py [asr|avsr]"""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tailored-avsr_amd")]
import torch
import bench
bench.WORKLOAD = sys.argv[1] if len(sys.argv) > 1 else "asr"
model = bench.build_product_model().cuda().train()
batch = bench.make_batch(32, 1234, "cuda")
params = [p for p in model.parameters()]
def step():
    for p in params: p.grad = None
    model(*batch)[0].backward()
for _ in range(3): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host issue time {1e3*(t1-t0)/5:.2f} ms/step, wall {1e3*(t2-t0)/5:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
