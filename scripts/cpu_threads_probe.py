"""how many host threads does the CPU oracle want?  One fwd+bwd step of the audio-only (batch 4) and AV (batch 4) oracle models by thread count."""
import copy, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
import bench as BN
from oracle.av import build_avsr_oracle
from oracle.model import build_asr_oracle
from tavsr.utils.tokens import CHAR_ENGLISH
for wl, build in (("asr", build_asr_oracle), ("avsr", build_avsr_oracle)):
    BN.WORKLOAD = wl
    torch.manual_seed(0)
    model = build(copy.deepcopy(BN.make_conf()), CHAR_ENGLISH).train()
    batch = BN.make_batch(4, 1234, "cpu")
    def step():
        for p in model.parameters():
            p.grad = None
        model(*batch)[0].backward()
    for th in (128, 64, 32, 16):
        torch.set_num_threads(th)
        step()
        ts = []
        for _ in range(2):
            t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
        print(f"{wl} batch 4, {th:3d} threads: {4 / min(ts):.3f} utt/s ({min(ts):.2f} s per step)", flush=True)
