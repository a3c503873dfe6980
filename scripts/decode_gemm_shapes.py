"""per-shape GEMM timing of the beam-search scorer step (eager launches, HIP events): which shapes bound the step"""
import argparse, copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tailored-avsr_amd")]
import torch
import bench_decode as B
from tavsr import ops
from tavsr.inference import beam_search as PBS
from tavsr.lm.transformer_lm import TransformerLM
from tavsr.tasks.avsr import AVSRTask
PBS.GRAPH_STEP = False
dev = torch.device("cuda", 0)
conf = B.make_conf()
torch.manual_seed(1)
model = AVSRTask.build_model(argparse.Namespace(**copy.deepcopy(conf))).eval().to(dev)
lm = TransformerLM(len(conf["token_list"]), **B.LM_CONF).eval().to(dev)
search = PBS.BatchBeamSearch(model, lm, **B.SEARCH)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
with torch.no_grad():
    enc, olens = model.encode(*B.make_utts(nb, 1234, dev))
    search.decode(enc[:8], olens[:8], nbest=1)
    torch.cuda.synchronize()
    prof = ops.GemmProfile(by_shape=True)
    ops.PROFILE = prof
    search.decode(enc, olens, nbest=1)
    ops.PROFILE = None
s = prof.summary()
tot = sum(v["seconds"] for v in s.values())
print(f"GEMM time {tot*1e3:.1f} ms over the search")
for k, v in sorted(s.items(), key=lambda kv: -kv[1]["seconds"])[:24]:
    print(f"{v['calls']:6d} {v['seconds']*1e3:8.2f} ms {v['seconds']/v['calls']*1e6:7.1f} us {v['flops']/v['seconds']/1e12:6.1f} TF/s  {k}")
