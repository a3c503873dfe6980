"""host-side profile of one batched search (where the per-step wall time goes): cProfile over BatchBeamSearch.decode"""
import argparse, copy, cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tailored-avsr_amd")]
import torch
import bench_decode as B
from tavsr.inference.beam_search import BatchBeamSearch
from tavsr.lm.transformer_lm import TransformerLM
from tavsr.tasks.avsr import AVSRTask
dev = torch.device("cuda", 0)
conf = B.make_conf()
torch.manual_seed(1)
model = AVSRTask.build_model(argparse.Namespace(**copy.deepcopy(conf))).eval().to(dev)
lm = TransformerLM(len(conf["token_list"]), **B.LM_CONF).eval().to(dev)
search = BatchBeamSearch(model, lm, **B.SEARCH)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
with torch.no_grad():
    enc, olens = model.encode(*B.make_utts(nb, 1234, dev))
    search.decode(enc[:8], olens[:8], nbest=1)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    search.decode(enc, olens, nbest=1)
    torch.cuda.synchronize()
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
