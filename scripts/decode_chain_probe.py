"""How long are the two scorer chains of a batch-1 search step, and how well do they overlap?  The same search with the
decoder's launches, the LM's launches, or neither removed from the captured step (the removed scorer answers with a constant
tensor recorded beforehand: no launches), and with both on one queue.  Device time per token = decode wall time / steps."""
import argparse, copy, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
import bench_decode as BD
from tavsr.inference import beam_search as B
from tavsr.lm.transformer_lm import TransformerLM
from tavsr.tasks.avsr import AVSRTask

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--short", action="store_true", help="only: both / LM only / decoder only / neither / one queue")
args = ap.parse_args()
dev = torch.device("cuda:0")
conf = BD.make_conf()
torch.manual_seed(1)
model = AVSRTask.build_model(argparse.Namespace(**copy.deepcopy(conf))).eval().to(dev)
lm = TransformerLM(len(conf["token_list"]), **BD.LM_CONF).eval().to(dev)
search = B.BatchBeamSearch(model, lm, **BD.SEARCH)
batch = BD.make_utts(args.batch, 1234, dev)
with torch.no_grad():
    enc, olens = model.encode(*batch)
steps_seen = []
orig_replay = torch.cuda.CUDAGraph.replay
def counting_replay(self):
    steps_seen[-1] += 1
    return orig_replay(self)
torch.cuda.CUDAGraph.replay = counting_replay

def timed(tag):
    search._captured = None          # (a captured step is re-used while the shape repeats: every variant captures its own)
    for _ in range(2):
        steps_seen.append(0)
        search.decode(enc, olens, nbest=1)
    ts = []
    for _ in range(3):
        steps_seen.append(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            search.decode(enc, olens, nbest=1)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / max(1, steps_seen[-1]))
    print(f"{tag:46s} {1e6 * min(ts):8.1f} us per token ({steps_seen[-1]} tokens)", flush=True)

timed("both scorers, two queues")
B.RECORD_QUEUE = not B.RECORD_QUEUE
search._copy_q = None
timed(f"both scorers, records on a queue of their own = {B.RECORD_QUEUE}")
B.RECORD_QUEUE = not B.RECORD_QUEUE
search._copy_q = None
timed("both scorers, two queues (again)")
if not args.short:
  B.CTC_BESIDE_SCORERS = not B.CTC_BESIDE_SCORERS
  timed(f"both scorers, CTC prefix scores beside the scorers = {B.CTC_BESIDE_SCORERS}")
  B.CTC_BESIDE_SCORERS = not B.CTC_BESIDE_SCORERS
  timed("both scorers, two queues (again)")
  from tavsr._lib import lib as _lib
  for mode, what in ((0, "four waves per item"), (1, "one wave per item"), (3, "the plan")):
    _lib().tavsr_tree_attn_tune(mode)
    timed(f"both scorers, tree attention: {what}")
  from tavsr import ops as _ops
  for ff in (4,):
    _ops.ROWLIN_KSPLIT = ff
    timed(f"both scorers, feed-forward closing projection in {ff} K slice(s)")
dec_step, lm_step = search.dec_step.step, search.lm_step.step
N, V = search.K * args.batch, search.V
const = torch.full((N, V), -3.7, device=dev)
def fake_dec(i, tok, anc, dyn=None, **score):
    out = torch.empty(N, V, device=dev)
    out.copy_(const)
    return out
def fake_lm(i, tok, anc, dyn=None, logits_only=False, **score):
    return const
search.dec_step.step = fake_dec
timed("LM chain only (decoder = one copy launch)")
search.dec_step.step = dec_step
search.lm_step.step = fake_lm
timed("decoder chain only (LM = constant)")
search.dec_step.step = fake_dec
timed("neither (beam update + CTC prefix only)")
search.dec_step.step, search.lm_step.step = dec_step, lm_step
B.SCORERS_PARALLEL = False
timed("both scorers, one queue")
