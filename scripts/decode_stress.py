"""Is the captured search step reproducible?  The same 4 s utterance decoded REPS times by one search object (captured step re-used):
every decode must return the first one's hypotheses and scores bit for bit.  Prints one line; exit code 1 on a mismatch."""
import argparse, copy, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
import bench_decode as BD
from tavsr.inference import beam_search as B
from tavsr.lm.transformer_lm import TransformerLM
from tavsr.tasks.avsr import AVSRTask
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for flag in sys.argv[2:]:              # e.g. CTC_BESIDE_SCORERS=0 RECORD_QUEUE=0: module constants of the search, flipped for this run
    k, v = flag.split("=")
    setattr(B, k, v not in ("0", "False"))
dev = torch.device("cuda:0")
conf = BD.make_conf()
torch.manual_seed(1)
model = AVSRTask.build_model(argparse.Namespace(**copy.deepcopy(conf))).eval().to(dev)
lm = TransformerLM(len(conf["token_list"]), **BD.LM_CONF).eval().to(dev)
search = B.BatchBeamSearch(model, lm, **BD.SEARCH)
encode = B.CapturedEncode(model)
bad = 0
with torch.no_grad():
    for seed in (1234, 1235):
        batch = BD.make_utts(1, seed, dev)
        first = None
        for r in range(reps):
            enc, olens = encode(*batch)
            hyps = search.decode(enc, olens, nbest=3)
            if first is None:
                first = hyps
            elif hyps != first:
                bad += 1
                print(f"seed {seed} rep {r}: differs from rep 0: {len(hyps[0][0][0])} vs {len(first[0][0][0])} tokens, scores {hyps[0][0][1]} vs {first[0][0][1]}", flush=True)
print(f"decode stress: {2 * reps} decodes, {bad} differ (CTC beside = {B.CTC_BESIDE_SCORERS}, record queue = {B.RECORD_QUEUE})", flush=True)
sys.exit(1 if bad else 0)
