"""tree attention step of the search (tavsr_tree_attn_step) at the LM's and the decoder's shapes: us per call in a captured chain of 20
calls, by number of keys (the position in the utterance)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from rowlin_bench import chain_us
from tavsr import ops
from tavsr._lib import lib
N, K = 10, 10
gen = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=gen)
for name, H, dk in (("LM 8 x 64", 8, 64), ("decoder 4 x 64", 4, 64)):
    D = H * dk
    steps = 300
    kpool, vpool = r(steps * N, D), r(steps * N, D)
    qkv = r(N, 3 * D)
    for nkeys in (1, 10, 33, 50, 65, 100, 300):
        # a beam: hypotheses share all but their last few ancestors
        anc = (torch.arange(steps, device="cuda").view(1, steps) * N + torch.zeros(N, 1, device="cuda", dtype=torch.long)).to(torch.int32)
        anc[:, max(0, nkeys - 4):] += torch.arange(N, device="cuda", dtype=torch.int32).view(N, 1)
        anc = anc.contiguous()
        out = torch.empty(N, D, device="cuda")
        row = []
        for mode in (2, 1, 0):
            lib().tavsr_tree_attn_tune(mode)
            row.append(chain_us(lambda: ops.tree_attn_step(qkv[:, :D], kpool, vpool, anc, nkeys, H, dk, out=out, k_new=qkv[:, D:2 * D], v_new=qkv[:, 2 * D:], group=K)))
        lib().tavsr_tree_attn_tune(3)
        print(f"{name:16s} keys {nkeys:4d}: four items per workgroup {row[0]:6.2f} us   one {row[1]:6.2f} us   four waves per item {row[2]:6.2f} us", flush=True)
