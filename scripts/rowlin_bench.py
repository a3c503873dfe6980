"""one-token Linear launches of the search step (tavsr_rowlin) at their shapes, 10 hypothesis rows: us per call in a captured
chain of 20 dependent calls."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
from tavsr import ops
from tavsr._lib import lib

def chain_us(fn, n=20, reps=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n)

N = 10
gen = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=gen)
shapes = [("LM q/k/v (LN)", 512, 1536, True, None), ("LM out (+res)", 512, 512, False, None), ("LM FFN1 (LN, relu)", 512, 2048, True, "relu"),
          ("LM FFN2 (+res)", 2048, 512, False, None), ("dec q/k/v (LN)", 256, 768, True, None), ("dec out (+res)", 256, 256, False, None),
          ("dec FFN1 (LN, relu)", 256, 2048, True, "relu"), ("dec FFN2 (+res)", 2048, 256, False, None)]
for rnd in range(2):
    for name, K, Nout, ln, act in shapes:
        x, w, b = r(N, K), r(Nout, K) / K ** 0.5, r(Nout)
        gam, bet = 1 + 0.1 * r(K), 0.1 * r(K)
        res = None if ln else r(N, Nout)
        out = torch.empty(N, Nout, device="cuda")
        row = [chain_us(lambda: ops.rowlin(x, w, b, ln=(gam, bet, 1e-12) if ln else None, act=act, res=res, out=out))] * 2
        if K == 2048:
            for ks in (2, 4, 8):
                row.append(chain_us(lambda: ops.rowlin(x, w, b, res=res, ksplit=ks)))
            print(f"round {rnd} {name:22s} K dealt to 2 / 4 / 8 blocks (partial tensors out): {row[2]:6.2f} / {row[3]:6.2f} / {row[4]:6.2f} us", flush=True)
        print(f"round {rnd} {name:22s} K {K:5d} -> {Nout:5d}: {row[0]:6.2f} us", flush=True)
