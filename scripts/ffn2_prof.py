"""Driver for rocprofv3 runs over the feed-forward block: a few eager launches of the GEMM path and of tavsr_ffn2_fwd
(plans from argv: "G,NS" ...), M = 3168, hidden 2048, eval form."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
from tavsr import functional as F_  # noqa: E402
from tavsr import ops  # noqa: E402

D, N1, M = 256, 2048, int(os.environ.get("FFN2_M", "3168"))
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
ln_w, ln_b, w1, b1, w2, b2 = 1 + 0.1 * r(D), 0.1 * r(D), r(N1, D) / 16, 0.1 * r(N1), r(D, N1) / 45, 0.1 * r(D)
x = r(M, D)
for it in range(12):
    ops.FFN2 = False
    F_._FFN.fwd(x, ln_w, ln_b, w1, b1, w2, b2, "swish", 0.5, save=False)
    ops.FFN2 = True
    for cfg in sys.argv[1:]:
        os.environ["TAVSR_FFN2_CFG"] = cfg
        ops.ffn2_fwd(x, ln_w, ln_b, 1e-12, w1, b1, w2, b2, "swish", 0.5, save=False)
torch.cuda.synchronize()
