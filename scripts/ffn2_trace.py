"""Phase time stamps of the streaming feed-forward kernel (TAVSR_FFN2_DBG=2): where a workgroup's time goes."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
os.environ["TAVSR_FFN2_DBG"] = "2"
from tavsr import ops  # noqa: E402
from tavsr._lib import lib  # noqa: E402

D, N1, M = 256, 2048, int(os.environ.get("FFN2_M", "3168"))
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
ln_w, ln_b, w1, b1, w2, b2 = 1 + 0.1 * r(D), 0.1 * r(D), r(N1, D) / 16, 0.1 * r(N1), r(D, N1) / 45, 0.1 * r(D)
x = r(M, D)
for cfg in sys.argv[1:] or ["10,4"]:
    os.environ["TAVSR_FFN2_CFG"] = cfg
    G = int(cfg.split(",")[0]) * ((M + 127) // 128)
    for _ in range(20):
        ops.ffn2_fwd(x, ln_w, ln_b, 1e-12, w1, b1, w2, b2, "swish", 0.5, save=False)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (1024 * 16))()
    lib().tavsr_ffn2_trace_read(buf, 1024 * 16)
    t = np.array(buf, dtype=np.int64).reshape(1024, 16)[:G]
    t0 = t[:, 0].min()
    names = ["start", "-", "prologue (x, LN, ring)", "units", "epilogue", "-", "drain"]
    print(f"cfg {cfg}: kernel span {(t[:, 6].max() - t0) / 100:.1f} us; per-workgroup phases in us (median / max), 100 MHz clock")
    print(f"  start skew       {np.median(t[:, 0] - t0) / 100:7.2f} / {(t[:, 0] - t0).max() / 100:7.2f}")
    d = (t[:, 1] - t[:, 0]) / 100.0
    print(f"    setup + issue of 36 LDS-DMA {np.median(d):7.2f} / {d.max():7.2f}")
    d = (t[:, 5] - t[:, 1]) / 100.0
    print(f"    wait for the rows           {np.median(d):7.2f} / {d.max():7.2f}")
    d = (t[:, 2] - t[:, 5]) / 100.0
    print(f"    LayerNorm, barriers, ring   {np.median(d):7.2f} / {d.max():7.2f}")
    for i, j in ((2, 0), (3, 2), (4, 3), (6, 4)):
        d = (t[:, i] - t[:, j]) / 100.0
        print(f"  {names[i]:24s} {np.median(d):7.2f} / {d.max():7.2f}")
    per_unit = (t[:, 3] - t[:, 2]) / 100.0 / np.maximum(t[:, 7], 1)
    print(f"  per unit (seg 0) {np.median(per_unit):7.2f} / {per_unit.max():7.2f}   units in seg 0: {t[:, 7].min()}..{t[:, 7].max()}")
    cyc = (t[:, 8 + 3] - t[:, 8 + 2]) / np.maximum(t[:, 7], 1)
    ghz = (t[:, 8 + 3] - t[:, 8 + 2]) / np.maximum(t[:, 3] - t[:, 2], 1) / 10.0
    print(f"  shader cycles per unit {np.median(cyc):9.0f} (16384 = MFMA-bound), in-loop clock {np.median(ghz):.2f} GHz (min {ghz.min():.2f}, max {ghz.max():.2f})")
    tot = (t[:, 6] - t[:, 0]) / 100.0
    print(f"  workgroup total  {np.median(tot):7.2f} / {tot.max():7.2f}")
