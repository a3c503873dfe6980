"""The stem's pooling kernels alone at the bench shape (32 clips x 100 frames, 44 x 44 x 64 after the Conv3d): BatchNorm + Swish +
MaxPool forward, its backward (reduce + apply), replayed from a captured graph.  TAVSR_LIB selects the library build."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
from tavsr import ops  # noqa: E402
from tavsr._lib import lib, check  # noqa: E402
import ctypes as C  # noqa: E402


def timed(fn, n=5, reps=4):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            g.replay()
        b.record()
        torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * reps) * 1e3


def main():
    N, H, W, Cc = 3200, 44, 44, 64
    torch.manual_seed(0)
    x = torch.randn(N, H, W, Cc, device="cuda")
    mean, rstd = torch.randn(Cc, device="cuda") * 0.1, torch.rand(Cc, device="cuda") + 0.5
    gamma, beta = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.1
    x2 = x.view(N * H * W, Cc)
    y, idx, Ho, Wo = ops.bn_act_maxpool3x3s2_fwd(x2, mean, rstd, gamma, beta, "swish", N, H, W, Cc)
    f = timed(lambda: ops.bn_act_maxpool3x3s2_fwd(x2, mean, rstd, gamma, beta, "swish", N, H, W, Cc))
    dpool = torch.randn_like(y)
    b = timed(lambda: ops.bn_bwd_pooled(dpool, idx, x2, mean, rstd, gamma, beta, N, H, W, act="swish"))
    # one of the stride-2 3x3 convolutions' patch copies (layer2.0: 3200 x 22 x 22 x 64 -> [N 11 11][9 x 64])
    xs = torch.randn(N * 22 * 22, 64, device="cuda")
    c = timed(lambda: ops.im2col2d(xs, N, 22, 22, 64, 3, 3, 2, 1))
    for a in ("relu", None):
        fa = timed(lambda: ops.bn_act_maxpool3x3s2_fwd(x2, mean, rstd, gamma, beta, a, N, H, W, Cc))
        ba = timed(lambda: ops.bn_bwd_pooled(dpool, idx, x2, mean, rstd, gamma, beta, N, H, W, act=a))
        print(f"  act={a}: forward {fa:.0f} us, backward {ba:.0f} us")
    pf = timed(lambda: ops.maxpool3x3s2_fwd(x2, N, H, W, Cc))
    print(f"  plain maxpool forward (no BatchNorm / activation) {pf:.0f} us")
    gb = x.numel() * 4 / 1e9
    print(f"lib={os.environ.get('TAVSR_LIB', 'in-tree')}: bn+swish+maxpool forward {f:.0f} us ({(gb * 1.3125) / f * 1e3:.2f} TB/s algorithmic), "
          f"backward (reduce + apply) {b:.0f} us, im2col 3x3/s2 of a 22x22x64 map {c:.0f} us")


if __name__ == "__main__":
    main()
