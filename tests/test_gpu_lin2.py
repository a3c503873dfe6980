"""GPU: linear layers of a d_model = 256 input as one streaming launch (csrc/lin2.hip, tavsr_lin2_fwd through the C ABI) against
fp64 torch of espnet's projections (linear_q / linear_k / linear_v of one attention input written into one [M, 768] buffer;
cgMLP channel_proj1 + GELU with the pre-activations kept): every plan of the unit split, row counts that leave whole waves
without a valid row, column windows of wider outputs."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, b, tol):
    a, b = a.double(), b.double()
    assert a.shape == b.shape
    err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert err < tol, err


@pytest.fixture(autouse=True)
def _default_plan():
    os.environ.pop("TAVSR_LIN2_WPB", None)
    yield
    os.environ.pop("TAVSR_LIN2_WPB", None)


@pytest.mark.parametrize("M,wpb", [(3168, None), (3168, "10"), (3168, "3"), (6400, None), (100, None), (64, None), (31, "1"), (1312, None)])
def test_lin2_qkv_projections(M, wpb):
    from tavsr import ops
    if wpb:
        os.environ["TAVSR_LIN2_WPB"] = wpb
    D = 256
    g = torch.Generator(device="cuda").manual_seed(M)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x = r(M, D)
    ws = [r(D, D) / 16 for _ in range(3)]
    bs = [0.1 * r(D), None, 0.1 * r(D)]
    out = torch.full((M, 3 * D + 64), 7.0, device="cuda")           # wider than the three windows: the rest stays untouched
    ops.lin2_fwd(x, [(w, b, out, j * D, None) for j, (w, b) in enumerate(zip(ws, bs))])
    for j, (w, b) in enumerate(zip(ws, bs)):
        ref = x.double() @ w.double().t() + (0 if b is None else b.double())
        _close(out[:, j * D:(j + 1) * D], ref, 3e-6)
    assert bool((out[:, 3 * D:] == 7.0).all())
    out2 = torch.empty(M, 3 * D + 64, device="cuda")
    ops.lin2_fwd(x, [(w, b, out2, j * D, None) for j, (w, b) in enumerate(zip(ws, bs))])
    assert torch.equal(out2[:, :3 * D], out[:, :3 * D])                # run-to-run reproducible


@pytest.mark.parametrize("M,N,act,wpb", [(3168, 2048, "gelu", None), (3168, 2048, "gelu", "7"), (200, 2048, "swish", None),
                                         (515, 1056, "relu", "2"), (999, 32, None, None)])
def test_lin2_activation_and_preactivations(M, N, act, wpb):
    from tavsr import ops
    if wpb:
        os.environ["TAVSR_LIN2_WPB"] = wpb
    D = 256
    g = torch.Generator(device="cuda").manual_seed(N + M)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    big = r(M, 3 * D)
    x = big[:, D:2 * D]                                              # strided input rows
    w, b = r(N, D) / 16, 0.1 * r(N)
    out, z = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ops.lin2_fwd(x, [(w, b, out, 0, z)], act=act)
    zr = x.double() @ w.double().t() + b.double()
    fn = {"gelu": F.gelu, "relu": torch.relu, "swish": lambda t: t * torch.sigmoid(t), None: lambda t: t}[act]
    _close(z, zr, 3e-6)
    _close(out, fn(zr), 3e-6)
    out2 = torch.empty(M, N, device="cuda")
    ops.lin2_fwd(x, [(w, b, out2, 0, None)], act=act)                # without the pre-activations: same values
    _close(out2, out, 1e-6)


def test_linear_wrappers_route_to_the_streaming_launch_when_enabled_and_agree_with_the_gemm():
    from tavsr import ops
    M, D = 777, 256
    g = torch.Generator(device="cuda").manual_seed(1)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x, w, b = r(M, D), r(2048, D) / 16, 0.1 * r(2048)
    ws = [(r(D, D) / 16, 0.1 * r(D), j * D) for j in range(3)]
    res = []
    keep = ops.LIN2
    try:
        for on in (True, False):
            ops.LIN2 = on
            y, z = ops.linear(x, w, b, act="gelu", save_z=True)
            qkv = ops.linear_group(x, ws, ops.empty(M, 3 * D, like=x))
            res.append((y, z, qkv))
    finally:
        ops.LIN2 = keep
    for a, c in zip(*res):
        _close(a, c, 3e-6)
