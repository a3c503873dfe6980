"""GPU: one Branchformer layer forward AND backward as ONE C call each (tavsr_branchformer_layer_fwd / _bwd, csrc/layer.hip)
against the Python sequencing of the same launches (functional.BranchformerLayerFn; autograd of
src/encoder/branchformer/encoder_layer.py:153-321) - outputs and every gradient BIT-equal, with the recipe's dropout on (same
tokens, same masks) and in eval, full and ragged lengths."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, tol):
    a, b = a.double(), b.double()
    err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert err < tol, err


def _run(layer_c, train, B, T, lens):
    from tavsr import ops
    from tavsr.encoder.branchformer.encoder import MyBranchformerEncoder
    from tavsr.layers import RelPositionalEncoding
    keep = ops.LAYER_C
    ops.LAYER_C = layer_c
    try:
        torch.manual_seed(0)
        D = 256
        p = 0.1 if train else 0.0
        enc = MyBranchformerEncoder(input_size=D, num_blocks=2, input_layer=None, dropout_rate=p, positional_dropout_rate=0.0,
                                    attention_dropout_rate=p, ffn_activation_type="swish", merge_method="learned_ave").cuda()
        enc.train(train)
        g = torch.Generator(device="cuda").manual_seed(5)
        x = torch.randn(B, T, D, device="cuda", generator=g)
        xs, pos = RelPositionalEncoding(D, 0.0)(x)
        xs = xs.detach().requires_grad_(train)
        mask = (torch.arange(T, device="cuda")[None, :] < lens[:, None])[:, None, :]
        ops.manual_seed(123)
        h = (xs, pos)
        with torch.set_grad_enabled(train):
            for layer in enc.encoders:
                h, mask = layer(h, mask)
        y = h[0]
        grads = []
        if train:
            (y * torch.randn(B, T, D, device="cuda", generator=g)).sum().backward()
            grads = [xs.grad] + [q.grad for _, q in sorted(enc.named_parameters()) if q.grad is not None]
        return y.detach(), grads
    finally:
        ops.LAYER_C = keep


@pytest.mark.parametrize("B,T", [(4, 99), (3, 40), (2, 150), (1, 5)])
@pytest.mark.parametrize("train", [False, True])
def test_layer_forward_in_c_equals_python_sequencing(B, T, train):
    lens = torch.tensor([T, max(1, (2 * T) // 3), max(1, T // 2), T][:B], device="cuda")
    y_c, g_c = _run(True, train, B, T, lens)
    y_p, g_p = _run(False, train, B, T, lens)
    assert torch.equal(y_c, y_p)                     # the same launches with the same arguments
    assert len(g_c) == len(g_p) and len(g_c) == (0 if not train else len(g_c))
    for a, b in zip(g_c, g_p):
        assert torch.equal(a, b), float((a - b).abs().max())      # same launches, same grouping, same order


def test_layer_in_c_is_used_and_refuses_foreign_shapes():
    """the fast path takes the recipe form; a descriptor outside the kernels' range is refused with TAVSR_EUNSUPPORTED
    (the Python sequencing then runs: e.g. d_model 512)"""
    import ctypes as C
    from tavsr._lib import BfLayerDesc, lib
    d = BfLayerDesc()
    d.B, d.T, d.D, d.H, d.ffn_units, d.cg_units, d.cg_kernel = 2, 50, 512, 8, 2048, 2048, 31
    fn = lib().tavsr_branchformer_layer_ws
    fn.restype = C.c_int64
    assert fn(C.byref(d)) == 0
    assert lib().tavsr_branchformer_layer_fwd(C.byref(d), None) != 0
    d.D, d.H = 256, 4
    assert fn(C.byref(d)) > 0
    # the backward entry wants the state a save = 1 forward call kept
    from tavsr._lib import BfLayerBwdDesc
    b = BfLayerBwdDesc()
    b.fwd = C.pointer(d)
    assert lib().tavsr_branchformer_layer_bwd(C.byref(b), None) != 0
    assert b"save" in lib().tavsr_last_error_string()


@pytest.mark.parametrize("ua,uv", [([True], [False]), ([False], [True]), ([True], [True]), ([False], [False])])
@pytest.mark.parametrize("train", [False, True])
def test_tailored_layer_forward_in_c_equals_python_sequencing(ua, uv, train):
    """tavsr_tailored_layer_fwd (both modality streams of a TailoredEncoderLayer, the video stream on the second queue;
    src/encoder/audiovisual/tailored/encoder_layer.py:118-274) against functional_av.TailoredLayerFn's Python sequencing: outputs
    and - through the shared Python backward, which runs on the state the C call kept - every gradient bit-equal."""
    from tavsr import ops
    from tavsr.encoder.audiovisual.tailored.encoder import TailoredEncoder
    from tavsr.layers import RelPositionalEncoding

    def run(layer_c):
        keep = ops.LAYER_C
        ops.LAYER_C = layer_c
        try:
            torch.manual_seed(0)
            B, T, D = 3, 50, 256
            p = 0.1 if train else 0.0
            enc = TailoredEncoder("rel_pos", "latest", num_blocks=1, dropout_rate=p, positional_dropout_rate=0.0, attention_dropout_rate=p,
                                  acoustic_use_attn=ua, visual_use_attn=uv).cuda()
            enc.train(train)
            layer = enc.encoders[0]
            g = torch.Generator(device="cuda").manual_seed(5)
            a, v = torch.randn(B, T, D, device="cuda", generator=g), torch.randn(B, T, D, device="cuda", generator=g)
            pe = RelPositionalEncoding(D, 0.0)
            xa, pos = pe(a)
            xv, _ = pe(v)
            xa, xv = xa.detach().requires_grad_(train), xv.detach().requires_grad_(train)
            alens, vlens = torch.tensor([T, 37, 20], device="cuda"), torch.tensor([T, T, 31], device="cuda")
            ar = torch.arange(T, device="cuda")[None, :]
            am, vm = (ar < alens[:, None])[:, None, :], (ar < vlens[:, None])[:, None, :]
            ops.manual_seed(321)
            ops.rng_step_begin(a.device)
            with torch.set_grad_enabled(train):
                (ya, _), _, (yv, _), _ = layer((xa, pos), am, (xv, pos), vm)
            grads = []
            if train:
                ((ya * torch.randn(B, T, D, device="cuda", generator=g)).sum() + (yv * torch.randn(B, T, D, device="cuda", generator=g)).sum()).backward()
                grads = [xa.grad, xv.grad] + [q.grad for _, q in sorted(layer.named_parameters()) if q.grad is not None]
            return ya.detach(), yv.detach(), grads
        finally:
            ops.LAYER_C = keep

    ya_c, yv_c, g_c = run(True)
    ya_p, yv_p, g_p = run(False)
    assert torch.equal(ya_c, ya_p) and torch.equal(yv_c, yv_p)
    assert len(g_c) == len(g_p)
    for x, y in zip(g_c, g_p):
        assert torch.equal(x, y), float((x - y).abs().max())
