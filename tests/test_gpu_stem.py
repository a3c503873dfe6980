"""GPU: the Conv3d stem as an implicit GEMM (tavsr_gemm_desc.conv_mode 4 / 5: every patch element is a 4-byte LDS-DMA
gather, no patch matrix) against torch's conv3d in fp64 - Conv3d(1, 64, (5,7,7), stride (1,2,2), padding (2,3,3), no bias)
of src/frontend/conv3d_resnet18/conv3d_resnet18.py:48-57 - forward and weight gradient, with clip borders in time and space
inside the tiles, and against the im2col route the library keeps as fallback."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,T,H,W", [(2, 3, 88, 88), (1, 4, 88, 88), (2, 5, 24, 40), (3, 2, 16, 16)])
def test_stem_implicit_gemm_matches_conv3d(B, T, H, W):
    from tavsr import ops
    g = torch.Generator(device="cuda").manual_seed(B * 100 + T)
    x = torch.randn(B, T, H, W, device="cuda", generator=g)
    w = torch.randn(64, 1, 5, 7, 7, device="cuda", generator=g) / 15
    assert ops.stem_implicit_ok(x), (B, T, H, W)
    w0 = torch.zeros(64, 256, device="cuda")
    w0[:, :245] = w.reshape(64, 245)
    z, Ho, Wo = ops.stem_conv_fwd(x, w0)
    ref = torch.nn.functional.conv3d(x.double().unsqueeze(1), w.double(), stride=(1, 2, 2), padding=(2, 3, 3))
    assert (Ho, Wo) == tuple(ref.shape[-2:])
    ref2 = ref.permute(0, 2, 3, 4, 1).reshape(-1, 64)              # rows (clip, t, ho, wo), channels last
    assert float((z.double() - ref2).abs().max() / ref2.abs().max()) < 2e-6
    # the fallback route computes the same product from the patch matrix
    col, _, _ = ops.im2col_stem(x)
    z2 = ops.linear(col, w0)
    assert float((z - z2).abs().max() / z2.abs().max()) < 2e-6
    # weight gradient
    dz = torch.randn(z.shape, device="cuda", generator=g)
    gw = ops.stem_conv_dw(dz, x)
    xr = x.double().unsqueeze(1)
    wr = w.double().requires_grad_(True)
    torch.nn.functional.conv3d(xr, wr, stride=(1, 2, 2), padding=(2, 3, 3)).backward(
        dz.double().view(B, T, Ho, Wo, 64).permute(0, 4, 1, 2, 3))
    gref = wr.grad.reshape(64, 245)
    assert float((gw[:, :245].double() - gref).abs().max() / gref.abs().max()) < 5e-6
    assert float(gw[:, 245:].abs().max()) == 0.0


@pytest.mark.parametrize("B,T,H,W", [(2, 3, 88, 88), (1, 4, 88, 88), (2, 5, 24, 40), (3, 2, 16, 16), (1, 2, 88, 88)])
def test_stem_on_padded_clips_matches_conv3d(B, T, H, W):
    """conv_mode 6 / 7: zero-padded clips, taps laid out 35 x 8 (K = N = 288), ordinary 16-byte LDS-DMA from 8-byte aligned
    sources - forward and weight gradient vs torch conv3d in fp64; output rows narrower than a K-step (Wo = 8, 20) walk over
    several rows / frames / clips per gather step."""
    from tavsr import ops
    g = torch.Generator(device="cuda").manual_seed(B * 10 + T)
    x = torch.randn(B, T, H, W, device="cuda", generator=g)
    w = torch.randn(64, 1, 5, 7, 7, device="cuda", generator=g) / 15
    assert ops.stem_pad16_ok(x)
    xp = ops.stem_pad(x)
    assert xp.shape == (B, T + 5, H + 6, W + 8)
    w288 = ops.stem_weight_288(w)
    assert w288.shape == (64, 288) and float(w288[:, 7::8].abs().max()) == 0.0 and float(w288[:, 280:].abs().max()) == 0.0
    z, Ho, Wo = ops.stem_conv_fwd_pad16(xp, w288, B, T, H, W)
    ref = torch.nn.functional.conv3d(x.double().unsqueeze(1), w.double(), stride=(1, 2, 2), padding=(2, 3, 3))
    ref2 = ref.permute(0, 2, 3, 4, 1).reshape(-1, 64)
    assert float((z.double() - ref2).abs().max() / ref2.abs().max()) < 2e-6
    dz = torch.randn(z.shape, device="cuda", generator=g)
    gw = ops.stem_weight_grad_from_288(ops.stem_conv_dw_pad16(dz, xp, T, H, W), w.shape)
    wr = w.double().requires_grad_(True)
    torch.nn.functional.conv3d(x.double().unsqueeze(1), wr, stride=(1, 2, 2), padding=(2, 3, 3)).backward(
        dz.double().view(B, T, Ho, Wo, 64).permute(0, 4, 1, 2, 3))
    assert float((gw.double() - wr.grad).abs().max() / wr.grad.abs().max()) < 5e-6


def test_stem_falls_back_for_shapes_the_gather_loader_does_not_take():
    from tavsr import ops
    x = torch.randn(1, 3, 88, 88, device="cuda")            # 3 * 1936 output pixels: not whole 32-row K steps
    assert not ops.stem_implicit_ok(x)
    with pytest.raises(RuntimeError):
        ops.stem_conv_dw(torch.randn(3 * 1936, 64, device="cuda"), x)
