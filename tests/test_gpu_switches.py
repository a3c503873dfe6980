"""Every route selector that is left in the package, flipped once on the small training models.

Round 3 ended with ~20 ``TAVSR_*`` environment switches whose cross product nobody ran.  Round 4 removed the losers' code
(v1 feed-forward chain, streaming K = 256 projections, joint two-stream FFN, the one-workgroup-per-utterance merge
specialisation, the planner's tuning variables); what is left are module constants that pick between a fused launch and the
launches it replaces for shapes it does not take.  Each of them is flipped here, alone, and the training step (loss and every
gradient) must agree with the default route: the reference has ONE way to compute
``MyBranchformerEncoderLayer.forward`` (src/encoder/branchformer/encoder_layer.py:153-321) and ``ESPnetAVSRModel.forward``
(src/models/avsr_espnet_model.py:211-367), so every route is the same function.
"""
import argparse
import importlib

import pytest
import torch

from helpers import AVSR_YAML, asr_conf, avsr_conf, grad_ok

pytestmark = pytest.mark.gpu

ASR_SWITCHES = [("tavsr.ops", "FFN2", False), ("tavsr.ops", "FFN2_BWD", False), ("tavsr.ops", "MERGE_ROWS", False),
                ("tavsr.ops", "MERGE_PROJ", False), ("tavsr.ops", "MERGE_ROWDOT", False), ("tavsr.ops", "LN_BWD_DROP", False), ("tavsr.ops", "LAYER_C", False),
                ("tavsr.ops", "FFN2_BWD_LN", False), ("tavsr.ops", "WGRAD_BESIDE", False), ("tavsr.functional", "_POS_DW_BESIDE", False),
                ("tavsr.ops", "ATTN_FUSED", False), ("tavsr.ops", "CSGU_FUSED", False), ("tavsr.ops", "CSGU_STATS_IN_GEMM", False),
                ("tavsr.ops", "CGMLP_ACT_BWD_FUSED", False), ("tavsr.functional", "CONV2_IMPLICIT", False),
                ("tavsr.models.espnet_model", "LOSS_BRANCH", False), ("tavsr._lib", "SINGLE_STREAM", True)]
AV_SWITCHES = [("tavsr.functional_av", "FRONT_PAIR", False), ("tavsr.functional_av", "STEM_POOL_FUSED", False),
               ("tavsr.ops", "STEM_PAD16", False), ("tavsr.ops", "STEM_IMPLICIT", False), ("tavsr.ops", "FFN2", False),
               ("tavsr.ops", "MERGE_ROWS", False), ("tavsr.ops", "ATTN_FUSED", False), ("tavsr.ops", "CSGU_FUSED", False),
               ("tavsr.ops", "WGRAD_BESIDE", False), ("tavsr.functional_av", "_FRONT_WGRAD_BESIDE", False)]


def _model(workload, dropout):
    from oracle.model import synth
    if workload == "avsr":
        from tavsr.tasks.avsr import AVSRTask
        from test_gpu_av import _bench_batch
        conf = avsr_conf(AVSR_YAML, num_blocks=2, dec_blocks=1)
        model = AVSRTask.build_model(argparse.Namespace(**conf))
        batch = [t.cuda() for t in _bench_batch(2)]
    else:
        from tavsr.tasks.asr import ASRTask
        model = ASRTask.build_model(argparse.Namespace(**asr_conf(num_blocks=2, dec_blocks=1, dropout=dropout)))
        text = synth((4, 30), seed=2, kind="int", lo=1, hi=40)
        lens = torch.tensor([400, 372, 333, 251])
        batch = [synth((4, 400, 80), seed=1).cuda(), lens.cuda(), text.cuda(), torch.full((4,), 30).cuda()]
    torch.manual_seed(0)
    model = model.cuda().train()
    return model, batch, [p for p in model.parameters() if p.requires_grad]


def _step(model, batch, params):
    from tavsr import ops
    ops.manual_seed(4242)
    torch.manual_seed(3)
    for p in params:
        p.grad = None
    loss = model(*batch)[0]
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), [p.grad.detach().clone() for p in params]


def _flip_and_compare(workload, switches, dropout):
    model, batch, params = _model(workload, dropout)
    ref = _step(model, batch, params)
    names = [n for n, _ in model.named_parameters()]
    for mod, name, value in switches:
        m = importlib.import_module(mod)
        keep = getattr(m, name)
        assert keep != value, (mod, name)
        setattr(m, name, value)
        try:
            got = _step(model, batch, params)
        finally:
            setattr(m, name, keep)
        assert abs(got[0] - ref[0]) < 1e-4 * abs(ref[0]), (name, got[0], ref[0])
        bad = [n for n, a, b in zip(names, got[1], ref[1]) if not grad_ok(a, b, 2e-3)]
        assert not bad, (name, bad[:6])


def test_every_route_selector_of_the_audio_only_step():
    _flip_and_compare("asr", ASR_SWITCHES, 0.0)


def test_the_mask_preserving_routes_agree_under_dropout():
    """the fused launches draw the masks of the launches they replace (same Philox counters): also equal with dropout on"""
    _flip_and_compare("asr", [s for s in ASR_SWITCHES if s[1] in ("FFN2", "FFN2_BWD", "MERGE_PROJ", "MERGE_ROWDOT", "LN_BWD_DROP", "LAYER_C",
                                                                   "FFN2_BWD_LN", "WGRAD_BESIDE", "_POS_DW_BESIDE", "CSGU_FUSED", "LOSS_BRANCH", "SINGLE_STREAM")], 0.1)


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_launch_scheduling_switches_do_not_change_a_bit(dropout):
    """Round-5 switches that re-arrange launches without touching arithmetic: ``functional._POS_DW_BESIDE`` (the positional rows' gradient, ``linear_pos``'s
    weight gradient and the two positional bias sums with the layer's other weight gradients), ``FFN2_BWD_LN`` (the feed-forward block's LayerNorm backward
    sums dn's partials in the order of the finishing launch it replaces) and ``WGRAD_BESIDE`` (the layers' weight gradients on the side
    queue, joined at the end of the autograd pass) - under the Python sequencing and the C-side sequencer alike: every gradient bit-equal."""
    from tavsr import ops
    model, batch, params = _model("asr", dropout)
    names = [n for n, _ in model.named_parameters()]
    from tavsr import functional as F_
    keep = (ops.LAYER_C, ops.FFN2_BWD_LN, ops.WGRAD_BESIDE, F_._POS_DW_BESIDE)
    try:
        for layer_c in (False, True):
            ops.LAYER_C, ops.FFN2_BWD_LN, ops.WGRAD_BESIDE, F_._POS_DW_BESIDE = layer_c, True, True, True
            ref = _step(model, batch, params)
            for ln, beside, pos in ((False, True, True), (True, False, True), (False, False, True), (True, True, False)):
                ops.FFN2_BWD_LN, ops.WGRAD_BESIDE, F_._POS_DW_BESIDE = ln, beside, pos      # (pos: the positional chain with the weight gradients)
                got = _step(model, batch, params)
                assert got[0] == ref[0]
                bad = [n for n, a, b in zip(names, got[1], ref[1]) if not torch.equal(a, b)]
                assert not bad, (layer_c, ln, beside, pos, bad[:6])
    finally:
        ops.LAYER_C, ops.FFN2_BWD_LN, ops.WGRAD_BESIDE, F_._POS_DW_BESIDE = keep


def test_every_route_selector_of_the_audio_visual_step():
    _flip_and_compare("avsr", AV_SWITCHES, 0.0)


def test_weight_gradients_beside_the_chain_do_not_change_a_bit_of_the_audio_visual_step():
    """``WGRAD_BESIDE`` on the AV model: the lip front-end's 17 weight-gradient convolutions, Conv2dSubsampling's two and the decoder's first
    flush run on side queues, joined at the end of the autograd pass - same kernels, same plans, another queue: every gradient bit-equal."""
    from tavsr import ops
    model, batch, params = _model("avsr", 0.0)
    names = [n for n, _ in model.named_parameters()]
    keep = ops.WGRAD_BESIDE
    try:
        ops.WGRAD_BESIDE = True
        ref = _step(model, batch, params)
        ops.WGRAD_BESIDE = False
        got = _step(model, batch, params)
    finally:
        ops.WGRAD_BESIDE = keep
    assert got[0] == ref[0]
    bad = [n for n, a, b in zip(names, got[1], ref[1]) if not torch.equal(a, b)]
    assert not bad, bad[:6]
