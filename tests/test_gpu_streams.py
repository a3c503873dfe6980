"""Results must not depend on how a step's launches are spread over HIP streams.

The reference runs its whole step on ONE queue (src/models/espnet_model.py:258-356, avsr_espnet_model.py:253-367 on torch's
current stream).  The package forks independent sections onto side streams (``ops.BranchScope``, the C-side layer sequencer,
autograd replaying the CTC branch on its forward stream); every such fork must be invisible in the results:

* ``TAVSR_SINGLE_STREAM`` (``_lib.SINGLE_STREAM``): the same step with every fork disabled is BIT-equal to the forked one;
* the race amplifier (``ops.arm_race_probe``): a spin kernel at the head of every forked section ("body": the side stream
  falls behind - a main-pool block freed too early would be overwritten under its readers) or right behind every join
  ("join": the owner falls behind - a side-pool block handed out again too early would be overwritten under the owner's readers)
  leaves loss and gradients bit-equal, in eager launches and inside a captured + replayed hipGraph.
"""
import argparse

import pytest
import torch

from helpers import AVSR_YAML, TOKENS_EN, asr_conf, avsr_conf

pytestmark = pytest.mark.gpu


def _setup(workload):
    from oracle.model import synth
    if workload == "avsr":
        from tavsr.tasks.avsr import AVSRTask
        from test_gpu_av import _bench_batch
        conf = avsr_conf(AVSR_YAML, num_blocks=2, dec_blocks=2)
        for k in ("dropout_rate", "positional_dropout_rate", "attention_dropout_rate"):
            conf["encoder_conf"][k] = 0.1
        model = AVSRTask.build_model(argparse.Namespace(**conf))
        batch = [t.cuda() for t in _bench_batch(4)]
    else:
        from tavsr.tasks.asr import ASRTask
        model = ASRTask.build_model(argparse.Namespace(**asr_conf(num_blocks=3, dec_blocks=2, dropout=0.1)))
        text = synth((6, 30), seed=2, kind="int", lo=1, hi=40)
        lens = torch.tensor([400, 400, 372, 333, 400, 251])
        batch = [synth((6, 400, 80), seed=1).cuda(), lens.cuda(), text.cuda(), torch.full((6,), 30).cuda()]
    torch.manual_seed(0)
    model = model.cuda().train()
    return model, batch, [p for p in model.parameters() if p.requires_grad]


def _step(model, batch, params):
    """one training step from a fixed generator state: (loss, gradients)"""
    from tavsr import ops
    ops.manual_seed(20261004)          # device generator of the dropout masks
    torch.manual_seed(7)               # host coins (stochastic depth)
    for p in params:
        p.grad = None
    loss = model(*batch)[0]
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach().clone(), [p.grad.detach().clone() for p in params]


def _assert_same(model, ref, got, what):
    assert torch.equal(ref[0], got[0]), (what, float(ref[0]), float(got[0]))
    bad = [n for (n, _), a, b in zip(model.named_parameters(), ref[1], got[1]) if not torch.equal(a, b)]
    assert not bad, (what, len(bad), bad[:8])


@pytest.fixture
def _streams_restored():
    from tavsr import _lib, ops
    yield
    _lib.SINGLE_STREAM = False
    ops.arm_race_probe(0.0, "alt")


@pytest.mark.parametrize("workload", ["asr", "avsr"])
def test_single_stream_step_is_bit_equal_to_the_forked_step(workload, _streams_restored):
    from tavsr import _lib
    model, batch, params = _setup(workload)
    _lib.SINGLE_STREAM = True
    one = _step(model, batch, params)
    assert torch.isfinite(one[0])
    _lib.SINGLE_STREAM = False
    forked = _step(model, batch, params)
    _assert_same(model, one, forked, "forked vs single stream")
    assert _lib._FORKED, "the forked run did not fork"


@pytest.mark.parametrize("mode", ["body", "join", "alt"])
@pytest.mark.parametrize("workload", ["asr", "avsr"])
def test_race_probe_leaves_the_eager_step_bit_equal(workload, mode, _streams_restored):
    from tavsr import _lib, ops
    model, batch, params = _setup(workload)
    _lib.SINGLE_STREAM = True
    ref = _step(model, batch, params)
    _lib.SINGLE_STREAM = False
    ops.arm_race_probe(300.0, mode)
    for rep in range(2):      # (second pass: the allocator's pools are warm - blocks are re-used, which is what a race needs)
        _assert_same(model, ref, _step(model, batch, params), f"eager, probe {mode}, pass {rep}")


@pytest.mark.parametrize("mode", ["body", "join"])
@pytest.mark.parametrize("workload", ["asr", "avsr"])
def test_race_probe_leaves_the_captured_step_bit_equal(workload, mode, _streams_restored):
    from tavsr import _lib, ops
    model, batch, params = _setup(workload)
    _lib.SINGLE_STREAM = True
    ref = _step(model, batch, params)
    _lib.SINGLE_STREAM = False
    ops.arm_race_probe(200.0, mode)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):           # warm-up on a side stream, as torch.cuda.graph wants it
        _step(model, batch, params)
    torch.cuda.current_stream().wait_stream(side)
    for p in params:
        p.grad = None
    ops.manual_seed(20261004)
    torch.manual_seed(7)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = model(*batch)[0]
        loss.backward()
    for rep in range(2):
        ops.manual_seed(20261004)          # the generator state is device memory: every replay starts from the same one
        g.replay()
        torch.cuda.synchronize()
        _assert_same(model, ref, (loss.detach(), [p.grad for p in params]), f"captured, probe {mode}, replay {rep}")


@pytest.mark.parametrize("mode", ["body", "join"])
def test_race_probe_leaves_the_beam_search_unchanged(mode, _streams_restored):
    """the decoder and the LM score side by side on two streams (inference/beam_search.py): same hypotheses and scores
    with either of them delayed"""
    from oracle.model import fill_parameters_, synth
    from tavsr import ops
    from tavsr.inference.beam_search import BatchBeamSearch
    from tavsr.lm.transformer_lm import TransformerLM
    from tavsr.tasks.asr import ASRTask
    from test_beam_search import LM_KW
    conf = asr_conf(num_blocks=2, dec_blocks=2)
    conf["token_list"] = TOKENS_EN
    pm = ASRTask.build_model(argparse.Namespace(**conf)).eval()
    fill_parameters_(pm, seed=5)
    plm = TransformerLM(len(TOKENS_EN), **LM_KW).eval()
    fill_parameters_(plm, seed=6)
    pm, plm = pm.cuda(), plm.cuda()
    x = synth((3, 160, 80), seed=7).cuda()
    lens = torch.tensor([160, 120, 88]).cuda()
    with torch.no_grad():
        enc, olens = pm.encode(x, lens)
        ref = BatchBeamSearch(pm, plm, 10, 0.1, 0.6, 0.5).decode(enc, olens)
        ops.arm_race_probe(100.0, mode)
        got = BatchBeamSearch(pm, plm, 10, 0.1, 0.6, 0.5).decode(enc, olens)
    for u in range(3):
        assert [h[0] for h in ref[u]] == [h[0] for h in got[u]], u
        assert [h[1] for h in ref[u]] == [h[1] for h in got[u]], u


def test_gradient_hooks_on_a_forked_stream_are_ordered_before_the_bucket_pack(_streams_restored):
    """the CTC branch's parameters accumulate on the forked stream and their hooks fire there; with
    accum_grad > 1 and small buckets a hook on either stream may complete a bucket - the pack must follow both streams.
    One rank cannot exchange anything, so the bucket machinery is driven by hand: what is checked is the packed flat buffer."""
    from tavsr import dp, ops
    model, batch, params = _setup("asr")
    ref1 = _step(model, batch, params)
    gb = dp.GradBuckets(params, bucket_bytes=1 << 16)
    packed = {}

    def fake_issue(i, flat):
        packed[i] = flat
        gb._works[i] = "rccl"
    gb._issue_flat = fake_issue
    # hooks as attach_overlap_hooks installs them (world 1 would return early)
    index = {id(p): i for i, b in enumerate(gb.buckets) for p in b}

    def hook(p):
        if not gb._armed:
            return
        st = torch.cuda.current_stream()
        gb._hook_streams.setdefault(st.cuda_stream, st)
        i = index[id(p)]
        gb._pending[i] -= 1
        while gb._next < len(gb.buckets) and gb._pending[gb._next] == 0:
            gb._launch_bucket(gb._next)
            gb._next += 1
    handles = [p.register_post_accumulate_grad_hook(hook) for p in params]
    try:
        ops.arm_race_probe(300.0, "body")
        ops.manual_seed(20261004)
        torch.manual_seed(7)
        for p in params:
            p.grad = None
        model(*batch)[0].backward()              # micro-batch 1 (no window: hooks idle)
        gb._pending = [len(b) for b in gb.buckets]
        gb._next, gb._hook_streams, gb._armed = 0, {}, True
        ops.manual_seed(20261004)
        torch.manual_seed(7)
        model(*batch)[0].backward()              # micro-batch 2, inside the window: gradients accumulate, hooks pack
        gb._armed = False
        torch.cuda.synchronize()
    finally:
        for h in handles:
            h.remove()
    assert len(packed) == len(gb.buckets)
    assert len(gb._hook_streams) >= 2, "no hook fired on a forked stream: the test does not exercise the ordering"
    for i, bucket in enumerate(gb.buckets):
        off = 0
        for p in bucket:
            want = 2.0 * ref1[1][[id(q) for q in params].index(id(p))]
            got = packed[i][off: off + p.numel()].view_as(p)
            assert torch.equal(got, want), (i, p.shape)
            off += (p.numel() + 3) // 4 * 4


def test_the_amplifier_catches_the_races_the_rules_prevent(_streams_restored):
    """Detection power of the instrument: with both allocator rules switched off (``_lib.RULES_OFF``: no record_stream at the
    pointer boundary, no wait on the owner at the head of a backward node) the delayed AV step does NOT reproduce the
    single-stream gradients - blocks of the main pool are re-used under their forked readers; with the rules on it does
    (measured: 112-135 mismatching gradient tensors over three passes against 0)."""
    from tavsr import _lib, ops
    model, batch, params = _setup("avsr")
    _lib.SINGLE_STREAM = True
    ref = _step(model, batch, params)
    _lib.SINGLE_STREAM = False
    try:
        counts = {}
        for off in (True, False):
            _lib.RULES_OFF = off
            bad = 0
            for mode in ("body", "join"):
                ops.arm_race_probe(300.0, mode)
                for _ in range(3):
                    got = _step(model, batch, params)
                    bad += sum(1 for a, b in zip(ref[1], got[1]) if not torch.equal(a, b)) + (0 if torch.equal(ref[0], got[0]) else 1)
            counts[off] = bad
    finally:
        _lib.RULES_OFF = False
    assert counts[False] == 0, counts
    if counts[True] == 0:      # (a race is a matter of timing: not provoking one on some box says nothing about the product)
        pytest.skip("the amplifier provoked no race with the rules off on this box")


def test_the_side_of_a_fork_is_never_a_stream_of_torchs_pool():
    """torch hands ``torch.cuda.Stream()`` out of a round-robin pool of 32 per device: a side stream taken from it re-appears, 32 streams
    later, as somebody's main or capturing stream - and the fork registry (``_lib._FORKED``, keyed by raw handles) then made a hipGraph
    capture wait for a stream outside it (a segfault in the first replay, in whatever test happened to draw the alias).  Side streams are
    the package's own (tavsr_stream_create): none of them is ever handed out by the pool."""
    from tavsr import _lib, ops
    pool = {torch.cuda.Stream().cuda_stream for _ in range(70)}      # both priorities' pools wrap around well before 70
    assert len(pool) <= 64
    mains = [torch.cuda.Stream() for _ in range(3)] + [torch.cuda.current_stream()]
    sides = [ops.branch_stream(m, slot) for m in mains for slot in (0, 1)]
    assert len({s.cuda_stream for s in sides}) >= 7                 # (the three pool streams may alias each other; their sides do not)
    assert not {s.cuda_stream for s in sides} & pool
    assert not set(_lib._FORKED) & pool
