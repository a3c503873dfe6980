"""CPU, world_size 2, gloo: the data-parallel gradient exchange (tavsr.dp) - bucket plan, parameter
broadcast, summed-then-averaged gradients identical on every rank."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT


def _worker(rank, world, port, ret):
    sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
    from tavsr import dp
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, _, w = dp.init_from_env("gloo")
    torch.manual_seed(100 + rank)  # different init per rank on purpose
    model = torch.nn.Sequential(torch.nn.Linear(64, 300), torch.nn.Linear(300, 7), torch.nn.LayerNorm(7))
    buckets = dp.GradBuckets(model.parameters(), bucket_bytes=5_000)  # forces several buckets
    assert len(buckets.buckets) >= 2
    buckets.broadcast_parameters(0)
    torch.manual_seed(7 + rank)
    x = torch.randn(5, 64)
    model(x).square().sum().backward()
    local = [p.grad.clone() for p in model.parameters()]
    buckets.allreduce_mean()
    ret[rank] = dict(params=[p.detach().clone() for p in model.parameters()], local=local,
                     avg=[p.grad.clone() for p in model.parameters()])
    dist.barrier()
    dist.destroy_process_group()


def test_dp_allreduce_mean_world2():
    world, port = 2, 29731
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    for pa, pb in zip(a["params"], b["params"]):
        assert torch.equal(pa, pb)                      # broadcast made the replicas identical
    for ga, gb, la, lb in zip(a["avg"], b["avg"], a["local"], b["local"]):
        assert torch.equal(ga, gb)                      # every rank holds the same reduced gradient
        assert torch.allclose(ga, (la + lb) / 2, atol=1e-6)


def _two_phase_worker(rank, world, port, ret):
    sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
    from tavsr import dp
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dp.init_from_env("gloo")

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.head = torch.nn.Linear(30, 7)              # registered first, used last: the plain plan puts it last
            self.stem = torch.nn.Sequential(torch.nn.Linear(64, 300), torch.nn.Tanh(), torch.nn.Linear(300, 30))
            self.both = torch.nn.Parameter(torch.ones(30))   # used below AND above the cut

        def forward(self, x):
            h = self.stem(x) * self.both
            h = dp.cut(h)
            return (self.head(h * self.both)).square().sum()

    torch.manual_seed(3)
    model = Net()
    params = list(model.parameters())
    buckets = dp.GradBuckets(params, bucket_bytes=5_000)
    buckets.broadcast_parameters(0)
    torch.manual_seed(7 + rank)
    x = torch.randn(5, 64)
    model(x).backward()                                      # no plan active: dp.cut is the identity
    local = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    two = dp.TwoPhaseBackward()
    with two.forward():
        loss = model(x)
    late = two.late_params(params)
    n_ready = buckets.replan(late)
    assert 0 < n_ready < len(buckets.buckets)
    assert {id(p) for b in buckets.buckets[n_ready:] for p in b} == {id(p) for p in late}
    two.phase_a(loss)
    early_done = all(p.grad is not None for b in buckets.buckets[:n_ready] for p in b)
    stem_untouched = all(p.grad is None for p in model.stem.parameters())
    buckets.launch_prefix(n_ready)                           # the head's buckets leave before the second phase runs
    issued = buckets._next
    two.phase_b()
    split = [p.grad.clone() for p in params]
    buckets.allreduce_mean()
    ret[rank] = dict(local=local, split=split, avg=[p.grad.clone() for p in params], n_ready=n_ready, issued=issued,
                     late=[n for n, p in model.named_parameters() if any(p is q for q in late)],
                     early_done=early_done, stem_untouched=stem_untouched)
    dist.barrier()
    dist.destroy_process_group()


def test_dp_two_phase_backward_buckets_world2():
    """TwoPhaseBackward + GradBuckets.replan/launch_prefix (bench.py's N > 1 step, here without graphs): the split backward
    gives the gradients of one backward pass, the parameters below the cut (and the one used on both sides) form the late
    buckets, the early buckets are issued before the second phase, and the reduced gradients are the rank mean."""
    world, port = 2, 29771
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_two_phase_worker, args=(world, port, ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    assert sorted(a["late"]) == ["both", "stem.0.bias", "stem.0.weight", "stem.2.bias", "stem.2.weight"]
    for r in (a, b):
        assert r["early_done"] and r["stem_untouched"] and r["issued"] == r["n_ready"]
        for g, s in zip(r["local"], r["split"]):
            assert torch.allclose(g, s, rtol=1e-6, atol=1e-6)
    for ga, gb, la, lb in zip(a["avg"], b["avg"], a["local"], b["local"]):
        assert torch.equal(ga, gb)
        assert torch.allclose(ga, (la + lb) / 2, rtol=1e-5, atol=1e-6)


def _hook_order_worker(rank, world, port, ret):
    """overlap hooks with DIFFERENT gradient sets per rank: rank 1 skips the middle layer (a stochastic-depth coin that fell
    differently), so its bucket never completes there.  Collectives must still be issued 0, 1, 2, ... on every rank."""
    sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
    from tavsr import dp
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dp.init_from_env("gloo")
    torch.manual_seed(5)
    layers = [torch.nn.Linear(16, 16) for _ in range(4)]       # equal-sized buckets: a mis-paired collective would NOT hang,
    params = [p for l in layers for p in l.parameters()]       # it would silently sum the wrong tensors
    buckets = dp.GradBuckets(params, bucket_bytes=16 * 16 * 4)
    assert len(buckets.buckets) == 4
    foreign = params[-1].register_post_accumulate_grad_hook(lambda p: None)      # somebody else's hook, there before this package's
    buckets.attach_overlap_hooks()
    # (ops.wgrad_may_go_beside: a layer may leave its weight gradients in flight on a side queue only if every hook on its parameters is this
    # package's own - those fence before they pack; a parameter that already carried a hook keeps the layer's gradients in place)
    from tavsr import ops
    assert all(getattr(p, "_tavsr_hooks_fence", False) for p in params[:-1]) and not params[-1]._tavsr_hooks_fence
    assert ops.wgrad_may_go_beside(params[:-1]) and not ops.wgrad_may_go_beside(params)
    foreign.remove()
    issued = []
    launch = buckets._launch_bucket_cpu
    buckets._launch_bucket_cpu = lambda i: (issued.append(i), launch(i))[1]
    torch.manual_seed(11 + rank)
    x = torch.randn(3, 16)
    outs = []
    for step in range(3):
        for p in params:
            p.grad = None
        h = x * (step + 1)
        for j, l in enumerate(layers):
            if not (rank == 1 and j == 2 and step != 1):       # rank 1 drops layer 2 in steps 0 and 2
                h = l(h)
        del issued[:]
        buckets.begin_step()
        h.square().sum().backward()
        if step == 2:                                          # a window that is never closed by allreduce_mean ...
            for p in params:
                p.grad = None
            h = x
            for l in layers:
                h = l(h)
            buckets.begin_step()                               # ... then a new one: the stale one is completed and dropped
            h.square().sum().backward()
        local = [None if p.grad is None else p.grad.clone() for p in params]
        buckets.allreduce_mean()
        outs.append(dict(local=local, avg=[p.grad.clone() for p in params], issued=list(issued)))
    ret[rank] = outs
    dist.barrier()
    dist.destroy_process_group()


def test_dp_overlap_hooks_issue_buckets_in_fixed_order_world2():
    world, port = 2, 29761
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_hook_order_worker, args=(world, port, ret), nprocs=world, join=True)
    for step in range(3):
        a, b = ret[0][step], ret[1][step]
        assert a["issued"][-4:] == [0, 1, 2, 3] and b["issued"][-4:] == [0, 1, 2, 3], (a["issued"], b["issued"])
        for ga, gb, la, lb in zip(a["avg"], b["avg"], a["local"], b["local"]):
            la = torch.zeros_like(ga) if la is None else la
            lb = torch.zeros_like(ga) if lb is None else lb
            assert torch.equal(ga, gb)
            assert torch.allclose(ga, (la + lb) / 2, atol=1e-6)


def _gpu_worker(rank, world, port, ret):
    """two ranks sharing cuda:0 over gloo (RCCL needs one device per rank): exercises the HIP pack/unpack path."""
    sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
    from tavsr import dp
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dp.init_from_env("gloo")
    torch.cuda.set_device(0)
    torch.manual_seed(100)
    model = torch.nn.Sequential(torch.nn.Linear(64, 301), torch.nn.Linear(301, 7), torch.nn.LayerNorm(7)).cuda()
    buckets = dp.GradBuckets(model.parameters(), bucket_bytes=5_000)
    assert len(buckets.buckets) >= 2
    torch.manual_seed(7 + rank)
    x = torch.randn(5, 64).cuda()
    outs = []
    for step in range(2):                                # second step reuses the pointer tables / flat buffers
        for p in model.parameters():
            p.grad = None
        model(x * (step + 1)).square().sum().backward()
        local = [p.grad.clone().cpu() for p in model.parameters()]
        buckets.allreduce_mean()
        torch.cuda.synchronize()
        outs.append(dict(local=local, avg=[p.grad.clone().cpu() for p in model.parameters()]))
    ret[rank] = outs
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_dp_allreduce_mean_world2_hip_buckets():
    world, port = 2, 29741
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gpu_worker, args=(world, port, ret), nprocs=world, join=True)
    for step in range(2):
        a, b = ret[0][step], ret[1][step]
        for ga, gb, la, lb in zip(a["avg"], b["avg"], a["local"], b["local"]):
            assert torch.equal(ga, gb)
            assert torch.allclose(ga, (la + lb) / 2, atol=1e-6)


def _real_model_worker(rank, world, port, ret):
    """two ranks sharing cuda:0 over gloo, the real 2-layer ASR model: each rank steps its half of a batch of four through
    the product's training harness (overlap hooks on, fused Adam under Noam)."""
    import argparse
    sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    from helpers import asr_conf
    from oracle.model import fill_parameters_, synth
    from tavsr import dp
    from tavsr.tasks.asr import ASRTask
    from tavsr.train import get_noam_scheduler, training
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dp.init_from_env("gloo")
    torch.cuda.set_device(0)

    def build(seed):
        m = ASRTask.build_model(argparse.Namespace(**asr_conf(num_blocks=2, dec_blocks=1)))
        fill_parameters_(m, seed=seed)
        return m.cuda().train()

    speech = synth((4, 100, 80), seed=8)
    slens = torch.tensor([100, 100, 100, 100])
    text = synth((4, 6), seed=9, kind="int", lo=1, hi=40)
    tlens = torch.tensor([6, 6, 6, 6])
    half = slice(2 * rank, 2 * rank + 2)
    mine = dict(speech=speech[half], speech_lengths=slens[half], text=text[half], text_lengths=tlens[half])

    model = build(7 + rank)                                   # different init per rank: the broadcast must fix it
    buckets = dp.GradBuckets(model.parameters(), bucket_bytes=4 << 20)
    assert len(buckets.buckets) >= 3
    buckets.broadcast_parameters(0)
    buckets.attach_overlap_hooks()
    for p in model.parameters():
        p.grad = None
    buckets.begin_step()
    model(**{k: v.cuda() for k, v in mine.items()})[0].backward()      # hooks enqueue the buckets during this backward
    buckets.allreduce_mean()
    torch.cuda.synchronize()
    avg = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}
    out = dict(avg=avg)
    if rank == 0:                                             # the same model on the concatenated batch, one process
        ref = build(7)
        ref(speech.cuda(), slens.cuda(), text.cuda(), tlens.cuda())[0].backward()
        out["cat"] = {n: p.grad.detach().cpu().clone() for n, p in ref.named_parameters()}
    opt = get_noam_scheduler(model.parameters(), 0.05, 256, 10)
    for _ in range(3):
        training(model, [mine], opt, None, 1, buckets=buckets)
    torch.cuda.synchronize()
    out["params"] = {n: p.detach().cpu().clone() for n, p in model.named_parameters()}
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_dp_real_model_equals_single_process_on_the_concatenated_batch():
    """the two-rank averaged gradients equal the single-process gradients of the concatenated batch, and
    after three optimizer steps the replicas' parameters are bitwise equal."""
    world, port = 2, 29751
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_real_model_worker, args=(world, port, ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    for n, g in a["avg"].items():
        assert torch.equal(g, b["avg"][n]), n                  # both ranks hold the same reduced gradient
        ref = a["cat"][n]
        scale = float(ref.abs().max())
        if scale < 1e-6:
            assert float(g.abs().max()) < 1e-5, n
        else:
            assert float((g - ref).abs().max()) / scale < 2e-4, (n, float((g - ref).abs().max()) / scale)
    for n, p in a["params"].items():
        assert torch.equal(p, b["params"][n]), n               # replicas stay bitwise identical


@pytest.mark.gpu
def test_rccl_c_abi_single_rank_roundtrip():
    """tavsr_dp_* on a one-rank communicator (the box has one GPU): id, init, in-place all-reduce and broadcast on a
    stream, destroy - the calls the multi-GPU path makes, with RCCL resolved by dlopen."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
    from tavsr._lib import check, lib
    torch.cuda.set_device(0)
    ident = (C.c_char * 128)()
    check(lib().tavsr_dp_unique_id(ident), "tavsr_dp_unique_id")
    check(lib().tavsr_dp_init(0, 1, ident), "tavsr_dp_init")
    assert lib().tavsr_dp_world() == 1
    x = torch.randn(1 << 20, device="cuda")
    want = x.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    check(lib().tavsr_dp_allreduce(C.c_void_p(x.data_ptr()), C.c_int64(x.numel()), C.c_void_p(s.cuda_stream)), "tavsr_dp_allreduce")
    check(lib().tavsr_dp_broadcast(C.c_void_p(x.data_ptr()), C.c_int64(x.numel()), 0, C.c_void_p(s.cuda_stream)), "tavsr_dp_broadcast")
    s.synchronize()
    assert torch.equal(x, want)
    check(lib().tavsr_dp_destroy(), "tavsr_dp_destroy")
    assert lib().tavsr_dp_world() == 0


@pytest.mark.gpu
def test_two_graph_step_with_the_real_collective_between_the_replays_world1():
    """The N > 1 step's flow on one GPU, with RCCL itself (a one-rank communicator, tavsr.dp.init_from_env(force_rccl=True)):
    graph A (forward + backward above the cut) -> pack + tavsr_dp_allreduce of the early buckets on the communication stream
    -> graph B (backward below the cut) -> the late buckets -> unpack.  Three replays; loss and every gradient stay bit-equal
    to one eager loss.backward() (a one-rank sum is the identity, 1 / world = 1), the eager hook-driven exchange too."""
    import argparse
    sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import asr_conf
    from oracle.model import synth
    from tavsr import dp, ops
    from tavsr.tasks.asr import ASRTask
    torch.cuda.set_device(0)
    model = ASRTask.build_model(argparse.Namespace(**asr_conf(num_blocks=4, dec_blocks=2, dropout=0.1)))
    torch.manual_seed(0)
    model = model.cuda().train()
    params = [p for p in model.parameters() if p.requires_grad]
    text = synth((8, 30), seed=2, kind="int", lo=1, hi=40)
    batch = [synth((8, 400, 80), seed=1).cuda(), torch.full((8,), 400).cuda(), text.cuda(), torch.full((8,), 30).cuda()]

    def eager():
        ops.manual_seed(99)
        for p in params:
            p.grad = None
        loss = model(*batch)[0]
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), [p.grad.detach().clone() for p in params]

    ref_loss, ref = eager()
    assert not dp.FORCE_WORLD1
    dp.init_from_env(seed=None, force_rccl=True)
    try:
        assert dp.RCCL_ABI and dp.FORCE_WORLD1 and dp._exchange_on()
        buckets = dp.GradBuckets(params, bucket_bytes=8 << 20)
        buckets.attach_overlap_hooks()
        # eager loop: the hooks enqueue the collectives under the backward pass
        ops.manual_seed(99)
        for p in params:
            p.grad = None
        buckets.begin_step()
        loss = model(*batch)[0]
        loss.backward()
        buckets.allreduce_mean()
        torch.cuda.synchronize()
        assert torch.equal(loss.detach(), ref_loss)
        assert all(torch.equal(p.grad, g) for p, g in zip(params, ref))
        # (the eager pass's graph must be gone before a capture: its AccumulateGrad nodes belong to the default stream, and
        # autograd re-uses a parameter's node for as long as anything keeps it alive)
        del loss
        # captured loop: two graphs around the cut, the exchange between and behind them
        buckets.overlap = False
        two = dp.TwoPhaseBackward()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for p in params:
                p.grad = None
            with two.forward():
                warm = model(*batch)[0]
            late = two.late_params(params)
            two.phase_a(warm)
            two.phase_b()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert late and len(late) < len(params)
        for p in params:
            p.grad = None
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga):
            with two.forward():
                static_loss = model(*batch)[0]
            two.phase_a(static_loss)
        with torch.cuda.graph(gb, pool=ga.pool()):
            two.phase_b()
        n_ready = buckets.replan(late)
        assert 0 < n_ready < len(buckets.buckets)
        for rep in range(3):
            ops.manual_seed(99)
            ga.replay()
            buckets.launch_prefix(n_ready)
            gb.replay()
            buckets.allreduce_mean()
            torch.cuda.synchronize()
            assert torch.equal(static_loss.detach(), ref_loss), rep
            bad = [n for (n, p), g in zip(model.named_parameters(), ref) if not torch.equal(p.grad, g)]
            assert not bad, (rep, bad[:6])
    finally:
        dp.shutdown()
    assert not dp.RCCL_ABI and not dp.FORCE_WORLD1


def test_two_phase_backward_refuses_a_loss_that_bypasses_the_cut():
    """with an intermediate-CTC tap below the cut the loss reaches the lower half of the graph without passing a
    detached leaf; phase_a would walk (and free) it and phase_b run it again.  TwoPhaseBackward.phase_a refuses such a step;
    the clean split still equals one backward pass."""
    sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
    from tavsr import dp
    w0, w1, w2 = (torch.randn(4, 4, requires_grad=True) for _ in range(3))
    x = torch.randn(3, 4)

    def net(bypass):
        h0 = torch.tanh(x @ w0)
        h = torch.tanh(h0 @ w1)
        t = dp.cut(h)
        y = torch.tanh(t @ w2).sum()
        if bypass == "at the cut":
            return y + (h * h).sum()
        if bypass == "one layer below":           # a tap strictly below the cut (layer 3 of 12 under a cut at 6)
            return y + (h0 * h0).sum()
        if bypass == "shared parameter":          # legal: w2 used above the cut, w1 again above it - no NODE of the lower graph is shared
            return y + torch.tanh(t @ w1).sum()
        return y

    two = dp.TwoPhaseBackward()
    for how in ("at the cut", "one layer below"):
        with two.forward():
            loss = net(how)
        assert two.split and two.bypassed(loss), how
        with pytest.raises(RuntimeError, match="below dp.cut"):
            two.phase_a(loss)
        w0.grad = w1.grad = w2.grad = None
    with two.forward():
        loss = net("shared parameter")
    assert two.split and not two.bypassed(loss)
    two.phase_a(loss)
    two.phase_b()
    ga = [w.grad.clone() for w in (w0, w1, w2)]
    w0.grad = w1.grad = w2.grad = None
    net("shared parameter").backward()
    assert all(torch.allclose(a, w.grad, atol=1e-6) for a, w in zip(ga, (w0, w1, w2)))
    w0.grad = w1.grad = w2.grad = None
    w1.grad = w2.grad = None
    with two.forward():
        loss = net(False)
    assert two.split and not two.bypassed(loss)
    assert [id(p) for p in two.late_params([w0, w1, w2])] == [id(w0), id(w1)]
    two.phase_a(loss)
    assert w0.grad is None and w1.grad is None and w2.grad is not None
    two.phase_b()
    g1, g2 = w1.grad.clone(), w2.grad.clone()
    w1.grad = w2.grad = None
    net(False).backward()
    assert torch.equal(g1, w1.grad) and torch.equal(g2, w2.grad)


def test_replan_drops_the_old_plans_staging_and_keeps_the_communication_stream():
    sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
    from tavsr import dp
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (5, 7, 3, 9)]
    gb = dp.GradBuckets(ps, bucket_bytes=32)
    n0 = len(gb.buckets)
    gb._hostbuf[0] = torch.zeros(3)
    gb._comm = "stream"
    n_early = gb.replan([ps[0]])
    assert gb._hostbuf == {} and gb._comm == "stream"
    assert 0 < n_early <= len(gb.buckets) and any(ps[0] is p for p in gb.buckets[-1])
    assert sum(len(b) for b in gb.buckets) == 4 and n0 >= 1
