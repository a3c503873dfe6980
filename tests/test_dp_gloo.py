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
