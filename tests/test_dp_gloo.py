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
