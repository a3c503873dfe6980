"""CPU: the launcher contract of bench.py / bench_decode.py - `python bench.py --gpus N` without WORLD_SIZE starts its N ranks
itself (a child `torch.distributed.run` on 127.0.0.1, before anything touches the GPU) and leaves with the child's exit code."""
import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script", ["bench", "bench_decode"])
def test_plain_gpus_n_self_launches_its_ranks(script, monkeypatch):
    sys.path.insert(0, ROOT)
    mod = importlib.import_module(script)
    seen = {}

    def fake_call(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = list(cmd), dict(env or {})
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", [script + ".py", "--gpus", "4"])
    with pytest.raises(SystemExit) as ex:
        mod.main()
    assert ex.value.code == 7                                    # the child's exit code
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == script + ".py" and cmd[-2:] == ["--gpus", "4"]
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"


def _report_worker(rank, world, port, ret):
    sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
    from tavsr import dp
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dp.init_from_env("gloo")
    ret[rank] = dp.world_report((rank + 1) * 2**30)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_bench_line_reports_the_communicator_and_every_ranks_peak_memory():
    """first-contact aids of the N > 1 run: bench.py's line carries ``rccl_world`` = tavsr_dp_world() (0 under
    the gloo rig: no RCCL communicator exists, and the line says so) and ``hbm_peak_gb_per_rank`` for every rank."""
    import torch.multiprocessing as mp
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "dp.world_report(" in src
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_report_worker, args=(2, 29757, ret), nprocs=2, join=True)
    for r in (0, 1):
        rep = ret[r]
        assert rep["rccl_world"] == 0 and rep["dist_world"] == 2 and rep["dist_backend"] == "gloo"
        assert rep["hbm_peak_gb_per_rank"] == [1.0, 2.0]
