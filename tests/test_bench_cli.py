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
