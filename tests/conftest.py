import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tailored-avsr_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle the parity tests compare against runs on the host: torch sizes its thread pool by the HOST's cores, a pool box
    # gives a job a share of them (16 CPUs for one GPU), and the oracle on 128 threads over that share is 6 - 10x slower than on 16
    # (scripts/cpu_threads_probe.py).  A cap, not a setting: fewer cores stay as they are.
    import torch
    torch.set_num_threads(min(16, torch.get_num_threads()))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def _poison_device_memory(gib: float, value: float) -> None:
    """Fill ``gib`` GiB of device memory with ``value`` and hand it back to torch's caching allocator: later ``torch.empty``
    blocks then start out holding it, as they would on a box whose memory still holds another job's data."""
    import torch
    if not torch.cuda.is_available():
        return
    blocks = [torch.full((int(256 * 2**20 / 4),), value, device="cuda") for _ in range(int(gib * 4))]
    blocks += [torch.full((n,), value, device="cuda") for n in (2**22, 2**20, 2**18, 2**16, 2**14, 2**12, 2**10, 256) for _ in range(64)]
    torch.cuda.synchronize()
    del blocks


@pytest.fixture(autouse=True)
def _poisoned_allocator():
    """TAVSR_POISON=nan|big: every GPU test starts with the allocator's cached blocks holding NaN / 3e38 - a kernel that reads
    memory it (or its producer) never wrote, and uses the value, shows up as a parity failure instead of passing on zero-filled
    fresh memory."""
    mode = os.environ.get("TAVSR_POISON")
    if mode:
        _poison_device_memory(float(os.environ.get("TAVSR_POISON_GIB", "6")), float("nan") if mode == "nan" else 3.0e38)
    yield
