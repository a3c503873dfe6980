"""__graft_entry__.smoke() with the caching allocator's blocks pre-filled with NaN (see tests/conftest.py::_poisoned_allocator)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402
import __graft_entry__ as g  # noqa: E402

conftest._poison_device_memory(8, float("nan") if (sys.argv[1:] or ["nan"])[0] == "nan" else 3.0e38)
g.smoke()
