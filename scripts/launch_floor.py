"""What a dependent launch costs in a captured chain at the grids of a batch-1 search step (tavsr_probe_launch): nothing at all,
one memory round trip + store, and that plus a barrier and a second dependent read.  The floor under tavsr_rowlin / tree attention."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
from tavsr._lib import addr, check, lib, stream

def chain_us(fn, n=40, reps=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n)

buf = torch.zeros(64 << 20, device="cuda")          # 256 MB: successive launches of a chain do not find their lines in the L2s
for grid, block in ((1, 64), (32, 256), (80, 64), (96, 512), (128, 512), (128, 1024), (256, 256), (1024, 256)):
    row = []
    for kind in (0, 1, 2):
        off = [0]
        def fn():
            off[0] = (off[0] + grid * block * 4 + 4096) % (buf.numel() - grid * block * 8)
            check(lib().tavsr_probe_launch(kind, grid, block, C.c_void_p(addr(buf, off[0])), grid * block * 4, stream()), "probe")
        row.append(chain_us(fn))
    print(f"grid {grid:5d} x {block:4d} threads: empty {row[0]:5.2f} us   load+store {row[1]:5.2f} us   load, barrier, load, store {row[2]:5.2f} us", flush=True)
