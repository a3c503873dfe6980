"""a phase boundary inside one launch (grid barrier, XCD-hierarchical) against a launch boundary, at the shapes of a layer seam: 256
workgroups, every one publishing 4 KB .. 128 KB that another one reads in the next phase (the feed-forward block's partial outputs are 128
rows x 256 floats = 128 KB per workgroup).  us per boundary in a captured chain."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
from tavsr._lib import addr, check, lib, stream

G, NPH, CH = 256, 9, 8
ctl = torch.zeros(2304 // 4, dtype=torch.int32, device="cuda")

def timed(kind, per_wg, reps=100, shift=97, flags=0):
    kind = kind + 256 * (shift + 1) + (flags << 24)
    buf = torch.zeros(NPH * G * per_wg, device="cuda")
    def fn():
        check(lib().tavsr_probe_seam(kind, G, C.c_int64(per_wg), NPH, C.c_void_p(addr(buf)), C.c_void_p(addr(ctl)), C.c_uint32(0), stream()), "seam")
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(CH):              # CH chains of NPH phases per replay: the replay's own fixed cost is shared by 72 phases
            fn()
    for _ in range(reps):                # warm-up: the chip leaves its idle clocks
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    kind &= 255
    ok = flags != 0 or bool((buf.view(NPH, G, per_wg)[NPH - 1, :, 0] == NPH - 1).all())
    return e0.elapsed_time(e1) * 1e3 / (reps * CH), ok

for per_wg in (1024, 8192, 32768):
    t0, ok0 = timed(0, per_wg)
    t1, ok1 = timed(1, per_wg)
    print(f"{per_wg * 4 // 1024:4d} KB per workgroup: {NPH} phases as launches {t0:7.1f} us ({t0 / NPH:5.2f} per phase), as one launch with {NPH - 1} grid "
          f"barriers {t1:7.1f} us -> a barrier seam costs {(t1 - t0) / (NPH - 1):+5.2f} us more than a launch seam   (results ok: {ok0} / {ok1})", flush=True)

for shift, what in ((0, "its OWN slab of the phase before"), (8, "the slab of a workgroup on the same XCD"), (97, "the slab of a workgroup on another XCD")):
    t0 = timed(0, 8192, shift=shift)[0]
    t1 = timed(1, 8192, shift=shift)[0]
    print(f"32 KB per workgroup, a phase reads {what}: launches {t0 / NPH:5.2f} us per phase, fused {t1 / NPH:5.2f} us per phase + barrier", flush=True)

# what does ONE phase-0 launch (writes only, nothing depends on anything) cost in the same kind of chain?
def chain_of(nph, per_wg, reps=100):
    buf = torch.zeros(NPH * G * per_wg, device="cuda")
    def fn():
        check(lib().tavsr_probe_seam(0, G, C.c_int64(per_wg), nph, C.c_void_p(addr(buf)), C.c_void_p(addr(ctl)), C.c_uint32(0), stream()), "seam")
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(72 // nph):
            fn()
    for _ in range(reps):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * 72)
for nph in (1, 2, 9):
    print(f"chains of {nph} phase(s), 4 KB per workgroup, 72 launches per replay: {chain_of(nph, 1024):5.2f} us per launch", flush=True)

print(f"barriers only (no phase work): {timed(1, 1024, flags=1)[0] / (NPH - 1):5.2f} us per grid barrier; empty launches: {timed(0, 1024, flags=1)[0] / NPH:5.2f} us each", flush=True)
print(f"a phase that reads what was written a whole chain ago (32 KB per workgroup): launches {timed(0, 8192, flags=2)[0] / NPH:5.2f} us per phase, "
      f"fused {timed(1, 8192, flags=2)[0] / NPH:5.2f} us per phase + barrier", flush=True)
