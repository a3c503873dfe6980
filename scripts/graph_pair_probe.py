"""Two independent chains of launches (the search step's LM | decoder): how a replayed hipGraph and its alternatives run them.
  (a) ONE captured graph with a fork (chain A on the side stream, chain B on the capturing stream, join)   - what the search step does
  (b) two graphs, replayed one after the other by one host thread on two streams
  (c) two graphs, replayed by two host threads at the same time
Each launch is tavsr_spin(us): one wave polling the clock, so a chain of n launches is n * us of device time whatever else runs.
Prints the wall time of one iteration (median of 20) against the chains' own lengths."""
import os, sys, threading, statistics, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
from tavsr import ops

dev = torch.device("cuda:0")
main, sa, sb = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()


_X = torch.randn(640, 512, device=dev)
_W1, _W2 = torch.randn(2048, 512, device=dev) * 0.02, torch.randn(512, 2048, device=dev) * 0.02
_H, _Y = torch.empty(640, 2048, device=dev), torch.empty(640, 512, device=dev)
KIND = "spin"


def chain(n, us):
    """n launches: spins of `us`, or (KIND = "gemm") the two Linears of a batched one-token step's feed-forward block, alternating"""
    for j in range(n):
        if KIND == "spin":
            ops.spin(us)
        elif j % 2 == 0:
            ops.gemm(640, 2048, 512, _X, 512, _W1, 512, _H, 2048)
        else:
            ops.gemm(640, 512, 2048, _H, 2048, _W2, 2048, _Y, 512)


def timed(fn, iters=20):
    ts = []
    for _ in range(iters + 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    return statistics.median(ts[3:])


for na, nb, us in ((130, 60, 3.0), (130, 60, 15.0), (30, 30, 100.0), (130, 60, -1.0)):
    KIND = "spin" if us > 0 else "gemm"
    with torch.cuda.stream(main):
        chain(2, 1.0)
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=main):
            with ops.BranchScope(True) as br:
                chain(na, us)
            chain(nb, us)
            br.join()
    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga, stream=sa):
        chain(na, us)
    with torch.cuda.graph(gb, stream=sb):
        chain(nb, us)

    def one():
        with torch.cuda.stream(main):
            g1.replay()

    def two_seq():
        with torch.cuda.stream(sa):
            ga.replay()
        with torch.cuda.stream(sb):
            gb.replay()

    go, done = threading.Event(), threading.Event()
    stop = [False]

    def worker():
        torch.cuda.set_device(0)
        while True:
            go.wait(); go.clear()
            if stop[0]:
                return
            with torch.cuda.stream(sa):
                ga.replay()
            done.set()

    th = threading.Thread(target=worker, daemon=True)
    th.start()

    def two_thr():
        go.set()
        with torch.cuda.stream(sb):
            gb.replay()
        done.wait(); done.clear()

    ta = timed(lambda: (torch.cuda.set_stream(sa), ga.replay()))
    tb = timed(lambda: (torch.cuda.set_stream(sb), gb.replay()))
    torch.cuda.set_stream(torch.cuda.default_stream())
    print(f"chains {na} | {nb} launches of {us:5.1f} us: A alone {ta:8.1f} us  B alone {tb:8.1f} us   one graph with a fork {timed(one):8.1f} us   "
          f"two graphs, one host thread {timed(two_seq):8.1f} us   two graphs, two host threads {timed(two_thr):8.1f} us", flush=True)
    stop[0] = True
    go.set()
    th.join()

# ---- back-to-back replays of ONE captured chain against two captures of the same chain replayed alternately (does a replay of an executable
# graph that is still running wait on the host?)
for n, us in ((60, 3.0), (190, 3.0), (190, 12.0)):
    KIND = "spin"
    gs = []
    for _ in range(2):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            chain(n, us)
        gs.append(g)

    def same(reps=40):
        with torch.cuda.stream(main):
            for _ in range(reps):
                gs[0].replay()

    def alternate(reps=40):
        with torch.cuda.stream(main):
            for r in range(reps):
                gs[r & 1].replay()

    def same_with_event(reps=40):       # what the search's loop does between two replays: an event, a wait on another stream
        with torch.cuda.stream(main):
            for _ in range(reps):
                gs[0].replay()
                e = torch.cuda.Event(); e.record(main); sa.wait_event(e)

    def with_record_only(reps=40):
        with torch.cuda.stream(main):
            for _ in range(reps):
                gs[0].replay()
                e = torch.cuda.Event(); e.record(main)

    def with_eager_launch(reps=40):
        with torch.cuda.stream(main):
            for _ in range(reps):
                gs[0].replay()
                ops.spin(1.0)

    dsrc, pdst = torch.zeros(32, device=dev), torch.zeros(32).pin_memory()

    def with_copy_on_main(reps=40):
        with torch.cuda.stream(main):
            for _ in range(reps):
                gs[0].replay()
                pdst.copy_(dsrc, non_blocking=True)

    try:
        e_ext = torch.cuda.Event(external=True)
        g_ext = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_ext, stream=main):
            chain(n, us)
            e_ext.record(torch.cuda.current_stream())

        def with_external_event(reps=40):       # the event is a node of the graph; another stream waits for it after every replay
            with torch.cuda.stream(main):
                for _ in range(reps):
                    g_ext.replay()
                    sa.wait_event(e_ext)
        t_ext = f"{timed(with_external_event, 5) / 40:8.1f} us"
    except Exception as exc:      # (no external events in this torch / HIP)
        t_ext = f"n/a ({type(exc).__name__})"
    print(f"   an event recorded INSIDE the captured graph (external) that another stream waits for after every replay {t_ext}", flush=True)
    print(f"   between two replays: an event record only {timed(with_record_only, 5) / 40:8.1f} us   an eager launch {timed(with_eager_launch, 5) / 40:8.1f} us   "
          f"a 128-byte copy to pinned memory on the same stream {timed(with_copy_on_main, 5) / 40:8.1f} us", flush=True)
    print(f"chain of {n} launches of {us:4.1f} us, 40 replays back to back, per replay: the same executable graph {timed(same, 5) / 40:8.1f} us   "
          f"two captures alternately {timed(alternate, 5) / 40:8.1f} us   same + event record / wait {timed(same_with_event, 5) / 40:8.1f} us", flush=True)
