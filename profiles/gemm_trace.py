"""Per-workgroup phase timeline of the LDS-DMA GEMM kernel (debug build with -DTAVSR_GEMM_TRACE, see
scripts/gpu_trace.sh).  For each shape: one traced launch; prints the launch span and, over the workgroups,
start offset / prologue (entry -> first tile landed) / K loop / epilogue in microseconds (wall_clock64, 100 MHz).
usage: TAVSR_LIB=.../lib_trace/libtavsr_hip.so python profiles/gemm_trace.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tailored-avsr_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tavsr import _lib, ops  # noqa: E402

SHAPES = [("NT", 3168, 2048, 256, (4, 1)), ("NT", 3168, 2048, 256, (8, 1)), ("NT", 3168, 256, 2048, (4, 1)),
          ("NT", 3168, 256, 2048, (8, 4)), ("NT", 3168, 256, 2048, (8, 5)), ("NT", 3168, 256, 256, (4, 1)),
          ("NN", 3168, 256, 2048, (4, 1)), ("TN", 2048, 256, 3168, (4, 4))]


def pct(x):
    return " ".join(f"{np.percentile(x, q):7.2f}" for q in (0, 10, 50, 90, 100))


def main():
    L = _lib.lib()
    L.tavsr_gemm_trace_read.restype = C.c_int
    buf = np.zeros((1 << 15, 6), dtype=np.uint64)
    for mode, M, N, K, force in SHAPES:
        a = torch.randn(M, K, device="cuda")
        b = torch.randn(K, N, device="cuda")
        A = a.t().contiguous() if mode == "TN" else a
        B = b if mode != "NT" else b.t().contiguous()
        Cout = torch.empty(M, N, device="cuda")
        kw = dict(a_kmajor=mode == "TN", b_kmajor=mode != "NT")
        for _ in range(3):
            ops.gemm(M, N, K, A, A.stride(0), B, B.stride(0), Cout, N, force=force, **kw)
        torch.cuda.synchronize()
        L.tavsr_gemm_trace_read(buf.ctypes.data_as(C.c_void_p), buf.shape[0])     # reset
        ops.gemm(M, N, K, A, A.stride(0), B, B.stride(0), Cout, N, force=force, **kw)
        n = L.tavsr_gemm_trace_read(buf.ctypes.data_as(C.c_void_p), buf.shape[0])
        t = buf[:n, :4].astype(np.int64)
        t0 = t[:, 0].min()
        us = (t - t0) / 100.0
        hw = buf[:n, 4]
        xcc = (hw >> np.uint64(32)) & np.uint64(0xF)
        hwid = hw & np.uint64(0xFFFFFFFF)
        cu = (hwid >> np.uint64(8)) & np.uint64(0xF)
        sh = (hwid >> np.uint64(12)) & np.uint64(0x1)
        se = (hwid >> np.uint64(13)) & np.uint64(0x7)
        cuid = (xcc * np.uint64(8) + se) * np.uint64(32) + sh * np.uint64(16) + cu
        ncu = len(np.unique(cuid))
        per_cu = np.bincount(np.unique(cuid, return_inverse=True)[1])
        print(f"== {mode} M={M} N={N} K={K} cfg/split={force}: {n} workgroups on {ncu} CUs "
              f"(per CU min/med/max {per_cu.min()}/{int(np.median(per_cu))}/{per_cu.max()}), span {us[:, 3].max():.2f} us")
        print(f"   percentiles            {'p0':>7} {'p10':>7} {'p50':>7} {'p90':>7} {'p100':>7}")
        print(f"   start offset           {pct(us[:, 0])}")
        print(f"   prologue (first tile)  {pct(us[:, 1] - us[:, 0])}")
        print(f"   K loop                 {pct(us[:, 2] - us[:, 1])}")
        print(f"   epilogue               {pct(us[:, 3] - us[:, 2])}")
        print(f"   lifetime               {pct(us[:, 3] - us[:, 0])}")
        print(f"   end                    {pct(us[:, 3])}")
        cyc = buf[:n, 5].astype(np.float64)
        loop_us = np.maximum(us[:, 2] - us[:, 1], 1e-3)
        nk = max(1, (K // force[1]) // 32 - 1)
        print(f"   K-loop cycles/K-step   {pct(cyc / nk)}   shader clock GHz {pct(cyc / loop_us / 1e3)}", flush=True)


if __name__ == "__main__":
    main()
