"""cgMLP backward kernels that evaluate the GELU derivative: LayerNorm backward of the gate half (tavsr_layernorm_bwd_act) and the
depthwise-convolution / gate backward (tavsr_dwconv_gate_bwd_act), M = 3168 rows x 1024 channels (BASELINE configs[1] layer)."""
import os, sys, torch
ROOT = "/root/repo" if os.path.exists("/root/repo/scripts") else os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from ffn2_bench import timed
from tavsr import ops
B, T, Cn = 32, 99, 1024
M = B * T
g = torch.randn(M, 2 * Cn, device="cuda"); z = torch.randn(M, 2 * Cn, device="cuda")
gam = torch.randn(Cn, device="cuda"); bet = torch.randn(Cn, device="cuda")
gn, mean, rstd = ops.layernorm_fwd(g[:, Cn:], gam, bet, 1e-12)
dgn = torch.randn(M, Cn, device="cuda"); dg = torch.empty_like(g)
print("layernorm_bwd_act (gate half)   %.1f us" % timed(lambda: ops.layernorm_bwd_act(dgn, g[:, Cn:], mean, rstd, gam, z[:, Cn:], "gelu", dx=dg[:, Cn:])), flush=True)
print("layernorm_bwd     (no act)      %.1f us" % timed(lambda: ops.layernorm_bwd(dgn, g[:, Cn:], mean, rstd, gam, dx=dg[:, Cn:])), flush=True)
w = torch.randn(Cn, 31, device="cuda") / 8; cb = torch.randn(Cn, device="cuda")
u, conv = ops.dwconv_gate_fwd(gn, g[:, :Cn], w, cb, B, T)
du = torch.randn(M, Cn, device="cuda")
print("dwconv_gate_bwd_act             %.1f us" % timed(lambda: ops.dwconv_gate_bwd(du, gn, g[:, :Cn], conv, w, dg[:, :Cn], B, T, zr=z[:, :Cn])), flush=True)
print("dwconv_gate_bwd (no act)        %.1f us" % timed(lambda: ops.dwconv_gate_bwd(du, gn, g[:, :Cn], conv, w, dg[:, :Cn], B, T)), flush=True)
x = torch.randn(M, Cn, device="cuda")
print("act_bwd gelu [M, 1024]          %.1f us" % timed(lambda: ops.act_bwd_(x, z[:, :Cn].contiguous(), "gelu")), flush=True)
