import os, sys, torch
ROOT = "/root/repo" if os.path.exists("/root/repo/scripts") else os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from ffn2_bench import timed
from tavsr import ops
M, D = 3168, 256
x = torch.randn(M, D, device="cuda"); w1 = torch.randn(2048, D, device="cuda") / 16; b1 = torch.randn(2048, device="cuda")
for act in (None, "relu", "swish", "gelu"):
    print(act, "fwd %.1f us" % timed(lambda: ops.linear(x, w1, b1, act=act)), "fwd+z %.1f us" % timed(lambda: ops.linear(x, w1, b1, act=act, save_z=True)), flush=True)
dy = torch.randn(M, 1024, device="cuda"); w2 = torch.randn(256, 1024, device="cuda") / 32
z = torch.randn(M, 2048, device="cuda"); dg = torch.randn(M, 2048, device="cuda")
for act in ("relu", "gelu"):
    print(act, "act_bwd %.1f us" % timed(lambda: ops.act_bwd_(dg, z, act)), flush=True)
