import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tailored-avsr_amd")
os.chdir("/root/repo")
import torch
import test_gpu_streams as T
from tavsr import _lib, ops
for wl in ("asr", "avsr"):
    model, batch, params = T._setup(wl)
    _lib.SINGLE_STREAM = True
    ref = T._step(model, batch, params)
    _lib.SINGLE_STREAM = False
    for rules_off in (False, True):
        _lib.RULES_OFF = rules_off
        for mode in ("body", "join"):
            ops.arm_race_probe(300.0, mode)
            bad_total = 0
            for rep in range(3):
                got = T._step(model, batch, params)
                bad = sum(1 for a, b in zip(ref[1], got[1]) if not torch.equal(a, b))
                bad_total += bad + (0 if torch.equal(ref[0], got[0]) else 1)
            print(wl, "rules_off" if rules_off else "rules_on ", mode, "mismatching tensors over 3 passes:", bad_total, flush=True)
    _lib.RULES_OFF = False
    ops.arm_race_probe(0.0, "alt")
