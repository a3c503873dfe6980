"""Times the fused attention core (tavsr_attn_fwd / tavsr_attn_bwd) at the encoder's shape (B 32, H 4, T 99, d_k 64, rel-pos,
attention dropout 0.1) and the decoder's (self: T 41 causal; source: 41 x 99).  hipGraph replay of 10 calls."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from ffn2_bench import timed  # noqa: E402
from tavsr import ops  # noqa: E402


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    H, dk = 4, 64
    for name, B, T1, T2, pos, causal, p in (("encoder rel-pos", 32, 99, 99, True, False, 0.1), ("encoder rel-pos eval", 32, 99, 99, True, False, 0.0),
                                            ("AV encoder T 100", 32, 100, 100, True, False, 0.1),
                                            ("decoder self", 32, 41, 41, False, True, 0.0), ("decoder source", 32, 41, 99, False, False, 0.0)):
        D = H * dk
        q, kv, ko, vo = r(B * T1, D), r(B * T2, 2 * D), 0, D
        P = r(2 * T1 - 1, D) if pos else None
        u, v = (0.1 * r(D), 0.1 * r(D)) if pos else (None, None)
        klens = torch.full((B,), T2, dtype=torch.int64, device="cuda")
        ops.manual_seed(3)
        fwd = lambda: ops.attn_fwd(q, 0, kv, ko, kv, vo, B, T1, T2, H, dk, klens=klens, causal=causal, pos=P, bias_u=u, bias_v=v, p_drop=p)
        ctx, lse, tok = fwd()
        dctx = r(B * T1, D)
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        bwd = lambda: ops.attn_bwd(dctx, ctx, lse, tok, q, 0, kv, ko, kv, vo, B, T1, T2, H, dk, dq, 0, dkv, ko, dkv, vo, klens=klens,
                                   causal=causal, pos=P, bias_u=u, bias_v=v)
        print(f"{name:24s} fwd {timed(fwd):7.1f} us   bwd {timed(bwd):7.1f} us", flush=True)


if __name__ == "__main__":
    main()
