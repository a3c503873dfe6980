"""Run-to-run reproducibility of the AV training step (one process, fixed inputs, dropout 0): every parameter gradient of
iteration k must equal iteration 0 bit for bit.  Prints the parameters that differ.  usage: python scripts/av_repro_check.py [B] [iters]"""
import argparse
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tailored-avsr_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402

from helpers import AVSR_YAML, TOKENS_EN, avsr_conf  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    from tavsr.tasks.avsr import AVSRTask
    kind = os.environ.get("REPRO_MODEL", "tailored")         # tailored | conventional | asr
    if kind == "asr":
        from helpers import asr_conf
        from tavsr.tasks.asr import ASRTask as AVSRTask      # noqa: F811  (same build_model interface)
        conf = asr_conf(num_blocks=12, dec_blocks=6)
    else:
        from helpers import AVSR_CONV_YAML
        conf = avsr_conf(AVSR_YAML if kind == "tailored" else AVSR_CONV_YAML, num_blocks=12, dec_blocks=6)
    conf["token_list"] = list(TOKENS_EN)
    torch.manual_seed(0)
    model = AVSRTask.build_model(argparse.Namespace(**copy.deepcopy(conf))).cuda().train()
    g = torch.Generator().manual_seed(1)
    audio, video = torch.randn(B, 400, 80, generator=g).cuda(), torch.randn(B, 100, 88, 88, generator=g).cuda()
    alens = torch.tensor([400 - 20 * (i % 3) for i in range(B)]).cuda()
    vlens = torch.tensor([100 - 5 * (i % 3) for i in range(B)]).cuda()
    text = torch.randint(1, 40, (B, 40), generator=g)
    tlens = torch.tensor([40 - (i % 7) for i in range(B)])
    for i, l in enumerate(tlens):
        text[i, l:] = -1
    text, tlens = text.cuda(), tlens.cuda()
    ref, bad = None, {}
    cold = os.environ.get("REPRO_COLD") == "1"          # a fresh model (same seed) and an emptied allocator cache per iteration
    for it in range(iters):
        if cold and it > 0:
            del model
            torch.cuda.empty_cache()
            torch.manual_seed(0)
            model = AVSRTask.build_model(argparse.Namespace(**copy.deepcopy(conf))).cuda().train()
        for p in model.parameters():
            p.grad = None
        loss = model(audio, alens, text, tlens)[0] if kind == "asr" else model(audio, alens, video, vlens, text, tlens)[0]
        loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        if ref is None:
            ref, loss0 = grads, float(loss)
            continue
        if float(loss) != loss0:
            print(f"iter {it}: loss {float(loss)!r} != {loss0!r}")
        for n, gr in grads.items():
            if not torch.equal(gr, ref[n]):
                d = float((gr - ref[n]).abs().max() / ref[n].abs().max().clamp_min(1e-30))
                bad.setdefault(n, []).append((it, d))
    flags = {k: v for k, v in os.environ.items() if k.startswith("TAVSR_")}
    print(f"B={B} iters={iters} flags={flags}: {len(bad)} parameters differ between runs")
    names = [n for n, _ in model.named_parameters()]
    groups = {}
    for n, v in bad.items():
        top = ".".join(n.split(".")[:3])
        g_ = groups.setdefault(top, [0, 0.0, set()])
        g_[0] += 1
        g_[1] = max(g_[1], max(d for _, d in v))
        g_[2] |= {it for it, _ in v}
    for top, (cnt, mx, its) in groups.items():
        print(f"  {top}: {cnt} tensors, max rel diff {mx:.3e}, iterations {sorted(its)}")
    good = [n for n in names if n not in bad]
    print("  unaffected groups:", sorted({".".join(n.split(".")[:2]) for n in good}))


if __name__ == "__main__":
    main()
