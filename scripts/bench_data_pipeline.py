"""throughput of the batch-assembly and scoring kernels against their bounds"""
import os, random, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tailored-avsr_amd")]
import torch
from tavsr import ops
from tavsr.transforms import video_transforms as PV
from tavsr.utils.avsr_dataloader import avsr_data_processing
from tavsr.evaluation.bootstrap_wer import pair_distances

dev = "cuda"
torch.manual_seed(0); random.seed(0)
B, T = 32, 100
samples = [{"sample_id": str(i), "audio": 0.1 * torch.randn(1, 64000, device=dev),
            "video": torch.randint(0, 256, (T, 96, 96), dtype=torch.uint8, device=dev), "transcription": "HOLA QUE TAL"} for i in range(B)]
tr = PV.Compose([PV.Normalise(0.0, 250.0), PV.Normalise(0.421, 0.165), PV.TimeMasking(fps=25, max_seconds=0.4), PV.RandomCrop((88, 88)),
                 PV.RandomHorizontalFlip(0.5)])
tok = types.SimpleNamespace(text2tokens=list)
conv = types.SimpleNamespace(tokens2ids=lambda t: [ord(c) % 40 for c in t])
cfg = types.SimpleNamespace(model_conf={"ignore_id": -1})
for _ in range(3):
    b = avsr_data_processing(samples, None, tr, tok, conv, cfg)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    b = avsr_data_processing(samples, None, tr, tok, conv, cfg)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"collate + augment, batch {B} x {T} frames 96x96 -> {tuple(b['video'].shape)}: {dt*1e3:.2f} ms / batch ({B/dt:.0f} clips/s), host-bound")
# kernel alone: one clip render, event-timed
clip = tr(PV.VideoClip(samples[0]["video"]))
out = torch.empty(T, 88, 88, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
clip.render(out)
torch.cuda.synchronize()
e0.record()
for _ in range(50):
    clip.render(out)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
byt = T * 88 * 88 * 5
print(f"tavsr_video_prep (+ mean frame), one clip: {us:.1f} us, {byt/us/1e3:.1f} GB/s algorithmic (1 B in + 4 B out per pixel)")
# edit distance: 2000 sentence pairs of ~12 words / ~60 characters
rng = random.Random(1)
words = [bytes([97 + rng.randrange(26)]) * rng.randint(1, 6) for _ in range(300)]
pairs_w = [([rng.choice(words) for _ in range(rng.randint(3, 20))], [rng.choice(words) for _ in range(rng.randint(3, 20))]) for _ in range(2000)]
pairs_c = [([bytes([c]) for c in b" ".join(a)], [bytes([c]) for c in b" ".join(h)]) for a, h in pairs_w]
for name, pairs in (("words", pairs_w), ("characters", pairs_c)):
    pair_distances(pairs); torch.cuda.synchronize()
    t0 = time.perf_counter(); d, l = pair_distances(pairs); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    cells = sum(len(a) * len(h) for a, h in pairs)
    print(f"edit distance, 2000 pairs ({name}): {dt*1e3:.2f} ms incl. host packing, {cells/1e6:.1f} M lattice cells")
d32, l32 = d, l
torch.cuda.synchronize(); t0 = time.perf_counter(); r = ops.bootstrap_rates(d32, l32, 1000, 0); torch.cuda.synchronize()
print(f"bootstrap 1000 x 2000 resamples: {(time.perf_counter()-t0)*1e3:.2f} ms")
