"""which host-side ops issue the ~130 device-to-device copies of a training step (rocprof: __amd_rocclr_copyBuffer)?  One eager step of the
12-layer audio-only model (and the AV one) under torch.profiler with stacks: aten::copy_ / clone / contiguous by caller."""
import argparse, collections, copy, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tailored-avsr_amd"))
import bench as BN
for wl in ("asr", "avsr"):
    BN.WORKLOAD = wl
    torch.manual_seed(0)
    model = BN.build_product_model().cuda().train()
    batch = BN.make_batch(8 if wl == "avsr" else 32, 1234, "cuda")
    params = [p for p in model.parameters() if p.requires_grad]
    def step():
        for p in params:
            p.grad = None
        loss = model(*batch)[0]
        loss.backward()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    cnt = collections.Counter()
    for ev in prof.events():
        if ev.name in ("aten::copy_", "aten::clone", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::cat", "aten::index_select"):
            st = [s for s in (ev.stack or []) if "tavsr" in s or "bench" in s or "autograd" in s]
            where = st[0] if st else ("(autograd engine)" if not ev.stack else ev.stack[0])
            cnt[(ev.name, str(ev.input_shapes)[:60], where[:110])] += 1
    print(f"== {wl}: {sum(cnt.values())} copy-like ops in one step")
    for (name, shp, where), n in cnt.most_common(25):
        print(f"{n:5d}  {name:16s} {shp:60s} {where}")
    del model
    torch.cuda.empty_cache()
