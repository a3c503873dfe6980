"""BASELINE config 5: hybrid CTC/attention beam search (beam 10, ctc 0.1) + Transformer LM (configs/LM/lm-english.yaml:
16 x 512, 8 heads, lm_weight 0.6, length bonus 0.5) on the tailored AV-Branchformer, synthetic 4 s utterances
(mel 400 x 80 + 100 lip frames 88 x 88), random-init weights.  Real-time factor = wall time to decode / 4 s of speech.

    python bench_decode.py --gpus 1 --utterances 256 --batch 64
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench_decode.py --gpus N ...

Replicas only: rank r decodes utterances r, r + N, ...; no collective on the data path.  Utterances are
decoded in batches of ``--batch``; an utterance's latency is the wall time of its batch (encoder + search), its RTF
that latency / 4 s.  Prints ONE JSON line: value = p50 RTF over all utterances (lower is better), plus the throughput
RTF (total wall / total audio) and the CPU oracle's RTF on a bounded sample.  bench.py stays the headline benchmark."""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "tailored-avsr_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

T_IN, N_MEL, T_VID, HW, DUR_S = 400, 80, 100, 88, 4.0
LM_CONF = dict(pos_enc=None, embed_unit=128, att_unit=512, head=8, unit=2048, layer=16, dropout_rate=0.0)
SEARCH = dict(beam_size=10, ctc_weight=0.1, lm_weight=0.6, penalty=0.5)


def make_conf():
    conf = yaml.safe_load(open(os.path.join(PKG, "configs", "avsr_tailored_transformer_ctc_english.yaml")))
    conf.update(acoustic_input_size=N_MEL, visual_input_size=None, specaug=None)
    from tavsr.utils.tokens import CHAR_ENGLISH
    conf["token_list"] = list(CHAR_ENGLISH)
    return conf


def make_utts(n, seed, device):
    g = torch.Generator().manual_seed(seed)
    audio = torch.randn(n, T_IN, N_MEL, generator=g)
    video = torch.randn(n, T_VID, HW, HW, generator=g)
    return (audio.to(device), torch.full((n,), T_IN, dtype=torch.int64, device=device), video.to(device),
            torch.full((n,), T_VID, dtype=torch.int64, device=device))


def cpu_baseline(model_state, lm_state, n_utt=1):
    """oracle (CPU restatement of the reference's decode path) on the host cores: same weights (the product's state
    dicts load into the oracle classes key for key), same search settings."""
    from oracle import beam_search as BS
    from oracle.av import build_avsr_oracle
    conf = make_conf()
    model = build_avsr_oracle(copy.deepcopy(conf), conf["token_list"]).eval()
    model.load_state_dict(model_state)
    lm = BS.TransformerLMOracle(len(conf["token_list"]), **LM_CONF).eval()
    lm.load_state_dict(lm_state)
    batch = make_utts(n_utt, 99, "cpu")
    host = torch.get_num_threads()
    # torch sizes its pool by the HOST's cores; a pool box hands a job a share of them (16 CPUs for one GPU) and 128 threads over that
    # share run the oracle 6 - 10x slower (scripts/cpu_threads_probe.py; the search alone - ~100 one-token steps on [10, 512] rows - 2 - 3x)
    threads = min(16, host)
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    try:
        with torch.no_grad():
            enc, olens = model.encode(*batch)
            for u in range(n_utt):
                BS.build_beam_search(model, lm, SEARCH["beam_size"], SEARCH["ctc_weight"], SEARCH["lm_weight"],
                                     SEARCH["penalty"]).forward(enc[u, : int(olens[u])])
    finally:
        torch.set_num_threads(host)
    el = time.perf_counter() - t0
    return {"value": round(el / (n_utt * DUR_S), 4), "unit": "RTF", "cores": threads, "kind": "port",
            "sample": f"{n_utt} utterance(s) of 4 s, encoder + beam-{SEARCH['beam_size']} + LM on the CPU oracle, {threads} threads, {el:.1f} s"}


def build(dev):
    """the config-5 stack: tailored AV model + 16 x 512 LM (random-init, seed 1), the batched search and the captured encoder"""
    from tavsr.inference.beam_search import BatchBeamSearch, CapturedEncode
    from tavsr.lm.transformer_lm import TransformerLM
    from tavsr.tasks.avsr import AVSRTask
    conf = make_conf()
    torch.manual_seed(1)                     # random-init weights (no checkpoints offline), the same on every rank
    model = AVSRTask.build_model(argparse.Namespace(**copy.deepcopy(conf))).eval()
    lm = TransformerLM(len(conf["token_list"]), **LM_CONF).eval()
    model, lm = model.to(dev), lm.to(dev)
    search = BatchBeamSearch(model, lm, **SEARCH)
    encode = CapturedEncode(model)           # what tavsr.inference.Speech2Text does: one hipGraph per input shape
    return model, lm, search, encode


def timed_decode(search, encode, dev, utterances, batch_size, rank=0, world=1, warm_full=True):
    """decode ``utterances`` synthetic 4 s clips (this rank's share) in batches of ``batch_size``, inputs resident in HBM before the
    clock starts; -> (per-utterance latencies [s], wall [s], encoder [s], search [s], tokens decoded).  The warm-up batch has the
    timed batches' size (a serving process in steady state: the search's captured step exists when the clock starts);
    ``warm_full=False`` is rounds 1-4's protocol - a warm-up of at most 8 utterances, so that the first timed batch of a larger
    size pays the ~10 ms capture of its step graph (``--cold-capture``)."""
    mine = list(range(rank, utterances, world))
    batches = [mine[i:i + batch_size] for i in range(0, len(mine), batch_size)]

    def run(batch):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            enc, olens = encode(*batch)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            hyps = search.decode(enc, olens, nbest=1)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1, hyps

    # warm-up: allocator pools, lazy module state, the captured step - and, for small batches, the chip's clocks: a batch-1 search keeps
    # a few compute units busy, and the first utterances after an idle spell decode ~14 % slower per token than the following ones
    # (p50 0.0157 with a p90 of 0.0182 over 16 utterances behind ONE warm-up utterance) - a serving process is past that
    for _ in range(1 if not warm_full else max(1, 8 // batch_size)):
        run(make_utts(batch_size if warm_full else min(batch_size, 8), 7, dev))
    data = [make_utts(len(b), 1234 + rank * 1000 + bi, dev) for bi, b in enumerate(batches)]   # resident in HBM
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    T0 = time.perf_counter()
    lat, enc_s, dec_s, ntok = [], 0.0, 0.0, 0
    for b, batch in zip(batches, data):
        e, d, hyps = run(batch)
        enc_s += e
        dec_s += d
        lat += [e + d] * len(b)
        ntok += sum(len(h[0][0]) - 2 for h in hyps if h)
    torch.cuda.synchronize()
    return lat, time.perf_counter() - T0, enc_s, dec_s, ntok


def driver_record(dev, n_b1=8, n_b64=128, cpu=True):
    """the ``decode`` object of bench.py's default line (BASELINE configs[4] on the driver's record): batch-1 p50 RTF over ``n_b1``
    utterances of 4 s and batch-64 throughput over ``n_b64``, one model / LM / search object, same protocol as this file's main()."""
    model, lm, search, encode = build(dev)
    lat1, wall1, e1, d1, tok1 = timed_decode(search, encode, dev, n_b1, 1)
    rtf1 = np.array(lat1) / DUR_S
    lat64, wall64, e64, d64, tok64 = timed_decode(search, encode, dev, n_b64, 64)
    out = {"workload": "BASELINE configs[4]: tailored AV-Branchformer 12L + 6L decoder, beam 10, ctc 0.1, Transformer LM 16x512 "
                       "(lm 0.6), length bonus 0.5, synthetic 4 s utterances, random-init weights",
           "batch1": {"rtf_p50": round(float(np.percentile(rtf1, 50)), 4), "rtf_p90": round(float(np.percentile(rtf1, 90)), 4),
                      "unit": "RTF (latency / 4 s)", "utterances": len(lat1), "encoder_ms_per_utt": round(1e3 * e1 / len(lat1), 2),
                      "search_ms_per_utt": round(1e3 * d1 / len(lat1), 2), "tokens_per_utt": round(tok1 / len(lat1), 1),
                      "search_us_per_token": round(1e6 * d1 / max(tok1, 1), 1)},
           "batch64": {"utterances_per_s": round(len(lat64) / wall64, 2), "utterances": len(lat64), "wall_s": round(wall64, 3),
                       "rtf_p50": round(float(np.percentile(np.array(lat64) / DUR_S, 50)), 4),
                       "encoder_s": round(e64, 3), "search_s": round(d64, 3), "tokens_decoded": int(tok64)},
           "higher_is_better": False, "dtype": "f32", "data": "synthetic"}
    if cpu:
        out["cpu_baseline"] = cpu_baseline({k: v.cpu() for k, v in model.state_dict().items()}, {k: v.cpu() for k, v in lm.state_dict().items()})
    del model, lm, search, encode
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--utterances", type=int, default=256)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cold-capture", action="store_true", help="warm up with at most 8 utterances (the protocol of rounds 1-4)")
    ap.add_argument("--driver-record", action="store_true",
                    help="print the `decode` object of bench.py's default line (batch-1 p50 RTF over 8 utterances, batch-64 utt/s over 128) and exit")
    args = ap.parse_args()

    if args.driver_record:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        print(json.dumps(driver_record(dev)), flush=True)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench_decode.py --gpus N`: start the N replicas as a child torch.distributed.run BEFORE this process
        # touches the GPU, and leave with its exit code (rank 0 prints the line)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    from tavsr import dp

    rank, local, world = dp.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    model, lm, search, encode = build(dev)
    lat, wall, enc_s, dec_s, ntok = timed_decode(search, encode, dev, args.utterances, args.batch, rank, world,
                                                     warm_full=not args.cold_capture)
    stats = torch.tensor([wall, float(len(lat)), enc_s, dec_s, float(ntok)], dtype=torch.float64, device=dev)
    if world > 1:
        allw = [torch.zeros_like(stats) for _ in range(world)]
        torch.distributed.all_gather(allw, stats)
        lats = [None] * world
        torch.distributed.all_gather_object(lats, lat)
        lat = [x for l in lats for x in l]
        wall = max(float(a[0]) for a in allw)
        enc_s, dec_s, ntok = (sum(float(a[k]) for a in allw) for k in (2, 3, 4))
    if rank == 0:
        rtf = np.array(lat) / DUR_S
        out = {
            "metric": "decode_rtf_p50", "value": round(float(np.percentile(rtf, 50)), 4), "unit": "RTF (latency / 4 s)",
            "n_gpus": world, "higher_is_better": False, "scaling": "replicas", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "utterances": len(lat), "batch": args.batch, "warm_up": "cold capture" if args.cold_capture else "full batch",
            "rtf_p90": round(float(np.percentile(rtf, 90)), 4),
            "throughput_rtf": round(wall / (len(lat) * DUR_S), 6),
            "utterances_per_s": round(len(lat) / wall, 2), "wall_s": round(wall, 3),
            "encoder_s": round(enc_s, 3), "search_s": round(dec_s, 3), "tokens_decoded": int(ntok),
            "config": {"workload": "BASELINE configs[4]: tailored AV-Branchformer 12L + 6L decoder, beam 10, ctc 0.1, "
                                   "Transformer LM 16x512 (lm 0.6), length bonus 0.5, 4 s utterances, batched search",
                       "parallelism": f"replicas x{world}"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline({k: v.cpu() for k, v in model.state_dict().items()},
                                               {k: v.cpu() for k, v in lm.state_dict().items()})
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
