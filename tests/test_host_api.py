"""CPU: host-side mirror of the reference API (no GPU, no compute calls): config surface, registry,
state_dict keys, error behaviour, and that the C-ABI library loads and exports every declared symbol."""
import argparse
import ctypes
import os
import re

import pytest
import torch

from helpers import ROOT, TOKENS_EN, asr_conf, golden


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "tavsr.h")).read()
    names = set(re.findall(r"\b(tavsr_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 30
    lib = ctypes.CDLL(os.path.join(ROOT, "tailored-avsr_amd", "tavsr", "lib", "libtavsr_hip.so"))
    for n in sorted(names):
        assert hasattr(lib, n), n
    assert lib.tavsr_version() >= 1


def test_state_dict_keys_match_reference():
    from tavsr.tasks.asr import ASRTask
    from tavsr.utils.tokens import CHAR_ENGLISH
    assert CHAR_ENGLISH == TOKENS_EN and len(CHAR_ENGLISH) == 41
    g = golden("asr_model_3L")
    m = ASRTask.build_model(argparse.Namespace(**asr_conf(num_blocks=3, dec_blocks=2)))
    assert sorted(m.state_dict().keys()) == list(g["keys"])
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"])
    full = ASRTask.build_model(argparse.Namespace(**asr_conf()))
    assert abs(sum(p.numel() for p in full.parameters()) / 1e6 - 51.2) < 0.1  # README table: 51.2 M


def test_layer_variants_keys():
    from tavsr.encoder.branchformer.encoder import MyBranchformerEncoder
    for tag, kw in {"learned": dict(merge_method="learned_ave"), "fixed": dict(merge_method="fixed_ave", cgmlp_weight=0.3),
                    "fixed_attn_only": dict(merge_method="fixed_ave", cgmlp_weight=0.0),
                    "fixed_mlp_only": dict(merge_method="fixed_ave", cgmlp_weight=1.0),
                    "concat": dict(merge_method="concat")}.items():
        enc = MyBranchformerEncoder(input_size=256, num_blocks=1, input_layer=None, ffn_activation_type="swish", **kw)
        assert sorted(enc.encoders[0].state_dict().keys()) == list(golden(f"bf_layer_{tag}")["keys"]), tag


def test_no_cpu_fallback_and_error_behaviour():
    from tavsr._lib import TavsrError
    from tavsr.encoder.branchformer.encoder import MyBranchformerEncoder
    from tavsr.layers import TooShortUttError
    enc = MyBranchformerEncoder(input_size=80, num_blocks=1, input_layer="conv2d", dropout_rate=0.0,
                                positional_dropout_rate=0.0, attention_dropout_rate=0.0).eval()
    with pytest.raises(TavsrError):
        enc(torch.randn(1, 40, 80), torch.tensor([40]))
    with pytest.raises(TooShortUttError):          # encoder.py:356-363
        enc(torch.randn(1, 6, 80), torch.tensor([6]))
    with pytest.raises(ValueError):                # encoder.py:203
        MyBranchformerEncoder(input_layer="bogus")
    with pytest.raises(ValueError):                # encoder_layer.py:148
        MyBranchformerEncoder(input_layer=None, merge_method="bogus")
    layer = MyBranchformerEncoder(input_layer=None, num_blocks=1).encoders[0]
    with pytest.raises(NotImplementedError):       # encoder_layer.py:168-169
        layer((torch.zeros(1, 4, 256), torch.zeros(1, 7, 256)), None, cache=torch.zeros(1))
    # the conventional wrapper: sub-encoder choice and its intermediate-CTC assertions (conventional/encoder.py:96-111, :211-217)
    from tavsr.encoder.audiovisual.conventional.encoder import ConventionalEncoder
    sub = dict(encoder_class_type="branchformer", pos_enc_layer_type="rel_pos", rel_pos_type="latest", num_blocks=2, input_layer=None)
    ConventionalEncoder(256, dict(sub), dict(sub), interctc_layer_idx=[1], interctc_use_conditioning=True)
    with pytest.raises(ValueError):
        ConventionalEncoder(256, dict(sub, encoder_class_type="conformer"), dict(sub))
    with pytest.raises(ValueError):
        ConventionalEncoder(256, dict(sub, encoder_class_type="bogus"), dict(sub))
    with pytest.raises(AssertionError):            # layer index outside 1 .. num_blocks - 1
        ConventionalEncoder(256, dict(sub), dict(sub), interctc_layer_idx=[2])
    with pytest.raises(AssertionError):            # audio-visual conditioning without conditioning
        ConventionalEncoder(256, dict(sub), dict(sub), interctc_layer_idx=[1], audiovisual_interctc_conditioning=True)
    with pytest.raises(AssertionError):            # intermediate CTC belongs to the wrapper, not to a wrapped encoder
        ConventionalEncoder(256, dict(sub, interctc_layer_idx=[1]), dict(sub))


def test_yaml_overrides_grammar(tmp_path):
    from tavsr.utils.config import load_config, override_yaml
    from helpers import ASR_YAML
    cfg = load_config(ASR_YAML, ["encoder_conf:dropout_rate:0.0", "encoder_conf:num_blocks:6", "input_size:80"]
                      if False else ["encoder_conf:dropout_rate:0.0", "encoder_conf:num_blocks:6"])
    assert cfg.encoder_conf["dropout_rate"] == 0.0 and cfg.encoder_conf["num_blocks"] == 6
    d = override_yaml({"a": {"flag": True, "n": 3}, "s": "x"}, ["a:flag:false", "a:n:5", "s:y"])
    assert d == {"a": {"flag": False, "n": 5}, "s": "y"}


def test_add_sos_eos_matches_espnet_semantics():
    from oracle.leaves import add_sos_eos as ref
    from tavsr.models.espnet_model import add_sos_eos
    ys = torch.tensor([[3, 4, 5, -1], [7, 8, 9, 10], [1, -1, -1, -1]])
    lens = torch.tensor([3, 4, 1])
    a, b = add_sos_eos(ys, lens, 40, 40, -1)
    c, d = ref(ys, 40, 40, -1)
    assert torch.equal(a, c) and torch.equal(b, d)


def test_c_oracle_ctc_greedy_matches_python_oracle():
    import numpy as np
    from oracle.leaves import ctc_greedy_collapse
    so = os.path.join(ROOT, "oracle", "_build", "liboracle_ctc.so")
    if not os.path.exists(so):
        import subprocess
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "oracle", "ctc_greedy.c")])
    lib = ctypes.CDLL(so)
    lib.oracle_ctc_greedy.restype = ctypes.c_int64
    g = golden("cfg1_wav_greedy")
    rng = np.random.default_rng(0)
    logits = rng.standard_normal((49, 41)).astype(np.float32)
    logits[5, 7] = logits[5, 3] = 9.0
    ids = np.zeros(49, dtype=np.int64)
    hyp = np.zeros(49, dtype=np.int64)
    n = lib.oracle_ctc_greedy(logits.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(49), ctypes.c_int64(41),
                              ctypes.c_int64(0), ids.ctypes.data_as(ctypes.c_void_p), hyp.ctypes.data_as(ctypes.c_void_p))
    t = torch.from_numpy(logits)
    assert ids.tolist() == t.argmax(-1).tolist()
    assert hyp[:n].tolist() == ctc_greedy_collapse(t.argmax(-1))
    # and on the reference's own config-1 ids
    ref_ids = torch.from_numpy(g["ids"][0])
    assert ctc_greedy_collapse(ref_ids) == g["hyp"].tolist()


def test_every_recipe_yaml_builds_through_the_task_registries():
    """the recipe files under tailored-avsr_amd/configs (values of the reference's configs/{ASR,VSR,AVSR}/*english*.yaml)
    share one top-level key set per task and build on the host through the same registries the reference uses
    (avsr_main.py:185: AVSRTask for ``task: avsr``, ASRTask for asr / vsr)."""
    import copy
    import glob

    import yaml
    from tavsr.tasks.asr import ASRTask
    from tavsr.tasks.avsr import AVSRTask
    files = sorted(glob.glob(os.path.join(ROOT, "tailored-avsr_amd", "configs", "*.yaml")))
    assert len(files) >= 6
    keys = {}
    for f in files:
        conf = yaml.safe_load(open(f))
        keys.setdefault(conf["task"], []).append(set(conf))
        conf["token_list"] = list(TOKENS_EN)
        task = AVSRTask if conf["task"] == "avsr" else ASRTask
        model = task.build_model(argparse.Namespace(**copy.deepcopy(conf)))
        n = sum(p.numel() for p in model.parameters()) / 1e6
        enc = conf["encoder_conf"]
        if conf["task"] == "vsr":
            assert type(model.frontend).__name__ == "Conv3dResNet18" and model.encoder.embed[0].in_features == 512
            assert 50 < n < 70, (f, n)            # ResNet-18 front-end (11.2 M) + the 12-layer encoder / 6-layer decoder
        if isinstance(enc.get("cgmlp_weight"), list):       # the *_tailored recipes: a dead branch is not built at all
            for w, layer in zip(enc["cgmlp_weight"], model.encoder.encoders):
                assert (layer.cgmlp is None) == (w == 0.0) and (layer.attn is None) == (w == 1.0)
    for task, sets in keys.items():
        assert all(s == sets[0] for s in sets), task


def test_stream_safety_rules_are_not_bypassed():
    """_lib.py's two allocator rules live at the one place where a tensor becomes a raw pointer (``ptr`` / ``addr``) and at
    the head of every autograd backward (``guarded``): no module takes ``.data_ptr()`` behind their back (alignment tests and
    identity asserts aside), every ``Function.backward`` of the package is guarded, and no hand-kept ``keep()`` list is left."""
    pkg = os.path.join(ROOT, "tailored-avsr_amd", "tavsr")
    offenders, unguarded, keeps = [], [], []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py"):
                continue
            path = os.path.join(dirpath, f)
            lines = open(path).read().split("\n")
            for i, ln in enumerate(lines):
                if ".data_ptr()" in ln and f != "_lib.py" and "% 16" not in ln and "assert" not in ln:
                    offenders.append(f"{path}:{i + 1}")
                if re.match(r"\s+def backward\(", ln) and "@guarded" not in lines[i - 1]:
                    unguarded.append(f"{path}:{i + 1}")
                if re.search(r"\bbr\.keep\(", ln):
                    keeps.append(f"{path}:{i + 1}")
    assert not offenders, offenders
    assert not unguarded, unguarded
    assert not keeps, keeps


def test_spanish_recipes_and_language_model_recipes():
    """the Spanish recipes (configs/*spanish*.yaml: the reference's values, 37-character inventory of
    src/tokenizers/char/spanish.txt) build with their own token list, and the LM recipes (configs/lm/*.yaml,
    the reference's configs/LM/lm-{english,spanish}.yaml) build through ``LMTask`` with espnet2's ``lm.*`` key prefix."""
    import copy

    import yaml
    from tavsr.tasks.asr import ASRTask
    from tavsr.tasks.lm import LMTask
    from tavsr.utils.tokens import CHAR_SPANISH, load_token_list
    assert len(CHAR_SPANISH) == 37 and CHAR_SPANISH[:3] == ["<blank>", "<unk>", "<space>"] and CHAR_SPANISH[-1] == "<sos/eos>"
    assert load_token_list("char/spanish") == CHAR_SPANISH and "Ñ" in CHAR_SPANISH and "'" not in CHAR_SPANISH
    cfg = os.path.join(ROOT, "tailored-avsr_amd", "configs")
    conf = yaml.safe_load(open(os.path.join(cfg, "asr_branchformer_transformer_ctc_spanish.yaml")))
    assert conf["inference_conf"]["beam_size"] == 30 and conf["inference_conf"]["lm_weight"] == 0.4
    model = ASRTask.build_model(argparse.Namespace(**copy.deepcopy(conf)))
    assert model.ctc.ctc_lo.weight.shape[0] == 37 and model.decoder.output_layer.weight.shape[0] == 37
    for lang, V in (("english", 41), ("spanish", 37)):
        lc = yaml.safe_load(open(os.path.join(cfg, "lm", f"lm_{lang}.yaml")))
        lm = LMTask.build_model(argparse.Namespace(**lc))
        sd = lm.state_dict()
        assert all(k.startswith("lm.") for k in sd) and sd["lm.embed.weight"].shape == (V, 128) and sd["lm.decoder.weight"].shape == (V, 512)
        assert len(lm.lm.encoder.encoders) == 16
    with pytest.raises(ValueError):
        LMTask.build_model(argparse.Namespace(lm="rnn", lm_conf={}, token_list="char/english"))


def test_no_one_token_linear_variant_spills_registers():
    """`rowlin_kernel<32,8,256>` carried 1.5 KB of scratch per lane.  A one-token Linear is ONE round trip to memory by
    design; a spilled operand makes it two.  The launch plan now only instantiates variants that fit (csrc/decode.hip:rowlin_launch;
    `tavsr_rowlin_ok` tells the host which calls exist): hipcc's own resource report of every `rowlin_kernel` instance says 0 bytes."""
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not on this machine")
    csrc = os.path.join(ROOT, "tailored-avsr_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                            "-c", os.path.join(csrc, "decode.hip"), "-o", os.path.join(tmp, "d.o"), "-Rpass-analysis=kernel-resource-usage"],
                           capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", r.stderr)
    scratch = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    assert len(names) == len(scratch) and any("rowlin_kernel" in n for n in names)
    bad = [(n, s) for n, s in zip(names, scratch) if "rowlin_kernel" in n and s > 0]
    assert not bad, bad


def test_weight_gradients_only_go_beside_the_chain_when_nobody_reads_them_inside_the_pass():
    """``ops.wgrad_may_go_beside``: a layer leaves its weight gradients in flight on the side queue (joined at the end of the autograd
    pass) only if AccumulateGrad will KEEP the tensors - no parameter has a ``.grad`` to add into - and nobody's hook reads them where
    they are accumulated; this package's own bucket hooks (``dp.GradBuckets.attach_overlap_hooks``) fence themselves and are let through;
    one queue (``TAVSR_SINGLE_STREAM``) and ``WGRAD_BESIDE = False`` switch it off."""
    import torch
    from tavsr import _lib, ops
    ps = [torch.nn.Parameter(torch.zeros(3)) for _ in range(3)]
    assert ops.wgrad_may_go_beside(ps + [None])
    ps[1].grad = torch.zeros(3)                       # a second micro-batch: AccumulateGrad adds in place, on arrival
    assert not ops.wgrad_may_go_beside(ps)
    ps[1].grad = None
    h = ps[2].register_hook(lambda g: g)              # a tensor hook sees the gradient inside the pass
    assert not ops.wgrad_may_go_beside(ps)
    h.remove()
    assert ops.wgrad_may_go_beside(ps)
    h = ps[0].register_post_accumulate_grad_hook(lambda p: None)      # somebody else's hook
    assert not ops.wgrad_may_go_beside(ps)
    ps[0]._tavsr_hooks_fence = True                   # ... this package's own (dp marks its parameters)
    assert ops.wgrad_may_go_beside(ps)
    keep = (ops.WGRAD_BESIDE, _lib.SINGLE_STREAM)
    try:
        ops.WGRAD_BESIDE = False
        assert not ops.wgrad_may_go_beside(ps)
        ops.WGRAD_BESIDE, _lib.SINGLE_STREAM = True, True
        assert not ops.wgrad_may_go_beside(ps)
    finally:
        ops.WGRAD_BESIDE, _lib.SINGLE_STREAM = keep
    ops.wgrad_fence()                                 # nothing open: a no-op, also without a GPU


def test_weight_gradient_group_deals_many_layers_problems_into_launches_of_one_reduction_length():
    """``ops.WgradGroup._chunks`` (the decoder hands it the weight gradients of all its layers at once): every problem in exactly one grouped
    launch, at most 12 problems and 5 x 256 tiles of 64 x 64 per launch, one reduction length (rows of dy) per launch - a launch lasts as long as
    its longest tile -; up to 12 problems stay one launch in the order they were added (the encoder layer's group, which the C-side
    sequencer mirrors)."""
    import torch
    from tavsr.ops import WgradGroup
    def prob(rows, n, k):
        return (torch.empty(rows, n), torch.empty(rows, k), 1.0, None, None)
    layer = [prob(1312, 2048, 256), prob(1312, 256, 2048)] + [prob(1312, 256, 256)] * 6 + [prob(3168, 256, 256)] * 2     # a decoder layer
    items = [p for _ in range(6) for p in layer]
    chunks = WgradGroup._chunks(items)
    assert sum(len(c) for c in chunks) == len(items) and sorted(map(id, (p for c in chunks for p in c))) == sorted(map(id, items))
    for c in chunks:
        assert 1 <= len(c) <= WgradGroup.MAX
        assert len({p[0].shape[0] for p in c}) == 1
        tiles = sum(WgradGroup._tiles(p) for p in c)
        assert tiles <= 1280 or len(c) == 1
    assert [p[0].shape[0] for c in chunks for p in c] == sorted((p[0].shape[0] for p in items), reverse=True)     # longest reductions first
    small = layer[:10]
    assert WgradGroup._chunks(small) == [small] and WgradGroup._chunks([]) == []
