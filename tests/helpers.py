"""Shared test helpers (config handling, golden loading). Tests may import oracle/."""
import copy
import os

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASR_YAML = os.path.join(ROOT, "tailored-avsr_amd", "configs", "asr_branchformer_transformer_ctc_english.yaml")

TOKENS_EN = (["<blank>", "<unk>", "'"] + [str(d) for d in range(10)] + ["<space>"]
             + [chr(c) for c in range(ord("A"), ord("Z") + 1)] + ["<sos/eos>"])


def golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)


def asr_conf(num_blocks=12, dropout=0.0, dec_blocks=6, **enc_over):
    """Same edits as oracle/gen_golden.py:asr_conf, on the repo's own YAML."""
    conf = yaml.safe_load(open(ASR_YAML))
    conf["input_size"] = 80
    conf["specaug"] = None
    conf["encoder_conf"]["num_blocks"] = num_blocks
    conf["decoder_conf"]["num_blocks"] = dec_blocks
    for k in ("dropout_rate", "positional_dropout_rate", "attention_dropout_rate"):
        conf["encoder_conf"][k] = dropout
    for k in ("dropout_rate", "positional_dropout_rate", "self_attention_dropout_rate", "src_attention_dropout_rate"):
        conf["decoder_conf"][k] = dropout
    conf["ctc_conf"]["dropout_rate"] = dropout
    conf["encoder_conf"].update(enc_over)
    return copy.deepcopy(conf)


def rel_err(a, b):
    a = torch.as_tensor(a).detach().to(torch.float64).flatten()
    b = torch.as_tensor(b).detach().to(torch.float64).flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_rel(a, b):
    """max |a-b| / max|b| : the '1e-4 rel' activation criterion of BASELINE.json north_star."""
    a = torch.as_tensor(a).detach().to(torch.float64)
    b = torch.as_tensor(b).detach().to(torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def grad_ok(a, b, tol):
    """relative L2 check that tolerates analytically-zero gradients (e.g. linear_k.bias, pooling biases:
    softmax shift invariance), whose fp32 values are rounding noise on both sides."""
    a = torch.as_tensor(a).detach().to(torch.float64).flatten()
    b = torch.as_tensor(b).detach().to(torch.float64).flatten()
    if float(b.abs().max()) < 1e-6:
        return float(a.abs().max()) < 1e-5
    return float((a - b).norm() / b.norm()) < tol


AVSR_YAML = os.path.join(ROOT, "tailored-avsr_amd", "configs", "avsr_tailored_transformer_ctc_english.yaml")
AVSR_CONV_YAML = os.path.join(ROOT, "tailored-avsr_amd", "configs", "avsr_conventional_transformer_ctc_english.yaml")


def _zero_dropout(d):
    for k, v in d.items():
        if isinstance(v, dict):
            _zero_dropout(v)
        elif k.endswith("dropout_rate"):
            d[k] = 0.0


def avsr_conf(yaml_path=AVSR_YAML, num_blocks=12, dec_blocks=6, visual_input_size=None, **enc_over):
    """Same edits as oracle/gen_golden.py:avsr_conf, on the repo's own YAML."""
    conf = yaml.safe_load(open(yaml_path))
    conf["acoustic_input_size"] = 80
    conf["visual_input_size"] = visual_input_size
    conf["specaug"] = None
    _zero_dropout(conf)
    if conf["encoder"] == "tailored":
        conf["encoder_conf"]["num_blocks"] = num_blocks
        conf["encoder_conf"]["acoustic_use_attn"] = conf["encoder_conf"]["acoustic_use_attn"][:num_blocks]
        conf["encoder_conf"]["visual_use_attn"] = conf["encoder_conf"]["visual_use_attn"][:num_blocks]
    else:
        conf["encoder_conf"]["acoustic_encoder_conf"]["num_blocks"] = num_blocks
        conf["encoder_conf"]["visual_encoder_conf"]["num_blocks"] = num_blocks
    conf["decoder_conf"]["num_blocks"] = dec_blocks
    conf["encoder_conf"].update(enc_over)
    return copy.deepcopy(conf)


def relu_gated_tol(name, tol):
    """Tolerance for a gradient of the product against the reference: the two 3x3 convolutions of Conv2dSubsampling sit
    behind ReLU masks over ~0.4 M pre-activations computed as K = 2304 fp32 sums; a unit whose pre-activation is 0 to
    within the rounding of the summation ORDER (observed: 1.5e-8 against values of order 1) is on in one order and off
    in another, and one such unit moves these gradients by ~3e-3 of their norm (1 / sqrt(active units)).  The reference's
    own order is one arbitrary choice; the product's depends on its K split.  Everything else keeps ``tol``."""
    return max(tol, 1e-2) if (".embed.conv." in name or name.startswith("embed.conv.")) else tol
