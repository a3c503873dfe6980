"""ORACLE tooling (build container only): import the *reference's own* Python modules
from /root/reference with ``oracle.leaves`` standing in for the absent ``espnet==202402``.

This exists only to generate ``tests/golden/*.npz`` (``oracle/gen_golden.py``) and to
pin the oracle's composite restatement against the reference's own wiring.  Nothing
here is used on the GPU box (``/root/reference`` does not exist there) and nothing from
the reference is copied: its modules are imported from where they lie.

Mechanism: a meta-path finder fabricates every ``espnet*`` / ``espnet2*`` / ``typeguard``
module on demand.  Names listed in ``_REAL`` resolve to ``oracle.leaves``
implementations; any other attribute resolves to an inert placeholder class (the
reference imports ~100 espnet classes it never instantiates for the shipped configs).
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import sys
import types

import torch

from . import leaves as L

REFERENCE_ROOT = "/root/reference"


class _Abs(torch.nn.Module):
    """Placeholder base for the espnet2 ``Abs*`` interfaces (pure ABCs upstream)."""

    def output_size(self):
        return self._output_size


def _placeholder(name):
    return type(name, (_Abs,), {"__module__": "oracle._shim.placeholder"})


class _ClassChoices:
    """espnet2.train.class_choices.ClassChoices: name -> class registry."""

    def __init__(self, name, classes, type_check=None, default=None, optional=False):
        self.name, self.classes, self.default, self.optional = name, dict(classes), default, optional

    def choices(self):
        return tuple(self.classes) + ((None,) if self.optional else ())

    def get_class(self, name):
        if name is None or (isinstance(name, str) and name.lower() in ("none", "null", "nil")):
            if not self.optional:
                raise ValueError(f"{self.name} must not be None")
            return None
        if name.lower() in self.classes:
            return self.classes[name.lower()]
        raise ValueError(f"--{self.name} must be one of {self.choices()}: {name}")

    def add_arguments(self, parser):
        pass


class _AbsTask:
    """Only what ``build_model`` touches."""


_REAL = {
    "typeguard": {"check_argument_types": lambda *a, **k: True, "check_return_type": lambda *a, **k: True},
    "espnet.nets.pytorch_backend.transformer.layer_norm": {"LayerNorm": L.LayerNorm},
    "espnet.nets.pytorch_backend.conformer.swish": {"Swish": L.Swish},
    "espnet.nets.pytorch_backend.transformer.positionwise_feed_forward": {"PositionwiseFeedForward": L.PositionwiseFeedForward},
    "espnet.nets.pytorch_backend.nets_utils": {
        "get_activation": L.get_activation, "make_pad_mask": L.make_pad_mask,
        "th_accuracy": L.th_accuracy, "pad_list": L.pad_list,
    },
    "espnet.nets.pytorch_backend.transformer.repeat": {"repeat": L.repeat, "MultiSequential": L.MultiSequential},
    "espnet.nets.pytorch_backend.transformer.embedding": {
        "PositionalEncoding": L.PositionalEncoding, "RelPositionalEncoding": L.RelPositionalEncoding,
        "ScaledPositionalEncoding": L.ScaledPositionalEncoding,
    },
    "espnet.nets.pytorch_backend.transformer.attention": {
        "MultiHeadedAttention": L.MultiHeadedAttention,
        "RelPositionMultiHeadedAttention": L.RelPositionMultiHeadedAttention,
    },
    "espnet.nets.pytorch_backend.transformer.subsampling": {
        "Conv2dSubsampling": L.Conv2dSubsampling, "Conv2dSubsampling1": L.Conv2dSubsampling1,
        "Conv2dSubsampling2": L.Conv2dSubsampling2, "Conv2dSubsampling6": L.Conv2dSubsampling6,
        "Conv2dSubsampling8": L.Conv2dSubsampling8, "Conv1dSubsampling2": L.Conv1dSubsampling2,
        "Conv1dSubsampling3": L.Conv1dSubsampling3,
        "TooShortUttError": L.TooShortUttError, "check_short_utt": L.check_short_utt,
    },
    "espnet.nets.pytorch_backend.transformer.subsampling_without_posenc": {"Conv2dSubsamplingWOPosEnc": L.Conv2dSubsamplingWOPosEnc},
    "espnet.nets.pytorch_backend.transformer.label_smoothing_loss": {"LabelSmoothingLoss": L.LabelSmoothingLoss},
    "espnet.nets.pytorch_backend.transformer.add_sos_eos": {"add_sos_eos": L.add_sos_eos},
    "espnet.nets.pytorch_backend.transformer.mask": {"subsequent_mask": L.subsequent_mask},
    "espnet.nets.e2e_asr_common": {"ErrorCalculator": L.ErrorCalculator},
    "espnet2.asr.layers.cgmlp": {"ConvolutionalGatingMLP": L.ConvolutionalGatingMLP},
    "espnet2.asr.layers.fastformer": {"FastSelfAttention": L.FastSelfAttention},
    "espnet2.asr.decoder.transformer_decoder": {"TransformerDecoder": L.TransformerDecoder},
    "espnet2.asr.frontend.default": {"DefaultFrontend": L.DefaultFrontend},
    "espnet2.layers.utterance_mvn": {"UtteranceMVN": L.UtteranceMVN},
    "espnet2.torch_utils.device_funcs": {"force_gatherable": L.force_gatherable},
    "espnet2.torch_utils.initialize": {"initialize": lambda model, init: None},
    "espnet2.train.class_choices": {"ClassChoices": _ClassChoices},
    "espnet2.tasks.abs_task": {"AbsTask": _AbsTask},
    "espnet2.text.phoneme_tokenizer": {"g2p_choices": [None]},
    "espnet2.utils.types": {
        "float_or_none": float, "int_or_none": int, "str2bool": bool, "str_or_none": str,
    },
}


class _FakeModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        real = _REAL.get(self.__name__, {})
        if self.__name__ == "espnet2.asr.ctc" and name == "CTC":
            # espnet2's CTC is the class the reference vendors verbatim as src/ctc/ctc.py:
            # the AVSR task imports the espnet2 name, so hand it the reference's own copy
            from src.ctc.ctc import CTC as value
        elif name in real:
            value = real[name]
        else:
            value = _placeholder(name)
        setattr(self, name, value)
        return value


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    PREFIXES = ("espnet", "espnet2", "typeguard")

    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".")[0]
        if root in self.PREFIXES:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _FakeModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


_installed = False


def install():
    """Make ``import src.<...>`` resolve to /root/reference/src over the leaf stand-ins."""
    global _installed
    if _installed:
        return
    import os

    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference checkout not present: the shim is build-container only")
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, REFERENCE_ROOT)
    _installed = True
