"""``ASRTask.build_model`` - drop-in for src/tasks/asr.py:482-619: same YAML keys, string registries
restricted to the classes on the hot path (anything else raises ValueError naming the key)."""
from __future__ import annotations

import argparse

from ..ctc.ctc import CTC
from ..decoder.transformer_decoder import TransformerDecoder
from ..encoder.branchformer.encoder import MyBranchformerEncoder
from ..frontend.conv3d_resnet18 import Conv3dResNet18
from ..frontend.default import DefaultFrontend
from ..models.espnet_model import ESPnetASRModel, UtteranceMVN
from ..specaug.specaug import SpecAug
from ..utils.tokens import load_token_list

frontend_choices = {"default": DefaultFrontend, "conv3dresnet18": Conv3dResNet18}      # src/tasks/asr.py:93-103
specaug_choices = {"specaug": SpecAug}
encoder_choices = {"branchformer": MyBranchformerEncoder}
decoder_choices = {"transformer": TransformerDecoder}
normalize_choices = {"utterance_mvn": UtteranceMVN}
model_choices = {"espnet": ESPnetASRModel}


def _pick(table, name, what):
    if name not in table:
        raise ValueError(f"--{what} must be one of {tuple(table)}: {name}")
    return table[name]


class ASRTask:
    @classmethod
    def build_model(cls, args: argparse.Namespace) -> ESPnetASRModel:
        token_list = load_token_list(args.token_list)
        args.token_list = list(token_list)
        vocab_size = len(token_list)
        if args.input_size is None:       # waveform in: the log-mel frontend is part of the model (src/tasks/asr.py:500-512)
            frontend = _pick(frontend_choices, getattr(args, "frontend", "default"), "frontend")(
                **(getattr(args, "frontend_conf", None) or {}))
            input_size = frontend.output_size()
        else:
            frontend, input_size = None, args.input_size
        specaug = None
        if getattr(args, "specaug", None) is not None:
            specaug = _pick(specaug_choices, args.specaug, "specaug")(**(getattr(args, "specaug_conf", None) or {}))
        normalize = None
        if getattr(args, "normalize", None) is not None:
            normalize = _pick(normalize_choices, args.normalize, "normalize")(**(args.normalize_conf or {}))
        encoder = _pick(encoder_choices, args.encoder, "encoder")(input_size=input_size, **args.encoder_conf)
        decoder = None
        if getattr(args, "decoder", None) is not None:
            decoder = _pick(decoder_choices, args.decoder, "decoder")(
                vocab_size=vocab_size, encoder_output_size=encoder.output_size(), **args.decoder_conf)
        ctc = CTC(odim=vocab_size, encoder_output_size=encoder.output_size(), **args.ctc_conf)
        model_class = model_choices.get(getattr(args, "model", "espnet"), ESPnetASRModel)
        model = model_class(vocab_size=vocab_size, frontend=frontend, specaug=specaug, normalize=normalize,
                            preencoder=None, encoder=encoder, postencoder=None, decoder=decoder, ctc=ctc,
                            joint_network=None, token_list=token_list, **args.model_conf)
        if getattr(args, "init", None) is not None:
            raise NotImplementedError("init: null in every shipped config")
        return model
