"""``LMTask.build_model`` - the language-model recipes (configs/lm/*.yaml; the reference's configs/LM/lm-{english,spanish}.yaml
read by lm_main.py:78-95 and by src/inference/avsr_inference.py:141-176 through espnet2's ``LMTask``): ``lm: transformer`` with
``lm_conf`` -> an ``ESPnetLanguageModel``-shaped holder whose ``lm`` is the ``TransformerLM`` the beam search scores with
(``state_dict`` keys ``lm.embed.*``, ``lm.encoder.*``, ``lm.decoder.*`` as espnet2's).  Training the LM is out of scope
(lm_main.py cannot run as shipped: SURVEY section 2 #14); this is the builder the decode path needs."""
from __future__ import annotations

import argparse

import torch

from ..lm.transformer_lm import TransformerLM
from ..utils.tokens import load_token_list

lm_choices = {"transformer": TransformerLM}


class ESPnetLanguageModel(torch.nn.Module):
    def __init__(self, lm: torch.nn.Module, vocab_size: int, ignore_id: int = 0):
        super().__init__()
        self.lm, self.sos, self.eos, self.ignore_id = lm, vocab_size - 1, vocab_size - 1, ignore_id


class LMTask:
    @classmethod
    def build_model(cls, args: argparse.Namespace) -> ESPnetLanguageModel:
        token_list = load_token_list(args.token_list)
        args.token_list = list(token_list)
        name = getattr(args, "lm", "transformer")
        if name not in lm_choices:
            raise ValueError(f"--lm must be one of {tuple(lm_choices)}: {name}")
        lm = lm_choices[name](len(token_list), **(getattr(args, "lm_conf", None) or {}))
        return ESPnetLanguageModel(lm, len(token_list), **(getattr(args, "model_conf", None) or {}))
