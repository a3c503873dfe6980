"""``AVSRTask.build_model`` - drop-in for src/tasks/avsr.py:506-718: same YAML keys, string registries restricted to
the classes on the hot path (anything else raises ValueError naming the key)."""
from __future__ import annotations

import argparse

from ..audiovisual_fusion.adaptive_audiovisual_fusion import AdaptiveAudioVisualFusion
from ..ctc.ctc import CTC
from ..decoder.transformer_decoder import TransformerDecoder
from ..embedding_for_avsr.default import DefaultEmbeddingLayerForAVSR
from ..encoder.audiovisual.conventional.encoder import ConventionalEncoder
from ..encoder.audiovisual.tailored.encoder import TailoredEncoder
from ..frontend.conv3d_resnet18 import Conv3dResNet18
from ..frontend.default import DefaultFrontend
from ..models.avsr_espnet_model import ESPnetAVSRModel
from ..models.espnet_model import UtteranceMVN
from ..specaug.specaug import SpecAug
from ..utils.tokens import load_token_list
from .asr import _pick

visual_frontend_choices = {"conv3dresnet18": Conv3dResNet18}
acoustic_frontend_choices = {"default": DefaultFrontend}
specaug_choices = {"specaug": SpecAug}
acoustic_embed_choices = {"default": DefaultEmbeddingLayerForAVSR}
visual_embed_choices = {"default": DefaultEmbeddingLayerForAVSR}
encoder_choices = {"tailored": TailoredEncoder, "conventional": ConventionalEncoder}
audiovisual_fusion_choices = {"adaptive": AdaptiveAudioVisualFusion}
decoder_choices = {"transformer": TransformerDecoder}
normalize_choices = {"utterance_mvn": UtteranceMVN}
model_choices = {"espnet": ESPnetAVSRModel}


class AVSRTask:
    @classmethod
    def build_model(cls, args: argparse.Namespace) -> ESPnetAVSRModel:
        token_list = load_token_list(args.token_list)
        args.token_list = list(token_list)
        vocab_size = len(token_list)
        if args.acoustic_input_size is None:      # waveform in (src/tasks/avsr.py:524-536)
            acoustic_frontend = _pick(acoustic_frontend_choices, getattr(args, "acoustic_frontend", "default"),
                                      "acoustic_frontend")(**(getattr(args, "acoustic_frontend_conf", None) or {}))
            acoustic_input_size = acoustic_frontend.output_size()
        else:
            acoustic_frontend, acoustic_input_size = None, args.acoustic_input_size
        if args.visual_input_size is None:
            visual_frontend = _pick(visual_frontend_choices, args.visual_frontend, "visual_frontend")(
                **(args.visual_frontend_conf or {}))
            visual_input_size = visual_frontend.output_size()
        else:
            visual_frontend, visual_input_size = None, args.visual_input_size
        specaug = None
        if getattr(args, "specaug", None) is not None:
            specaug = _pick(specaug_choices, args.specaug, "specaug")(**(getattr(args, "specaug_conf", None) or {}))
        normalize = None
        if getattr(args, "normalize", None) is not None:
            normalize = _pick(normalize_choices, args.normalize, "normalize")(**(args.normalize_conf or {}))
        d = args.encoder_conf["output_size"]
        acoustic_embed = _pick(acoustic_embed_choices, args.acoustic_embed, "acoustic_embed")(
            input_size=acoustic_input_size, output_size=d, **args.acoustic_embed_conf)
        visual_embed = _pick(visual_embed_choices, args.visual_embed, "visual_embed")(
            input_size=visual_input_size, output_size=d, **args.visual_embed_conf)
        assert acoustic_embed._rel_pos_type == visual_embed._rel_pos_type
        assert acoustic_embed._pos_enc_layer_type == visual_embed._pos_enc_layer_type
        assert acoustic_embed.output_size() == visual_embed.output_size()
        encoder_class = _pick(encoder_choices, args.encoder, "encoder")
        if encoder_class is TailoredEncoder:
            encoder = encoder_class(embed_pos_enc_layer_type=acoustic_embed._pos_enc_layer_type,
                                    embed_rel_pos_type=acoustic_embed._rel_pos_type, **args.encoder_conf)
        else:
            encoder = encoder_class(input_size=acoustic_embed.output_size(),
                                    embed_pos_enc_layer_type=acoustic_embed._pos_enc_layer_type,
                                    embed_rel_pos_type=acoustic_embed._rel_pos_type, **args.encoder_conf)
        fusion = _pick(audiovisual_fusion_choices, args.audiovisual_fusion, "audiovisual_fusion")(
            input_size=encoder.output_size(), **args.audiovisual_fusion_conf)
        encoder_output_size = fusion.output_size()
        decoder = None
        if getattr(args, "decoder", None) is not None:
            decoder = _pick(decoder_choices, args.decoder, "decoder")(
                vocab_size=vocab_size, encoder_output_size=encoder_output_size, **args.decoder_conf)
        ctc = CTC(odim=vocab_size, encoder_output_size=encoder_output_size, **args.ctc_conf)
        model_class = model_choices.get(getattr(args, "model", "espnet"), ESPnetAVSRModel)
        model = model_class(vocab_size=vocab_size, token_list=token_list, specaug=specaug, normalize=normalize,
                            acoustic_frontend=acoustic_frontend, visual_frontend=visual_frontend, acoustic_preencoder=None,
                            visual_preencoder=None, acoustic_embed=acoustic_embed, visual_embed=visual_embed, encoder=encoder,
                            audiovisual_fusion=fusion, postencoder=None, decoder=decoder, ctc=ctc, joint_network=None,
                            **args.model_conf)
        if getattr(args, "init", None) is not None:
            raise NotImplementedError("init: null in every shipped config")
        return model
