"""``Conv3dResNet18`` - drop-in for src/frontend/conv3d_resnet18/conv3d_resnet18.py:39-97 (+ modules/resnet.py).

Same constructor, ``forward(speech, speech_lengths)``, ``output_size()`` and state_dict keys (``frontend3D.0.weight``,
``frontend3D.1.*``, ``trunk.layer{1..4}.{0,1}.{conv1,bn1,conv2,bn2,downsample.{0,1}}.*``, BatchNorm running buffers and
``num_batches_tracked`` included).  The torch.nn modules below are parameter containers only: forward and backward run
as ONE autograd node on the gfx950 kernels (``tavsr.functional_av.VisualFrontendFn``).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn

from .. import functional_av as FA


class _Swish(nn.Module):      # placeholder so that Sequential indices match the reference (no parameters)
    pass


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, activation_type="swish"):
        super().__init__()
        assert activation_type in ["relu", "prelu", "swish"]
        if activation_type != "swish":
            raise ValueError("the HIP path covers activation_type='swish' (all shipped configs)")
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu1, self.relu2 = _Swish(), _Swish()
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class ResNet(nn.Module):
    def __init__(self, block=BasicBlock, layers=(2, 2, 2, 2), activation_type="swish"):
        super().__init__()
        if tuple(layers) != (2, 2, 2, 2):
            raise ValueError("the HIP path covers the ResNet-18 layout [2, 2, 2, 2]")
        self.inplanes = 64
        self.activation_type = activation_type
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, activation_type=self.activation_type)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes, activation_type=self.activation_type) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


class Conv3dResNet18(nn.Module):
    def __init__(self, activation_type="swish"):
        super().__init__()
        self.frontend3D = nn.Sequential(
            nn.Conv3d(1, 64, kernel_size=(5, 7, 7), stride=(1, 2, 2), padding=(2, 3, 3), bias=False),
            nn.BatchNorm3d(64), _Swish(), nn.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1)))
        self.trunk = ResNet(BasicBlock, [2, 2, 2, 2], activation_type=activation_type)
        self._names = FA.frontend_param_names()

    def output_size(self) -> int:
        return self.trunk.layer4[1].conv2.out_channels

    EVAL_CHUNK = 32

    def forward(self, speech: torch.Tensor, speech_lengths: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """speech (B, T, 88, 88) lip crops -> (B, T, 512) per-frame features."""
        if speech.dim() != 4:
            raise ValueError(f"expected (batch, time, height, width), got {tuple(speech.shape)}")
        params = dict(self.named_parameters())
        cfg = dict(names=self._names, buffers=dict(self.named_buffers()), training=self.training)
        P = [params[n] for n in self._names]
        if not self.training and not torch.is_grad_enabled() and speech.shape[0] > self.EVAL_CHUNK:
            # eval: BatchNorm uses the running statistics, so utterances are independent - decode batches of hundreds
            # of clips go through in slices (the stem's patch matrix is 197 MB per clip)
            feats = torch.cat([FA.VisualFrontendFn.apply(speech[i:i + self.EVAL_CHUNK], cfg, *P)
                               for i in range(0, speech.shape[0], self.EVAL_CHUNK)], dim=0)
        else:
            feats = FA.VisualFrontendFn.apply(speech, cfg, *P)
        return feats, speech_lengths
