"""``MyBranchformerEncoderLayer`` - drop-in for src/encoder/branchformer/encoder_layer.py:49-321.

Same constructor, forward signature, parameter names and introspection attributes
(``weight_global`` / ``weight_local``, read by src/scripts/study_branches.py:44-45); the forward
and backward run as one autograd node on hand-written gfx950 kernels
(``tavsr.functional.BranchformerLayerFn``).
"""
from __future__ import annotations

from typing import Optional

import torch

from ... import functional as F_
from ...layers import LayerNorm


class MyBranchformerEncoderLayer(torch.nn.Module):
    def __init__(self, size: int, attn: Optional[torch.nn.Module], cgmlp: Optional[torch.nn.Module],
                 feed_forward_macaron: Optional[torch.nn.Module], feed_forward: Optional[torch.nn.Module],
                 dropout_rate: float, merge_method: str, cgmlp_weight: float = 0.5,
                 attn_branch_drop_rate: float = 0.0, stochastic_depth_rate: float = 0.0):
        super().__init__()
        assert (attn is not None) or (cgmlp is not None), "At least one branch should be valid"
        if feed_forward_macaron is None or feed_forward is None:
            # the reference itself crashes without the macaron FFN
            raise ValueError("macaron=True with both feed-forward modules is required")
        self.size = size
        self.attn, self.cgmlp = attn, cgmlp
        self.feed_forward_macaron, self.feed_forward = feed_forward_macaron, feed_forward
        self.ff_scale = 0.5
        self.merge_method = merge_method
        self.cgmlp_weight = cgmlp_weight
        self.attn_branch_drop_rate = attn_branch_drop_rate
        self.stochastic_depth_rate = stochastic_depth_rate
        self.use_two_branches = (attn is not None) and (cgmlp is not None)
        self.norm_ff_macaron = LayerNorm(size)
        if attn is not None:
            self.norm_mha = LayerNorm(size)
        if cgmlp is not None:
            self.norm_mlp = LayerNorm(size)
        self.norm_ff = LayerNorm(size)
        self.norm_final = LayerNorm(size)
        self.dropout_rate = dropout_rate
        if self.use_two_branches:
            if merge_method == "concat":
                self.merge_proj = torch.nn.Linear(size + size, size)
            elif merge_method == "learned_ave":
                self.pooling_proj1 = torch.nn.Linear(size, 1)
                self.pooling_proj2 = torch.nn.Linear(size, 1)
                self.weight_proj1 = torch.nn.Linear(size, 1)
                self.weight_proj2 = torch.nn.Linear(size, 1)
                self.merge_proj = torch.nn.Linear(size, size)
            elif merge_method == "fixed_ave":
                assert 0.0 <= cgmlp_weight <= 1.0, "cgmlp weight should be between 0.0 and 1.0"
                if cgmlp_weight == 0.0:
                    self.use_two_branches = False
                    self.cgmlp = None
                    self.norm_mlp = None
                elif cgmlp_weight == 1.0:
                    self.use_two_branches = False
                    self.attn = None
                    self.norm_mha = None
                self.merge_proj = torch.nn.Linear(size, size)
            else:
                raise ValueError(f"unknown merge method: {merge_method}")
        else:
            self.merge_proj = torch.nn.Identity()
        self.weight_global = None
        self.weight_local = None

    # ---- gather parameters in the order BranchformerLayerFn expects (None for absent modules)
    def _params(self):
        from ..._lib import cached_params
        return cached_params(self, F_.BF_PARAM_NAMES)

    def _active_dropout(self) -> bool:
        rates = [self.dropout_rate]
        if self.attn is not None:
            rates.append(self.attn.dropout_rate)
        return self.training and any(r > 0 for r in rates)

    def forward(self, x_input, mask, cache=None, lens: Optional[torch.Tensor] = None):
        """x_input: (x[B,T,size], pos_emb[1,2T-1,size]); mask: (B,1,T) bool prefix mask or None.

        ``lens`` (int64 [B], valid frames) may be supplied by the encoder to avoid recomputing it from
        the mask; masks on this path are always length prefixes (encoder.py:345)."""
        if cache is not None:
            raise NotImplementedError("cache is not None, which is not tested")
        if not isinstance(x_input, tuple):
            raise NotImplementedError("the HIP path implements the rel_pos form: x_input = (x, pos_emb)")
        x, pos_emb = x_input
        coeff = 1.0
        if self.training and self.stochastic_depth_rate > 0:
            skip = torch.rand(1).item() < self.stochastic_depth_rate
            coeff = 1.0 / (1 - self.stochastic_depth_rate)
            if skip:
                return (x, pos_emb), mask
        merge = self.merge_method
        cgmlp_weight = self.cgmlp_weight
        dropped = False
        if (self.training and self.use_two_branches and merge == "learned_ave" and self.attn_branch_drop_rate > 0
                and torch.rand(1).item() < self.attn_branch_drop_rate):
            # encoder_layer.py:233-240: the attention branch is dropped for this step: w1, w2 = 0.0, 1.0 - the merge is a
            # constant-weight one (the branch still runs and receives zero gradients, the pooling / weight projections
            # are not part of the step)
            merge, cgmlp_weight, dropped = "fixed_ave", 1.0, True
        if lens is None and mask is not None:
            lens = mask.squeeze(1).sum(-1).to(torch.int64)
        cfg = dict(heads=self.attn.h if self.attn is not None else 1, ffn_act=self.feed_forward.activation,
                   merge=merge, has_attn=self.attn is not None, has_mlp=self.cgmlp is not None,
                   cgmlp_weight=cgmlp_weight, coeff=coeff,
                   merge_identity=isinstance(self.merge_proj, torch.nn.Identity),
                   # train-mode dropout (self.dropout / PositionwiseFeedForward / csgu: dropout_rate; attention
                   # probabilities: attention_dropout_rate); masks come from the device-resident generator (ops.dropout)
                   p=self.dropout_rate if self.training else 0.0,
                   p_att=(self.attn.dropout_rate if (self.training and self.attn is not None) else 0.0))
        y = F_.grad_apply(F_.BranchformerLayerFn, x, pos_emb, lens, cfg, *self._params())
        w = cfg.get("_last_w")
        if dropped:
            self.weight_global, self.weight_local = 0.0, 1.0
        elif w is not None:  # (B,2) -> the reference's (B,1,1) views (encoder_layer.py:286-289)
            self.weight_global = w[:, 0].view(-1, 1, 1)
            self.weight_local = w[:, 1].view(-1, 1, 1)
        return (y, pos_emb), mask
