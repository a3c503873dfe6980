"""Checkpoint helpers with the reference's names, file layout and semantics (src/utils/model_checkpoint.py:6-131):
``model_<suffix>.pth`` files holding a plain ``state_dict`` under ``<output_dir>/models/``, module-wise loading by key
prefix, FairSeq-style checkpoint averaging, module freezing.  Files written by either side load on the other (the
state_dict keys are asserted equal to the reference's in the parity tests).  Host-side file I/O: no kernels.

Table-driven: which branch of the hybrid model a sub-module belongs to, and the error raised when that branch does not
exist for the given CTC weight, live in ``_BRANCH``."""
from __future__ import annotations

import csv
import os
from collections import OrderedDict
from typing import Callable, Dict, Iterable, List, Optional, Tuple

import torch

# sub-module -> (branch exists for this ctc weight?, error text of the reference when it does not)
_BRANCH: Dict[str, Tuple[Callable[[float], bool], str]] = {
    "decoder": (lambda w: w < 1.0, "Attention-based decoding branch"),
    "ctc": (lambda w: w > 0.0, "CTC-based decoding branch"),
}
_LOADABLE = ("frontend", "encoder", "decoder", "ctc")
_FROZEN_NAME = {"frontend": "Frontend", "encoder": "Encoder", "decoder": "Attention-based Decoder", "ctc": "CTC-based Decoder"}


def _require_branch(module: str, weight: float, article: str) -> None:
    rule = _BRANCH.get(module)
    if rule is not None and not rule[0](weight):
        raise RuntimeError(f"The end-to-end model does not have {article} {rule[1]}!")


def _select(checkpoint, keep: Callable[[str], Optional[str]]) -> "OrderedDict[str, torch.Tensor]":
    out = OrderedDict()
    for key, value in checkpoint.items():
        new_key = keep(key)
        if new_key is not None:
            out[new_key] = value
    return out


def load_frontend_lrw(e2e, checkpoint, module_name: str) -> None:
    """the LRW-pretrained visual frontend: every ``trunk`` / ``frontend3D`` entry except the TCN head (:6-16)."""
    weights = _select(checkpoint, lambda k: k if ("tcn_trunk" not in k and ("trunk" in k or "frontend3D" in k)) else None)
    target = {"frontend": "frontend", "visual_frontend": "visual_frontend"}.get(module_name)
    if target is not None:
        getattr(e2e, target).load_state_dict(weights)


def load_module(e2e, module: str, checkpoint, ctc_weight: float) -> None:
    """one sub-module from a whole-model checkpoint: keys containing ``<module>.`` with that prefix removed (:18-43)."""
    prefix = module + "."
    weights = _select(checkpoint, lambda k: k.replace(prefix, "") if prefix in k else None)
    if module in _LOADABLE:
        _require_branch(module, ctc_weight, "an" if module == "decoder" else "a")
        getattr(e2e, module).load_state_dict(weights)


def load_e2e(e2e, modules: Iterable[str], checkpoint_path: str, ctc_weight: float) -> None:
    """(:45-66) ``entire-e2e`` loads everything non-strictly; otherwise the listed modules; LRW files: frontend only."""
    if not checkpoint_path:
        print("No checkpoint given: the end-to-end model starts from its random initialisation.")
        return
    wanted = list(modules)
    checkpoint = torch.load(checkpoint_path, map_location="cpu")
    if "entire-e2e" in wanted:
        print(f"[checkpoint] whole model <- {checkpoint_path} (non-strict)")
        e2e.load_state_dict(checkpoint, strict=False)
        return
    from_lrw = "LRW" in checkpoint_path
    for module in wanted:
        if from_lrw:
            assert module in ["frontend", "visual_frontend"], \
                "When loading from the LRW model, it is only possible loading the frontend."
            load_frontend_lrw(e2e, checkpoint, module)
        else:
            load_module(e2e, module, checkpoint, ctc_weight)
        print(f"[checkpoint] {module} <- {checkpoint_path}")


def average_model(e2e, checkpoint_paths: List[str]) -> None:
    """(:68-89) entry-wise mean of the checkpoints (sum in file order, one true division), loaded strictly."""
    states = (torch.load(path, map_location="cpu") for path in checkpoint_paths)
    total: Dict[str, torch.Tensor] = {}
    for state in states:
        for name, tensor in state.items():
            total[name] = total[name] + tensor if name in total else tensor.clone()
    count = len(checkpoint_paths)
    e2e.load_state_dict({name: torch.div(tensor, count) for name, tensor in total.items()})


def set_bn_eval(module) -> None:
    if isinstance(module, torch.nn.modules.batchnorm._BatchNorm):
        module.eval()


def freeze_e2e(e2e, modules: Iterable[str], mtlalpha: float) -> None:
    """(:95-121).  Quirk kept: for ``ctc`` the reference sets a misspelt attribute (``requieres_grad``), so the CTC head
    stays trainable; the same (non-)effect here."""
    wanted = list(modules)
    if "no-frozen" in wanted:
        print("Nothing is frozen: every parameter of the model is trained.")
        return
    for module in wanted:
        if module not in _FROZEN_NAME:
            continue
        _require_branch(module, mtlalpha, "a")
        if module != "ctc":                                   # see the docstring: the reference's CTC freeze has no effect
            for param in getattr(e2e, module).parameters():
                param.requires_grad = False
        print(f"[freeze] {_FROZEN_NAME[module]}")


def save_model(output_dir: str, model, suffix: str) -> str:
    folder = output_dir + "/models/"
    os.makedirs(folder, exist_ok=True)
    path = os.path.join(folder, f"model_{suffix}.pth")
    torch.save(model.state_dict(), path)
    print(f"[checkpoint] {path} written")
    return path


def save_val_stats(output_dir: str, val_stats) -> None:
    """``val_stats.csv`` with the columns pandas writes for the reference (:133-136): index, model_check_path, cer."""
    with open(os.path.join(output_dir, "val_stats.csv"), "w", newline="") as f:
        writer = csv.writer(f)
        writer.writerow(["", "model_check_path", "cer"])
        writer.writerows([i, path, cer] for i, (path, cer) in enumerate(val_stats))
