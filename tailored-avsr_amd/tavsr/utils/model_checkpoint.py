"""Checkpoint helpers with the reference's names, file layout and semantics (src/utils/model_checkpoint.py:6-131):
``model_<suffix>.pth`` files holding a plain ``state_dict`` under ``<output_dir>/models/``, module-wise loading by key
prefix, FairSeq-style checkpoint averaging, module freezing.  Files written by either side load on the other (the
state_dict keys are asserted equal to the reference's in the parity tests).  Host-side file I/O: no kernels."""
from __future__ import annotations

import csv
import os
from collections import OrderedDict
from typing import Iterable, List

import torch


def load_frontend_lrw(e2e, checkpoint, module_name: str) -> None:
    """the LRW-pretrained visual frontend: every ``trunk`` / ``frontend3D`` entry except the TCN head (:6-16)."""
    picked = OrderedDict((k, v) for k, v in checkpoint.items()
                         if "tcn_trunk" not in k and ("trunk" in k or "frontend3D" in k))
    if module_name == "frontend":
        e2e.frontend.load_state_dict(picked)
    elif module_name == "visual_frontend":
        e2e.visual_frontend.load_state_dict(picked)


def load_module(e2e, module: str, checkpoint, ctc_weight: float) -> None:
    """one sub-module from a whole-model checkpoint: keys containing ``<module>.`` with that prefix removed (:18-43)."""
    sub = OrderedDict((k.replace(module + ".", ""), v) for k, v in checkpoint.items() if module + "." in k)
    if module == "frontend":
        e2e.frontend.load_state_dict(sub)
    if module == "encoder":
        e2e.encoder.load_state_dict(sub)
    if module == "decoder":
        if not ctc_weight < 1.0:
            raise RuntimeError("The end-to-end model does not have an Attention-based decoding branch!")
        e2e.decoder.load_state_dict(sub)
    if module == "ctc":
        if not ctc_weight > 0.0:
            raise RuntimeError("The end-to-end model does not have a CTC-based decoding branch!")
        e2e.ctc.load_state_dict(sub)


def load_e2e(e2e, modules: Iterable[str], checkpoint_path: str, ctc_weight: float) -> None:
    """(:45-66) ``entire-e2e`` loads everything non-strictly; otherwise the listed modules; LRW files: frontend only."""
    if checkpoint_path == "":
        print("Training the end-to-end model from scratch!")
        return
    checkpoint = torch.load(checkpoint_path, map_location="cpu")
    modules = list(modules)
    if "entire-e2e" in modules:
        print(f"Loading the entire E2E system from {checkpoint_path}")
        e2e.load_state_dict(checkpoint, strict=False)
        return
    for module in modules:
        if "LRW" in checkpoint_path:
            assert module in ["frontend", "visual_frontend"], \
                "When loading from the LRW model, it is only possible loading the frontend."
            print(f"Loading pre-trained visual frontend from {checkpoint_path}")
            load_frontend_lrw(e2e, checkpoint, module)
        else:
            print(f"Loading pre-trained {module} from {checkpoint_path}.")
            load_module(e2e, module, checkpoint, ctc_weight)


def average_model(e2e, checkpoint_paths: List[str]) -> None:
    """(:68-89) entry-wise mean of the checkpoints (sum in file order, one true division), loaded strictly."""
    total = {}
    for path in checkpoint_paths:
        for k, p in torch.load(path, map_location="cpu").items():
            total[k] = p.clone() if k not in total else total[k] + p
    n = len(checkpoint_paths)
    e2e.load_state_dict({k: torch.div(v, n) for k, v in total.items()})


def set_bn_eval(module) -> None:
    if isinstance(module, torch.nn.modules.batchnorm._BatchNorm):
        module.eval()


def freeze_e2e(e2e, modules: Iterable[str], mtlalpha: float) -> None:
    """(:95-121).  Quirk kept: for ``ctc`` the reference sets a misspelt attribute (``requieres_grad``), so the CTC head
    stays trainable; the same (non-)effect here."""
    modules = list(modules)
    if "no-frozen" in modules:
        print("The entire E2E system will be trained")
        return
    for module in modules:
        if module == "frontend":
            for p in e2e.frontend.parameters():
                p.requires_grad = False
            print("The Frontend is frozen!!")
        elif module == "encoder":
            for p in e2e.encoder.parameters():
                p.requires_grad = False
            print("The Encoder is frozen!!")
        elif module == "decoder":
            if not mtlalpha < 1.0:
                raise RuntimeError("The end-to-end model does not have a Attention-based decoding branch!")
            for p in e2e.decoder.parameters():
                p.requires_grad = False
            print("The Attention-based Decoder is frozen!!")
        elif module == "ctc":
            if not mtlalpha > 0.0:
                raise RuntimeError("The end-to-end model does not have a CTC-based decoding branch!")
            print("The CTC-based Decoder is frozen!!")


def save_model(output_dir: str, model, suffix: str) -> str:
    dst_root = output_dir + "/models/"
    os.makedirs(dst_root, exist_ok=True)
    dst_path = os.path.join(dst_root, "model_" + suffix + ".pth")
    print(f"Saving model in {dst_path} ...")
    torch.save(model.state_dict(), dst_path)
    return dst_path


def save_val_stats(output_dir: str, val_stats) -> None:
    """``val_stats.csv`` with the columns pandas writes for the reference (:133-136): index, model_check_path, cer."""
    with open(os.path.join(output_dir, "val_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["", "model_check_path", "cer"])
        for i, (path, cer) in enumerate(val_stats):
            w.writerow([i, path, cer])
