"""Checkpoint compatibility with the reference's drivers.

The reference writes `{'epoch', 'model_state_dict', 'optimizer_state_dict', 'config'[, 'best_map']}` with `torch.save`
(ref src/train_detect.py:769-792) and reads it back with `torch.load` (`:834-841`, src/eval.py, src/inference.py).
The build's modules keep the reference's 243 state-dict keys, and `training.FusedAdamW` exports / imports
`torch.optim.AdamW`'s state layout, so files move both ways.  Loading uses `weights_only=True`: nothing in the file is
executed (a reference checkpoint holds tensors and plain containers only)."""
from __future__ import annotations

import os
from typing import Any, Dict, Optional

import torch


def save_checkpoint(path, model, optimizer=None, epoch: int = 0, config: Optional[Dict[str, Any]] = None,
                    best_map: Optional[float] = None) -> None:
    ckpt = {"epoch": int(epoch), "model_state_dict": model.state_dict(), "config": dict(config or {})}
    if optimizer is not None:
        ckpt["optimizer_state_dict"] = optimizer.state_dict()
    if best_map is not None:
        ckpt["best_map"] = float(best_map)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(ckpt, path)


def load_checkpoint(path, model=None, optimizer=None, map_location="cpu") -> Dict[str, Any]:
    """Returns the checkpoint dict; loads `model` / `optimizer` in place when given (ref src/train_detect.py:834-841)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    for key in ("epoch", "model_state_dict"):
        if key not in ckpt:
            raise KeyError(f"{path}: not a detector checkpoint (no {key!r})")
    if model is not None:
        model.load_state_dict(ckpt["model_state_dict"])
    if optimizer is not None:
        if "optimizer_state_dict" not in ckpt:
            raise KeyError(f"{path}: checkpoint holds no optimizer state")
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    return ckpt
