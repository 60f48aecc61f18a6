"""Checkpoint surface (reference: neuralnets/io.py): one file = {"state_dict", "class_name", "config"}
written with torch.save; loading re-instantiates `class_(**config)` and loads strictly."""
from __future__ import annotations

from typing import Any, Container, Protocol

import torch


class SavableModel(Protocol):
    def state_dict(self) -> dict[str, Any]: ...
    def get_config(self) -> dict[str, Any]: ...
    def load_state_dict(self, d: dict[str, Any], strict: bool) -> None: ...


class InvalidFileFormatError(Exception):
    pass


def complement_lightning_checkpoint(model: SavableModel, checkpoint: dict[str, Any]) -> None:
    assert "state_dict" in checkpoint
    checkpoint["class_name"] = type(model).__name__
    checkpoint["config"] = model.get_config()


def save_model(model: SavableModel, filename: str) -> None:
    contents = {"state_dict": model.state_dict()}
    complement_lightning_checkpoint(model, contents)
    torch.save(contents, filename)


def load_model(filename: str, class_candidates: Container[type]):
    contents = torch.load(filename, weights_only=True)
    missing = [k for k in ("state_dict", "class_name", "config") if k not in contents]
    if missing:
        raise InvalidFileFormatError(f"Bad dict contents. Got {list(contents.keys())}")
    by_name = {c.__name__: c for c in class_candidates}
    instance = by_name[contents["class_name"]](**contents["config"])
    instance.load_state_dict(contents["state_dict"], strict=True)
    return instance
