"""Reference: datatransformation/tensors/normalization.py:19-24."""
import torch


def whiten_image(image: torch.Tensor):
    return image - 0.5


def unwhiten_image(image: torch.Tensor):
    return image + 0.5
