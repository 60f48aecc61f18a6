"""CPU restatement (numpy) of the on-GPU intensity augmentation chain - SURVEY.md §8 row f3.

TEST INFRASTRUCTURE (part of oracle/): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.

PARITY UNPINNED.  The reference builds the chain from kornia classes (trackertraincode/pipelines.py:508-532,
datatransformation/batch/intensity.py:9-64; `kornia` unpinned in requirements.txt:7) and kornia is not installed here,
so neither golden vectors nor a reference run can anchor these functions.  They restate kornia's published algorithms
(kornia.enhance.equalize/_scale_channel, posterize, adjust_gamma, adjust_contrast, adjust_brightness,
kornia.filters.gaussian_blur2d with border_type='reflect', RandomGaussianNoise.apply_transform) for float images in
[0,1]; the random selection (which operations fire for which sample) is a host-side matter and is tested statistically.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
EQUALIZE, POSTERIZE_BITS, GAMMA, CONTRAST, BRIGHTNESS, BLUR, NOISE_STD, NPARAMS = 0, 1, 2, 3, 4, 5, 6, 8


def equalize(img: np.ndarray) -> np.ndarray:
    """kornia.enhance.equalize -> _scale_channel: 256-bin histogram of im*255 (torch.histc over [0,255]),
    step = (sum(non-empty bins) - last non-empty bin) // 255, lut = (cumsum + step//2) // step shifted by one bin."""
    im = (img.astype(F32) * F32(255.0)).astype(F32)
    inside = (im >= 0) & (im <= 255)
    bins = np.minimum((im[inside] * F32(256.0 / 255.0)).astype(np.int64), 255)
    histo = np.bincount(bins.ravel(), minlength=256).astype(np.int64)
    nonzero = histo[histo != 0]
    step = (int(nonzero.sum()) - int(nonzero[-1])) // 255 if nonzero.size else 0
    if step == 0:
        return img.astype(F32)
    lut = (np.cumsum(histo) + step // 2) // step
    lut = np.clip(np.concatenate([[0], lut[:-1]]), 0, 255)
    idx = np.clip(im.astype(np.int64), 0, 255)  # im.long(): truncation
    return (lut[idx].astype(F32) / F32(255.0)).astype(F32)


def posterize(img, bits: int):
    """kornia.enhance.posterize: uint8(im*255) with the low (8-bits) bits cleared."""
    if bits <= 0 or bits >= 8:
        return img.astype(F32)
    u = np.clip((img.astype(F32) * F32(255.0)).astype(np.int64), 0, 255)
    return (((u >> (8 - bits)) << (8 - bits)).astype(F32) / F32(255.0)).astype(F32)


def adjust_gamma(img, gamma):
    return np.clip(np.power(img.astype(F32), F32(gamma), dtype=F32), 0, 1).astype(F32)


def adjust_contrast(img, factor):
    """kornia.enhance.adjust_contrast (multiplicative form used by RandomContrast)."""
    return np.clip(img.astype(F32) * F32(factor), 0, 1).astype(F32)


def adjust_brightness(img, factor):
    """RandomBrightness(brightness=(lo,hi)) calls adjust_brightness(input, factor - 1): an additive shift."""
    return np.clip(img.astype(F32) + (F32(factor) - F32(1.0)), 0, 1).astype(F32)


def gaussian_blur5(img, sigma=1.5):
    """kornia.filters.gaussian_blur2d((5,5), (sigma,sigma), border_type='reflect'), separable."""
    k = np.arange(5) - 2
    g = np.exp(-(k.astype(F32) ** 2) / F32(2.0 * sigma * sigma)).astype(F32)
    g = (g / g.sum(dtype=F32)).astype(F32)
    H, W = img.shape

    def refl(i, n):
        i = np.abs(i)
        return np.where(i >= n, 2 * n - 2 - i, i)

    cols = refl(np.arange(W)[None, :] + k[:, None], W)  # [5, W]
    tmp = np.zeros_like(img, dtype=F32)
    for t in range(5):
        tmp = tmp + g[t] * img[:, cols[t]].astype(F32)
    rows = refl(np.arange(H)[None, :] + k[:, None], H)
    out = np.zeros_like(img, dtype=F32)
    for t in range(5):
        out = out + g[t] * tmp[rows[t], :]
    return out.astype(F32)


def augment(x: np.ndarray, params: np.ndarray, noise: np.ndarray | None, out_shift: float = 0.0) -> np.ndarray:
    """x [B,H,W] in [0,1]; params [B,NPARAMS]; noise [B,H,W] standard normal or None.  The order of the operations is
    the order of the reference's container (pipelines.py:510-528)."""
    out = np.empty_like(x, dtype=F32)
    for n in range(x.shape[0]):
        p, v = params[n], x[n].astype(F32)
        if p[EQUALIZE] > 0:
            v = equalize(v)
        v = posterize(v, int(p[POSTERIZE_BITS]))
        if p[GAMMA] > 0:
            v = adjust_gamma(v, p[GAMMA])
        if p[CONTRAST] > 0:
            v = adjust_contrast(v, p[CONTRAST])
        if p[BRIGHTNESS] > 0:
            v = adjust_brightness(v, p[BRIGHTNESS])
        if p[BLUR] > 0:
            v = gaussian_blur5(v)
        if noise is not None and p[NOISE_STD] > 0:
            v = v + F32(p[NOISE_STD]) * noise[n].astype(F32)
        out[n] = np.clip(v, 0, 1) + F32(out_shift)
    return out
