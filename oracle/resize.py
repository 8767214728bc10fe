"""Oracle restatement of the pipeline's anti-aliased resize (the first pre-loop stage).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Follows
``/root/reference/pipeline/pipeline_stable_video_diffusion_controlnet.py:604-712`` (``_resize_with_antialiasing``,
``_gaussian``, ``_gaussian_blur2d``, ``_filter2d``, ``_compute_padding``): a separable Gaussian blur with reflect padding
whose sigma follows the down-scaling factor, then bicubic interpolation with ``align_corners=True``.  Pinned by
``tests/golden/resize.npz`` (outputs of the reference function itself).  The restatement is written with explicit index /
weight arithmetic (no ``conv2d`` / ``interpolate`` call), which is also the form the HIP kernels take.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def blur_params(h: int, w: int, size):
    """``:610-628``: (sigma_y, sigma_x), (ks_y, ks_x) - sigma = max((factor - 1) / 2, 0.001), ks = max(int(4 sigma), 3) made odd."""
    factors = (h / size[0], w / size[1])
    sigmas = (max((factors[0] - 1.0) / 2.0, 0.001), max((factors[1] - 1.0) / 2.0, 0.001))
    ks = [int(max(2.0 * 2 * sigmas[0], 3)), int(max(2.0 * 2 * sigmas[1], 3))]
    ks = [k + 1 if k % 2 == 0 else k for k in ks]
    return sigmas, tuple(ks)


def gaussian_taps(window: int, sigma: float) -> torch.Tensor:
    """``_gaussian`` (``:676-689``) in fp32 like the reference: exp(-x^2 / (2 sigma^2)) / sum."""
    s = torch.tensor(sigma, dtype=torch.float32)
    x = torch.arange(window, dtype=torch.float32) - window // 2
    if window % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2 * s.pow(2.0)))
    return g / g.sum()


def _reflect(i: torch.Tensor, n: int) -> torch.Tensor:
    """index of ``F.pad(mode="reflect")``: -1 -> 1, n -> n - 2."""
    i = i.abs()
    return torch.where(i >= n, 2 * (n - 1) - i, i)


def blur_1d(x: torch.Tensor, taps: torch.Tensor, dim: int) -> torch.Tensor:
    """``_filter2d`` along one axis (``:651-673``): pad_front = (k - 1) // 2, reflect, cross-correlation."""
    n, k = x.shape[dim], taps.numel()
    front = (k - 1) // 2
    idx = torch.arange(n)
    out = torch.zeros_like(x)
    for t in range(k):
        out = out + taps[t] * x.index_select(dim, _reflect(idx + t - front, n))
    return out


def cubic_weights(t: torch.Tensor, a: float = -0.75):
    """PyTorch's bicubic convolution coefficients (A = -0.75) for the taps at -1, 0, 1, 2 around floor(x)."""
    def c1(x):  # |x| <= 1
        return ((a + 2) * x - (a + 3)) * x * x + 1
    def c2(x):  # 1 < |x| < 2
        return ((a * x - 5 * a) * x + 8 * a) * x - 4 * a
    return c2(t + 1), c1(t), c1(1 - t), c2(2 - t)


def bicubic_align_corners(x: torch.Tensor, size) -> torch.Tensor:
    """``F.interpolate(mode="bicubic", align_corners=True)``: src = dst * (in - 1) / (out - 1), taps clamped to the border."""
    h, w = x.shape[-2:]
    oh, ow = size

    def axis(n_in, n_out):
        scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
        src = torch.arange(n_out, dtype=torch.float32) * np.float32(scale)
        i0 = torch.floor(src)
        t = src - i0
        i0 = i0.long()
        idx = [(i0 + d).clamp(0, n_in - 1) for d in (-1, 0, 1, 2)]
        return idx, cubic_weights(t)

    iy, wy = axis(h, oh)
    ix, wx = axis(w, ow)
    rows = sum(wy[k].view(-1, 1) * x.index_select(-2, iy[k]) for k in range(4))       # [..., oh, w]
    return sum(wx[k].view(1, -1) * rows.index_select(-1, ix[k]) for k in range(4))     # [..., oh, ow]


def resize_with_antialiasing(x: torch.Tensor, size) -> torch.Tensor:
    if x.ndim == 3:
        x = x.unsqueeze(0)
    h, w = x.shape[-2:]
    sigmas, ks = blur_params(h, w, size)
    x = blur_1d(x, gaussian_taps(ks[1], sigmas[1]), dim=-1)            # out_x first (:705), then y
    x = blur_1d(x, gaussian_taps(ks[0], sigmas[0]), dim=-2)
    return bicubic_align_corners(x, size)
