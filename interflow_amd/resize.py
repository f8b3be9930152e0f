"""Antialiased bicubic down-scaling of the LR-consistency loss (`train.py:99-100`):
`resize(input, scale_factors=1/4, interp_method=cubic, antialiasing=True)` of the reference's `utils/resize_right.py`
(ResizeRight), for the arguments the training loop uses (last two dims, zero 'constant' padding, no by_convs).

ResizeRight computes, per resized dimension, for every output position a 16-tap field of view and normalised weights
(`resize_right.py:71-123`), then gathers and sums.  The same linear map is a dense matrix A [out, in] (taps that fall on
the zero padding drop out), so the 2-D resize is `A_h @ X @ A_w^T`: two plain GEMMs per call (rocBLAS through
torch.matmul), differentiable for free.  The weights follow the reference's formulas and float32 operation order.
"""
from __future__ import annotations

from math import ceil

import torch


def cubic(x):
    """utils/interp_methods.py:35-43"""
    absx = torch.abs(x)
    absx2, absx3 = absx ** 2, absx ** 3
    return ((1.5 * absx3 - 2.5 * absx2 + 1.) * (absx <= 1.).to(x.dtype) +
            (-0.5 * absx3 + 2.5 * absx2 - 4. * absx + 2.) * ((1. < absx) & (absx <= 2.)).to(x.dtype))


def resize_matrix(in_sz: int, scale: float, device=None, support_sz: float = 4.0) -> torch.Tensor:
    """The [out_sz, in_sz] matrix of one dimension of ResizeRight (antialiasing on, pad_mode 'constant')."""
    eps = torch.finfo(torch.float32).eps
    scale = float(scale)
    out_sz = ceil(scale * in_sz)                                                     # resize_right.py:302-304
    out_coordinates = torch.arange(out_sz, device=device)
    grid = (out_coordinates / scale + (in_sz - 1) / 2 - (out_sz - 1) / (2 * scale))  # :126-137
    if scale < 1.0:                                                                  # :343-355 antialiasing
        interp = lambda a: scale * cubic(scale * a)
        sup = support_sz / scale
    else:
        interp, sup = cubic, support_sz
    left = (grid - sup / 2 - eps).ceil().long()                                      # :140-149
    fov = left[:, None] + torch.arange(ceil(sup - eps), device=device)
    weights = interp(grid[:, None] - fov)                                            # :201-212 (the pad shift cancels)
    s = weights.sum(1, keepdim=True)
    s[s == 0] = 1
    weights = weights / s
    A = torch.zeros(out_sz, in_sz, dtype=weights.dtype, device=device)
    ok = (fov >= 0) & (fov < in_sz)                                                  # taps on the zero padding contribute 0
    rows = torch.arange(out_sz, device=device)[:, None].expand_as(fov)
    A.index_put_((rows[ok], fov[ok]), weights[ok], accumulate=True)
    return A


_cache = {}


def resize_last2(x: torch.Tensor, scale: float = 0.25) -> torch.Tensor:
    """resize(x, scale_factors=scale, interp_method=cubic, antialiasing=True) over the last two dims."""
    H, W = x.shape[-2:]
    key = (H, W, float(scale), str(x.device))
    if key not in _cache:
        if len(_cache) > 16:
            _cache.clear()
        _cache[key] = (resize_matrix(H, scale, x.device), resize_matrix(W, scale, x.device))
    Ah, Aw = _cache[key]
    # the reference resizes dim -2 first, then dim -1 (dims sorted by scale, stable)
    return torch.matmul(torch.matmul(Ah, x), Aw.t())
