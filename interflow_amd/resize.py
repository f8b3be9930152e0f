"""Antialiased bicubic down-scaling of the LR-consistency loss (`train.py:99-100`):
`resize(input, scale_factors=1/4, interp_method=cubic, antialiasing=True)` of the reference's `utils/resize_right.py`
(ResizeRight), for the arguments the training loop uses (last two dims, zero 'constant' padding, no by_convs).

ResizeRight computes, per resized dimension, for every output position a 16-tap field of view and normalised weights
(`resize_right.py:71-123`), then gathers and sums.  Here the fields of view and weights are built once per size with the
reference's formulas and float32 operation order (`resize_taps`); `ifl_resize_sep` (csrc/layout.hip) evaluates the two
dimensions in the reference's order in one kernel, and the backward pass is the same kernel on the transposed tables
(each input sample reaches at most 5 outputs at x1/4).  `resize_matrix` is the same map as a dense [out, in] matrix (tests,
and the source of the transposed tables).  GPU tensors only: like the rest of the package there is no CPU evaluation.
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


def _banded_tables(A: torch.Tensor):
    """Dense [out, in] matrix with banded rows -> (idx [out, T] int32, w [out, T] float32): per row the columns from its first
    to its last non-zero (T = the widest row), zero weights and clamped indices beyond a row's own band."""
    out_sz, in_sz = A.shape
    nz = A != 0
    cols = torch.arange(in_sz)
    first = torch.where(nz, cols, in_sz).min(dim=1).values.clamp(max=in_sz - 1)
    last = torch.where(nz, cols, -1).max(dim=1).values.clamp(min=0)
    T = int((last - first).max().item()) + 1
    idx = first[:, None] + torch.arange(T)
    w = torch.where(idx < in_sz, A.gather(1, idx.clamp(max=in_sz - 1)), torch.zeros(()))
    return idx.clamp(max=in_sz - 1).to(torch.int32).contiguous(), w.to(torch.float32).contiguous()


def resize_taps(in_sz: int, scale: float):
    """Forward and adjoint tap tables of one dimension (CPU tensors): ((idx, w) for out <- in, (idx, w) for in <- out)."""
    A = resize_matrix(in_sz, scale)
    return _banded_tables(A), _banded_tables(A.t().contiguous())


_cache = {}


def _tables(H, W, scale, device):
    key = (H, W, float(scale), str(device))
    if key not in _cache:
        if len(_cache) > 16:
            _cache.clear()
        fh, bh = resize_taps(H, scale)
        fw, bw = resize_taps(W, scale)
        _cache[key] = tuple(tuple(t.to(device) for t in pair) for pair in (fh, fw, bh, bw))
    return _cache[key]


def _launch(x3, Hout, Wout, th, tw):
    from . import hip
    N, Hin, Win = x3.shape
    out = torch.empty(N, Hout, Wout, dtype=torch.float32, device=x3.device)
    with torch.cuda.device(x3.device):
        for n0 in range(0, N, 65535):
            xs, os_ = x3[n0:n0 + 65535], out[n0:n0 + 65535]
            hip.check(hip.lib().ifl_resize_sep(hip._ptr(xs), hip._ptr(os_), xs.shape[0], Hin, Win, Hout, Wout, hip._ptr(th[0]), hip._ptr(th[1]),
                                               th[0].shape[1], hip._ptr(tw[0]), hip._ptr(tw[1]), tw[0].shape[1], hip._stream()),
                      "ifl_resize_sep")
    return out


class _ResizeSep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        from . import hip
        x = hip._dev_f32(x, "resize input")
        H, W = x.shape[-2:]
        fh, fw, bh, bw = _tables(H, W, scale, x.device)
        ctx.meta = (x.shape, bh, bw)
        out = _launch(x.reshape(-1, H, W), fh[0].shape[0], fw[0].shape[0], fh, fw)
        return out.reshape(*x.shape[:-2], fh[0].shape[0], fw[0].shape[0])

    @staticmethod
    def backward(ctx, g):
        shape, bh, bw = ctx.meta
        g = g.contiguous().float()
        dx = _launch(g.reshape(-1, g.shape[-2], g.shape[-1]), shape[-2], shape[-1], bh, bw)
        return dx.reshape(shape), None


def resize_last2(x: torch.Tensor, scale: float = 0.25) -> torch.Tensor:
    """resize(x, scale_factors=scale, interp_method=cubic, antialiasing=True) over the last two dims (dimension -2 first,
    then -1, as the reference), on the HIP kernel, differentiable."""
    return _ResizeSep.apply(x, float(scale))
