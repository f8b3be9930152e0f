"""Differentiable convolution on the HIP kernels (training path).

`conv2d_ring` is a `torch.autograd.Function` around the library's generic implicit-GEMM convolution
(`ifl_conv2d`, csrc/convg.hip): forward, data gradient (the same kernel with transposed, flipped weights) and weight
gradient (`ifl_conv2d_wgrad`).  It replaces, for gradient-enabled passes, the `F.conv2d` calls of the reference's
`flow.Conv2d` / `flow.Conv2dZeros` (flow.py:25-82) -- 99.8 % of the path's FLOPs.

Tensors are "ringed channels-last": [B, h+2, w+2, C] float32 with a zero one-pixel ring (the convs' 'same' padding) and C a
multiple of 32.  `to_ring` / `from_ring` convert from/to NCHW; both are differentiable torch ops (layout plumbing).
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from . import hip


def pad32(c: int) -> int:
    return (c + 31) // 32 * 32


def to_ring(x_nchw: torch.Tensor, cpad: int | None = None) -> torch.Tensor:
    """NCHW -> ringed channels-last with the channel count padded to `cpad` (default: next multiple of 32)."""
    B, C, H, W = x_nchw.shape
    cpad = pad32(C) if cpad is None else cpad
    return F.pad(x_nchw.permute(0, 2, 3, 1), (0, cpad - C, 1, 1, 1, 1)).contiguous()


def from_ring(x_ring: torch.Tensor, c: int) -> torch.Tensor:
    """ringed channels-last -> NCHW, keeping the first c channels."""
    return x_ring[:, 1:-1, 1:-1, :c].permute(0, 3, 1, 2)


def ring_of(interior: torch.Tensor) -> torch.Tensor:
    """[B, h, w, C] channels-last interior -> ringed tensor (zero ring)."""
    return F.pad(interior, (0, 0, 1, 1, 1, 1))


def pack_weight(wp: torch.Tensor) -> torch.Tensor:
    """[cout_pad, cin_pad, taps] -> fragment order of ifl_conv2d (see interflow_hip.h), on the tensor's device."""
    cout, cin, taps = wp.shape
    return wp.reshape(cout // 32, 32, cin // 32, 4, 2, 4, taps).permute(0, 2, 6, 3, 4, 1, 5).contiguous()


def _padded_weight(weight: torch.Tensor, splits, cout_pad: int) -> torch.Tensor:
    """weight [Cout, sum(real channels), kh, kw] -> [cout_pad, sum(padded channels), taps]; `splits` = [(real, padded)]
    per source, in the order the reference concatenates them (torch.cat((z1, ft), 1))."""
    cout = weight.shape[0]
    taps = weight.shape[2] * weight.shape[3]
    w = weight.reshape(cout, -1, taps)
    parts, o = [], 0
    for real, padded in splits:
        parts.append(F.pad(w[:, o:o + real], (0, 0, 0, padded - real)))
        o += real
    return F.pad(torch.cat(parts, 1), (0, 0, 0, 0, 0, cout_pad - cout))


def _packed(weight: torch.Tensor, splits, cout_pad: int, transposed: bool):
    """Fragment-order copy of `weight` for ifl_conv2d, cached ON the parameter object per version (an optimizer step or a
    load bumps `_version`; a cache keyed by address would outlive the tensor): forward weights, or -- `transposed` -- the
    spatially flipped transpose [cin_pad, cout_pad, taps] whose source ranges are the data-gradient convolutions' weights."""
    cache = getattr(weight, "_ifl_pack", None)
    if cache is None or cache[0] != (weight._version, weight.data_ptr()):
        cache = ((weight._version, weight.data_ptr()), {})
        try:
            weight._ifl_pack = cache
        except AttributeError:                    # not attribute-capable (should not happen for Parameters): no caching
            pass
    key = (tuple(splits), cout_pad, transposed)
    hit = cache[1].get(key)
    if hit is not None:
        return hit
    wpad = _padded_weight(weight.detach().float(), splits, cout_pad)
    if not transposed:
        out = pack_weight(wpad)
    else:
        wt = wpad.flip(2).transpose(0, 1).contiguous()            # Wt[ci][co][tap] = W[co][ci][taps-1-tap]
        out, o = [], 0
        for _, padded in splits:
            out.append(pack_weight(wt[o:o + padded].contiguous()))
            o += padded
    cache[1][key] = out
    return out


def _launch_conv(srcs, wpacked, bias, out, cout, taps, B, h, w):
    d = hip.IflConvDesc()
    d.nsrc = len(srcs)
    for i, s in enumerate(srcs):
        d.src[i] = s.data_ptr()
        d.src_stride[i] = s.shape[3]
        d.src_off[i] = 0
        d.src_ch[i] = s.shape[3]
    d.w = wpacked.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.out = out.data_ptr()
    d.out_stride, d.out_off, d.cout, d.taps, d.act = out.shape[3], 0, cout, taps, 0
    d.alpha = d.beta = 1.0
    with torch.cuda.device(out.device):
        hip.check(hip.lib().ifl_conv2d(ctypes.byref(d), B, h, w, hip._stream()), "ifl_conv2d")


class _ConvRing(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, bias, real_channels, *srcs):
        srcs = [hip._dev_f32(s, "conv source") for s in srcs]
        B, hp, wp_, _ = srcs[0].shape
        h, w = hp - 2, wp_ - 2
        cout = weight.shape[0]
        cout_pad = pad32(cout)
        taps = weight.shape[2] * weight.shape[3]
        assert taps in (1, 9) and len(srcs) in (1, 2) and len(real_channels) == len(srcs)
        for s in srcs:
            assert s.shape[:3] == srcs[0].shape[:3] and s.shape[3] % 32 == 0, tuple(s.shape)
        splits = [(r, s.shape[3]) for r, s in zip(real_channels, srcs)]
        assert sum(r for r, _ in splits) == weight.shape[1], (splits, tuple(weight.shape))
        bpad = F.pad(bias.detach().float(), (0, cout_pad - cout)).contiguous() if bias is not None else None
        out = torch.zeros(B, hp, wp_, cout_pad, dtype=torch.float32, device=srcs[0].device)
        _launch_conv(srcs, _packed(weight, splits, cout_pad, False), bpad, out, cout_pad, taps, B, h, w)
        ctx.save_for_backward(weight, *srcs)
        ctx.meta = (splits, cout, cout_pad, taps, B, h, w, bias is not None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        weight, *srcs = ctx.saved_tensors
        splits, cout, cout_pad, taps, B, h, w, has_bias = ctx.meta
        L = hip.lib()
        # a fresh contiguous [B, h+2, w+2, cout_pad] buffer (the kernels index it as such, whatever strides gout has):
        # zero ring and zero padded channels, whatever arrives
        g = torch.zeros(gout.shape, dtype=torch.float32, device=gout.device)
        assert g.is_contiguous()
        g[:, 1:-1, 1:-1, :cout] = gout[:, 1:-1, 1:-1, :cout]
        cin_pad = sum(p for _, p in splits)
        grads = [None, None, None]
        if ctx.needs_input_grad[0]:
            dW = torch.zeros(cout_pad, cin_pad, taps, dtype=torch.float32, device=g.device)
            c0 = 0
            with torch.cuda.device(g.device):
                for s in srcs:
                    hip.check(L.ifl_conv2d_wgrad(hip._ptr(s), s.shape[3], 0, s.shape[3], hip._ptr(g), cout_pad, 0, cout_pad,
                                                 hip._ptr(dW), cin_pad, c0 // 32, taps, B, h, w, hip._stream()),
                              "ifl_conv2d_wgrad")
                    c0 += s.shape[3]
            parts, o = [], 0
            for real, padded in splits:
                parts.append(dW[:cout, o:o + real])
                o += padded
            grads[0] = torch.cat(parts, 1).reshape(weight.shape).to(weight.dtype)
        if has_bias and ctx.needs_input_grad[1]:
            grads[1] = g[:, 1:-1, 1:-1, :cout].sum(dim=(0, 1, 2))
        need_src = [ctx.needs_input_grad[3 + i] for i in range(len(srcs))]
        if any(need_src):
            # data gradient = the same convolution with W^T, spatially flipped:  Wt[ci][co][tap] = W[co][ci][taps-1-tap]
            wts = _packed(weight, splits, cout_pad, True)
            for i, s in enumerate(srcs):
                padded = s.shape[3]
                if need_src[i]:
                    gi = torch.zeros(s.shape, dtype=torch.float32, device=s.device)
                    _launch_conv([g], wts[i], None, gi, padded, taps, B, h, w)
                    grads.append(gi)
                else:
                    grads.append(None)
        else:
            grads += [None] * len(srcs)
        return tuple(grads)


def conv2d_ring(srcs, real_channels, weight, bias=None):
    """Stride-1 'same' convolution (3x3 or 1x1) of the channel-wise concatenation of `srcs` (ringed channels-last tensors
    whose channel counts are multiples of 32; the first `real_channels[i]` channels of source i are the reference's
    channels, the rest zero) with `weight` [Cout, sum(real), k, k] (+ bias).  Returns a ringed tensor with pad32(Cout)
    channels (zero ring, zero padded channels)."""
    return _ConvRing.apply(weight, bias, tuple(real_channels), *srcs)


# ------------------------------------------------------------------------------------------------ fused per-pixel operators
class _AnRelu(torch.autograd.Function):
    """y = relu((x + bias) * exp(logs)) on a ringed 64-channel tensor (ifl_an_relu_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x, bias, logs):
        x = hip._dev_f32(x, "an_relu input")
        B, hp, wp_, C = x.shape
        b, l = bias.detach().reshape(-1).float().contiguous(), logs.detach().reshape(-1).float().contiguous()
        y = torch.zeros(x.shape, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            hip.check(hip.lib().ifl_an_relu_fwd(hip._ptr(x), hip._ptr(y), hip._ptr(b), hip._ptr(l), B, hp - 2, wp_ - 2, C, hip._stream()),
                      "ifl_an_relu_fwd")
        ctx.save_for_backward(x, b, l)
        ctx.shapes = (bias.shape, logs.shape)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, b, l = ctx.saved_tensors
        B, hp, wp_, C = x.shape
        L = hip.lib()
        gy = gy.contiguous()
        dx = torch.zeros(x.shape, dtype=torch.float32, device=x.device)
        nblk = L.ifl_an_relu_blocks(B, hp - 2, wp_ - 2)
        part = torch.empty(nblk, 2 * C, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            hip.check(L.ifl_an_relu_bwd(hip._ptr(x), hip._ptr(gy), hip._ptr(dx), hip._ptr(b), hip._ptr(l), hip._ptr(part),
                                        B, hp - 2, wp_ - 2, C, hip._stream()), "ifl_an_relu_bwd")
        s = part.sum(dim=0)
        return dx, s[:C].reshape(ctx.shapes[0]), s[C:].reshape(ctx.shapes[1])


def an_relu_ring(x_ring, bias, logs):
    """ReLU(ActNorm(x)) of flow.Conv2d + the following nn.ReLU, ringed 64-channel tensors in and out."""
    return _AnRelu.apply(x_ring, bias, logs)


class _Coupling(torch.autograd.Function):
    """(z', sum of +-log scale per sample) of one affine coupling (ifl_coupling_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, z, h_ring, logs3, eps, branch, c0, n):
        z = hip._dev_f32(z, "coupling z")
        h_ring = hip._dev_f32(h_ring, "coupling h")
        B, h, w, C = z.shape
        assert h_ring.shape[:3] == (B, h + 2, w + 2), (tuple(z.shape), tuple(h_ring.shape))
        l3 = logs3.detach().reshape(-1).float().contiguous()
        assert l3.numel() == 2 * n
        L = hip.lib()
        nblk = L.ifl_coupling_blocks(h, w)
        zo = torch.empty_like(z)
        ldp = torch.empty(B, nblk, dtype=torch.float32, device=z.device)
        with torch.cuda.device(z.device):
            hip.check(L.ifl_coupling_fwd(hip._ptr(z), hip._ptr(zo), hip._ptr(h_ring), h_ring.shape[3], hip._ptr(l3), float(eps), branch,
                                         c0, n, hip._ptr(ldp), B, h, w, C, hip._stream()), "ifl_coupling_fwd")
        ctx.save_for_backward(z, h_ring, l3)
        ctx.meta = (float(eps), branch, c0, n, logs3.shape)
        return zo, ldp.sum(dim=1)

    @staticmethod
    @once_differentiable
    def backward(ctx, gz, gld):
        z, h_ring, l3 = ctx.saved_tensors
        eps, branch, c0, n, lshape = ctx.meta
        B, h, w, C = z.shape
        L = hip.lib()
        nblk = L.ifl_coupling_blocks(h, w)
        gz = gz.contiguous()
        gld = gld.contiguous().float() if gld is not None else None
        dz = torch.empty_like(z)
        dh = torch.zeros(h_ring.shape, dtype=torch.float32, device=z.device)
        lp = torch.empty(B * nblk, 2 * n, dtype=torch.float32, device=z.device)
        with torch.cuda.device(z.device):
            hip.check(L.ifl_coupling_bwd(hip._ptr(z), hip._ptr(gz), hip._ptr(gld), hip._ptr(h_ring), h_ring.shape[3], hip._ptr(l3), eps,
                                         branch, c0, n, hip._ptr(dz), hip._ptr(dh), hip._ptr(lp), B, h, w, C, hip._stream()),
                      "ifl_coupling_bwd")
        return dz, dh, lp.sum(dim=0).reshape(lshape), None, None, None, None


def coupling_apply(z_cl, h_ring, logs3, eps, branch, c0, n):
    """Affine coupling on channels [c0, c0+n) of the contiguous channels-last z with the raw Conv2dZeros output `h_ring`
    (ringed); returns (z', per-sample signed sum of log scale)."""
    return _Coupling.apply(z_cl.contiguous(), h_ring, logs3, eps, branch, c0, n)
