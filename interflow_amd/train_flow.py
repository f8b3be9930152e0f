"""Gradient-enabled pass of the flow (training path, first version).

When autograd has to flow through `FlowUpsamplerNet` (training: `train.py:73,85-86,94`), the stack is evaluated layer by
layer as a differentiable graph instead of by the fused inference kernels:

* every convolution of the coupling networks -- 99.8 % of the FLOPs -- runs on the HIP implicit-GEMM kernels through
  `train_ops.conv2d_ring` (forward, data gradient and weight gradient all on the matrix cores);
* the per-pixel glue (ActNorm, ReLU, sigmoid, the affine coupling, the C x C channel mix of the invertible 1x1 conv and the
  log-det sums) are ATen ops on channels-last GPU tensors for now -- they are bandwidth-bound, a fraction of a percent of
  the arithmetic, and are due to be replaced by fused backward kernels.

The arithmetic follows the reference line by line (citations below); `tests/test_gpu_train.py` checks values and gradients
against CPU autograd of the oracle.  Data-dependent ActNorm initialisation (FlowActNorms.py:29-43) happens here, on the first
gradient-enabled pass of a module in training mode whose bias is still all zero.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import hip
from . import train_ops as T
from .synth import N_RRDB_CH


# ------------------------------------------------------------------------------------------------ ActNorm
def actnorm_init(an, x_cl):
    """FlowActNorms.py:29-43 on a channels-last tensor [B,h,w,C]."""
    if not an.training:
        return
    if bool((an.bias != 0).any()):
        an.inited = True
        return
    with torch.no_grad():
        bias = x_cl.mean(dim=(0, 1, 2)) * -1.0
        var = ((x_cl + bias) ** 2).mean(dim=(0, 1, 2))
        logs = torch.log(an.scale / (torch.sqrt(var) + 1e-6))
        an.bias.data.copy_(bias.reshape(an.bias.shape))
        an.logs.data.copy_(logs.reshape(an.logs.shape))
        an.inited = True


def actnorm(an, x_cl, logdet=None, reverse=False):
    """FlowActNorms.py:45-97"""
    if not an.inited:
        actnorm_init(an, x_cl)
    assert x_cl.shape[-1] == an.num_features, (tuple(x_cl.shape), an.num_features)          # :104-109
    b, logs = an.bias.reshape(-1), an.logs.reshape(-1)
    if not reverse:
        x = (x_cl + b) * torch.exp(logs)
    else:
        x = x_cl * torch.exp(-logs) - b
    if logdet is not None:
        d = logs.sum() * (x_cl.shape[1] * x_cl.shape[2])
        logdet = logdet - d if reverse else logdet + d
    return x, logdet


# ------------------------------------------------------------------------------------------------ invertible 1x1 conv
def invconv_weight(ic, reverse):
    """Permutations.py:76-101 -> (W [C,C], sum of log|det| per pixel)."""
    if not ic.LU:
        dl = torch.slogdet(ic.weight)[1]
        w = ic.weight if not reverse else torch.inverse(ic.weight.double()).float()
        return w, dl
    C = ic.w_shape[0]
    dev = ic.l.device
    l_mask = torch.tril(torch.ones(C, C, device=dev), -1)
    eye = torch.eye(C, device=dev)
    l = ic.l * l_mask + eye
    u = ic.u * l_mask.transpose(0, 1).contiguous() + torch.diag(ic.sign_s * torch.exp(ic.log_s))
    dl = ic.log_s.sum()
    if not reverse:
        w = torch.matmul(ic.p, torch.matmul(l, u))
    else:
        l = torch.inverse(l.double()).float()
        u = torch.inverse(u.double()).float()
        w = torch.matmul(u, torch.matmul(l, ic.p.t()))      # p is a permutation matrix: p.inverse() == p^T exactly
        # (torch.linalg.inv of a permutation matrix fails on this ROCm build: its LU runs without pivoting)
    return w, dl


def invconv(ic, x_cl, logdet=None, reverse=False):
    """Permutations.py:103-123: z = conv1x1(x, W) = x @ W^T on channels-last."""
    w, dl = invconv_weight(ic, reverse)
    z = torch.matmul(x_cl, w.t())
    if logdet is not None:
        d = dl * (x_cl.shape[1] * x_cl.shape[2])
        logdet = logdet - d if reverse else logdet + d
    return z, logdet


# ------------------------------------------------------------------------------------------------ coupling networks
def coupling_nn(seq, srcs, reals):
    """FlowAffineCouplingsAblation.py:248-256: Conv2d 3x3 + ActNorm, ReLU, Conv2d 1x1 + ActNorm, ReLU, Conv2dZeros 3x3.
    `srcs`: ringed channels-last inputs whose concatenation is the network's input; returns h [B,h,w,Cout] channels-last."""
    c1, c2, c3 = seq[0], seq[2], seq[4]
    p1 = T.conv2d_ring(srcs, reals, c1.weight)[:, 1:-1, 1:-1, :]                               # flow.py:25-64 (no bias)
    a1, _ = actnorm(c1.actnorm, p1)
    a1 = F.relu(a1)
    p2 = T.conv2d_ring([T.ring_of(a1)], [a1.shape[-1]], c2.weight)[:, 1:-1, 1:-1, :]
    a2, _ = actnorm(c2.actnorm, p2)
    a2 = F.relu(a2)
    cout = c3.weight.shape[0]
    h = T.conv2d_ring([T.ring_of(a2)], [a2.shape[-1]], c3.weight, c3.bias)[:, 1:-1, 1:-1, :cout]
    return h * torch.exp(c3.logs.reshape(-1) * c3.logscale_factor)                               # flow.py:80-82


def _scale_shift(h, eps):
    shift, scale = h[..., 0::2], h[..., 1::2]                                                    # thops.py:54-55 "cross"
    return torch.sigmoid(scale + 2.) + eps, shift                                                # Ablation.py:229


def _sum_log(scale):
    return torch.log(scale).sum(dim=(1, 2, 3))                                                   # Ablation.py:216-217


def coupling(aff, z_cl, logdet, reverse, ft_ring):
    """CondAffineSeparatedAndCond.forward, FlowAffineCouplingsAblation.py:159-208 (multReverse flips the flag, :164-165)."""
    if aff.mult_reverse:
        reverse = not reverse
    C = aff.in_channels
    assert z_cl.shape[-1] == C, (z_cl.shape[-1], C)
    ch = aff.channels_for_nn
    eps = aff.affine_eps

    def f_affine(z1):
        z1r = T.ring_of(F.pad(z1, (0, T.pad32(ch) - ch)))
        return _scale_shift(coupling_nn(aff.fAffine, [z1r, ft_ring], [ch, N_RRDB_CH]), eps)

    if not reverse:                                                                              # :167-186
        scale_ft, shift_ft = _scale_shift(coupling_nn(aff.fFeatures, [ft_ring], [N_RRDB_CH]), eps)
        z = (z_cl + shift_ft) * scale_ft
        logdet = logdet + _sum_log(scale_ft)
        z1, z2 = z[..., :ch], z[..., ch:]
        scale, shift = f_affine(z1)
        z2 = (z2 + shift) * scale
        logdet = logdet + _sum_log(scale)
        return torch.cat([z1, z2], dim=-1), logdet
    z1, z2 = z_cl[..., :ch], z_cl[..., ch:]                                                      # :187-207
    scale, shift = f_affine(z1)
    z2 = z2 / scale - shift
    z = torch.cat([z1, z2], dim=-1)
    logdet = logdet - _sum_log(scale)
    scale_ft, shift_ft = _scale_shift(coupling_nn(aff.fFeatures, [ft_ring], [N_RRDB_CH]), eps)
    z = z / scale_ft - shift_ft
    logdet = logdet - _sum_log(scale_ft)
    return z, logdet


# ------------------------------------------------------------------------------------------------ steps and the stack
def flow_step(step, z_cl, logdet, reverse, ft_ring):
    """FlowStep.normal_flow / reverse_flow, FlowStep.py:114-196."""
    has = hasattr(step, "affine")
    if not reverse:
        z, logdet = actnorm(step.actnorm, z_cl, logdet, False)
        z, logdet = invconv(step.invconv, z, logdet, False)
        if has:
            z, logdet = coupling(step.affine, z, logdet, False, ft_ring)
        return z, logdet
    z = z_cl
    if has:
        z, logdet = coupling(step.affine, z, logdet, True, ft_ring)
    z, logdet = invconv(step.invconv, z, logdet, True)
    z, logdet = actnorm(step.actnorm, z, logdet, True)
    return z, logdet


def squeeze_cl(x_cl):
    """flow.py:163-176 on channels-last: out[b,h,w,4c+2i+j] = in[b,2h+i,2w+j,c]."""
    B, H, W, C = x_cl.shape
    assert H % 2 == 0 and W % 2 == 0, "{}".format((H, W, 2))
    return x_cl.reshape(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 5, 2, 4).reshape(B, H // 2, W // 2, C * 4)


def unsqueeze_cl(x_cl):
    """flow.py:179-193"""
    B, H, W, C = x_cl.shape
    assert C % 4 == 0, "{}".format(C)
    return x_cl.reshape(B, H, W, C // 4, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(B, H * 2, W * 2, C // 4)


def run(fu, src, rrdbResults, logdet, test_input_size, decode):
    """FlowUpsamplerNet.encode / decode (FlowUpsamplerNet.py:217-304) as a differentiable graph.  src is NCHW."""
    from .flow_modules import _as_logdet
    src = hip._dev_f32(src, "flow input") if not src.requires_grad else src
    if not src.is_cuda:
        raise hip.InterflowHipError(f"flow input is on {src.device}; the HIP path needs a GPU tensor (no CPU fallback)")
    B = src.shape[0]
    H = src.shape[2] * 4 if decode else src.shape[2]
    if not decode:
        assert src.shape[1] == 3, src.shape
        assert src.shape[2] % 4 == 0 and src.shape[3] % 4 == 0, "{}".format((src.shape[2], src.shape[3], 2))
    feats = fu._level_features(rrdbResults, H, test_input_size, decode)
    rings = {}

    def ft_ring(level):                       # one ringed copy of a level's features serves its 16 steps (autograd sums)
        if level not in rings:
            ft = feats[level]
            assert ft.shape[1] == N_RRDB_CH, tuple(ft.shape)
            rings[level] = T.to_ring(ft.float())
        return rings[level]

    ld = _as_logdet(logdet, B, src.device)
    z = src.float().permute(0, 2, 3, 1)
    kinds = [k for k, _ in fu._kinds]
    if not decode:
        level = 0
        for layer, kind in zip(fu.layers, kinds):
            if kind == "squeeze":
                z = squeeze_cl(z)
                level += 1
            else:
                z, ld = flow_step(layer, z, ld, False, ft_ring(level) if kind == "stepc" else None)
    else:
        assert src.shape[1] == fu.C, (src.shape[1], fu.C)
        level = sum(1 for k in kinds if k == "squeeze")
        for layer, kind in zip(reversed(list(fu.layers)), reversed(kinds)):
            if kind == "squeeze":
                z = unsqueeze_cl(z)
                level -= 1
            else:
                z, ld = flow_step(layer, z, ld, True, ft_ring(level) if kind == "stepc" else None)
    return z.permute(0, 3, 1, 2).contiguous(), ld


# ------------------------------------------------------------------------------------------------ prior (flow.py:138-161)
LOG2PI = 1.8378770664093453


def gaussian_logp(mean, cov, x):
    """Gaussian.logp / likelihood, flow.py:138-161, differentiable (x is NCHW)."""
    if mean is None and cov is None:
        return (-0.5 * (x.shape[1] * LOG2PI + (x ** 2).sum(dim=1))).sum(dim=(1, 2))             # standard normal
    x_ = x.permute(0, 2, 3, 1)
    x_c = x_ - mean
    k = x.shape[1]
    cov_inv = torch.inverse(cov)
    logdet = torch.slogdet(cov)[1]
    lik = -0.5 * (logdet + k * LOG2PI + (torch.matmul(x_c, cov_inv) * x_c).sum(dim=3))
    return lik.sum(dim=(1, 2))
