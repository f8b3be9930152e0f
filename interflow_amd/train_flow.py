"""Gradient-enabled pass of the flow (training path, first version).

When autograd has to flow through `FlowUpsamplerNet` (training: `train.py:73,85-86,94`), the stack is evaluated layer by
layer as a differentiable graph instead of by the fused inference kernels:

* every convolution of the coupling networks -- 99.8 % of the FLOPs -- runs on the HIP implicit-GEMM kernels through
  `train_ops.conv2d_ring` (forward, data gradient and weight gradient all on the matrix cores);
* the per-pixel operators of the coupling networks -- ActNorm + ReLU, and exp(3 logs) / sigmoid / the affine coupling / its sum
  of log scale -- are fused HIP kernels with hand-written backward kernels (`train_ops.an_relu_ring`, `coupling_apply`:
  csrc/train_glue.hip); what is left to ATen is the flow-level ActNorm and the C x C channel mix of the invertible 1x1 conv
  (a handful of launches per step), whose LU parameters go through torch autograd.

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


FUSED_LEVELS = True      # encode on the level drivers of train_fused.py when they apply (tests switch it off to compare)


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
def _conv_actnorm_relu(conv, srcs, reals):
    """flow.Conv2d (no bias) + its ActNorm + the following ReLU, ringed in and out.  Data-dependent initialisation of the
    ActNorm (flow.py:57-63) from the convolution's interior on the first training pass."""
    p = T.conv2d_ring(srcs, reals, conv.weight)
    an = conv.actnorm
    if not an.inited:
        actnorm_init(an, p[:, 1:-1, 1:-1, :].detach())
    assert p.shape[-1] == an.num_features, (tuple(p.shape), an.num_features)
    return T.an_relu_ring(p, an.bias, an.logs)


def coupling_nn(seq, srcs, reals):
    """FlowAffineCouplingsAblation.py:248-256: Conv2d 3x3 + ActNorm, ReLU, Conv2d 1x1 + ActNorm, ReLU, Conv2dZeros 3x3.
    `srcs`: ringed channels-last inputs whose concatenation is the network's input.  Returns the RAW output of the last
    convolution (conv + bias, before the exp(3 logs) factor of flow.py:80-82, which the coupling kernel applies) as a ringed
    tensor with pad32(Cout) channels."""
    c1, c2, c3 = seq[0], seq[2], seq[4]
    a1 = _conv_actnorm_relu(c1, srcs, reals)                                                    # flow.py:25-64
    a2 = _conv_actnorm_relu(c2, [a1], [a1.shape[-1]])
    return T.conv2d_ring([a2], [a2.shape[-1]], c3.weight, c3.bias)


def coupling(aff, z_cl, logdet, reverse, ft_ring):
    """CondAffineSeparatedAndCond.forward, FlowAffineCouplingsAblation.py:159-208 (multReverse flips the flag, :164-165)."""
    if aff.mult_reverse:
        reverse = not reverse
    C = aff.in_channels
    assert z_cl.shape[-1] == C, (z_cl.shape[-1], C)
    ch = aff.channels_for_nn
    eps = aff.affine_eps
    assert aff.fAffine[4].logscale_factor == 3 and aff.fFeatures[4].logscale_factor == 3

    def f_affine(z):
        z1r = F.pad(z[..., :ch], (0, T.pad32(ch) - ch, 1, 1, 1, 1))
        return coupling_nn(aff.fAffine, [z1r, ft_ring], [ch, N_RRDB_CH])

    def f_features():
        return coupling_nn(aff.fFeatures, [ft_ring], [N_RRDB_CH])

    if not reverse:                                                                              # :167-186
        z, ld = T.coupling_apply(z_cl, f_features(), aff.fFeatures[4].logs, eps, 0, 0, C)
        logdet = logdet + ld
        z, ld = T.coupling_apply(z, f_affine(z), aff.fAffine[4].logs, eps, 0, ch, C - ch)
        return z, logdet + ld
    z, ld = T.coupling_apply(z_cl, f_affine(z_cl), aff.fAffine[4].logs, eps, 1, ch, C - ch)      # :187-207
    logdet = logdet + ld
    z, ld = T.coupling_apply(z, f_features(), aff.fFeatures[4].logs, eps, 1, 0, C)
    return z, logdet + ld


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


def _split(layer, z_cl, ld, reverse, epses, y_onehot, eps_std):
    """A Split2d layer inside the stack (FlowUpsamplerNet.py:243-245,263-269 / :292-295,306-311); the operator works on NCHW."""
    zn = z_cl.permute(0, 3, 1, 2).contiguous()
    if y_onehot is None:
        raise ValueError("Split2d layers need y_onehot: one class label (0 / 1) per sample (Split.py:56-60)")
    if not reverse:
        zn, ld, eps = layer(zn, ld, reverse=False, y_onehot=y_onehot)
        if isinstance(epses, list):
            epses.append(eps)
    else:
        zn, ld = layer(zn, logdet=ld, reverse=True, eps=epses.pop() if isinstance(epses, list) else None, eps_std=eps_std,
                       y_onehot=y_onehot)
    return zn.permute(0, 2, 3, 1), ld


def run(fu, src, rrdbResults, logdet, test_input_size, decode, epses=None, y_onehot=None, eps_std=None):
    """FlowUpsamplerNet.encode / decode (FlowUpsamplerNet.py:217-304) layer by layer: a differentiable graph when autograd
    records, and the evaluation of every configuration the fused stack driver is not built for (more than two levels,
    Split2d layers; `epses`, `y_onehot`, `eps_std` belong to those).  src is NCHW."""
    from .flow_modules import _as_logdet
    src = hip._dev_f32(src, "flow input") if not src.requires_grad else src
    if not src.is_cuda:
        raise hip.InterflowHipError(f"flow input is on {src.device}; the HIP path needs a GPU tensor (no CPU fallback)")
    B = src.shape[0]
    H = src.shape[2] << sum(1 for k, _ in fu._kinds if k == "squeeze") if decode else src.shape[2]
    if not decode:
        assert src.shape[1] == 3, src.shape
        n_sq = sum(1 for k, _ in fu._kinds if k == "squeeze")
        assert src.shape[2] % (1 << n_sq) == 0 and src.shape[3] % (1 << n_sq) == 0, "{}".format((src.shape[2], src.shape[3], 2))
    feats = fu._level_features(rrdbResults, H, test_input_size, decode)
    rings = {}

    def ft_ring(level):                       # one ringed copy of a level's features serves its 16 steps (autograd sums)
        if level not in rings:
            ft = feats[level]
            assert ft.shape[1] == N_RRDB_CH, tuple(ft.shape)
            rings[level] = T.to_ring(ft.float())
        return rings[level]

    ld = _as_logdet(logdet, B, src.device)
    if not decode and FUSED_LEVELS:
        from . import train_fused
        if train_fused.usable(fu, feats):
            out = train_fused.run_encode(fu, src, feats, ld)
            if out is not None:
                return out
    if decode and FUSED_LEVELS:
        from . import train_fused
        if train_fused.usable(fu, feats) and src.shape[1] == fu.C:
            out = train_fused.run_decode(fu, src, feats, ld)
            if out is not None:
                return out
    z = src.float().permute(0, 2, 3, 1)
    kinds = [k for k, _ in fu._kinds]
    if not decode:
        level = 0
        for layer, kind in zip(fu.layers, kinds):
            if kind == "squeeze":
                z = squeeze_cl(z)
                level += 1
            elif kind == "split":
                z, ld = _split(layer, z, ld, False, epses, y_onehot, eps_std)
            else:
                z, ld = flow_step(layer, z, ld, False, ft_ring(level) if kind == "stepc" else None)
    else:
        assert src.shape[1] == fu.C, (src.shape[1], fu.C)
        level = sum(1 for k in kinds if k == "squeeze")
        for layer, kind in zip(reversed(list(fu.layers)), reversed(kinds)):
            if kind == "squeeze":
                z = unsqueeze_cl(z)
                level -= 1
            elif kind == "split":
                z, ld = _split(layer, z, ld, True, epses, y_onehot, eps_std)
            else:
                z, ld = flow_step(layer, z, ld, True, ft_ring(level) if kind == "stepc" else None)
    return z.permute(0, 3, 1, 2).contiguous(), ld


# ------------------------------------------------------------------------------------------------ prior (flow.py:138-161)
LOG2PI = 1.8378770664093453


class _InvLogdet(torch.autograd.Function):
    """(cov^-1, log|det cov|) of the 48x48 class covariance with a hand-written backward.  torch.inverse and the backward
    pass of torch.slogdet read an error flag on the host: a device synchronisation per class in the forward pass and another
    at the very start of the backward pass, where the host then waits for the whole forward to drain before it can issue
    anything else.  inv_ex / slogdet's forward do not; the gradients are  d cov = -inv^T d_inv inv^T + d_logdet inv^T.
    A singular covariance gives inf/nan (the reference raises); the training step's loss guard then drops the update."""

    @staticmethod
    def forward(ctx, cov):
        inv = torch.linalg.inv_ex(cov)[0]
        ctx.save_for_backward(inv)
        return inv, torch.linalg.slogdet(cov)[1]

    @staticmethod
    def backward(ctx, g_inv, g_ld):
        inv, = ctx.saved_tensors
        it = inv.transpose(-1, -2)
        g = torch.zeros_like(inv)
        if g_inv is not None:
            g = g - torch.matmul(it, torch.matmul(g_inv, it))
        if g_ld is not None:
            g = g + g_ld[..., None, None] * it
        return g


def gaussian_logp_classes(means, covs, x):
    """Gaussian.logp (flow.py:138-161) of x [n, C, h, w] under K class priors at once: means [K, C], covs [K, C, C] ->
    [n, K].  The same expressions as `gaussian_logp`, batched over the classes (one batched inverse / log-det and one
    batched quadratic form instead of K of each: the loss sits between the forward and the backward chain, nothing overlaps it)."""
    x_ = x.permute(0, 2, 3, 1)[None]                                              # [1, n, h, w, C]
    x_c = x_ - means[:, None, None, None, :]                                        # [K, n, h, w, C]
    k = x.shape[1]
    cov_inv, logdet = _InvLogdet.apply(covs)                                        # [K, C, C], [K]
    quad = (torch.matmul(x_c, cov_inv[:, None, None]) * x_c).sum(dim=4)             # [K, n, h, w]
    lik = -0.5 * (logdet[:, None, None, None] + k * LOG2PI + quad)
    return lik.sum(dim=(2, 3)).t()


def gaussian_logp(mean, cov, x):
    """Gaussian.logp / likelihood, flow.py:138-161, differentiable (x is NCHW)."""
    if mean is None and cov is None:
        return (-0.5 * (x.shape[1] * LOG2PI + (x ** 2).sum(dim=1))).sum(dim=(1, 2))             # standard normal
    x_ = x.permute(0, 2, 3, 1)
    x_c = x_ - mean
    k = x.shape[1]
    cov_inv, logdet = _InvLogdet.apply(cov)    # flow.py:146-149's torch.inverse / torch.slogdet without their host round trips
    lik = -0.5 * (logdet + k * LOG2PI + (torch.matmul(x_c, cov_inv) * x_c).sum(dim=3))
    return lik.sum(dim=(1, 2))
