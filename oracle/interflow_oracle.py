"""CPU oracle for the InterFlow conditional-flow forward/inverse path.

TEST INFRASTRUCTURE ONLY.  This is a CPU restatement, in plain PyTorch CPU ops, of the
reference algorithm (dongjinkim9/InterFlow, `interflow/models/...`), written as pure functions
over a state dict.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it; the product package `interflow_amd/` never does.

Pinning: the reference has no tests or golden vectors.  The oracle is pinned
against the reference itself, imported on CPU in the authoring container by
`tests/golden/make_golden.py`; the outputs are committed as fixtures under `tests/golden/` and
`tests/test_oracle_golden.py` replays them against this file.

Every function cites the reference file:line it restates (paths relative to
`/root/reference/interflow/`).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

LOG2PI = float(np.log(2 * np.pi))   # models/modules/flow.py:139


# ----------------------------------------------------------------------------- thops
def thops_sum(t, dims):
    """models/modules/thops.py:15-28 -- reduce the dims one at a time, ascending."""
    dims = sorted(dims)
    for d in dims:
        t = t.sum(dim=d, keepdim=True)
    for i, d in enumerate(dims):
        t = t.squeeze(d - i)
    return t


def pixels(t):
    """models/modules/thops.py:62-63"""
    return int(t.size(2) * t.size(3))


def split_feature(t, kind="split"):
    """models/modules/thops.py:47-55"""
    C = t.size(1)
    if kind == "split":
        return t[:, :C // 2], t[:, C // 2:]
    return t[:, 0::2], t[:, 1::2]


# ----------------------------------------------------------------------------- squeeze
def squeeze2d(x, factor=2):
    """models/modules/flow.py:163-176: out[b, c*f*f + i*f + j, h, w] = in[b, c, h*f+i, w*f+j]"""
    if factor == 1:
        return x
    B, C, H, W = x.shape
    assert H % factor == 0 and W % factor == 0, "{}".format((H, W, factor))
    x = x.view(B, C, H // factor, factor, W // factor, factor)
    x = x.permute(0, 1, 3, 5, 2, 4).contiguous()
    return x.view(B, C * factor * factor, H // factor, W // factor)


def unsqueeze2d(x, factor=2):
    """models/modules/flow.py:179-193"""
    if factor == 1:
        return x
    B, C, H, W = x.shape
    f2 = factor * factor
    assert C % f2 == 0, "{}".format(C)
    x = x.view(B, C // f2, factor, factor, H, W)
    x = x.permute(0, 1, 4, 2, 5, 3).contiguous()
    return x.view(B, C // f2, H * factor, W * factor)


# ----------------------------------------------------------------------------- actnorm
def actnorm(x, bias, logs, logdet=None, reverse=False):
    """models/modules/FlowActNorms.py:45-97 (eval mode, already initialised).
    fwd: (x + b) * exp(logs); rev: x * exp(-logs) - b; dlogdet = sum(logs) * H*W."""
    assert x.dim() == 4 and x.size(1) == bias.size(1)      # FlowActNorms.py:104-109
    if _TRAIN_INIT[0] and not bool((bias != 0).any()):     # FlowActNorms.py:29-35,47-48: training mode, first call
        b, l = actnorm_init(x)
        bias.copy_(b)                                      # the reference writes its parameters in place (:41-42)
        logs.copy_(l)
    if not reverse:
        x = x + bias
        x = x * torch.exp(logs)
    else:
        x = x * torch.exp(-logs)
    if logdet is not None:
        d = torch.sum(logs) * pixels(x)
        if reverse:
            d = d * -1
        logdet = logdet + d
    if reverse:
        x = x - bias
    return x, logdet


_TRAIN_INIT = [False]


class training_init:
    """`with training_init():` -- evaluate like the reference in train() mode on uninitialised ActNorms: every ActNorm
    (FlowActNorms.py:29-43; also the one inside each flow.Conv2d, flow.py:57-63) whose bias is still all zero takes its
    bias / logs from the statistics of the tensor it sees, writing them into the state dict's tensors in place."""

    def __enter__(self):
        self.prev, _TRAIN_INIT[0] = _TRAIN_INIT[0], True

    def __exit__(self, *a):
        _TRAIN_INIT[0] = self.prev


def actnorm_init(x, scale=1.0):
    """models/modules/FlowActNorms.py:37-43 data-dependent init -> (bias, logs)."""
    def mean_dims(t):
        for d in (0, 2, 3):
            t = t.mean(dim=d, keepdim=True)
        return t
    bias = mean_dims(x.clone()) * -1.0
    var = mean_dims((x.clone() + bias) ** 2)
    logs = torch.log(scale / (torch.sqrt(var) + 1e-6))
    return bias, logs


# ----------------------------------------------------------------------------- invertible 1x1 conv
def invconv_weight(p, l, u, log_s, sign_s, reverse):
    """models/modules/Permutations.py:88-101 (LU branch)."""
    C = p.shape[0]
    l_mask = torch.tril(torch.ones(C, C, dtype=p.dtype), -1)
    eye = torch.eye(C, dtype=p.dtype)
    lm = l * l_mask + eye
    um = u * l_mask.transpose(0, 1).contiguous() + torch.diag(sign_s * torch.exp(log_s))
    if not reverse:
        w = torch.matmul(p, torch.matmul(lm, um))
    else:
        li = torch.inverse(lm.double()).to(p.dtype)         # .float() in the reference; p.dtype so that a float64 evaluation
        ui = torch.inverse(um.double()).to(p.dtype)         # of the whole network (tests: numerical noise floor) stays float64
        w = torch.matmul(ui, torch.matmul(li, p.inverse()))
    return w


def invconv(x, p, l, u, log_s, sign_s, logdet=None, reverse=False):
    """models/modules/Permutations.py:103-123"""
    w = invconv_weight(p, l, u, log_s, sign_s, reverse)
    d = torch.sum(log_s) * pixels(x)
    z = F.conv2d(x, w.view(w.shape[0], w.shape[1], 1, 1))
    if logdet is not None:
        logdet = logdet + d if not reverse else logdet - d
    return z, logdet


def invconv_plain(x, weight, logdet=None, reverse=False):
    """models/modules/Permutations.py:78-86 (non-LU branch: a `weight` key in the state dict)."""
    d = torch.slogdet(weight)[1] * pixels(x)
    w = weight if not reverse else torch.inverse(weight.double()).to(weight.dtype)
    z = F.conv2d(x, w.view(w.shape[0], w.shape[1], 1, 1))
    if logdet is not None:
        logdet = logdet + d if not reverse else logdet - d
    return z, logdet


# ----------------------------------------------------------------------------- coupling network
def conv2d_actnorm(x, weight, an_bias, an_logs):
    """models/modules/flow.py:25-64: 'same' zero-pad conv, no bias, then ActNorm."""
    pad = [((k - 1) + 1) // 2 for k in weight.shape[2:]]
    y = F.conv2d(x, weight, None, 1, pad)
    y, _ = actnorm(y, an_bias, an_logs)
    return y


def conv2d_zeros(x, weight, bias, logs, logscale_factor=3):
    """models/modules/flow.py:67-82"""
    y = F.conv2d(x, weight, bias, 1, 1)
    return y * torch.exp(logs * logscale_factor)


def coupling_nn(x, sd, pfx):
    """models/modules/FlowAffineCouplingsAblation.py:248-256: Conv2d,ReLU,Conv2d(1x1),ReLU,Conv2dZeros"""
    h = conv2d_actnorm(x, sd[pfx + "0.weight"], sd[pfx + "0.actnorm.bias"], sd[pfx + "0.actnorm.logs"])
    h = F.relu(h)
    h = conv2d_actnorm(h, sd[pfx + "2.weight"], sd[pfx + "2.actnorm.bias"], sd[pfx + "2.actnorm.logs"])
    h = F.relu(h)
    return conv2d_zeros(h, sd[pfx + "4.weight"], sd[pfx + "4.bias"], sd[pfx + "4.logs"])


def scale_shift(h, eps):
    """FlowAffineCouplingsAblation.py:219-240: shift = h[:,0::2], scale = sigmoid(h[:,1::2]+2)+eps"""
    shift, scale = split_feature(h, "cross")
    return torch.sigmoid(scale + 2.0) + eps, shift


def get_logdet(scale):
    """FlowAffineCouplingsAblation.py:216-217"""
    return thops_sum(torch.log(scale), [1, 2, 3])


def cond_affine_separated_and_cond(z, logdet, reverse, ft, sd, pfx, eps=0.001, mult_reverse=True):
    """FlowAffineCouplingsAblation.py:159-208 (CondAffineSeparatedAndCond.forward).
    `mult_reverse` flips the flag (line 164-165), so encode runs the division branch."""
    C = z.shape[1]
    cn = C // 2
    if mult_reverse:
        reverse = not reverse
    if not reverse:
        assert z.shape[1] == C
        scaleFt, shiftFt = scale_shift(coupling_nn(ft, sd, pfx + "fFeatures."), eps)
        z = z + shiftFt
        z = z * scaleFt
        logdet = logdet + get_logdet(scaleFt)
        z1, z2 = z[:, :cn], z[:, cn:]
        scale, shift = scale_shift(coupling_nn(torch.cat([z1, ft], dim=1), sd, pfx + "fAffine."), eps)
        assert scale.shape[1] == z2.shape[1]
        z2 = z2 + shift
        z2 = z2 * scale
        logdet = logdet + get_logdet(scale)
        z = torch.cat((z1, z2), dim=1)
    else:
        z1, z2 = z[:, :cn], z[:, cn:]
        scale, shift = scale_shift(coupling_nn(torch.cat([z1, ft], dim=1), sd, pfx + "fAffine."), eps)
        assert scale.shape[1] == z2.shape[1]
        z2 = z2 / scale
        z2 = z2 - shift
        z = torch.cat((z1, z2), dim=1)
        logdet = logdet - get_logdet(scale)
        scaleFt, shiftFt = scale_shift(coupling_nn(ft, sd, pfx + "fFeatures."), eps)
        z = z / scaleFt
        z = z - shiftFt
        logdet = logdet - get_logdet(scaleFt)
    return z, logdet


# ----------------------------------------------------------------------------- flow step / stack
def _invconv_from_sd(z, logdet, reverse, sd, pfx):
    if pfx + "weight" in sd:
        return invconv_plain(z, sd[pfx + "weight"], logdet, reverse)
    return invconv(z, sd[pfx + "p"], sd[pfx + "l"], sd[pfx + "u"], sd[pfx + "log_s"], sd[pfx + "sign_s"],
                   logdet, reverse)


def flow_step(z, logdet, reverse, ft, sd, pfx, eps=0.001):
    """models/modules/FlowStep.py:114-196: actnorm -> invconv -> coupling, and the exact reverse."""
    has_coupling = (pfx + "affine.fAffine.0.weight") in sd
    if not reverse:
        z, logdet = actnorm(z, sd[pfx + "actnorm.bias"], sd[pfx + "actnorm.logs"], logdet, False)
        z, logdet = _invconv_from_sd(z, logdet, False, sd, pfx + "invconv.")
        if has_coupling:
            z, logdet = cond_affine_separated_and_cond(z, logdet, False, ft, sd, pfx + "affine.", eps)
    else:
        if has_coupling:
            z, logdet = cond_affine_separated_and_cond(z, logdet, True, ft, sd, pfx + "affine.", eps)
        z, logdet = _invconv_from_sd(z, logdet, True, sd, pfx + "invconv.")
        z, logdet = actnorm(z, sd[pfx + "actnorm.bias"], sd[pfx + "actnorm.logs"], logdet, True)
    return z, logdet


LEVEL_TO_NAME = {0: "fea_up4", 1: "fea_up2", 2: "fea_up1", 3: "fea_up0", 4: "fea_up-1"}   # FlowUpsamplerNet.py:66-73


def layer_kinds(sd, pfx):
    """Recover the layer list (FlowUpsamplerNet.py:94-105) from the state-dict keys: a layer index
    with no parameters is a squeeze."""
    idx = set()
    for k in sd:
        if k.startswith(pfx + "layers."):
            idx.add(int(k[len(pfx) + 7:].split(".")[0]))
    n = max(idx) + 1
    # squeeze layers hold no parameters: every missing index below the max, plus index 0; a Split2d (multi-level
    # configurations, FlowUpsamplerNet.py:149-167) is the layer with a `conv.weight` of its own
    return ["squeeze" if i not in idx else "split" if f"{pfx}layers.{i}.conv.weight" in sd else "step" for i in range(n)]


def _static_sizes(kinds, base=160):
    """output_shapes[i][2] of FlowUpsamplerNet.py:141-142,183-187 for the 160-pixel patch."""
    sizes, s = [], base
    for k in kinds:
        if k == "squeeze":
            s //= 2
        sizes.append(s)
    return sizes


def _level(test_input_size, size):
    """FlowUpsamplerNet.py:230-234 / 286-290"""
    ref = 160 if test_input_size is None else test_input_size
    return int(np.log(ref / size) / np.log(2))


def flow_encode(gt, rrdb_results, sd, pfx, logdet, test_input_size=None, eps=0.001, epses=None, y_onehot=None):
    """FlowUpsamplerNet.encode, FlowUpsamplerNet.py:217-255.  With Split2d layers (unconditional ones: `position` None,
    :156, :264) each split's eps is appended to `epses` when that is a list, and the list + [z] is returned (:251-255)."""
    kinds = layer_kinds(sd, pfx)
    static = _static_sizes(kinds)
    z = gt
    for i, kind in enumerate(kinds):
        size = static[i] if test_input_size is None else z.shape[2]
        level = _level(test_input_size, size)
        cond = rrdb_results[LEVEL_TO_NAME[level]]          # :239 (indexed for squeeze layers too)
        if kind == "squeeze":
            z = squeeze2d(z, 2)
        elif kind == "split":                               # :243-245, 263-269
            z, logdet, e = split2d(z, logdet, False, sd, f"{pfx}layers.{i}.", y_onehot)
            if isinstance(epses, list):
                epses.append(e)
        else:
            z, logdet = flow_step(z, logdet, False, cond, sd, f"{pfx}layers.{i}.", eps)
    if not isinstance(epses, list):
        return z, logdet
    epses.append(z)
    return epses, logdet


def flow_decode(z, rrdb_results, sd, pfx, logdet, test_input_size=None, eps=0.001, n_levels=2, epses=None, y_onehot=None):
    """FlowUpsamplerNet.decode, FlowUpsamplerNet.py:271-304; `epses` (a list: the encode's return value) supplies z and the
    eps of every Split2d, popped from the end of a copy (:206, :272, :306-311)."""
    kinds = layer_kinds(sd, pfx)
    static = _static_sizes(kinds)
    if isinstance(epses, list):
        epses = list(epses)
        z = epses.pop()
    level_conditionals = {lv: rrdb_results[LEVEL_TO_NAME[lv]] for lv in range(n_levels + 1)}   # :278-280
    for i in reversed(range(len(kinds))):
        size = static[i] if test_input_size is None else z.shape[2]
        level = _level(test_input_size, size)
        if kinds[i] == "squeeze":
            z = unsqueeze2d(z, 2)
        elif kinds[i] == "split":
            rrdb_results[LEVEL_TO_NAME[level]]              # :294 indexes the dict
            z, logdet = split2d(z, logdet, True, sd, f"{pfx}layers.{i}.", y_onehot, eps=epses.pop())
        else:
            z, logdet = flow_step(z, logdet, True, level_conditionals[level], sd, f"{pfx}layers.{i}.", eps)
    assert z.shape[1] == 3                                  # :303
    return z, logdet


# ----------------------------------------------------------------------------- prior / loss
def gaussian_likelihood(mean, cov, x):
    """models/modules/flow.py:141-156"""
    x = x.permute(0, 2, 3, 1)
    k = x.shape[-1]
    if mean is None and cov is None:
        return -0.5 * (k * LOG2PI + torch.sum(x ** 2, dim=-1))
    x = x - mean
    cov_inv = torch.inverse(cov)
    ld = torch.slogdet(cov)[1]
    return -0.5 * (ld + k * LOG2PI + torch.sum((x @ cov_inv) * x, dim=-1))


def gaussian_logp(mean, cov, x):
    """models/modules/flow.py:158-161"""
    return thops_sum(gaussian_likelihood(mean, cov, x), [1, 2])


def calculate_loss(z, logdet, npix, label, classes, mean_shifts, cov_shifts):
    """models/flow_model.py:119-173.  Returns (z, nll, logdet, ll_ib)."""
    C = z.shape[1]
    objective = logdet.clone()
    if not isinstance(label, torch.Tensor):
        label = torch.Tensor(label)
    assert z.shape[0] == label.shape[0], 'need one class label per datapoint'
    assert len(label.shape) == 1, 'labels must be one dimensional'
    dom = {'X': label == 0}
    dom.update({f'Y_{c}': label == c for c in classes})
    covs = [torch.matmul(c, c.T) for c in cov_shifts]
    ll = torch.zeros(z.shape[0])
    for i, (k, v) in enumerate(dom.items()):
        if k == 'X':
            ll[v] = gaussian_logp(None, None, z[v])
        else:
            ll[v] = gaussian_logp(mean_shifts[i - 1] * torch.ones(C), covs[i - 1], z[v])
    cls = torch.tensor(np.tile(classes, reps=z.shape[0]).reshape((z.shape[0], -1)))
    mask_label = (cls == label.reshape(-1, 1))
    ll_ib = torch.zeros(cls.shape)
    for i, (k, v) in enumerate(dom.items()):
        if k == 'X':
            ll[v] = gaussian_logp(None, None, z[v])
        else:
            ll_ib[:, i - 1] = gaussian_logp(mean_shifts[i - 1] * torch.ones(C), covs[i - 1], z)
    ll_ib = ll_ib / float(npix)
    ll_ib = torch.log_softmax(ll_ib, dim=1)
    ll_ib = torch.masked_select(ll_ib, mask_label)
    objective = objective + ll
    nll = (-objective) / float(np.log(2.) * npix)
    return z, nll, logdet, ll_ib


def get_pdf(z, scales, classes, mean_shifts, cov_shifts):
    """models/flow_model.py:202-215"""
    C = z.shape[1]
    npix = pixels(z)
    covs = [torch.matmul(c, c.T) for c in cov_shifts]
    out = []
    for s in scales:
        idx = classes.index(s)
        out.append(gaussian_logp(torch.ones(C) * mean_shifts[idx], covs[idx], z) / float(np.log(2.) * npix))
    return torch.stack(out, dim=0).reshape(-1)


# ----------------------------------------------------------------------------- conditioning encoder
def rrdb_forward(x, sd, pfx, nb=23, block_idxs=(1, 8, 15, 22)):
    """models/modules/RRDBNet_arch.py:85-148 (scale 4, fea_up0 enabled) -> results dict."""
    def conv(name, t):
        return F.conv2d(t, sd[pfx + name + ".weight"], sd[pfx + name + ".bias"], 1, 1)

    def lrelu(t):
        return F.leaky_relu(t, 0.2)

    def rdb(t, p):                                          # RRDBNet_arch.py:23-29
        x1 = lrelu(conv(p + "conv1", t))
        x2 = lrelu(conv(p + "conv2", torch.cat((t, x1), 1)))
        x3 = lrelu(conv(p + "conv3", torch.cat((t, x1, x2), 1)))
        x4 = lrelu(conv(p + "conv4", torch.cat((t, x1, x2, x3), 1)))
        x5 = conv(p + "conv5", torch.cat((t, x1, x2, x3, x4), 1))
        return x5 * 0.2 + t

    n, c, h, w = x.shape
    x = F.unfold(x, 4, stride=4).view(n, c * 16, h // 4, w // 4)      # :79-83 space2depth
    fea = conv("conv_first", x)
    blocks = {}
    for b in range(nb):
        t = fea
        for r in (1, 2, 3):
            t = rdb(t, f"RRDB_trunk.{b}.RDB{r}.")
        fea = t * 0.2 + fea                                 # :41-45
        if b in block_idxs:
            blocks[f"block_{b}"] = fea
    trunk = conv("trunk_conv", fea)
    last_lr_fea = fea + trunk
    # RRDBNet_arch.py:77,105-109: the LeakyReLU is inplace=True, so the tensors the reference
    # returns as 'fea_up2' / 'fea_up4' are the POST-activation values (aliasing quirk, kept).
    fea_up2 = lrelu(conv("upconv1", F.interpolate(last_lr_fea, scale_factor=2, mode='nearest')))
    fea = fea_up2
    fea_up4 = lrelu(conv("upconv2", F.interpolate(fea, scale_factor=2, mode='nearest')))
    fea = fea_up4
    out = conv("conv_last", lrelu(conv("HRconv", fea)))
    res = {'last_lr_fea': last_lr_fea, 'fea_up1': last_lr_fea, 'fea_up2': fea_up2, 'fea_up4': fea_up4,
           'fea_up8': None, 'fea_up16': None, 'fea_up32': None, 'out': out,
           'fea_up0': F.interpolate(last_lr_fea, scale_factor=1 / 2, mode='bilinear')}
    res.update(blocks)
    return res


def rrdb_preprocessing(lr, sd, pfx, block_idxs=(1, 8, 15, 22)):
    """models/modules/RRDBGlowNet_arch.py:217-246 (stackRRDB concat=True)."""
    res = rrdb_forward(lr, sd, pfx, block_idxs=block_idxs)
    concat = torch.cat([res[f"block_{i}"] for i in block_idxs], dim=1)
    for k in ['last_lr_fea', 'fea_up1', 'fea_up2', 'fea_up4', 'fea_up0']:
        h, w = res[k].shape[2:]
        res[k] = torch.cat([res[k], F.interpolate(concat, (h, w))], dim=1)
    return res


# ----------------------------------------------------------------------------- model API
def interpflow_forward(sd, x, cond=None, lr_enc=None, reverse=False, test_input_size=None,
                       noise=None, add_gt_noise=False, quant=256):
    """models/flow_model.py:83-116 without the loss.  `noise` replaces torch.rand (line 109) so
    the result is RNG-independent; returns (z_or_x, logdet)."""
    if lr_enc is None:
        lr_enc = rrdb_preprocessing(cond, sd, "arch.RRDB.")
    ref = cond if cond is not None else x
    if reverse:
        logdet = torch.zeros_like(ref[:, 0, 0, 0])
        if add_gt_noise:
            logdet = logdet - float(-np.log(quant) * pixels(ref))
        return flow_decode(x, lr_enc, sd, "flow_upsampler.", logdet, test_input_size)
    logdet = torch.zeros_like(x[:, 0, 0, 0])
    npix = pixels(x)
    if add_gt_noise:
        x = x + ((noise - 0.5) / quant)
        logdet = logdet + float(-np.log(quant) * npix)
    return flow_encode(x, lr_enc, sd, "flow_upsampler.", logdet, test_input_size)


def weighted_interpolate(sd, lr1, lr2, a, lr_enc):
    """models/flow_model.py:194-200 with add_noise=False and a shared lr_enc."""
    z1, _ = interpflow_forward(sd, lr1, lr_enc=lr_enc)
    z2, _ = interpflow_forward(sd, lr2, lr_enc=lr_enc)
    zt = (1 - a) * z1 + a * z2
    return interpflow_forward(sd, zt, cond=lr1, lr_enc=lr_enc, reverse=True, test_input_size=lr1.shape[2])[0]


def to_torch(sd_np):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}


# ----------------------------------------------------------------------------- ResizeRight (LR-consistency loss)
def _cubic(x):
    """utils/interp_methods.py:35-43"""
    absx = torch.abs(x)
    absx2, absx3 = absx ** 2, absx ** 3
    return ((1.5 * absx3 - 2.5 * absx2 + 1.) * (absx <= 1.).to(x.dtype) +
            (-0.5 * absx3 + 2.5 * absx2 - 4. * absx + 2.) * ((1. < absx) & (absx <= 2.)).to(x.dtype))


def resize_right(x, scale):
    """utils/resize_right.py:32-123 for resize(x, scale_factors=scale, interp_method=cubic, antialiasing=True): last two
    dims, gather formulation as the reference (field of view, zero 'constant' padding, weights normalised per output)."""
    eps = torch.finfo(torch.float32).eps
    out = x
    for dim in (x.ndim - 2, x.ndim - 1):
        in_sz = x.shape[dim]
        out_sz = math.ceil(scale * in_sz)                                                    # :302-304
        grid = torch.arange(out_sz) / float(scale) + (in_sz - 1) / 2 - (out_sz - 1) / (2 * float(scale))   # :126-137
        if scale < 1.0:                                                                      # :343-355
            interp, sup = (lambda a: scale * _cubic(scale * a)), 4 / scale
        else:
            interp, sup = _cubic, 4.0
        left = (grid - sup / 2 - eps).ceil().long()                                          # :140-149
        fov = left[:, None] + torch.arange(math.ceil(sup - eps))
        pad = [-fov[0, 0].item(), fov[-1, -1].item() - in_sz + 1]                            # :152-164
        fov, grid = fov + pad[0], grid + pad[0]
        w = interp(grid[:, None] - fov)                                                      # :201-212
        s = w.sum(1, keepdim=True)
        s[s == 0] = 1
        w = w / s
        t = out.transpose(dim, 0)                                                            # :215-247
        if pad[0] > 0 or pad[1] > 0:
            t = F.pad(t.transpose(0, -1), (max(pad[0], 0), max(pad[1], 0))).transpose(0, -1)
        if pad[0] < 0 or pad[1] < 0:                                                         # negative pad = crop
            t = t[(-pad[0] if pad[0] < 0 else 0):(t.shape[0] + pad[1] if pad[1] < 0 else t.shape[0])]
        nb = t[fov]
        out = (nb * w.reshape(*w.shape, *[1] * (x.ndim - 1))).sum(1).transpose(0, dim)
    return out


# ----------------------------------------------------------------------------- Split2d
def split2d(x, logdet, reverse, sd, pfx, y_onehot, ft=None, eps=None, logs_eps=0.0, tanh=False):
    """models/modules/Split.py:41-122 (Split2d.forward).  `sd[pfx + ...]`: conv.{weight,bias,logs}, mean_shift, cov_shift.
    Forward: x [B,C,H,W] -> (z1, logdet, eps); reverse (eps given): z1 -> (z, logdet).  y_onehot: 1-D tensor of 0 / 1 domain
    labels (0: standard normal, 1: N(mean_shift, I + cov_shift cov_shift^T))."""
    consume = sd[pfx + "conv.weight"].shape[0] // 2
    dom_x, dom_y = y_onehot == 0, y_onehot == 1

    def prior(z1):
        z = z1 if ft is None else torch.cat([z1, ft], dim=1)                                   # Split.py:41-45
        h = conv2d_zeros(z, sd[pfx + "conv.weight"], sd[pfx + "conv.bias"], sd[pfx + "conv.logs"])
        return split_feature(h, "cross")

    def exp_eps(logs):                                                                           # Split.py:47-51
        return (torch.exp(torch.tanh(logs)) if tanh else torch.exp(logs)) + logs_eps

    def ll_of(e):                                                                                # Split.py:77-84, 108-113
        cov = torch.eye(consume, dtype=e.dtype) + sd[pfx + "cov_shift"] @ sd[pfx + "cov_shift"].T
        ll = torch.zeros(e.shape[0], dtype=e.dtype)
        if dom_x.any():
            ll[dom_x] = gaussian_logp(None, None, e[dom_x])
        if dom_y.any():
            ll[dom_y] = gaussian_logp(sd[pfx + "mean_shift"], cov, e[dom_y])
        return ll

    if not reverse:
        npass = x.shape[1] - consume
        z1, z2 = x[:, :npass], x[:, npass:]                                                      # Split.py:123-125
        mean, logs = prior(z1)
        e = (z2 - mean) / (exp_eps(logs) + 1e-6)
        return z1, logdet + ll_of(e) - logs.sum(dim=[1, 2, 3]), e
    mean, logs = prior(x)
    z2 = mean + exp_eps(logs) * eps
    return torch.cat((x, z2), dim=1), logdet - ll_of(eps) + logs.sum(dim=[1, 2, 3])
