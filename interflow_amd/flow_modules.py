"""Host-side mirror of the reference's flow modules (`interflow/models/modules/*.py`).

Same class names, constructor arguments, parameter/buffer names and call signatures as the reference,
so `state_dict()` / `load_state_dict(strict=True)` interchange with it.  The modules only HOLD
parameters; the arithmetic of the hot path runs in libinterflow_hip.so:

* `FlowUpsamplerNet.forward` -> `ifl_flow_encode` / `ifl_flow_decode` (whole 38-layer stack natively);
* `FlowStep.forward`         -> `ifl_flow_step`;
* `SqueezeLayer.forward`     -> `ifl_squeeze2d` / `ifl_unsqueeze2d`.

Leaf modules that the reference calls one by one (ActNorm2d, InvertibleConv1x1, Conv2d, ...) are fused
into those kernels and have no forward of their own here.
"""
from __future__ import annotations

import ctypes
import weakref

import numpy as np
import torch
from torch import nn

from . import hip, pack
from .synth import HIDDEN, N_RRDB_CH


def opt_get(opt, keys, default=None):
    """utils/util.py:259-267"""
    if opt is None:
        return default
    ret = opt
    for k in keys:
        ret = ret.get(k, None)
        if ret is None:
            return default
    return ret


class _Fused(nn.Module):
    def forward(self, *a, **k):
        raise NotImplementedError(
            f"{type(self).__name__} is fused into the FlowStep / FlowUpsamplerNet HIP kernels; call those")


class ActNorm2d(_Fused):
    """FlowActNorms.py:7-109 (parameter holder)."""

    def __init__(self, num_features, scale=1.):
        super().__init__()
        self.register_parameter("bias", nn.Parameter(torch.zeros(1, num_features, 1, 1)))
        self.register_parameter("logs", nn.Parameter(torch.zeros(1, num_features, 1, 1)))
        self.num_features = num_features
        self.scale = float(scale)
        self.inited = False


class InvertibleConv1x1(_Fused):
    """Permutations.py:44-123 (parameter holder; same initialisation)."""

    def __init__(self, num_channels, LU_decomposed=False, min_singular=None):
        super().__init__()
        import scipy.linalg
        w_shape = [num_channels, num_channels]
        w_init = np.linalg.qr(np.random.randn(*w_shape))[0].astype(np.float32)
        if not LU_decomposed:
            self.register_parameter("weight", nn.Parameter(torch.Tensor(w_init)))
        else:
            np_p, np_l, np_u = scipy.linalg.lu(w_init)
            np_s = np.diag(np_u)
            self.register_buffer('p', torch.Tensor(np_p.astype(np.float32)))
            self.register_buffer('sign_s', torch.Tensor(np.sign(np_s).astype(np.float32)))
            self.l = nn.Parameter(torch.Tensor(np_l.astype(np.float32)))
            self.log_s = nn.Parameter(torch.Tensor(np.log(np.abs(np_s)).astype(np.float32)))
            self.u = nn.Parameter(torch.Tensor(np.triu(np_u, k=1).astype(np.float32)))
        self.w_shape = w_shape
        self.LU = LU_decomposed
        if min_singular is not None:
            raise NotImplementedError("min_singular clipping (Permutations.py:108-111) is not part of the shipped config")


class Conv2d(_Fused):
    """flow.py:25-64: 3x3 / 1x1 'same' conv without bias + ActNorm (parameter holder)."""

    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), weight_std=0.05):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *kernel_size).normal_(0.0, weight_std))
        self.actnorm = ActNorm2d(out_channels)


class Conv2dZeros(_Fused):
    """flow.py:67-82 (parameter holder)."""

    def __init__(self, in_channels, out_channels, logscale_factor=3):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(out_channels, in_channels, 3, 3))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.logscale_factor = logscale_factor
        self.register_parameter("logs", nn.Parameter(torch.zeros(out_channels, 1, 1)))


def _F(in_channels, out_channels, hidden_channels):
    """FlowAffineCouplingsAblation.py:248-256 -- indices 0, 2, 4 hold parameters (1, 3 are ReLUs)."""
    return nn.Sequential(Conv2d(in_channels, hidden_channels), nn.ReLU(inplace=False),
                         Conv2d(hidden_channels, hidden_channels, kernel_size=(1, 1)), nn.ReLU(inplace=False),
                         Conv2dZeros(hidden_channels, out_channels))


class CondAffineSeparatedAndCond(_Fused):
    """FlowAffineCouplingsAblation.py:123-157 (parameter holder)."""

    def __init__(self, in_channels, opt):
        super().__init__()
        self.need_features = True
        self.in_channels = in_channels
        self.in_channels_rrdb = N_RRDB_CH
        key = ['network_G', 'flow', 'CondAffineSeparatedAndCond']
        hidden = opt_get(opt, key + ['hidden_channels'])
        self.hidden_channels = HIDDEN if hidden is None else hidden
        self.affine_eps = opt_get(opt, key + ['eps'], 0.0001)
        self.mult_reverse = opt_get(opt, key + ['multReverse'], False)
        for flag in ('tanh', 'skip'):
            if opt_get(opt, key + [flag], False):
                raise NotImplementedError(f"CondAffineSeparatedAndCond option '{flag}' is outside the shipped config")
        if self.hidden_channels != HIDDEN or not self.mult_reverse:
            raise NotImplementedError("the HIP kernels implement hidden_channels=64, multReverse=True (flow_model.py:46)")
        self.channels_for_nn = in_channels // 2
        self.channels_for_co = in_channels - self.channels_for_nn
        self.fAffine = _F(self.channels_for_nn + self.in_channels_rrdb, self.channels_for_co * 2, self.hidden_channels)
        self.fFeatures = _F(self.in_channels_rrdb, in_channels * 2, self.hidden_channels)


class SqueezeLayer(nn.Module):
    """flow.py:196-207"""

    def __init__(self, factor):
        super().__init__()
        self.factor = factor

    def forward(self, input, logdet=None, reverse=False):
        if not reverse:
            return hip.squeeze2d(input, self.factor), logdet
        return hip.unsqueeze2d(input, self.factor), logdet


def _module_params_numpy(module, prefix=""):
    return {prefix + k: v.detach().to("cpu", torch.float32).numpy() for k, v in module.state_dict().items()}


def _version_key(module):
    return tuple((p.data_ptr(), p._version) for p in list(module.parameters()) + list(module.buffers()))


class FlowStep(nn.Module):
    """FlowStep.py:29-196: actnorm -> invconv -> coupling, and the exact reverse order."""

    def __init__(self, in_channels, hidden_channels, actnorm_scale=1.0, flow_permutation="invconv",
                 flow_coupling="additive", LU_decomposed=False, opt=None, idx=None, acOpt=None, normOpt=None,
                 position=None):
        super().__init__()
        if flow_permutation != "invconv" or normOpt:
            raise NotImplementedError("only flow_permutation='invconv' with ActNorm2d is in the shipped config")
        if flow_coupling not in ("noCoupling", "CondAffineSeparatedAndCond"):
            raise NotImplementedError(f"coupling '{flow_coupling}' is not reachable from the shipped config "
                                      "(SURVEY.md section 2, row 11)")
        self.flow_permutation = flow_permutation
        self.flow_coupling = flow_coupling
        self.position = position
        self.in_channels = in_channels
        self.actnorm = ActNorm2d(in_channels, actnorm_scale)
        self.invconv = InvertibleConv1x1(in_channels, LU_decomposed=LU_decomposed)
        if flow_coupling == "CondAffineSeparatedAndCond":
            self.affine = CondAffineSeparatedAndCond(in_channels=in_channels, opt=opt)
        self._packed = None

    def _pack(self, device):
        key = (_version_key(self), str(device), self.training)
        if self._packed is None or self._packed[0] != key:
            prm = _module_params_numpy(self, "layers.0.")
            has = hasattr(self, "affine")
            params, layers = pack.pack_flow(prm, [("stepc" if has else "step0", self.in_channels)],
                                            training=self.training)
            self._packed = (key, torch.from_numpy(params).to(device), layers)
        return self._packed[1], self._packed[2]

    def forward(self, input, logdet=None, reverse=False, rrdbResults=None):
        z = hip._dev_f32(input, "FlowStep input")
        B, C, h, w = z.shape
        assert C == self.in_channels, (C, self.in_channels)            # FlowActNorms.py:104-109
        has = hasattr(self, "affine")
        ft = None
        if has:
            ft = rrdbResults if isinstance(rrdbResults, torch.Tensor) else rrdbResults[self.position]   # FlowStep.py:13-15
            ft = hip._dev_f32(ft, "FlowStep rrdbResults")
            assert ft.shape == (B, N_RRDB_CH, h, w), (tuple(ft.shape), (B, N_RRDB_CH, h, w))
        params, layers = self._pack(z.device)
        lvl = 1 if C == 12 else 2
        ws = hip.Workspace.get(B, h << lvl, w << lvl, z.device)
        ws.generation += 1                      # a prepared conditioning in this workspace is overwritten
        out = torch.empty_like(z)
        ld_in = None if logdet is None else _as_logdet(logdet, B, z.device)
        ld_out = torch.empty(B, dtype=torch.float32, device=z.device)
        eps = float(self.affine.affine_eps) if has else 0.0
        with torch.cuda.device(z.device):
            hip.check(hip.lib().ifl_flow_step(layers, hip._ptr(params), ws.ptr, hip._ptr(z), hip._ptr(out),
                                              hip._ptr(ft), hip._ptr(ld_in), hip._ptr(ld_out), B, h, w,
                                              1 if reverse else 0, eps, hip._stream()), "ifl_flow_step")
            ws.mark_used()
        return out, (ld_out if logdet is not None else None)


class Split2d(nn.Module):
    """Split.py:13-125: the conditional split prior of multi-level flows.  A stand-alone
    operator on the HIP kernels -- its 3x3 Conv2dZeros runs on the library's generic convolution (train_ops.conv2d_ring:
    ifl_conv2d, and its gradients), the class log-densities on ifl_gaussian_logp (flow_model.Gaussian.logp); the per-pixel
    arithmetic between them is torch on the device.  The shipped configuration (L=2, flow_model.py:40-46) never creates
    one, so the fused FlowUpsamplerNet driver does not schedule it (it raises for configurations that would)."""

    def __init__(self, num_channels, logs_eps=0, cond_channels=0, position=None, consume_ratio=0.5, opt=None):
        super().__init__()
        self.num_channels_consume = int(round(num_channels * consume_ratio))
        self.num_channels_pass = num_channels - self.num_channels_consume
        self.conv = Conv2dZeros(in_channels=self.num_channels_pass + cond_channels, out_channels=self.num_channels_consume * 2)
        self.logs_eps = opt_get(opt, ['network_G', 'flow', 'split', 'eps'], logs_eps)
        self.position = position
        self.opt = opt
        C = self.num_channels_consume
        self.I = nn.Parameter(torch.eye(C), requires_grad=False)
        self.mean_shift = nn.Parameter(torch.zeros(C))
        self.cov_shift = nn.Parameter(torch.eye(C) * opt_get(opt, ['network_G', 'flow', 'shift', 'std_init_shift'], 1.0))

    def split2d_prior(self, z, ft):                                                       # Split.py:41-45
        from . import train_ops as T
        cin = z.shape[1] + (0 if ft is None else ft.shape[1])
        assert cin == self.conv.weight.shape[1], (cin, tuple(self.conv.weight.shape))
        src = z if ft is None else torch.cat([z, hip._dev_f32(ft, "Split2d ft")], dim=1)
        h = T.from_ring(T.conv2d_ring([T.to_ring(src)], [cin], self.conv.weight, self.conv.bias), self.conv.weight.shape[0])
        h = h * torch.exp(self.conv.logs * self.conv.logscale_factor)                     # flow.py:80-82
        return h[:, 0::2], h[:, 1::2]                                                      # thops.py:54-55 ("cross")

    def exp_eps(self, logs):                                                              # Split.py:47-51
        if opt_get(self.opt, ['network_G', 'flow', 'split', 'tanh'], False):
            return torch.exp(torch.tanh(logs)) + self.logs_eps
        return torch.exp(logs) + self.logs_eps

    def _class_ll(self, eps, y_onehot):                                                   # Split.py:74-84, 105-113
        from .flow_model import Gaussian
        dom_x, dom_y = y_onehot == 0, y_onehot == 1
        ll = torch.zeros(eps.shape[0], device=eps.device)
        if bool(dom_x.any()):
            ll[dom_x] = Gaussian.logp(None, None, eps[dom_x].contiguous())
        if bool(dom_y.any()):
            cov_shifted = self.I + torch.matmul(self.cov_shift, self.cov_shift.T)
            ll[dom_y] = Gaussian.logp(mean=self.mean_shift, cov=cov_shifted, x=eps[dom_y].contiguous(), key=self.cov_shift)
        return ll

    def forward(self, input, logdet=0., reverse=False, eps_std=None, eps=None, ft=None, y_onehot=None):
        x = hip._dev_f32(input, "Split2d input")
        if type(y_onehot) != torch.Tensor:
            y_onehot = torch.Tensor(y_onehot)
        y_onehot = y_onehot.to(x.device)
        assert len(y_onehot.shape) == 1, 'labels must be one dimensional'                  # Split.py:60
        if not reverse:
            z1, z2 = self.split_ratio(x)
            mean, logs = self.split2d_prior(z1, ft)
            eps = (z2 - mean) / (self.exp_eps(logs) + 1e-6)
            assert eps.shape[0] == y_onehot.shape[0], 'need one class label per datapoint'
            logdet = logdet + self._class_ll(eps, y_onehot) - logs.sum(dim=[1, 2, 3])
            return z1, logdet, eps
        z1 = x
        mean, logs = self.split2d_prior(z1, ft)
        if eps is None:                                                                    # Split.py:92-103: sample eps
            std = 1.0 if eps_std is None else eps_std
            eps = torch.randn(mean.shape, device=mean.device) * std
            dom_y = y_onehot == 1
            if bool(dom_y.any()):
                noise = torch.randn(mean.shape, device=mean.device)[dom_y] * std
                eps[dom_y] = eps[dom_y] + self.mean_shift.reshape(1, -1, 1, 1) + torch.matmul(self.cov_shift, noise.permute(3, 2, 1, 0)).permute(3, 2, 1, 0)   # Split.py:101: `.T` of a 4-D tensor
        else:
            eps = hip._dev_f32(eps, "Split2d eps")
        assert eps.shape[0] == y_onehot.shape[0], 'need one class label per datapoint'
        z2 = mean + self.exp_eps(logs) * eps
        z = torch.cat((z1, z2), dim=1)
        return z, logdet - self._class_ll(eps, y_onehot) + logs.sum(dim=[1, 2, 3])

    def split_ratio(self, input):                                                         # Split.py:123-125
        return input[:, :self.num_channels_pass, ...], input[:, self.num_channels_pass:, ...]


def _as_logdet(logdet, B, device):
    """The reference accepts a python float (default 0.) or a [B] tensor for logdet."""
    if isinstance(logdet, torch.Tensor):
        ld = logdet.to(device=device, dtype=torch.float32).reshape(-1)
        return (ld.expand(B) if ld.numel() == 1 else ld).contiguous()
    return torch.full((B,), float(logdet), dtype=torch.float32, device=device)


def _tensor_version(t):
    """torch's in-place write counter of a tensor, or None for tensors that keep none (made under torch.inference_mode):
    writes to those cannot be seen, so they never count as unchanged."""
    try:
        return t._version
    except RuntimeError:
        return None


class PreparedCond(dict):
    """`rrdbResults` whose conditioning-only part of every coupling network is already in a workspace
    (`FlowUpsamplerNet.prepare_cond`, `ifl_cond_prepare`).  Behaves as the plain dict everywhere; `FlowUpsamplerNet`
    recognises it and skips that part while the workspace and the parameters are unchanged, and silently takes the
    ordinary path otherwise.  Outputs are bit-identical either way."""

    def __init__(self, rrdbResults, ws, generation, params_key, level_feats, shape):
        super().__init__(rrdbResults)
        self._ws, self._generation, self._params_key = ws, generation, params_key
        self._level_feats, self._shape = level_feats, shape

    def usable(self, ws, params_key, feats, shape):
        return (ws is self._ws and ws.generation == self._generation and params_key == self._params_key
                and shape == self._shape and set(feats) == set(self._level_feats)
                and all(feats[lv] is self._level_feats[lv] for lv in feats))


class FlowUpsamplerNet(nn.Module):
    """FlowUpsamplerNet.py:28-311.  Builds the same layer list and runs encode / decode through the HIP library."""

    def __init__(self, image_shape, hidden_channels, K, L=None, actnorm_scale=1.0, flow_permutation=None,
                 flow_coupling="affine", LU_decomposed=False, opt=None):
        super().__init__()
        self.layers = nn.ModuleList()
        self.output_shapes = []
        self.L = opt_get(opt, ['network_G', 'flow', 'L'])
        self.K = opt_get(opt, ['network_G', 'flow', 'K'])
        if isinstance(self.K, int):
            self.K = [self.K] * (self.L + 1)
        self.opt = opt
        H, W, self.C = image_shape
        assert self.C == 1 or self.C == 3, "image_shape should be HWC, like (64, 64, 3)"      # :198-200
        if opt['scale'] != 4:
            raise NotImplementedError("only scale 4 (flow_model.py:40) is in the shipped config")
        self.levelToName = {0: 'fea_up4', 1: 'fea_up2', 2: 'fea_up1', 3: 'fea_up0', 4: 'fea_up-1'}   # :66-73
        for unsupported in (['network_G', 'flow', 'norm'], ['network_G', 'flow', 'condAff'],
                            ['network_G', 'flow', 'condFtAffine'], ['network_G', 'flow', 'levelConditional']):
            if opt_get(opt, unsupported):
                raise NotImplementedError(f"option {unsupported} is outside the shipped config")
        self._kinds = []
        n_extra = int(opt['network_G']['flow'].get('additionalFlowNoAffine', 0))
        for level in range(1, self.L + 1):
            self.C, H, W = self.C * 4, H // 2, W // 2                                      # :183-187
            self.layers.append(SqueezeLayer(factor=2))
            self.output_shapes.append([-1, self.C, H, W])
            self._kinds.append(("squeeze", self.C))
            for _ in range(n_extra):                                                       # :169-181
                self.layers.append(FlowStep(in_channels=self.C, hidden_channels=hidden_channels,
                                            actnorm_scale=actnorm_scale, flow_permutation='invconv',
                                            flow_coupling='noCoupling', LU_decomposed=LU_decomposed, opt=opt))
                self.output_shapes.append([-1, self.C, H, W])
                self._kinds.append(("step0", self.C))
            for k in range(self.K[level]):                                                 # :122-142
                self.layers.append(FlowStep(in_channels=self.C, hidden_channels=hidden_channels,
                                            actnorm_scale=actnorm_scale, flow_permutation='invconv',
                                            flow_coupling=flow_coupling, LU_decomposed=LU_decomposed, opt=opt,
                                            idx=k, position='fea_up{}'.format(opt['scale'] / (160 // H))))
                self.output_shapes.append([-1, self.C, H, W])
                self._kinds.append(("stepc", self.C))
            correction = 0 if opt_get(opt, ['network_G', 'flow', 'split', 'correct_splits'], False) else 1
            if opt_get(opt, ['network_G', 'flow', 'split', 'enable']) and level < self.L - correction:   # :149-167
                if opt_get(opt, ['network_G', 'flow', 'split', 'conditional']):
                    raise NotImplementedError("conditional Split2d: the reference's own driver cannot run it (FlowUpsamplerNet.py:264 "
                                              "indexes the level's feature TENSOR with the position name)")
                if opt_get(opt, ['network_G', 'flow', 'split', 'type'], 'Split2d') != 'Split2d':
                    raise NotImplementedError("split type other than 'Split2d' (the reference creates nothing for it either, :162)")
                split = Split2d(num_channels=self.C,
                                logs_eps=opt_get(opt, ['network_G', 'flow', 'split', 'logs_eps']) or 0, position=None,
                                cond_channels=opt_get(opt, ['network_G', 'flow', 'split', 'cond_channels']) or 0,
                                consume_ratio=opt_get(opt, ['network_G', 'flow', 'split', 'consume_ratio']) or 0.5, opt=opt)
                self.layers.append(split)
                self.output_shapes.append([-1, split.num_channels_pass, H, W])
                self._kinds.append(("split", self.C))            # C entering the split; half of it goes on
                self.C = split.num_channels_pass
        # The fused stack driver (ifl_flow_pass) is built for the shipped layout: channel counts 12 and 48 on two levels.
        # Anything else -- more levels, Split2d layers: the configurations of general SRFlow / DeFlow checkpoints -- runs layer
        # by layer on the library's generic kernels (train_flow.run: ifl_conv2d, fused ActNorm+ReLU and coupling kernels,
        # Split2d on ifl_conv2d + ifl_gaussian_logp), with or without autograd.
        self._general = any(k == "split" or (k != "squeeze" and c not in (12, 48)) for k, c in self._kinds)
        # dead module the reference registers and never calls (:107-110); kept for the state-dict contract
        n_aff = (len(opt_get(opt, ['network_G', 'flow', 'stackRRDB', 'blocks']) or []) + 1) * 64
        n_f = 2 * 3 * 64 // 2 // 2 if opt_get(opt, ['network_G', 'flow', 'split', 'enable']) else 2 * 3 * 64
        self.f = nn.Sequential(nn.Conv2d(n_aff, n_f, kernel_size=[3, 3], stride=1, padding=1, bias=True))
        self.H, self.W = H, W
        self.scaleH, self.scaleW = 160 / H, 160 / W
        self._packed = None
        self._last_cond, self._pass_no = None, 0      # conditioning tensors of the previous pass (weak), passes so far
        self.affine_eps = opt_get(opt, ['network_G', 'flow', 'CondAffineSeparatedAndCond', 'eps'], 0.0001)
        # "fp32" (reference arithmetic); "bf16": the coupling networks' convolutions that depend only on the conditioning (95 % of
        # the FLOPs) on the bf16 matrix cores, operands rounded to bf16, fp32 accumulate -- encode and decode still invert each
        # other to fp32 accuracy; "bf16_all": fAffine's z-dependent convolutions too (all of BASELINE.json configs[4]; the round
        # trip then holds to bf16 accuracy only, see IFL_FLOW_BF16_CHAIN in interflow_hip.h).  Not a state-dict item.
        self.conv_dtype = "fp32"

    def _flags(self):
        if self.conv_dtype not in ("fp32", "bf16", "bf16_all"):
            raise ValueError(f"conv_dtype {self.conv_dtype!r}: expected 'fp32', 'bf16' or 'bf16_all'")
        return {"fp32": 0, "bf16": hip.FLOW_BF16_CONV1, "bf16_all": hip.FLOW_BF16_CONV1 | hip.FLOW_BF16_CHAIN}[self.conv_dtype]

    # ---- packed parameters, rebuilt when any parameter changes ----
    def _pack(self, device):
        key = (_version_key(self.layers), str(device), self.training)
        if self._packed is None or self._packed[0] != key:
            prm = _module_params_numpy(self.layers, "layers.")
            params, layers = pack.pack_flow(prm, self._kinds, training=self.training)
            self._packed = (key, torch.from_numpy(params).to(device), layers)
        return self._packed[1], self._packed[2]

    def _level(self, test_input_size, size):
        ref = 160 if test_input_size is None else test_input_size                        # :230-234, :286-290
        return int(np.log(ref / size) / np.log(2))

    def _level_features(self, rrdbResults, H, test_input_size, decode):
        """Apply the reference's per-layer level rule (:228-239 encode, :278-297 decode) and return, per flow
        level (number of squeezes so far), the conditioning map its FlowSteps use.  Missing keys raise KeyError
        exactly where the reference would."""
        if decode:                                                                       # :278-280
            for lv in range(self.L + 1):
                rrdbResults[self.levelToName[lv]]
        feats = {}
        size_in, flow_level = H, 0
        for (kind, _), shape in zip(self._kinds, self.output_shapes):
            size_out = size_in // 2 if kind == "squeeze" else size_in
            if kind == "squeeze":
                flow_level += 1
            if test_input_size is None:
                size = shape[2]                       # static 160-patch shapes
            else:
                size = size_out if decode else size_in    # fl_fea.shape[2] when the layer runs
            level = self._level(test_input_size, size)
            if kind == "squeeze":
                if not decode:
                    rrdbResults[self.levelToName[level]]      # :239 indexes the dict for every layer
            elif kind == "split":
                rrdbResults[self.levelToName[level]]          # :239 / :294 index the dict; an unconditional split reads nothing of it
            else:
                if decode and level not in range(self.L + 1):
                    raise KeyError(level)                     # level_conditionals[level], :297
                ft = rrdbResults[self.levelToName[level]]
                if flow_level in feats and feats[flow_level] is not ft:
                    raise NotImplementedError("one conditioning map per flow level expected")
                feats[flow_level] = ft
            size_in = size_out
        return feats

    def forward(self, gt=None, rrdbResults=None, z=None, epses=None, logdet=0., reverse=False, eps_std=None,
                y_onehot=None, test_input_size=None):
        if reverse:
            epses_copy = [eps for eps in epses] if isinstance(epses, list) else epses     # :206: the caller's list is left alone
            return self.decode(rrdbResults, z, eps_std, epses=epses_copy, logdet=logdet, y_onehot=y_onehot,
                               test_input_size=test_input_size)
        assert gt is not None
        assert rrdbResults is not None
        return self.encode(gt, rrdbResults, logdet=logdet, epses=epses, y_onehot=y_onehot,
                           test_input_size=test_input_size)

    def _load_cond(self, ws, feats, B, H, W, early=False):
        L = hip.lib()
        ws.generation += 1
        for lv, ft in feats.items():
            ft = hip._dev_f32(ft, f"rrdbResults level {lv}")
            assert ft.shape == (B, N_RRDB_CH, H >> lv, W >> lv), (tuple(ft.shape), (B, N_RRDB_CH, H >> lv, W >> lv))
            if early:
                hip.check(L.ifl_load_cond_early(ws.ptr, B, H, W, lv, hip._ptr(ft), ws.done_handle()), "ifl_load_cond_early")
            else:
                hip.check(L.ifl_load_cond(ws.ptr, B, H, W, lv, hip._ptr(ft), hip._stream()), "ifl_load_cond")

    # ---- passes in a row on unchanged conditioning tensors (hip.set_cross_pass_overlap) ----
    def _cond_unchanged(self, feats):
        """True when every conditioning tensor of this pass is the very tensor object the previous pass loaded, unwritten since
        (torch's version counter) and usable as it is: the features were then complete before that pass was issued, which is
        what IFL_FLOW_COND_EARLY asks the caller to know."""
        last = self._last_cond
        if last is None or set(last) != set(feats):
            return False
        for lv, ft in feats.items():
            ref, version, ptr = last[lv]
            if (version is None or ref() is not ft or _tensor_version(ft) != version or ft.data_ptr() != ptr
                    or ft.dtype != torch.float32 or not ft.is_contiguous()):
                return False
        return True

    def _remember_cond(self, feats):
        self._last_cond = {lv: (weakref.ref(ft), _tensor_version(ft), ft.data_ptr()) for lv, ft in feats.items()}

    def prepare_cond(self, rrdbResults, batch, H, W, test_input_size=None):
        """Compute, once, the part of every coupling network that depends only on the conditioning features (the first
        3x3 conv over the 320 feature channels of fFeatures and fAffine; no counterpart call in the reference, which
        recomputes it in each of the passes that `weighted_interpolate_image`, `generate_images.py` and `train.py` run
        over one `lr_enc`).  Returns `rrdbResults` wrapped so that later encode/decode calls with the same batch and
        image size reuse it.  H, W: size of the image the flow encodes (= the decode output)."""
        if self._general:
            return rrdbResults                                    # the layer-by-layer evaluation prepares nothing ahead
        enc = self._level_features(rrdbResults, H, test_input_size, decode=False)
        dec = self._level_features(rrdbResults, H, test_input_size, decode=True)
        if set(enc) != set(dec) or any(enc[lv] is not dec[lv] for lv in enc):
            return rrdbResults                                    # encode and decode would pick different maps
        dev = next(iter(enc.values())).device
        params, layers = self._pack(dev)
        ws = hip.Workspace.get(batch, H, W, dev)
        with torch.cuda.device(dev):
            self._load_cond(ws, enc, batch, H, W)
            hip.check(hip.lib().ifl_cond_prepare(layers, len(layers), hip._ptr(params), ws.ptr, batch, H, W,
                                                 float(self.affine_eps), self._flags(), hip._stream()), "ifl_cond_prepare")
            ws.mark_used()
        return PreparedCond(rrdbResults, ws, ws.generation, (self._packed[0], self.conv_dtype), enc, (batch, H, W))

    def _needs_autograd(self, src, rrdbResults):
        """True when a gradient has to flow through this call: then the differentiable layer-by-layer path runs
        (train_flow.py) instead of the fused inference kernels, which record nothing for autograd."""
        if not torch.is_grad_enabled():
            return False
        if isinstance(src, torch.Tensor) and src.requires_grad:
            return True
        if any(isinstance(v, torch.Tensor) and v.requires_grad for v in rrdbResults.values()):
            return True
        return any(p.requires_grad for p in self.layers.parameters())

    def _actnorm_init_due(self):
        """Training mode with ActNorms whose data-dependent initialisation is still due (FlowActNorms.py:29-43; flow.py:57-63):
        the reference initialises them inside the first training-mode forward, with or without autograd; so does the
        layer-by-layer evaluation, which this pass then takes (the fused driver packs finished parameters only)."""
        if not self.training:
            return False
        from .training import _uninitialised_actnorms
        return bool(_uninitialised_actnorms(self))

    def _run(self, src, rrdbResults, logdet, test_input_size, decode, epses=None, y_onehot=None, eps_std=None):
        if self._general or self._needs_autograd(src, rrdbResults) or self._actnorm_init_due():
            from . import train_flow
            out, ld = train_flow.run(self, src, rrdbResults, logdet, test_input_size, decode, epses=epses, y_onehot=y_onehot,
                                     eps_std=eps_std)
            if decode:
                assert out.shape[1] == 3
            return out, ld
        src = hip._dev_f32(src, "flow input")
        B = src.shape[0]
        if decode:
            assert src.shape[1] == self.C, (src.shape[1], self.C)
            H, W = src.shape[2] * 4, src.shape[3] * 4
            out = torch.empty(B, 3, H, W, dtype=torch.float32, device=src.device)
        else:
            assert src.shape[1] == 3, src.shape
            H, W = src.shape[2], src.shape[3]
            assert H % 4 == 0 and W % 4 == 0, "{}".format((H, W, 2))                      # flow.py:172
            out = torch.empty(B, self.C, H // 4, W // 4, dtype=torch.float32, device=src.device)
        feats = self._level_features(rrdbResults, H, test_input_size, decode)
        params, layers = self._pack(src.device)
        L = hip.lib()
        ld_in = _as_logdet(logdet, B, src.device)
        ld_out = torch.empty(B, dtype=torch.float32, device=src.device)
        with torch.cuda.device(src.device):
            ws, prepared = None, False
            if isinstance(rrdbResults, PreparedCond):
                ws = hip.Workspace.get(B, H, W, src.device)           # where prepare_cond put it
                prepared = rrdbResults.usable(ws, (self._packed[0], self.conv_dtype), feats, (B, H, W))
            flags = self._flags()
            if prepared:
                flags |= hip.FLOW_COND_PREPARED
            else:
                # two workspaces in turn: the next pass may fill one while this pass's chain still reads the other
                overlap = hip.set_cross_pass_overlap() and not torch.cuda.is_current_stream_capturing()
                self._pass_no += 1
                ws = hip.Workspace.get(B, H, W, src.device, slot=self._pass_no & 1 if overlap else 0)
                early = overlap and self._cond_unchanged(feats)
                self._load_cond(ws, feats, B, H, W, early=early)
                self._remember_cond(feats)
                if early:
                    flags |= hip.FLOW_COND_EARLY
            hip.check(L.ifl_flow_pass(layers, len(layers), hip._ptr(params), ws.ptr, hip._ptr(src), hip._ptr(out),
                                      hip._ptr(ld_in), hip._ptr(ld_out), B, H, W, float(self.affine_eps),
                                      1 if decode else 0, flags, hip._stream()), "ifl_flow_pass")
            ws.mark_used()
        return out, ld_out

    def encode(self, gt, rrdbResults, logdet=0.0, epses=None, y_onehot=None, test_input_size=None):
        """:217-255.  With `epses` a list, every Split2d's eps is appended to it and the list + [z] is returned (:251-255)."""
        z, logdet = self._run(gt, rrdbResults, logdet, test_input_size, decode=False, epses=epses, y_onehot=y_onehot)
        if not isinstance(epses, list):
            return z, logdet
        epses.append(z)
        return epses, logdet

    def decode(self, rrdbResults, z, eps_std=None, epses=None, logdet=0.0, y_onehot=None, test_input_size=None):
        """:271-304.  A list `epses` supplies z (its last item, :272) and the eps of every Split2d (popped, :306-311); without
        one a Split2d draws its eps (Split.py:92-103)."""
        z = epses.pop() if isinstance(epses, list) else z
        sr, logdet = self._run(z, rrdbResults, logdet, test_input_size, decode=True, epses=epses, y_onehot=y_onehot,
                               eps_std=eps_std)
        assert sr.shape[1] == 3                                                           # :303
        return sr, logdet
