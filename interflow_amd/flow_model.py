"""Drop-in for the reference's model API (`interflow/models/flow_model.py`, `RRDBGlowNet_arch.py`,
`RRDBNet_arch.py`): same constructor, attributes, method signatures, return values and state-dict keys.

`InterpFlow.forward` / `interpolate_image` / `weighted_interpolate_image` / `calculate_loss` / `get_pdf`
run the flow, the latent prior and the latent interpolation through libinterflow_hip.so.  The
conditioning encoder (`RRDBNet`) runs through `ifl_rrdb_encode` (inference) or, when autograd
must reach its parameters, through the differentiable HIP convolution; there is no eager-PyTorch or CPU evaluation.
"""
from __future__ import annotations

import functools

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip
from .flow_modules import FlowUpsamplerNet, opt_get


# ------------------------------------------------------------------------------------- conditioning encoder
class ResidualDenseBlock_5C(nn.Module):
    """RRDBNet_arch.py:9-29"""

    def __init__(self, nf=64, gc=32, bias=True):
        super().__init__()
        self.conv1 = nn.Conv2d(nf, gc, 3, 1, 1, bias=bias)
        self.conv2 = nn.Conv2d(nf + gc, gc, 3, 1, 1, bias=bias)
        self.conv3 = nn.Conv2d(nf + 2 * gc, gc, 3, 1, 1, bias=bias)
        self.conv4 = nn.Conv2d(nf + 3 * gc, gc, 3, 1, 1, bias=bias)
        self.conv5 = nn.Conv2d(nf + 4 * gc, nf, 3, 1, 1, bias=bias)
        self.lrelu = nn.LeakyReLU(negative_slope=0.2, inplace=True)
        for m in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5):      # module_util.py:7-16, scale 0.1
            nn.init.kaiming_normal_(m.weight, a=0, mode='fan_in')
            m.weight.data *= 0.1
            m.bias.data.zero_()

    def forward(self, x):
        x1 = self.lrelu(self.conv1(x))
        x2 = self.lrelu(self.conv2(torch.cat((x, x1), 1)))
        x3 = self.lrelu(self.conv3(torch.cat((x, x1, x2), 1)))
        x4 = self.lrelu(self.conv4(torch.cat((x, x1, x2, x3), 1)))
        x5 = self.conv5(torch.cat((x, x1, x2, x3, x4), 1))
        return x5 * 0.2 + x


class RRDB(nn.Module):
    """RRDBNet_arch.py:32-45"""

    def __init__(self, nf, gc=32):
        super().__init__()
        self.RDB1 = ResidualDenseBlock_5C(nf, gc)
        self.RDB2 = ResidualDenseBlock_5C(nf, gc)
        self.RDB3 = ResidualDenseBlock_5C(nf, gc)

    def forward(self, x):
        return self.RDB3(self.RDB2(self.RDB1(x))) * 0.2 + x


class RRDBNet(nn.Module):
    """RRDBNet_arch.py:48-148 for scale 4."""

    def __init__(self, in_nc, out_nc, nf, nb, gc=32, scale=4, opt=None):
        super().__init__()
        self.opt = opt
        self.scale = scale
        if scale != 4:
            raise NotImplementedError("only scale 4 is in the shipped config")
        self.conv_first = nn.Conv2d(in_nc * 16, nf, 3, 1, 1, bias=True)
        self.RRDB_trunk = nn.Sequential(*[RRDB(nf=nf, gc=gc) for _ in range(nb)])
        self.trunk_conv = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.upconv1 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.upconv2 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.HRconv = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.conv_last = nn.Conv2d(nf, out_nc, 3, 1, 1, bias=True)

    @staticmethod
    def space2depth(x, block_size):
        n, c, h, w = x.size()                                                         # :79-83
        return F.unfold(x, block_size, stride=block_size).view(n, c * block_size ** 2, h // block_size, w // block_size)

    def _forward_hipconv(self, x, get_steps):
        """The same forward with every nn.Conv2d evaluated by the differentiable HIP convolution (train_ops.conv2d_ring:
        forward, data gradient and weight gradient on the library's kernels) on ringed channels-last tensors; the
        activations, residuals, concatenations and resizes are ATen glue.  Used when autograd must reach the encoder."""
        from . import train_ops as T

        def conv(m, t, cin):
            return T.conv2d_ring([t], [cin], m.weight, m.bias)

        def lrelu(t):
            return F.leaky_relu(t, 0.2)

        def rdb(m, t):                                                            # RRDBNet_arch.py:23-29
            x1 = lrelu(conv(m.conv1, t, 64))
            x2 = lrelu(conv(m.conv2, torch.cat((t, x1), -1), 96))
            x3 = lrelu(conv(m.conv3, torch.cat((t, x1, x2), -1), 128))
            x4 = lrelu(conv(m.conv4, torch.cat((t, x1, x2, x3), -1), 160))
            x5 = conv(m.conv5, torch.cat((t, x1, x2, x3, x4), -1), 192)
            return x5 * 0.2 + t

        def up2(t):                                                               # nearest x2 of a ringed tensor
            i = t[:, 1:-1, 1:-1, :]
            return T.ring_of(i.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2))

        nf = self.conv_first.weight.shape[0]
        fea = conv(self.conv_first, T.to_ring(self.space2depth(x, 4)), self.conv_first.weight.shape[1])
        block_idxs = opt_get(self.opt, ['network_G', 'flow', 'stackRRDB', 'blocks']) or []
        block_results = {}
        for idx, m in enumerate(self.RRDB_trunk.children()):
            fea = rdb(m.RDB3, rdb(m.RDB2, rdb(m.RDB1, fea))) * 0.2 + fea            # :41-45
            if idx in block_idxs:
                block_results["block_{}".format(idx)] = T.from_ring(fea, nf)
        last = fea + conv(self.trunk_conv, fea, nf)
        f2 = lrelu(conv(self.upconv1, up2(last), nf))
        f4 = lrelu(conv(self.upconv2, up2(f2), nf))
        out = T.from_ring(conv(self.conv_last, lrelu(conv(self.HRconv, f4, nf)), nf), self.conv_last.weight.shape[0])
        last_lr_fea, fea_up2, fea_up4 = T.from_ring(last, nf), T.from_ring(f2, nf), T.from_ring(f4, nf)
        results = {'last_lr_fea': last_lr_fea, 'fea_up1': last_lr_fea, 'fea_up2': fea_up2, 'fea_up4': fea_up4,
                   'fea_up8': None, 'fea_up16': None, 'fea_up32': None, 'out': out}
        if opt_get(self.opt, ['network_G', 'flow', 'fea_up0']) or False:
            results['fea_up0'] = F.interpolate(last_lr_fea, scale_factor=1 / 2, mode='bilinear')
        if opt_get(self.opt, ['network_G', 'flow', 'fea_up-1']) or False:
            results['fea_up-1'] = F.interpolate(last_lr_fea, scale_factor=1 / 4, mode='bilinear')
        if get_steps:
            results.update(block_results)
            return results
        return out

    def forward(self, x, get_steps=False):
        """RRDBNet_arch.py:85-148 with autograd: every convolution on the HIP kernels (`_forward_hipconv`).  There is no
        eager-PyTorch / CPU evaluation of the encoder in this package: other devices and dtypes raise."""
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32):
            raise hip.InterflowHipError(
                f"RRDBNet.forward: expected a float32 GPU tensor, got {getattr(x, 'dtype', type(x))} on "
                f"{getattr(x, 'device', '?')}; the HIP path has no CPU or other-dtype fallback")
        return self._forward_hipconv(x, get_steps)


RRDB_TRAIN_FUSED = True        # tests switch it off to compare with the convolution-by-convolution evaluation


class RRDBGlowNet(nn.Module):
    """RRDBGlowNet_arch.py:16-59,217-246.  The DeFlow-legacy normal_flow/reverse_flow (:81-214) are not called
    by InterFlow's drivers and are not provided."""

    def __init__(self, in_nc, out_nc, nf, nb, gc=32, scale=4, K=None, opt=None, step=None):
        super().__init__()
        self.opt = opt
        self.RRDB = RRDBNet(in_nc, out_nc, nf, nb, gc, scale, opt)
        hidden_channels = opt_get(opt, ['network_G', 'flow', 'hidden_channels']) or 64
        self.RRDB_training = True
        self.flowUpsamplerNet = FlowUpsamplerNet((160, 160, 3), hidden_channels, K,
                                                 flow_coupling=opt['network_G']['flow']['coupling'],
                                                 LU_decomposed=opt_get(opt, ['network_G', 'flow', 'LU_decomposed'], False),
                                                 opt=opt)
        C = self.flowUpsamplerNet.C
        self.I = nn.Parameter(torch.eye(C), requires_grad=False)
        self.mean_shift = nn.Parameter(torch.zeros(C))
        self.cov_shift = nn.Parameter(torch.eye(C) * opt_get(opt, ['network_G', 'flow', 'shift', 'std_init_shift'], 1.0))

    def set_rrdb_training(self, trainable):
        if self.RRDB_training != trainable:
            for p in self.RRDB.parameters():
                p.requires_grad = trainable
            self.RRDB_training = trainable
            return True
        return False

    def forward(self, *a, **k):
        raise NotImplementedError("RRDBGlowNet.forward (DeFlow legacy) is not used by InterFlow; call InterpFlow")

    def _needs_autograd(self):
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.RRDB.parameters())

    def _rrdb_hip(self, lr, outputs=None):
        """RRDBNet.forward(get_steps=True) + the stacking below, through ifl_rrdb_encode (inference only).  outputs: the
        feature maps the caller will read (None: all, as the reference); the others are None in the result and the layers
        only they need (upconv2, HRconv, conv_last for 'fea_up4' / 'out') are not run."""
        import ctypes
        from . import pack
        lr = hip._dev_f32(lr, "rrdbPreprocessing input")
        B, C, H, W = lr.shape
        assert C == 3, C
        assert H % 4 == 0 and W % 4 == 0, "{}".format((H, W, 4))           # F.unfold(..., 4).view(...) would fail
        block_idxs = list(opt_get(self.opt, ['network_G', 'flow', 'stackRRDB', 'blocks']) or [])
        nb = len(self.RRDB.RRDB_trunk)
        key = (tuple((p.data_ptr(), p._version) for p in self.RRDB.parameters()), str(lr.device))
        if getattr(self, "_rrdb_packed", None) is None or self._rrdb_packed[0] != key:
            prm = {k: v.detach().to("cpu", torch.float32).numpy() for k, v in self.RRDB.state_dict().items()}
            self._rrdb_packed = (key, torch.from_numpy(pack.pack_rrdb(prm, nb)).to(lr.device))
        params = self._rrdb_packed[1]
        L = hip.lib()
        wkey = (B, H, W, len(block_idxs), lr.device.index, torch.cuda.current_stream().cuda_stream)
        cache = self.__dict__.setdefault("_rrdb_ws", {})
        if wkey not in cache:
            if len(cache) >= 4:
                cache.pop(next(iter(cache)))
            n = L.ifl_rrdb_ws_bytes(B, H, W, len(block_idxs))
            if n < 0:
                hip.check(int(n), "ifl_rrdb_ws_bytes")
            buf = torch.empty(n, dtype=torch.uint8, device=lr.device)
            with torch.cuda.device(lr.device):
                hip.check(L.ifl_rrdb_ws_init(hip._ptr(buf), n, B, H, W, len(block_idxs), hip._stream()), "ifl_rrdb_ws_init")
            cache[wkey] = buf
        ws = cache[wkey]
        Ct = 64 * (1 + len(block_idxs))

        def new(c, h, w):
            return torch.empty(B, c, h, w, dtype=torch.float32, device=lr.device)
        want = (lambda k: True) if outputs is None else (lambda k: k in outputs)
        up1 = new(Ct, H // 4, W // 4)
        up2 = new(Ct, H // 2, W // 2) if want('fea_up2') else None
        up4 = new(Ct, H, W) if want('fea_up4') else None
        up0 = None
        if opt_get(self.opt, ['network_G', 'flow', 'fea_up0']) or False:
            if H // 8 == 0 or W // 8 == 0:                  # F.interpolate(scale_factor=1/2) of RRDBNet_arch.py:138 raises
                raise RuntimeError(f"Input and output sizes should be greater than 0, but got input (H: {H // 4}, "
                                   f"W: {W // 4}) output (H: {H // 8}, W: {W // 8})")
            up0 = new(Ct, H // 8, W // 8)
        out = new(3, H, W) if want('out') else None
        blocks = torch.empty(len(block_idxs), B, 64, H // 4, W // 4, dtype=torch.float32, device=lr.device)
        idx = (ctypes.c_int * max(len(block_idxs), 1))(*block_idxs)
        with torch.cuda.device(lr.device):
            hip.check(L.ifl_rrdb_encode(hip._ptr(params), nb, idx, len(block_idxs), hip._ptr(ws), hip._ptr(lr), hip._ptr(up1),
                                        hip._ptr(up2), hip._ptr(up4), hip._ptr(up0), hip._ptr(out), hip._ptr(blocks),
                                        B, H, W, hip._stream()), "ifl_rrdb_encode")
        res = {'last_lr_fea': up1, 'fea_up1': up1, 'fea_up2': up2, 'fea_up4': up4, 'fea_up8': None, 'fea_up16': None,
               'fea_up32': None, 'out': out}
        if up0 is not None:
            res['fea_up0'] = up0
        for i, k in enumerate(block_idxs):
            res["block_{}".format(k)] = blocks[i]
        return res

    def rrdbPreprocessing(self, lr, outputs=None):
        """RRDBGlowNet_arch.py:217-246.  outputs (an extension; default None = everything, as the reference): the names among
        'fea_up1', 'fea_up2', 'fea_up4', 'out' the caller will read -- the K=16, L=2 flow only reads 'fea_up1' and 'fea_up2', and
        the full-resolution tail (upconv2, HRconv, conv_last and a 262 MB 'fea_up4' at batch 8) is 6 % of the encoder pass."""
        concat_on = opt_get(self.opt, ['network_G', 'flow', 'stackRRDB', 'concat']) or False
        upn1 = opt_get(self.opt, ['network_G', 'flow', 'fea_up-1']) or False
        if not self._needs_autograd() and concat_on and not upn1:
            return self._rrdb_hip(lr, outputs)
        # autograd has to reach the encoder's parameters
        from . import train_rrdb
        if RRDB_TRAIN_FUSED and concat_on and train_rrdb.usable(self, lr, outputs):
            res = train_rrdb.encode(self, lr, outputs)         # forward and backward inside the library (csrc/rrdb.hip)
            if res is not None:
                return res
        # the differentiable evaluation convolution by convolution (every convolution on the HIP kernels)
        if not lr.is_cuda:
            raise hip.InterflowHipError(f"rrdbPreprocessing: tensor is on {lr.device}; there is no CPU path")
        rrdbResults = self.RRDB(lr, get_steps=True)
        block_idxs = opt_get(self.opt, ['network_G', 'flow', 'stackRRDB', 'blocks']) or []
        if len(block_idxs) > 0:
            concat = torch.cat([rrdbResults["block_{}".format(idx)] for idx in block_idxs], dim=1)
            if opt_get(self.opt, ['network_G', 'flow', 'stackRRDB', 'concat']) or False:
                keys = ['last_lr_fea', 'fea_up1', 'fea_up2', 'fea_up4']
                if 'fea_up0' in rrdbResults.keys():
                    keys.append('fea_up0')
                if 'fea_up-1' in rrdbResults.keys():
                    keys.append('fea_up-1')
                for k in keys:
                    h, w = rrdbResults[k].shape[2], rrdbResults[k].shape[3]
                    rrdbResults[k] = torch.cat([rrdbResults[k], F.interpolate(concat, (h, w))], dim=1)
        return rrdbResults


class DegradationNet(nn.Module):
    """flow_model.py:13-35 -- instantiated by the reference (:56) and never called; kept for the state dict."""

    def __init__(self, in_ch=320, out_ch=48, hidden_ch=64, L=3):
        super().__init__()
        layers = []
        for i in range(L):
            cin = in_ch if i == 0 else hidden_ch
            cout = out_ch if i == L - 1 else (64 if i == 0 else hidden_ch)
            layers += [nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1), nn.ReLU(inplace=True)]
        self.net = nn.Sequential(*layers)

    def forward(self, x):
        return self.net(x) + 0.5


# ------------------------------------------------------------------------------------- prior
def _h2d(t, device):
    """Host tensor -> device without a host/device synchronisation (pinned staging + non_blocking copy; a pageable copy blocks
    the host until the device has drained the stream)."""
    if t.device.type != "cpu" or device.type == "cpu":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


class Gaussian:
    """flow.py:138-161 on the HIP path."""
    Log2PI = float(np.log(2 * np.pi))

    @staticmethod
    def inverse_of(cov_fn, key, mean, C, device, tag=""):
        """(mean [C], cov_inv [C, C], ln|cov|) for the inference kernel, on `device`, cached per VERSION of the parameter
        `key` the covariance is made of (`cov_fn()` builds it): flow.py:146-149 inverts the covariance on every call, here
        the inverse (float64 `inv_ex` on the device, rounded once) and the log-determinant are computed when the parameter
        has changed and kept, so a loss evaluation in steady state reads nothing back from the device (one read of the
        log-determinant per parameter update; the kernel takes it by value)."""
        ver = (key._version, mean._version if isinstance(mean, torch.Tensor) else 0)
        ent = key.__dict__.get("_ifl_prior")             # kept on the parameter object itself, like the packed weights
        if ent is not None and ent[0] == ver and ent[1] == device and ent[2] == tag:
            return ent[3:]
        c64 = cov_fn().detach().to(device=device, dtype=torch.float64)
        inv = torch.linalg.inv_ex(c64)[0].to(torch.float32).contiguous()
        logdet = float(torch.linalg.slogdet(c64)[1])
        mean_t = mean.detach().to(device=device, dtype=torch.float32).reshape(-1)
        if mean_t.numel() == 1:
            mean_t = mean_t.expand(C)
        mean_t = mean_t.contiguous()
        key.__dict__["_ifl_prior"] = (ver, device, tag, mean_t, inv, logdet)
        return mean_t, inv, logdet

    @staticmethod
    def logp(mean, cov, x, key=None):
        """key: the parameter `cov` was computed from, when the caller has one -- the inverse is then cached per parameter
        version (`inverse_of`); without it the covariance is inverted on the host on every call, as the reference does."""
        if torch.is_grad_enabled() and (x.requires_grad or (cov is not None and (cov.requires_grad or mean.requires_grad))):
            from . import train_flow
            return train_flow.gaussian_logp(mean, cov, x)              # differentiable (training path)
        if mean is None and cov is None:
            return hip.gaussian_logp(x)
        if key is not None:
            mean_t, cov_inv, logdet = Gaussian.inverse_of(lambda: cov, key, mean, x.shape[1], x.device)
            return hip.gaussian_logp(x, mean_t, cov_inv, logdet)
        # torch.inverse / torch.slogdet of the 48x48 covariance: tiny host-side linear algebra, as the reference
        cov_c = cov.detach().to("cpu", torch.float32)
        cov_inv = torch.inverse(cov_c)
        logdet = float(torch.slogdet(cov_c)[1])
        mean_t = mean.detach().to(device=x.device, dtype=torch.float32).reshape(-1)
        if mean_t.numel() == 1:
            mean_t = mean_t.expand(x.shape[1])
        return hip.gaussian_logp(x, mean_t.contiguous(), cov_inv.to(x.device).contiguous(), logdet)


# ------------------------------------------------------------------------------------- model API
class InterpFlow(nn.Module):
    """flow_model.py:37-224"""

    def __init__(self, scales: list, mean_init: float = 0.001, std_init: float = 1.0):
        super().__init__()
        opt = {'scale': 4,
               'network_G': {'in_nc': 3, 'out_nc': 3, 'nf': 64, 'nb': 23, 'upscale': 4, 'train_RRDB': True, 'scale': 4,
                             'flow': {'K': 16, 'L': 2, 'noInitialInj': True, 'LU_decomposed': True,
                                      'coupling': 'CondAffineSeparatedAndCond', 'additionalFlowNoAffine': 2,
                                      'split': {'enable': True}, 'fea_up0': True,
                                      'stackRRDB': {'blocks': [1, 8, 15, 22], 'concat': True},
                                      'CondAffineSeparatedAndCond': {'eps': 0.001, 'multReverse': True}}}, }
        self.scale = 4
        self.arch = RRDBGlowNet(in_nc=3, out_nc=3, nf=64, nb=23, scale=self.scale, K=16, opt=opt, step=0)
        self.flow_upsampler = self.arch.flowUpsamplerNet
        self.rrdb = self.arch.RRDB
        self.degradation_net = DegradationNet()
        self.quant = 256
        self.classes = scales
        self.C = self.flow_upsampler.C
        self.I = nn.Parameter(torch.eye(self.C), requires_grad=False)
        num_class = len(self.classes)
        mean_interval = torch.linspace(-mean_init * (num_class // 2), mean_init * (num_class // 2), num_class)
        self.mean_shifts = [nn.Parameter(torch.zeros(1) + mean_interval[i]) for i in range(num_class)]
        self.cov_shifts = [nn.Parameter(torch.eye(self.C) * std_init) for _ in range(num_class)]
        for i, p in enumerate(self.mean_shifts):
            self.register_parameter(f"mean_shift_{i}", p)
        for i, p in enumerate(self.cov_shifts):
            self.register_parameter(f"cov_shift_{i}", p)

    def _apply(self, fn, *a, **k):
        # .to()/.cuda() replace Parameter objects' data in place, so the python lists stay valid; refresh anyway
        r = super()._apply(fn, *a, **k)
        self.mean_shifts = [getattr(self, f"mean_shift_{i}") for i in range(len(self.classes))]
        self.cov_shifts = [getattr(self, f"cov_shift_{i}") for i in range(len(self.classes))]
        return r

    def forward(self, input_value, cond_image, label=None, reverse=False, test_input_size=None, lr_enc=None,
                add_gt_noise=True, calc_loss=True):
        if reverse:
            if input_value is None:
                raise NotImplementedError
            lr, z = cond_image, input_value
            if lr_enc is None:
                lr_enc = self.arch.rrdbPreprocessing(lr)
            logdet = torch.zeros_like(lr[:, 0, 0, 0])
            if add_gt_noise:
                pixels = int(lr.size(2) * lr.size(3))
                logdet = logdet - float(-np.log(self.quant) * pixels)
            return self.flow_upsampler(rrdbResults=lr_enc, z=z, logdet=logdet, reverse=True, epses=None,
                                       y_onehot=None, test_input_size=test_input_size)
        lr, gt = cond_image, input_value
        logdet = torch.zeros_like(gt[:, 0, 0, 0])
        pixels = int(gt.size(2) * gt.size(3))
        if lr_enc is None:
            lr_enc = self.arch.rrdbPreprocessing(lr)
        if add_gt_noise:
            gt = gt + ((torch.rand(gt.shape, device=gt.device) - 0.5) / self.quant)
            logdet = logdet + float(-np.log(self.quant) * pixels)
        z, logdet = self.flow_upsampler(rrdbResults=lr_enc, gt=gt, logdet=logdet, reverse=False, epses=None,
                                        y_onehot=None)
        if not calc_loss:
            return z, logdet
        return self.calculate_loss(z, logdet, pixels, label)

    def _shifted(self, device):
        covs = [torch.matmul(c, c.T) for c in self.cov_shifts]
        means = [m * torch.ones(self.C, device=m.device) for m in self.mean_shifts]
        return means, covs

    def _prior(self, i, device):
        """(mean [C], cov_inv, ln|cov|) of class i's prior (flow_model.py:138-141: cov = cov_shift cov_shift^T, mean =
        mean_shift * 1), cached per version of the two parameters: no host read in steady state."""
        c = self.cov_shifts[i]
        return Gaussian.inverse_of(lambda: torch.matmul(c, c.T), c, self.mean_shifts[i], self.C, device)

    def _class_table(self, device):
        """The class scales as a device tensor (uploaded once per device: no host round trip per loss evaluation)."""
        t = self.__dict__.get("_classes_dev")
        if t is None or t.device != device:
            t = torch.tensor(np.asarray(self.classes), device=device)
            self.__dict__["_classes_dev"] = t
        return t

    def _class_loglik(self, z, label):
        """flow_model.py:143-160: (log-density of every sample under every class prior [n, cls], the label mask, the
        log-density `ll` the objective uses).  The reference ASSIGNS ll[label == c] per class, and the 'X' (label 0) rows are
        written again by the second loop, so a label-0 sample ends with the standard-normal log-density even when 0 is one
        of the classes.  No host tests on device data here (the reference's `any()` branch is a `where`)."""
        label = _h2d(label, z.device)
        if torch.is_grad_enabled() and (z.requires_grad or any(c.requires_grad for c in self.cov_shifts) or any(m.requires_grad for m in self.mean_shifts)):
            from . import train_flow
            means, covs = self._shifted(z.device)
            ll_all = train_flow.gaussian_logp_classes(torch.stack(means), torch.stack(covs), z)                  # (n, cls), all classes at once
        else:
            ll_all = torch.stack([hip.gaussian_logp(z, *self._prior(i, z.device)) for i in range(len(self.classes))], dim=1)  # (n, cls)
        mask_label = self._class_table(z.device)[None, :] == label.reshape(-1, 1)
        ll = torch.zeros(z.shape[0], device=z.device)
        for i in range(len(self.classes)):
            ll = torch.where(mask_label[:, i], ll_all[:, i], ll)
        ll = torch.where(label == 0, Gaussian.logp(None, None, z), ll)
        return ll_all, mask_label, ll

    def calculate_loss(self, epses, logdet, pixels, label):
        objective = logdet.clone()
        z = epses[-1] if isinstance(epses, (list, tuple)) else epses
        if type(label) != torch.Tensor:
            label = torch.Tensor(label)
        assert z.shape[0] == label.shape[0], 'need one class label per datapoint'
        assert len(label.shape) == 1, 'labels must be one dimensional'
        ll_all, mask_label, ll = self._class_loglik(z, label)
        ll_ib = ll_all / float(pixels)
        ll_ib = torch.log_softmax(ll_ib, dim=1)
        if label.device.type == "cpu" and z.device.type != "cpu":
            # labels that live on the host (train.py:69 builds them there): the selected (sample, class) pairs are known without
            # asking the device, so the selection is a gather with host-made indices instead of masked_select, whose output
            # size is a device-to-host read
            rows, cols = np.nonzero(np.asarray(self.classes)[None, :] == label.numpy().reshape(-1, 1))
            ll_ib = ll_ib[_h2d(torch.from_numpy(rows), z.device), _h2d(torch.from_numpy(cols), z.device)]
        else:
            ll_ib = torch.masked_select(ll_ib, mask_label)
        objective = objective + ll
        nll = (-objective) / float(np.log(2.) * pixels)
        if isinstance(epses, list):
            return epses, nll, logdet
        return z, nll, logdet, ll_ib

    def loss_terms(self, z, logdet, pixels, label):
        """(nll per sample, ib loss) = calculate_loss followed by train.py:74's -mean(ll_ib), with the mean over the selected
        entries written as a masked sum: masked_select's output shape depends on the data, i.e. a device-to-host round trip
        in the middle of the step, which keeps the host from issuing the backward pass while the forward still runs.
        Same values as calculate_loss (no selected entry at all: 0/0 = NaN, like the mean of an empty tensor)."""
        ll_all, mask_label, ll = self._class_loglik(z, label)
        ls = torch.log_softmax(ll_all / float(pixels), dim=1)
        ib_loss = -(ls * mask_label).sum() / mask_label.sum()
        nll = (-(logdet + ll)) / float(np.log(2.) * pixels)
        return nll, ib_loss

    def get_lr(self, input_lr, heat=1., test_input_size=None):
        z_shape = (input_lr.shape[0], input_lr.shape[1] * 16, input_lr.shape[2], input_lr.shape[3])
        z = torch.normal(mean=0, std=heat, size=z_shape).to(input_lr.device)
        return self.forward(z, input_lr, reverse=True, test_input_size=test_input_size)[0]

    def interpolate_image(self, input_lr1, scale1, input_lr2, scale2, target_scale, cond_gt):
        scale1 = torch.tensor(scale1).double()
        scale2 = torch.tensor(scale2).double()
        target_scale = torch.tensor(target_scale).double()
        lr_enc = self.arch.rrdbPreprocessing(cond_gt)          # computed once, the reference recomputes it 3 times
        lr_enc = self.flow_upsampler.prepare_cond(lr_enc, input_lr1.shape[0], input_lr1.shape[2], input_lr1.shape[3])
        z1, _ = self.forward(input_lr1, cond_gt, lr_enc=lr_enc, calc_loss=False)
        z2, _ = self.forward(input_lr2, cond_gt, lr_enc=lr_enc, calc_loss=False)
        a = (target_scale - scale2) / (scale1 - scale2)
        target_z = hip.lerp(z1, z2, float(a), float(1 - a))
        return self.forward(target_z, cond_gt, reverse=True, test_input_size=input_lr1.shape[2], lr_enc=lr_enc,
                            calc_loss=False)[0]

    def weighted_interpolate_image(self, input_lr1, input_lr2, a, cond_gt, add_noise=False):
        lr_enc = self.arch.rrdbPreprocessing(cond_gt)
        lr_enc = self.flow_upsampler.prepare_cond(lr_enc, input_lr1.shape[0], input_lr1.shape[2], input_lr1.shape[3])
        z1, _ = self.forward(input_lr1, cond_gt, lr_enc=lr_enc, add_gt_noise=add_noise, calc_loss=False)
        z2, _ = self.forward(input_lr2, cond_gt, lr_enc=lr_enc, add_gt_noise=add_noise, calc_loss=False)
        target_z = hip.lerp(z1, z2, 1 - a, a)
        return self.forward(target_z, cond_gt, reverse=True, add_gt_noise=add_noise, lr_enc=lr_enc,
                            test_input_size=input_lr1.shape[2], calc_loss=False)[0]

    def get_pdf(self, z: torch.Tensor, scales: list):
        pixels = int(z.size(2) * z.size(3))
        ll = []
        for scale in scales:
            index = self.classes.index(scale)
            ll.append(hip.gaussian_logp(z, *self._prior(index, z.device)) / float(np.log(2.) * pixels))
        return torch.stack(ll, dim=0).reshape(-1)

    def get_shifted_info(self):
        if len(self.mean_shifts) != 0:
            return np.stack([m.detach().cpu() for m in self.mean_shifts], axis=0), \
                np.stack([c.detach().cpu() for c in self.cov_shifts], axis=0), self.classes
        return None, None, None
