"""Deterministic weight / input synthesiser for tests and benchmarks.

There is no network in the build or on the GPU box, so the pretrained InterFlow checkpoint
(reference `interflow/README.md:15`) is not available.  This module produces a state dict with
exactly the reference's key names and shapes (reference `interflow/models/flow_model.py:53-80`
registers the flow and the RRDB encoder twice, under `arch.*` and under `flow_upsampler.*` /
`rrdb.*`; both aliases are emitted here) from a numpy seed, so the same weights exist in the
authoring container (where they are loaded into the imported reference to make golden vectors)
and on the GPU box (where they are loaded into the HIP path).

Everything random is drawn in float64 from `numpy.random.default_rng` and cast to float32 once.
The distributions mirror the reference's initialisers where that gives a numerically sane flow:

* invertible 1x1 conv: QR of a Gaussian matrix, then LU, exactly the recipe of
  reference `Permutations.py:48-65` (in float64, cast at the end);
* coupling-network `Conv2d`: N(0, 0.05^2) weights (reference `flow.py:53`); their ActNorm
  parameters are set analytically to what the reference's data-dependent init
  (`FlowActNorms.py:29-43`) would approximately give for unit-variance inputs, instead of zeros;
* `Conv2dZeros`: the reference zero-initialises these (`flow.py:77-78`), which makes every
  coupling constant; here they get small N(0, sigma^2) values so scale/shift vary per pixel;
* step-level ActNorm: log-scale chosen so that |z| stays O(1) through the 32 couplings.

This is test/bench infrastructure for the product path; it never touches `oracle/`.
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict

import numpy as np

HIDDEN = 64          # coupling-network width (reference FlowAffineCouplingsAblation.py:133)
N_RRDB_CH = 320      # conditioning channels (reference FlowUpsamplerNet.py:117-120: (4+1)*64)
K_STEPS = 16         # reference flow_model.py:42
N_LEVELS = 2         # reference flow_model.py:42
N_EXTRA = 2          # additionalFlowNoAffine, reference flow_model.py:43


def layer_plan(L: int = N_LEVELS, K: int = K_STEPS, extra: int = N_EXTRA, in_ch: int = 3, split=None):
    """List of (kind, C) in the order of `FlowUpsamplerNet.layers`
    (reference FlowUpsamplerNet.py:94-105): per level squeeze, `extra` no-coupling steps,
    K conditional steps.  kind in {"squeeze", "step0", "stepc"}; C = channels after the layer.
    split: None (the shipped configuration creates no Split2d), or {"correct_splits": bool}: a ("split", C entering it) after
    the steps of every level < L - (0 if correct_splits else 1) (FlowUpsamplerNet.py:149-167); half the channels go on."""
    plan = []
    C = in_ch
    for level in range(1, L + 1):
        C *= 4
        plan.append(("squeeze", C))
        for _ in range(extra):
            plan.append(("step0", C))
        for _ in range(K):
            plan.append(("stepc", C))
        if split is not None and level < L - (0 if split.get("correct_splits") else 1):
            plan.append(("split", C))
            C = C - int(round(C * 0.5))
    return plan


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).astype(np.float32))


def _invconv_params(rng, C):
    """reference Permutations.py:48-65, float64 until the final cast."""
    import scipy.linalg
    w_init = np.linalg.qr(rng.standard_normal((C, C)))[0]
    p, l, u = scipy.linalg.lu(w_init)
    s = np.diag(u)
    return {
        "p": _f32(p),
        "sign_s": _f32(np.sign(s)),
        "l": _f32(l),
        "log_s": _f32(np.log(np.abs(s))),
        "u": _f32(np.triu(u, k=1)),
    }


def _coupling_net(rng, prefix, cin, cout, in_ms, sd, zeros_std, zeros_bias_std=None):
    """One `F(...)` stack of reference FlowAffineCouplingsAblation.py:248-256.
    in_ms: assumed mean-square of each input channel (array of len cin), used to pick the
    ActNorm log-scale analytically."""
    w0 = rng.standard_normal((HIDDEN, cin, 3, 3)) * 0.05
    var0 = (w0 ** 2).sum(axis=(2, 3)) @ in_ms            # [HIDDEN] pre-activation variance
    sd[prefix + "0.weight"] = _f32(w0)
    sd[prefix + "0.actnorm.bias"] = _f32((rng.uniform(-0.3, 0.3, HIDDEN) * np.sqrt(var0)).reshape(1, HIDDEN, 1, 1))
    sd[prefix + "0.actnorm.logs"] = _f32((-0.5 * np.log(var0) + rng.uniform(-0.1, 0.1, HIDDEN)).reshape(1, HIDDEN, 1, 1))
    w2 = rng.standard_normal((HIDDEN, HIDDEN, 1, 1)) * 0.05
    var2 = (w2 ** 2).sum(axis=(1, 2, 3)) * 0.5           # E[relu(n)^2] = 1/2
    sd[prefix + "2.weight"] = _f32(w2)
    sd[prefix + "2.actnorm.bias"] = _f32((rng.uniform(-0.3, 0.3, HIDDEN) * np.sqrt(var2)).reshape(1, HIDDEN, 1, 1))
    sd[prefix + "2.actnorm.logs"] = _f32((-0.5 * np.log(var2) + rng.uniform(-0.1, 0.1, HIDDEN)).reshape(1, HIDDEN, 1, 1))
    sd[prefix + "4.weight"] = _f32(rng.standard_normal((cout, HIDDEN, 3, 3)) * zeros_std)
    sd[prefix + "4.bias"] = _f32(rng.standard_normal(cout) * (zeros_std * 4 if zeros_bias_std is None else zeros_bias_std))
    sd[prefix + "4.logs"] = _f32(rng.uniform(-0.1, 0.1, (cout, 1, 1)))


def synth_flow_state(seed: int = 0, L: int = N_LEVELS, K: int = K_STEPS, extra: int = N_EXTRA,
                     ft_rms: float = 1.0, zeros_std: float = 0.02, prefix: str = "",
                     zeros_bias_std=None, split=None) -> "OrderedDict[str, np.ndarray]":
    """State dict of a `FlowUpsamplerNet` (keys `layers.{i}.*`, `f.0.*`), as numpy float32.  zeros_std: std of the
    `Conv2dZeros` weights (how much the coupling scales vary from pixel to pixel); zeros_bias_std: std of their biases
    (how much they vary from channel to channel; default 4 * zeros_std)."""
    rng = np.random.default_rng([seed, 0xF10])
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    first = True
    # growth of |z| per conditional step in the encode direction when scale ~= sigmoid(2)+eps
    s_bar = 1.0 / (1.0 + np.exp(-2.0)) + 0.001
    growth = (1.0 / s_bar) * np.sqrt(0.5 * (1.0 + 1.0 / s_bar ** 2))
    for i, (kind, C) in enumerate(layer_plan(L, K, extra, split=split)):
        if kind == "squeeze":
            continue
        p = f"{prefix}layers.{i}."
        if kind == "split":   # Split2d (Split.py:13-39), unconditional: Conv2dZeros over the passed half, the shifted class prior
            cons = int(round(C * 0.5))
            sd[p + "conv.weight"] = _f32(rng.standard_normal((2 * cons, C - cons, 3, 3)) * 0.02)
            sd[p + "conv.bias"] = _f32(rng.standard_normal(2 * cons) * 0.1)
            sd[p + "conv.logs"] = _f32(rng.standard_normal((2 * cons, 1, 1)) * 0.1)
            sd[p + "I"] = _f32(np.eye(cons))
            sd[p + "mean_shift"] = _f32(rng.standard_normal(cons) * 0.2)
            sd[p + "cov_shift"] = _f32(np.eye(cons) * 0.7 + rng.standard_normal((cons, cons)) * 0.05)
            continue
        if first:   # raw pixels U[0,1): centre and whiten like the data-dependent init would
            bias = np.full(C, -0.5) + rng.uniform(-0.02, 0.02, C)
            logs = np.full(C, np.log(1.0 / np.sqrt(1.0 / 12.0))) + rng.uniform(-0.05, 0.05, C)
            first = False
        else:
            bias = rng.uniform(-0.1, 0.1, C)
            logs = rng.uniform(-0.05, 0.05, C) - (np.log(growth) if kind == "stepc" else 0.0)
        sd[p + "actnorm.bias"] = _f32(bias.reshape(1, C, 1, 1))
        sd[p + "actnorm.logs"] = _f32(logs.reshape(1, C, 1, 1))
        ic = _invconv_params(rng, C)
        for k in ("l", "log_s", "u", "p", "sign_s"):
            sd[p + "invconv." + k] = ic[k]
        if kind == "stepc":
            ms_ft = np.full(N_RRDB_CH, ft_rms ** 2)
            _coupling_net(rng, p + "affine.fAffine.", C // 2 + N_RRDB_CH, (C - C // 2) * 2,
                          np.concatenate([np.ones(C // 2), ms_ft]), sd, zeros_std, zeros_bias_std)
            _coupling_net(rng, p + "affine.fFeatures.", N_RRDB_CH, C * 2, ms_ft, sd, zeros_std, zeros_bias_std)
    # dead module registered by the reference (FlowUpsamplerNet.py:107-108), never called
    sd[f"{prefix}f.0.weight"] = _f32(rng.standard_normal((2 * 3 * 64 // 2 // 2, N_RRDB_CH, 3, 3)) * 0.01)
    sd[f"{prefix}f.0.bias"] = _f32(np.zeros(2 * 3 * 64 // 2 // 2))
    return sd


def synth_rrdb_state(seed: int = 0, nf: int = 64, nb: int = 23, gc: int = 32, in_nc: int = 3,
                     out_nc: int = 3, prefix: str = "", feat_gain: float = 1.0 / 12.0) -> "OrderedDict[str, np.ndarray]":
    """State dict of the conditioning encoder `RRDBNet` (reference RRDBNet_arch.py:48-77).
    Dense blocks: kaiming-normal * 0.1, zero bias (module_util.py:7-16); the other convs use
    PyTorch's default U(+-1/sqrt(fan_in)) for weight and bias.  With those initialisers the 23-block
    trunk amplifies U[0,1) images to features of rms ~11-15, far from what a trained encoder feeds the
    flow; `feat_gain` scales `conv_first` (the trunk is positively homogeneous: zero biases, LeakyReLU)
    so that the 320-channel conditioning maps come out with rms ~1."""
    rng = np.random.default_rng([seed, 0xBDB])
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def default_conv(name, cout, cin):
        bound = 1.0 / np.sqrt(cin * 9)
        sd[prefix + name + ".weight"] = _f32(rng.uniform(-bound, bound, (cout, cin, 3, 3)))
        sd[prefix + name + ".bias"] = _f32(rng.uniform(-bound, bound, cout))

    default_conv("conv_first", nf, in_nc * 16)
    for suffix in (".weight", ".bias"):
        sd[prefix + "conv_first" + suffix] = _f32(sd[prefix + "conv_first" + suffix].astype(np.float64) * feat_gain)
    for b in range(nb):
        for r in (1, 2, 3):
            for c in range(1, 6):
                cin = nf + (c - 1) * gc
                cout = gc if c < 5 else nf
                name = f"{prefix}RRDB_trunk.{b}.RDB{r}.conv{c}"
                sd[name + ".weight"] = _f32(rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (cin * 9)) * 0.1)
                sd[name + ".bias"] = _f32(np.zeros(cout))
    default_conv("trunk_conv", nf, nf)
    default_conv("upconv1", nf, nf)
    default_conv("upconv2", nf, nf)
    default_conv("HRconv", nf, nf)
    default_conv("conv_last", out_nc, nf)
    return sd


def synth_interpflow_state(scales, seed: int = 0, mean_init: float = 0.001, std_init: float = 1.0,
                           ft_rms: float = 1.0, zeros_std: float = 0.02, shift_noise: float = 0.05, zeros_bias_std=None):
    """Full `InterpFlow` state dict with the reference's aliased key set
    (reference flow_model.py:53-80, RRDBGlowNet_arch.py:49-59)."""
    rng = np.random.default_rng([seed, 0x1F])
    flow = synth_flow_state(seed, ft_rms=ft_rms, zeros_std=zeros_std, zeros_bias_std=zeros_bias_std)
    rrdb = synth_rrdb_state(seed)
    C = 48
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    n = len(scales)
    mean_interval = np.linspace(-mean_init * (n // 2), mean_init * (n // 2), n)
    sd["I"] = _f32(np.eye(C))
    for i in range(n):
        sd[f"mean_shift_{i}"] = _f32(np.array([mean_interval[i]]) + rng.uniform(-shift_noise, shift_noise, 1))
    for i in range(n):
        sd[f"cov_shift_{i}"] = _f32(np.eye(C) * std_init + rng.uniform(-shift_noise, shift_noise, (C, C)))
    sd["arch.I"] = _f32(np.eye(C))
    sd["arch.mean_shift"] = _f32(np.zeros(C))
    sd["arch.cov_shift"] = _f32(np.eye(C))
    for k, v in rrdb.items():
        sd["arch.RRDB." + k] = v
    for k, v in flow.items():
        sd["arch.flowUpsamplerNet." + k] = v
    # dead DegradationNet (reference flow_model.py:13-35,56)
    for idx, (co, ci) in zip((0, 2, 4), ((64, 320), (64, 64), (48, 64))):
        sd[f"degradation_net.net.{idx}.weight"] = _f32(rng.standard_normal((co, ci, 3, 3)) * 0.01)
        sd[f"degradation_net.net.{idx}.bias"] = _f32(np.zeros(co))
    for k, v in rrdb.items():
        sd["rrdb." + k] = v
    for k, v in flow.items():
        sd["flow_upsampler." + k] = v
    return sd


def synth_features(seed: int, B: int, H: int, W: int, rms: float = 1.0, keys=("fea_up4", "fea_up2", "fea_up1")):
    """Synthetic `rrdbResults` for an HxW image batch: 320-channel maps at H (fea_up4), H/2
    (fea_up2), H/4 (fea_up1) (reference RRDBGlowNet_arch.py:228-245), zero-mean uniform with the
    given rms.  Lets the flow be exercised without the conditioning encoder."""
    div = {"fea_up4": 1, "fea_up2": 2, "fea_up1": 4, "fea_up0": 8}
    out = {}
    a = rms * np.sqrt(3.0)
    for i, k in enumerate(keys):
        rng = np.random.default_rng([seed, 0xFEA, i])
        out[k] = rng.uniform(-a, a, (B, N_RRDB_CH, H // div[k], W // div[k])).astype(np.float32)
    return out


def synth_images(seed: int, B: int, H: int, W: int, C: int = 3):
    """U[0,1) float32 images, NCHW (SURVEY.md section 8(d) synthetic inputs)."""
    return np.random.default_rng([seed, 0x1336]).random((B, C, H, W), dtype=np.float64).astype(np.float32)


def state_digest(sd) -> str:
    """sha256 over the sorted (key, shape, bytes) of a state dict: lets a fixture check that the
    weights regenerated on another machine are bit-identical to those the goldens were made with."""
    h = hashlib.sha256()
    for k in sorted(sd):
        a = np.ascontiguousarray(sd[k])
        h.update(k.encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()
