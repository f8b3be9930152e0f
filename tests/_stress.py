"""The stress state of tests/golden/make_golden_stress.py: synthesised weights (second seed, wide coupling scales, feature
rms 2) whose ActNorm parameters are the ones the REFERENCE's data-dependent initialisation produced (stored in the fixture)."""
import os

import numpy as np

from interflow_amd import synth

SCALES = [2, 4]


def stress_fixture(golden_dir, seed=1):
    """seed 1: the full fixture (32x32, 160x160 batch 2, 512x512 samples); seed 2 (round 3): a third weight seed, 160x160 batch 1."""
    return np.load(os.path.join(golden_dir, f"stress_seed{seed}.npz"))


def stress_state(g):
    sd = synth.synth_interpflow_state(SCALES, seed=int(g["seed"]), ft_rms=float(g["ft_rms"]), zeros_std=float(g["zeros_std"]),
                                      zeros_bias_std=float(g["zeros_bias_std"]))
    n = 0
    for k in g.files:
        if k.startswith("an_"):
            for pre in ("flow_upsampler.", "arch.flowUpsamplerNet."):
                assert sd[pre + k[3:]].shape == g[k].shape
                sd[pre + k[3:]] = g[k]
            n += 1
    assert n == 328
    assert synth.state_digest(sd) == str(g["digest"]), "stress weights differ from the ones the fixture was made with"
    return sd


def elementwise_close(got, want, tol, what=""):
    """Two criteria, both asserted and both returned: the max-norm one (|err|max <= tol * |want|max) and an element-wise
    one that small elements cannot hide behind the tensor's largest: |err_i| <= tol * max(|want_i|, rms(want))."""
    import torch
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.isfinite(got).all(), what
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    if err.size == 0:
        return 0.0, 0.0
    e_max = float(err.max()) / max(float(np.abs(want).max()), 1e-30)
    rms = float(np.sqrt(np.mean(want.astype(np.float64) ** 2)))
    e_elem = float((err / np.maximum(np.abs(want), max(rms, 1e-30))).max())
    assert e_max <= tol, f"{what}: max-norm rel err {e_max:.3e} > {tol:.1e} (element-wise {e_elem:.3e})"
    assert e_elem <= tol, f"{what}: element-wise rel err {e_elem:.3e} > {tol:.1e} (max-norm {e_max:.3e})"
    return e_max, e_elem


def _errs(got, want):
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    rms = float(np.sqrt(np.mean(want.astype(np.float64) ** 2)))
    return float(err.max()) / max(float(np.abs(want).max()), 1e-30), float((err / np.maximum(np.abs(want), max(rms, 1e-30))).max())


NOISE_K = 4.0          # round 2: 8 (the F(4x4,3x3) form on the points 0, +-1, +-2 needed it); round 3's points need less than 3


def noise_aware_close(got, ref32, truth64, tol, what=""):
    """For quantities whose float32 evaluation is ill-conditioned (the stress regime: rounding amplified through 32 divisions
    by the coupling scale).  Errors are measured against a float64 evaluation of the same network (`truth64`, stored in the
    fixture); a result is accepted when it is within `tol` OR within NOISE_K times the REFERENCE's own float32 rounding noise
    in that quantity (its error against the same float64 evaluation), whichever is larger -- no implementation can be asked
    to agree with the reference more closely than the reference agrees with exact arithmetic.  Both the max-norm and the
    element-wise criterion.  Returns ((max-norm, element-wise) of `got`, (max-norm, element-wise) of the reference)."""
    import torch
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    assert got.shape == truth64.shape and np.isfinite(got).all(), what
    e, n = _errs(got, truth64), _errs(np.asarray(ref32), truth64)
    for i, name in enumerate(("max-norm", "element-wise")):
        allowed = max(tol, NOISE_K * n[i])
        assert e[i] <= allowed, (f"{what}: {name} rel err vs float64 {e[i]:.3e} > {allowed:.1e} "
                                 f"(tolerance {tol:.0e}, reference's own noise {n[i]:.2e})")
    return e, n


ROUNDTRIP_K = 2.0


def roundtrip_close(got, x, ref_roundtrip, tol, what=""):
    """decode(encode(x)) against x.  float32 itself limits this quantity in the stress regime: the reference's own round trip
    (stored in the fixture) is 4.3e-5 ... 7.4e-5 of |x|max away from x, and even a float64 decode of its float32 z is (7.2e-5 on
    the third seed).  Accepted: within `tol`, or within ROUNDTRIP_K = 2 times the reference's own round-trip error, whichever is
    larger, max-norm and element-wise.  Returns (ours, the reference's)."""
    import torch
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    x = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
    e, n = _errs(got, x), _errs(np.asarray(ref_roundtrip), x)
    for i, name in enumerate(("max-norm", "element-wise")):
        allowed = max(tol, ROUNDTRIP_K * n[i])
        assert np.isfinite(e[i]) and e[i] <= allowed, (f"{what}: {name} round-trip error {e[i]:.3e} > {allowed:.1e} "
                                                       f"(tolerance {tol:.0e}, the reference's own round trip {n[i]:.2e})")
    return e, n
