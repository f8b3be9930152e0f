"""Numerics of the Winograd forms of the bulk convolution, on the CPU oracle (the checker, not the product): how much fp32
rounding error do F(2x2,3x3) and F(4x4,3x3) add to z of a full 160x160 encode, compared with the direct fp32 form and a
float64 evaluation of the same network?  The bound asserted here is what lets the bulk kernel use F(4x4,3x3) inside the
1e-4 tolerance of tests/test_gpu_parity.py.

    python tests/test_wino_numerics.py        (prints the three errors)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # repository root
sys.path.insert(0, ROOT)
from interflow_amd import synth  # noqa: E402
from oracle import interflow_oracle as O  # noqa: E402

from interflow_amd import pack  # noqa: E402
AT4, G4, BT4 = pack.wino4_matrices()        # the library's points 0, +-3/4, +-3/2, inf (csrc/ifl_common.h)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def wino_conv(x, w, m):
    """'same' 3x3 conv of x [B,Ci,H,W] with w [Co,Ci,3,3] by F(m x m, 3x3); all data arithmetic in x.dtype."""
    BT, G, AT = (BT2, G2, AT2) if m == 2 else (BT4, G4, AT4)
    a = m + 2
    dt = x.dtype
    U = torch.from_numpy(np.einsum("ik,ockl,jl->ijoc", G, w.double().numpy(), G)).to(dt)     # host float64 -> dt
    BTt, ATt = torch.from_numpy(BT).to(dt), torch.from_numpy(AT).to(dt)
    B, Ci, H, W = x.shape
    th, tw = -(-H // m), -(-W // m)
    xp = F.pad(x, (1, tw * m + 1 - W, 1, th * m + 1 - H))
    win = xp.unfold(2, a, m).unfold(3, a, m)                     # B,Ci,th,tw,a,a
    V = torch.einsum("ik,bcyxkl,jl->ijbcyx", BTt, win, BTt)
    M = torch.einsum("ijoc,ijbcyx->ijboyx", U, V)
    Y = torch.einsum("ki,ijboyx,lj->boykxl", ATt, M, ATt)        # B,Co,th,m,tw,m
    return Y.reshape(B, -1, th * m, tw * m)[:, :, :H, :W]


MODE = {"m": 0}


def conv2d_actnorm(x, weight, an_bias, an_logs):
    pad = [((k - 1) + 1) // 2 for k in weight.shape[2:]]
    if MODE["m"] and weight.shape[2] == 3 and weight.shape[1] >= 320:
        nz = weight.shape[1] - 320
        y = wino_conv(x[:, nz:], weight[:, nz:], MODE["m"])
        if nz:
            y = y + F.conv2d(x[:, :nz], weight[:, :nz], None, 1, pad)
    else:
        y = F.conv2d(x, weight, None, 1, pad)
    y, _ = O.actnorm(y, an_bias, an_logs)
    return y


def errors():
    torch.set_num_threads(8)
    saved = O.conv2d_actnorm
    O.conv2d_actnorm = conv2d_actnorm
    try:
        return _errors()
    finally:
        O.conv2d_actnorm = saved
        MODE["m"] = 0


def _errors():
    sd_np = synth.synth_interpflow_state([2, 4], seed=0)
    B, H = 1, 160
    x = torch.from_numpy(synth.synth_images(1, B, H, H))
    feats = {k: torch.from_numpy(v) for k, v in synth.synth_features(2, B, H, H).items()}
    sd32 = O.to_torch(sd_np)
    sd64 = {k: v.double() for k, v in sd32.items()}
    with torch.no_grad():
        MODE["m"] = 0
        z64, ld64 = O.flow_encode(x.double(), {k: v.double() for k, v in feats.items()}, sd64, "flow_upsampler.", torch.zeros(B).double())
        ref = float(z64.abs().max())
        out = {}
        for m in (0, 2, 4):
            MODE["m"] = m
            z, ld = O.flow_encode(x, feats, sd32, "flow_upsampler.", torch.zeros(B))
            out[m] = (float((z.double() - z64).abs().max()) / ref, float(((ld.double() - ld64) / ld64).abs().max()))
    return out


def test_winograd_forms_stay_far_inside_the_tolerance():
    e = errors()
    for m, (ez, el) in e.items():
        print(f"F({m}x{m},3x3)" if m else "direct     ", f"z max|err|/max|z| vs float64 = {ez:.3e}   logdet rel {el:.2e}")
    assert e[0][0] < 1e-5 and e[2][0] < 1e-5 and e[4][0] < 2e-5           # tolerance of the parity tests: 1e-4
    assert all(el < 1e-5 for _, el in e.values())


if __name__ == "__main__":
    test_winograd_forms_stay_far_inside_the_tolerance()
