"""Which interpolation points should F(4x4,3x3) use?  CPU experiment behind conv1x.hip's choice (not product, not a test).

For a set of five finite points + infinity, builds (A^T, G, B^T) by Toom-Cook (evaluation / transposed interpolation), checks
them in float64 against the direct correlation, then measures the fp32 rounding error of a 320 -> 64 channel 3x3 'same'
convolution evaluated the way conv1_ft_wino4_kernel does (U rounded once from float64, V = B^T d B in fp32, sums over the
channels in fp32, Y = A^T M A in fp32) against a float64 direct evaluation, for inputs shaped like the conditioning features.

    python tools/exp_wino_points.py
"""
import itertools
import sys
from fractions import Fraction as Fr

import numpy as np
import torch


def toom_cook(points, m=4, r=3):
    """points: n-1 finite points (Fractions); returns AT [m,n], G [n,r], BT [n,n] as float64 (exact rationals rounded)."""
    n = m + r - 1
    assert len(points) == n - 1
    V = [[p ** k for k in range(n)] for p in points] + [[Fr(0)] * (n - 1) + [Fr(1)]]       # evaluation of degree n-1 polys
    # inverse by Gauss-Jordan in rationals
    a = [row[:] + [Fr(int(i == j)) for j in range(n)] for i, row in enumerate(V)]
    for c in range(n):
        piv = next(i for i in range(c, n) if a[i][c] != 0)
        a[c], a[piv] = a[piv], a[c]
        pv = a[c][c]
        a[c] = [x / pv for x in a[c]]
        for i in range(n):
            if i != c and a[i][c] != 0:
                f = a[i][c]
                a[i] = [x - f * y for x, y in zip(a[i], a[c])]
    C = [row[n:] for row in a]                                  # V^-1  [n, n]
    BT = [[C[j][i] for j in range(n)] for i in range(n)]        # (V^-1)^T
    G = [[p ** k for k in range(r)] for p in points] + [[Fr(0)] * (r - 1) + [Fr(1)]]
    AT = [[(p ** i) for p in points] + [Fr(int(i == m - 1))] for i in range(m)]
    return AT, G, BT


def rescale(AT, G, BT, mode):
    """Move per-position scale factors between G and B^T (their product per position is fixed)."""
    n = len(G)
    AT = [row[:] for row in AT]; G = [row[:] for row in G]; BT = [row[:] for row in BT]
    for k in range(n):
        if mode == "bt_int":           # make B^T's row integer with smallest entries: scale row by lcm of denominators
            den = 1
            for x in BT[k]:
                den = den * x.denominator // np.gcd(den, x.denominator)
            num = 0
            for x in BT[k]:
                num = np.gcd(num, abs(int(x * den)))
            s = Fr(int(den), int(num) if num else 1)
        elif mode == "g_unit":         # G's row has max |entry| = 1
            s = 1 / max(abs(x) for x in G[k])
            s = 1 / s
        else:
            s = Fr(1)
        BT[k] = [x * s for x in BT[k]]
        G[k] = [x / s for x in G[k]]
    return AT, G, BT


def f64(M):
    return np.array([[float(x) for x in row] for row in M], np.float64)


def wino_conv(x, w, AT, G, BT, dt=torch.float32, m=4):
    a = m + 2
    U = torch.from_numpy(np.einsum("ik,ockl,jl->ijoc", G, w.double().numpy(), G)).to(dt)
    BTt, ATt = torch.from_numpy(BT).to(dt), torch.from_numpy(AT).to(dt)
    B, Ci, H, W = x.shape
    th, tw = -(-H // m), -(-W // m)
    xp = torch.nn.functional.pad(x.to(dt), (1, tw * m + 1 - W, 1, th * m + 1 - H))
    win = xp.unfold(2, a, m).unfold(3, a, m)
    V = torch.einsum("ik,bcyxkl->bcyxil", BTt, win)            # columns first, then rows: two rounded stages as the kernel
    V = torch.einsum("bcyxil,jl->ijbcyx", V, BTt)
    M = torch.zeros(a, a, B, U.shape[2], th, tw, dtype=dt)
    for c0 in range(0, Ci, 8):                                  # sub-chunks of 8 channels accumulated in order
        M = M + torch.einsum("ijoc,ijbcyx->ijboyx", U[..., c0:c0 + 8], V[:, :, :, c0:c0 + 8])
    Y = torch.einsum("ki,ijboyx->kjboyx", ATt, M)
    Y = torch.einsum("kjboyx,lj->boykxl", Y, ATt)
    return Y.reshape(B, -1, th * m, tw * m)[:, :, :H, :W]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    B, Ci, Co, H = 1, 320, 64, 40
    # conditioning-like input: smooth-ish positive/negative features of rms 2; weights like flow.Conv2d's init (std 0.05)
    x = torch.randn(B, Ci, H, H) * 2.0
    x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode="reflect"), 3, 1) * 2 + 0.5 * torch.randn(B, Ci, 1, 1)
    w = torch.randn(Co, Ci, 3, 3) * 0.05
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    direct = torch.nn.functional.conv2d(x, w, padding=1)
    scale = float(ref.abs().max()); rms = float(ref.pow(2).mean().sqrt())
    print(f"direct fp32: max err / max {float((direct.double() - ref).abs().max()) / scale:.3e}   rms err / rms {float((direct.double() - ref).pow(2).mean().sqrt()) / rms:.3e}")
    h = Fr(1, 2)
    cands = {
        "0,+-1,+-2 (Lavin & Gray)": [Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-2)],
        "0,+-1,+-1/2": [Fr(0), Fr(1), Fr(-1), h, -h],
        "0,+-1,1/2,-2": [Fr(0), Fr(1), Fr(-1), h, Fr(-2)],
        "0,+-1,2,-1/2": [Fr(0), Fr(1), Fr(-1), Fr(2), -h],
        "0,+-1/2,+-2": [Fr(0), h, -h, Fr(2), Fr(-2)],
        "0,+-1/2,+-3/2": [Fr(0), h, -h, Fr(3, 2), Fr(-3, 2)],
        "0,+-3/4,+-3/2": [Fr(0), Fr(3, 4), Fr(-3, 4), Fr(3, 2), Fr(-3, 2)],
        "0,+-1,+-3/2": [Fr(0), Fr(1), Fr(-1), Fr(3, 2), Fr(-3, 2)],
        "0,+-2/3,+-3/2": [Fr(0), Fr(2, 3), Fr(-2, 3), Fr(3, 2), Fr(-3, 2)],
        "0,+-1/2,+-1,": None,
        "0,+-5/8,+-13/8": [Fr(0), Fr(5, 8), Fr(-5, 8), Fr(13, 8), Fr(-13, 8)],
        "0,+-3/4,+-4/3": [Fr(0), Fr(3, 4), Fr(-3, 4), Fr(4, 3), Fr(-4, 3)],
        "0,+-1/2,+-2 ": [Fr(0), h, -h, Fr(2), Fr(-2)],
        "0,+-1,1/2,-3": [Fr(0), Fr(1), Fr(-1), h, Fr(-3)],
    }
    for name, pts in cands.items():
        if pts is None:
            continue
        for mode in ("none", "bt_int"):
            AT, G, BT = rescale(*toom_cook(pts), mode)
            ATf, Gf, BTf = f64(AT), f64(G), f64(BT)
            chk = wino_conv(x[:, :16], w[:4, :16], ATf, Gf, BTf, torch.float64)
            assert float((chk - torch.nn.functional.conv2d(x[:, :16].double(), w[:4, :16].double(), padding=1)).abs().max()) < 1e-9, name
            y = wino_conv(x, w, ATf, Gf, BTf)
            e = (y.double() - ref)
            print(f"{name:28s} {mode:7s} max err / max {float(e.abs().max()) / scale:.3e}   rms err / rms {float(e.pow(2).mean().sqrt()) / rms:.3e}"
                  f"   max|BT| {np.abs(BTf).max():.3g} max|AT| {np.abs(ATf).max():.3g} max|G| {np.abs(Gf).max():.3g}")


if __name__ == "__main__":
    main()
