"""Host logic of the Winograd forms of the bulk convolution (CPU only): the packed filter transforms of pack.py, read
back through the layout conv1.hip / conv1x.hip document, reproduce the direct 3x3 'same' convolution of the reference
(flow.py:25-64) when combined with the data and output transforms in float64."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytest.importorskip("interflow_amd.hip")
from interflow_amd import hip, pack  # noqa: E402

AT4, _, BT4 = pack.wino4_matrices()        # points 0, +-3/4, +-3/2, inf (csrc/ifl_common.h)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def _lib_or_skip():
    try:
        return hip.lib()
    except hip.InterflowHipError as e:          # the library is built by __graft_entry__.build(); this test needs no GPU
        pytest.skip(str(e))


def _wino(x, U, BT, AT, m):
    """x [Ci,H,W] float64, U [a,a,Co,Ci] -> 'same' conv [Co,H,W] by F(m x m, 3x3)."""
    a = m + 2
    Ci, H, W = x.shape
    th, tw = -(-H // m), -(-W // m)
    xp = np.zeros((Ci, th * m + 2, tw * m + 2))
    xp[:, 1:H + 1, 1:W + 1] = x
    out = np.zeros((U.shape[2], th * m, tw * m))
    for ty in range(th):
        for tx in range(tw):
            d = xp[:, ty * m:ty * m + a, tx * m:tx * m + a]
            V = np.einsum("ik,ckl,jl->ijc", BT, d, BT)
            M = np.einsum("ijoc,ijc->ijo", U, V)
            out[:, ty * m:(ty + 1) * m, tx * m:(tx + 1) * m] = np.einsum("ki,ijo,lj->okl", AT, M, AT)
    return out[:, :H, :W]


@pytest.mark.parametrize("zoff", [0, 6, 24])
def test_packed_winograd_filters_reproduce_the_direct_convolution(zoff):
    _lib_or_skip()
    rng = np.random.default_rng(zoff)
    wt = rng.standard_normal((64, zoff + 320, 3, 3)).astype(np.float32)
    x = rng.standard_normal((320, 7, 9))
    ref = F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(wt[:, zoff:].astype(np.float64)), None, 1, 1)[0].numpy()
    # F(4x4,3x3): [mh][bi][bj][s][il][jl][hh][o][e] -> U[i = 3 bi + il][jc = 3 bj + jl][co = 32 mh + o][ci = 8 s + 4 hh + e]
    u = pack._pack_conv1_wino4(wt, zoff).reshape(2, 2, 2, 40, 3, 3, 2, 32, 4).astype(np.float64)
    U4 = u.transpose(1, 4, 2, 5, 0, 7, 3, 6, 8).reshape(6, 6, 64, 320)
    got = _wino(x, U4, BT4, AT4, 4)
    assert np.abs(got - ref).max() < 2e-4 * np.abs(ref).max()          # fp32 rounding of U only
    # F(2x2,3x3): [jc][cc][kq][i][mt][h][o][e] -> U[i][jc][co = 32 mt + o][ci = 32 cc + 8 kq + 4 h + e]
    u = pack._pack_conv1_wino(wt, zoff).reshape(4, 10, 4, 4, 2, 2, 32, 4).astype(np.float64)
    U2 = u.transpose(3, 0, 4, 6, 1, 2, 5, 7).reshape(4, 4, 64, 320)
    got = _wino(x, U2, BT2, AT2, 2)
    assert np.abs(got - ref).max() < 2e-5 * np.abs(ref).max()


def test_step_block_has_room_for_every_field():
    L = _lib_or_skip()
    for C in (12, 48):
        total = L.ifl_step_param_floats(C, 1)
        offs = sorted(L.ifl_step_field_off(C, f) for f in range(hip.F_W3F_WINO + 1))
        nf = -(-C // 8)
        assert offs[0] == 0 and all(b > a for a, b in zip(offs, offs[1:])) and offs[-1] + nf * 16 * 16 * 64 <= total
        assert L.ifl_step_field_off(C, hip.F_W3F_WINO) == offs[-1]
        assert L.ifl_step_field_off(C, hip.F_W3F_WINO + 1) == -1


def test_conv3_winograd_weights_reproduce_the_direct_convolution():
    """pack._pack_conv3_wino: U = G g G^T of the permuted Conv2dZeros rows.  Evaluating F(2x2,3x3) from the packed block with
    numpy (V = B^T d B, M = U (.) V summed over channels, Y = A^T M A) gives the direct 3x3 correlation of the same rows."""
    from interflow_amd import pack
    rng = np.random.default_rng(3)
    C = 12
    w3 = rng.standard_normal((2 * C, 64, 3, 3)).astype(np.float32)
    up = pack._pack_conv3_wino(w3, C, 1).reshape(2, 4, 16, 4, 16, 4)          # tile s pos q i e
    rows = pack._row_map(C, 1).reshape(2, 16)
    d = rng.standard_normal((64, 4, 4))
    BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
    AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
    V = np.einsum("ir,crk,jk->cij", BT, d, BT).reshape(64, 16)
    for tile in range(2):
        U = up[tile].transpose(1, 3, 0, 2, 4).reshape(16, 16, 64)             # pos i ch(s q e)
        M = np.einsum("pic,cp->ip", U.astype(np.float64), V).reshape(16, 4, 4)
        Y = np.einsum("ai,rij,bj->rab", AT, M, AT)                            # [row i][2][2]
        for i in range(16):
            r = rows[tile, i]
            want = np.zeros((2, 2)) if r < 0 else np.array([[np.sum(w3[r].astype(np.float64) * d[:, y:y + 3, x:x + 3]) for x in range(2)] for y in range(2)])
            np.testing.assert_allclose(Y[i], want, rtol=1e-5, atol=1e-5)
