"""Host-side parameter preparation for the HIP flow kernels.

Takes the reference-format parameters of `FlowUpsamplerNet.layers.*` (numpy float32 arrays keyed as in
the state dict) and writes the packed block each kernel reads.  Mirrors what the reference recomputes
on every call in Python (`InvertibleConv1x1.get_weight`, Permutations.py:76-101; `exp(logs)` of the
ActNorms) plus pure re-orderings of the convolution weights into MFMA fragment order.  The block layout
is owned by the C library (`ifl_step_field_off`); this file only fills it.
"""
from __future__ import annotations

import numpy as np

from . import hip
from .synth import HIDDEN, N_RRDB_CH

F32 = np.float32


def invconv_matrices(prm, pfx):
    """W, Winv (float32 [C,C]) and sum(log|det| per pixel) exactly as Permutations.py:76-101 builds them."""
    if pfx + "weight" in prm:                                   # non-LU branch, Permutations.py:78-86
        w = prm[pfx + "weight"].astype(F32)
        winv = np.linalg.inv(w.astype(np.float64)).astype(F32)
        return w, winv, F32(np.linalg.slogdet(w)[1])
    p, l, u = prm[pfx + "p"], prm[pfx + "l"], prm[pfx + "u"]
    log_s, sign_s = prm[pfx + "log_s"], prm[pfx + "sign_s"]
    C = p.shape[0]
    l_mask = np.tril(np.ones((C, C), F32), -1)
    eye = np.eye(C, dtype=F32)
    lm = (l * l_mask + eye).astype(F32)
    um = (u * l_mask.T + np.diag(sign_s * np.exp(log_s))).astype(F32)
    w = (p @ (lm @ um)).astype(F32)
    li = np.linalg.inv(lm.astype(np.float64)).astype(F32)      # Permutations.py:98-99: inverse in float64, cast
    ui = np.linalg.inv(um.astype(np.float64)).astype(F32)
    pinv = np.linalg.inv(p.astype(np.float64)).astype(F32)     # p is a permutation: exact
    winv = (ui @ (li @ pinv)).astype(F32)
    return w, winv, F32(np.sum(log_s, dtype=F32))


def _acc_channel_order():
    """Channel held by accumulator element (m, h, r) of the 64 x pixels MFMA block: m*32 + (r&3) + 8*(r>>2) + 4*h,
    flattened as [m][h][r]."""
    m, h, r = np.meshgrid(np.arange(2), np.arange(2), np.arange(16), indexing="ij")
    return (m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h).reshape(-1)


_ACC = _acc_channel_order()


def _pack_conv1(wt, zoff, kz):
    """wt [64, Cin, 3, 3] -> fragment order.  ft part [cc][tap][kq][mt][h][i][e], then z part [tap][kq][mt][h][i][e]."""
    wft = wt[:, zoff:zoff + N_RRDB_CH].reshape(2, 32, 10, 4, 2, 4, 9)           # mt i cc kq h e tap
    out = [np.ascontiguousarray(wft.transpose(2, 6, 3, 0, 4, 1, 5)).reshape(-1)]
    if kz:
        wz = np.zeros((HIDDEN, kz, 9), F32)
        wz[:, :zoff] = wt[:, :zoff].reshape(HIDDEN, zoff, 9)
        wz = wz.reshape(2, 32, kz // 8, 2, 4, 9)                                 # mt i kq h e tap
        out.append(np.ascontiguousarray(wz.transpose(5, 2, 0, 3, 1, 4)).reshape(-1))
    return np.concatenate(out)


_WINO_G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)


def _pack_conv1_wino(wt, zoff):
    """Conditioning part of wt [64, Cin, 3, 3] -> Winograd F(2x2,3x3) domain U = G g G^T (float64, rounded once to fp32),
    laid out [jc][cc][kq][i][mt][h][o][e]: position (i, jc) of the 4x4 domain, out channel mt*32 + o, in channel
    cc*32 + kq*8 + h*4 + e (conv1.hip, conv1_ft_wino_kernel)."""
    g = wt[:, zoff:zoff + N_RRDB_CH].astype(np.float64)                          # [64, 320, 3, 3]
    u = np.einsum("ia,ocab,jb->ocij", _WINO_G, g, _WINO_G).astype(F32)           # [64, 320, 4, 4]
    u = u.reshape(2, 32, 10, 4, 2, 4, 4, 4)                                      # mt o cc kq h e i jc
    return np.ascontiguousarray(u.transpose(7, 2, 3, 6, 0, 4, 1, 5)).reshape(-1)


def wino4_matrices(a=0.75, b=1.5):
    """(A^T [4,6], G [6,3], B^T [6,6]) of F(4x4,3x3) for the interpolation points 0, +-a, +-b, infinity, float64, the rows of
    B^T scaled to a leading coefficient of one -- the library's choice a = 3/4, b = 3/2 and scaling (csrc/ifl_common.h,
    IFL_W4_*; tools/exp_wino_points.py for why these points)."""
    A, B = a * a, b * b
    da, db = 1 / (2 * A * (A - B)), 1 / (2 * B * (B - A))
    G = np.array([[1 / (A * B), 0, 0], [da, a * da, A * da], [da, -a * da, A * da],
                  [db, b * db, B * db], [db, -b * db, B * db], [0, 0, 1]], np.float64)
    BT = np.array([[A * B, 0, -(A + B), 0, 1, 0], [0, -a * B, -B, a, 1, 0], [0, a * B, -B, -a, 1, 0],
                   [0, -b * A, -A, b, 1, 0], [0, b * A, -A, -b, 1, 0], [0, A * B, 0, -(A + B), 0, 1]], np.float64)
    AT = np.array([[1, 1, 1, 1, 1, 0], [0, a, -a, b, -b, 0], [0, A, A, B, B, 0], [0, a ** 3, -a ** 3, b ** 3, -b ** 3, 1]], np.float64)
    return AT, G, BT


_WINO4_G = wino4_matrices()[1]


def _pack_conv1_wino4(wt, zoff):
    """Conditioning part of wt [64, Cin, 3, 3] -> Winograd F(4x4,3x3) domain U = G g G^T (6x6, float64, rounded once to
    fp32), laid out [mh][bi][bj][s][il][jl][hh][o][e]: position (3 bi + il, 3 bj + jl), out channel mh*32 + o, in channel
    s*8 + hh*4 + e (conv1x.hip, conv1_ft_wino4_kernel: contiguous per wave, MFMA A-fragment order)."""
    g = wt[:, zoff:zoff + N_RRDB_CH].astype(np.float64)                          # [64, 320, 3, 3]
    u = np.einsum("ia,ocab,jb->ocij", _WINO4_G, g, _WINO4_G).astype(F32)         # [64, 320, 6, 6]
    u = u.reshape(2, 32, 40, 2, 4, 2, 3, 2, 3)                                   # mh o s hh e bi il bj jl
    return np.ascontiguousarray(u.transpose(0, 5, 7, 2, 6, 8, 3, 1, 4)).reshape(-1)


def to_bf16_bits(a):
    """float32 -> bfloat16 bit patterns (uint16), round to nearest even (what v_cvt_pk_bf16_f32 does for finite values)."""
    u = np.ascontiguousarray(a, F32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def _as_float_slots(bits16):
    """uint16 array (even length) -> float32 view, two bf16 per float slot of the parameter block."""
    return np.ascontiguousarray(bits16).view(F32)


def _pack_conv1_bf16(wt, zoff):
    """Conditioning part of wt [64, Cin, 3, 3] -> [cc][tap][ks][mt][h][i][j], in channel = cc*32 + ks*16 + 8h + j, as bf16."""
    wft = wt[:, zoff:zoff + N_RRDB_CH].reshape(2, 32, 10, 2, 2, 8, 9)            # mt i cc ks h j tap
    return _as_float_slots(to_bf16_bits(np.ascontiguousarray(wft.transpose(2, 6, 3, 0, 4, 1, 5)).reshape(-1)))


def _pack_conv2_bf16(w2):
    """w2 [64,64,1,1] -> [m2][m][s][h][i][j] = W[m2*32+i][m*32 + 16s + 8(j>>2) + 4h + (j&3)] (the accumulator's row order
    when a 32x32 tile is fed back as the B operand of v_mfma_f32_32x32x16_bf16), as bf16."""
    w = w2.reshape(HIDDEN, HIDDEN)
    m2, m, s_, h, i, j = np.meshgrid(np.arange(2), np.arange(2), np.arange(2), np.arange(2), np.arange(32), np.arange(8),
                                     indexing="ij")
    return _as_float_slots(to_bf16_bits(w[m2 * 32 + i, m * 32 + 16 * s_ + 8 * (j >> 2) + 4 * h + (j & 3)].reshape(-1)))


def _pack_actnorm64(bias, logs):
    b = bias.reshape(-1)[_ACC]
    s = np.exp(logs.reshape(-1).astype(F32))[_ACC]
    return np.concatenate([b, s]).astype(F32)


def _pack_conv2(w2):
    """w2 [64,64,1,1] -> [m2][m][gq][h][i][e] = W[m2*32+i][m*32 + e + 8*gq + 4*h]"""
    w = w2.reshape(HIDDEN, HIDDEN)
    m2, m, gq, h, i, e = np.meshgrid(np.arange(2), np.arange(2), np.arange(4), np.arange(2), np.arange(32),
                                     np.arange(4), indexing="ij")
    return w[m2 * 32 + i, m * 32 + e + 8 * gq + 4 * h].reshape(-1).astype(F32)


def _row_map(C, net):
    """Original Conv2dZeros output row for (tile, q, r), or -1.  net 0 = fAffine (rows 2*c2 + s for z2 channel
    c2), net 1 = fFeatures (rows 2*c + s).  slot sigma = 8*tile + 2*q + (r&1), s = r>>1 (0 shift, 1 scale)."""
    ch = C // 2
    ntile = -(-ch // 8) if net == 0 else -(-C // 8)
    rows = np.full((ntile, 4, 4), -1, np.int64)
    for t in range(ntile):
        for q in range(4):
            for r in range(4):
                sigma = 8 * t + 2 * q + (r & 1)
                s = r >> 1
                if net == 0:
                    if sigma < ch:
                        rows[t, q, r] = 2 * sigma + s
                elif sigma < C:
                    rows[t, q, r] = 2 * ((sigma + ch) % C) + s
    return rows


def _pack_conv3(w3, bias, logs, C, net):
    rows = _row_map(C, net)                                   # [tile][q][r]
    ntile = rows.shape[0]
    flat = rows.reshape(ntile, 16)                            # row index inside the tile = 4*q + r
    wsrc = np.concatenate([w3.reshape(w3.shape[0], HIDDEN, 9), np.zeros((1, HIDDEN, 9), F32)])
    wperm = wsrc[flat]                                        # [tile][16][64][9]; -1 picks the zero row
    # -> [tile][tap][sidx][kq4][i][e] with ci = 16*sidx + 4*kq4 + e
    wperm = wperm.reshape(ntile, 16, 4, 4, 4, 9)              # tile i sidx kq4 e tap
    wp = np.ascontiguousarray(wperm.transpose(0, 5, 2, 3, 1, 4)).reshape(-1).astype(F32)
    bsrc = np.concatenate([bias.reshape(-1), np.zeros(1, F32)])
    esrc = np.concatenate([np.exp((logs.reshape(-1) * F32(3.0)).astype(F32)), np.zeros(1, F32)])
    bp = np.stack([bsrc[rows], esrc[rows]], axis=1)           # [tile][2][q][r]
    return wp, bp.reshape(-1).astype(F32)


def _pack_conv3_wino(w3, C, net):
    """Conv2dZeros weight [rows, 64, 3, 3] -> Winograd F(2x2,3x3) domain U = G g G^T (4x4 positions; float64, rounded once to
    fp32) with the rows permuted as `_pack_conv3` does, laid out [row tile][s][pos][q][i][e]: position pos = 4 i_pos + jc,
    row i of the tile, in channel 16 s + 4 q + e (conv3w.hip: one contiguous 16 KB slab per (row tile, sub-chunk))."""
    rows = _row_map(C, net)
    ntile = rows.shape[0]
    flat = rows.reshape(ntile, 16)
    g = np.concatenate([w3.astype(np.float64), np.zeros((1, HIDDEN, 3, 3))])     # -1 picks the zero row
    u = np.einsum("ia,ocab,jb->ocij", _WINO_G, g, _WINO_G).astype(F32)           # [rows + 1, 64, 4, 4]
    u = u[flat]                                                                   # [tile][i 16][64][4][4]
    u = u.reshape(ntile, 16, 4, 4, 4, 16)                                        # tile i s q e pos
    return np.ascontiguousarray(u.transpose(0, 2, 5, 3, 1, 4)).reshape(-1)


def pack_step(prm, pfx, C, has_coupling, out, training=False):
    """Fill `out` (float32 view of the step's block).  Returns (sum_logs, sum_log_s)."""
    L = hip.lib()

    def put(field, arr):
        o = L.ifl_step_field_off(C, field)
        out[o:o + arr.size] = arr.reshape(-1)

    bias = prm[pfx + "actnorm.bias"].reshape(-1).astype(F32)
    logs = prm[pfx + "actnorm.logs"].reshape(-1).astype(F32)
    if training and not np.any(bias != 0):
        raise NotImplementedError(
            "the fused driver packs initialised ActNorm parameters only: FlowUpsamplerNet runs the data-dependent "
            "initialisation (FlowActNorms.py:29-43) in its first training-mode pass; here, load initialised parameters or call .eval()")
    w, winv, sum_log_s = invconv_matrices(prm, pfx + "invconv.")
    put(hip.F_ANIC_FWD, np.concatenate([bias, np.exp(logs), w.reshape(-1)]))
    put(hip.F_ANIC_INV, np.concatenate([bias, np.exp(-logs), winv.reshape(-1)]))
    if has_coupling:
        ch = C // 2
        kz = -(-ch // 8) * 8
        for net, name, fw1, fan1, fw2, fan2, fw3, fb3 in (
                (0, "affine.fAffine.", hip.F_W1A, hip.F_AN1A, hip.F_W2A, hip.F_AN2A, hip.F_W3A, hip.F_B3A),
                (1, "affine.fFeatures.", hip.F_W1F, hip.F_AN1F, hip.F_W2F, hip.F_AN2F, hip.F_W3F, hip.F_B3F)):
            q = pfx + name
            if training and not np.any(prm[q + "0.actnorm.bias"] != 0):
                raise NotImplementedError("uninitialised ActNorm inside a coupling network: see FlowUpsamplerNet._actnorm_init_due")
            put(fw1, _pack_conv1(prm[q + "0.weight"].astype(F32), ch if net == 0 else 0, kz if net == 0 else 0))
            put(fan1, _pack_actnorm64(prm[q + "0.actnorm.bias"], prm[q + "0.actnorm.logs"]))
            put(fw2, _pack_conv2(prm[q + "2.weight"].astype(F32)))
            put(fan2, _pack_actnorm64(prm[q + "2.actnorm.bias"], prm[q + "2.actnorm.logs"]))
            wp, bp = _pack_conv3(prm[q + "4.weight"].astype(F32), prm[q + "4.bias"].astype(F32),
                                 prm[q + "4.logs"].astype(F32), C, net)
            put(fw3, wp)
            put(fb3, bp)
            put(hip.F_W3A_WINO if net == 0 else hip.F_W3F_WINO, _pack_conv3_wino(prm[q + "4.weight"].astype(F32), C, net))
            put(hip.F_W1A_BF16 if net == 0 else hip.F_W1F_BF16,
                _pack_conv1_bf16(prm[q + "0.weight"].astype(F32), ch if net == 0 else 0))
            put(hip.F_W1A_WINO if net == 0 else hip.F_W1F_WINO,
                _pack_conv1_wino(prm[q + "0.weight"].astype(F32), ch if net == 0 else 0))
            put(hip.F_W1A_WINO4 if net == 0 else hip.F_W1F_WINO4,
                _pack_conv1_wino4(prm[q + "0.weight"].astype(F32), ch if net == 0 else 0))
            if net == 1:
                put(hip.F_W2F_BF16, _pack_conv2_bf16(prm[q + "2.weight"].astype(F32)))
    return float(np.sum(logs, dtype=F32)), float(sum_log_s)


def pack_flow(prm, kinds, pfx="", training=False):
    """kinds: list of (kind, C) per layer, kind in {"squeeze","step0","stepc"} (synth.layer_plan order).
    Returns (params float32 ndarray, ctypes array of IflLayer)."""
    L = hip.lib()
    offs, total = [], 0
    for kind, C in kinds:
        offs.append(total)
        if kind != "squeeze":
            total += L.ifl_step_param_floats(C, 1 if kind == "stepc" else 0)
    params = np.zeros(max(total, 1), F32)
    layers = (hip.IflLayer * len(kinds))()
    for i, (kind, C) in enumerate(kinds):
        layers[i].C = C
        layers[i].param_off = offs[i]
        if kind == "squeeze":
            layers[i].kind = 0
            continue
        has = kind == "stepc"
        layers[i].kind = 2 if has else 1
        n = L.ifl_step_param_floats(C, 1 if has else 0)
        sl, ss = pack_step(prm, f"{pfx}layers.{i}.", C, has, params[offs[i]:offs[i] + n], training)
        layers[i].sum_logs = sl
        layers[i].sum_log_s = ss
    return params, layers


# ------------------------------------------------------------------------------------- conditioning encoder
def pack_conv_generic(w, cin_pad, cout_pad):
    """nn.Conv2d weight [Cout, Cin, kh, kw] -> convg.hip fragment order
    [group of 32 outputs][chunk of 32 inputs][tap][kq][h][i][e], zero-padded to (cout_pad, cin_pad)."""
    cout, cin, kh, kw = w.shape
    taps = kh * kw
    wp = np.zeros((cout_pad, cin_pad, taps), F32)
    wp[:cout, :cin] = w.reshape(cout, cin, taps)
    wp = wp.reshape(cout_pad // 32, 32, cin_pad // 32, 4, 2, 4, taps)  # g i chunk kq h e tap
    return np.ascontiguousarray(wp.transpose(0, 2, 6, 3, 4, 1, 5)).reshape(-1)


def rrdb_conv_names(nb):
    """State-dict prefixes of the encoder's convolutions in execution order (RRDBNet_arch.py:85-125)."""
    names = ["conv_first"]
    for k in range(nb):
        for r in (1, 2, 3):
            names += [f"RRDB_trunk.{k}.RDB{r}.conv{c}" for c in (1, 2, 3, 4, 5)]
    return names + ["trunk_conv", "upconv1", "upconv2", "HRconv", "conv_last"]


def pack_rrdb(prm, nb, pfx=""):
    """prm: RRDBNet state dict as numpy arrays -> packed float32 block for ifl_rrdb_encode."""
    import ctypes
    L = hip.lib()
    names = rrdb_conv_names(nb)
    assert len(names) == L.ifl_rrdb_conv_count(nb)
    out = np.zeros(L.ifl_rrdb_param_floats(nb), F32)
    cin, cout = ctypes.c_int(), ctypes.c_int()
    wo, bo = ctypes.c_int64(), ctypes.c_int64()
    for i, name in enumerate(names):
        hip.check(L.ifl_rrdb_conv_desc(nb, i, ctypes.byref(cin), ctypes.byref(cout), ctypes.byref(wo), ctypes.byref(bo)),
                  "ifl_rrdb_conv_desc")
        w = prm[f"{pfx}{name}.weight"].astype(F32)
        b = prm[f"{pfx}{name}.bias"].astype(F32)
        assert w.shape[0] <= cout.value and w.shape[1] <= cin.value, (name, w.shape, cin.value, cout.value)
        packed = pack_conv_generic(w, cin.value, cout.value)
        out[wo.value:wo.value + packed.size] = packed
        out[bo.value:bo.value + b.size] = b
    return out
