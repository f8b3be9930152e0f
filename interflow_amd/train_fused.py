"""Gradient-enabled encode of the flow on the level drivers of the HIP library (training path, second version).

`train_flow.run` evaluates the stack layer by layer as a torch graph: about 40 autograd nodes and a few hundred small
launches per FlowStep, and the GPU idles while Python and autograd issue them (profiles/r02_train_kernel_stats.csv).
Here one `torch.autograd.Function` covers a whole flow level (2 + 16 FlowSteps): its forward is ONE library call
(`ifl_train_level_fwd`) after one re-layout launch for all of the level's convolution weights (`ifl_pack_table`), its
backward one call (`ifl_train_level_bwd`) plus one launch that brings the dense weight gradients into the parameters'
layout.  Activations, packed weights, gradients and partial sums live in an arena that is allocated once per shape
(`LevelPlan`); the descriptors that carry its pointers are built once.  What stays in torch: the squeeze between levels, the
composition of each step's per-pixel affine map from its ActNorm and LU parameters (batched over the level's steps, so
autograd handles the chain rule to bias / logs / l / u / log_s in a handful of nodes), and five reductions of per-block
partial sums per level.

Used by `train_flow.run` for the encode direction whenever it applies (`usable`): every ActNorm initialised, conditioning
features that need no gradient (frozen encoder), C in {12, 48}, a free arena.  Everything else (decode with gradients, the
trained-encoder case, the data-dependent ActNorm initialisation pass) stays on the layer-by-layer path.

Reference: FlowUpsamplerNet.encode (FlowUpsamplerNet.py:217-260), FlowStep.normal_flow (FlowStep.py:114-151),
CondAffineSeparatedAndCond.forward (FlowAffineCouplingsAblation.py:159-208), flow.Conv2d / Conv2dZeros (flow.py:25-82),
ActNorm2d (FlowActNorms.py:45-97), InvertibleConv1x1 (Permutations.py:76-123).
"""
from __future__ import annotations

import contextlib
import ctypes
import os
from ctypes import POINTER, Structure, c_float, c_int32, c_int64, c_void_p

import numpy as np
import torch

from . import hip
from . import train_ops as T
from .synth import N_RRDB_CH

MAX_PLANS_PER_SHAPE = 4        # forwards whose backward is still due, per level and shape (each holds an arena)

# Inside `accumulate_into_grad()` (training.train_step wraps loss.backward() in it) the level functions add the parameter
# gradients straight into the parameters' existing .grad tensors with one multi-tensor add per level and hand autograd None
# for them, instead of returning ~290 tensors per level that autograd clones or adds one by one.  Off by default: a plain
# `torch.autograd.grad` / `backward` keeps the usual semantics.
_direct_grad = False
# ... and do that whole weight-gradient phase of a level (batched weight gradients, the Winograd-domain GEMMs, re-layout,
# the multi-tensor add) on a second stream, so that it runs beside the next level's sequential chain of small kernels;
# leaving the `accumulate_into_grad()` block joins that stream.
SIDE_STREAM = os.environ.get("IFL_TRAIN_SIDE_STREAM", "1") != "0"
WINO_FORK = os.environ.get("IFL_TRAIN_WINO_FORK", "1") != "0"
_side = {}
_pending = set()


def _side_stream(device, which=0):
    key = (torch.device(device).index, which)
    if key not in _side:
        _side[key] = torch.cuda.Stream(device=device)
    return _side[key]


def join_side_streams():
    """The caller's current stream waits for the weight-gradient work issued on the side streams."""
    while _pending:
        s = _pending.pop()
        torch.cuda.current_stream(s.device).wait_stream(s)


# Data parallel: `on_ready(params, final)` (training.GradientAllReducer.mark_ready) is called right behind every such add, on the
# stream it was queued on, so that the reducer can launch a bucket's all-reduce while the backward pass goes on.  A step may run
# several flow passes over the same parameters (the consistency term: three encodes and a decode): every gradient-enabled
# forward of a level notes one contribution that is still due, every backward settles one, and a gradient is final -- may be
# exchanged -- when none is left.
_on_ready = None
_contrib = {}


def reset_contributions():
    """Start of a training step: graphs dropped without a backward pass (a skipped step) leave contributions behind."""
    _contrib.clear()


def _note_forward(params):
    if params:
        _contrib[id(params[0])] = _contrib.get(id(params[0]), 0) + 1


def _settle(params):
    """One contribution to these parameters' gradients has just been queued; tell the reducer (final: it was the last)."""
    if not params:
        return
    k = id(params[0])
    n = _contrib.get(k, 1) - 1
    _contrib[k] = max(n, 0)
    if _on_ready is not None:
        _on_ready(params, n <= 0)


class accumulate_into_grad:
    def __init__(self, on_ready=None):
        self.on_ready = on_ready

    def __enter__(self):
        global _direct_grad, _on_ready
        self.prev, _direct_grad = _direct_grad, True
        self.prev_cb, _on_ready = _on_ready, self.on_ready

    def __exit__(self, *exc):
        global _direct_grad, _on_ready
        _direct_grad = self.prev
        _on_ready = self.prev_cb
        join_side_streams()


class IflTrainNet(Structure):
    _fields_ = [(k, c_void_p) for k in ("w1", "w2", "w3", "w1t", "w2t", "w3t", "an1_bias", "an1_logs", "an2_bias", "an2_logs",
                                        "bias3", "logs3", "p1", "a1", "p2", "a2", "hraw", "dW1", "dW2", "dW3", "an1_part",
                                        "an2_part", "dlogs_part", "dbias_part", "ldpart", "dp1", "dp2", "dh")] + \
               [(k, c_int32) for k in ("c0", "n", "hstride", "has_z1", "p1_stride", "pad_")]


class IflTrainStep(Structure):
    _fields_ = [(k, c_void_p) for k in ("M", "c", "dM_part", "z_in", "z_a", "z_mid", "z_out", "z1")] + \
               [("cond", c_int32), ("pad_", c_int32), ("net", IflTrainNet * 2)]


class IflTrainLevel(Structure):
    _fields_ = [(k, c_int32) for k in ("B", "h", "w", "C", "ch", "nsteps", "branch", "ft_ch", "skip_step_wgrad", "groups")] + \
               [("eps", c_float), ("skip_ft_wgrad", c_int32), ("ft", c_void_p), ("z1ring", c_void_p), ("dz1ring", c_void_p),
                ("dz", c_void_p * 4), ("dh", c_void_p * 2), ("d64", c_void_p * 2), ("steps", POINTER(IflTrainStep)),
                ("anp", c_void_p), ("anp_step", c_int64), ("da_all", c_void_p), ("f_batched", c_int32), ("decode", c_int32),
                ("raw", c_void_p), ("raw_step", c_int64)]


assert ctypes.sizeof(IflTrainNet) == 248 and ctypes.sizeof(IflTrainStep) == 568 and ctypes.sizeof(IflTrainLevel) == 192


PACK_DTYPE = np.dtype([("src", "<u8"), ("dst", "<u8"), ("nelem", "<i8"), ("mode", "<i4"), ("cout_real", "<i4"),
                       ("cout_pad", "<i4"), ("taps", "<i4"), ("real0", "<i4"), ("pad0", "<i4"), ("real1", "<i4"),
                       ("pad1", "<i4"), ("a0", "<i4"), ("acnt", "<i4"), ("cin_src", "<i4"), ("pad_", "<i4")])
assert PACK_DTYPE.itemsize == 72
PACK_FWD, PACK_TRANSPOSED, PACK_VECTOR, PACK_GRAD, PACK_WINO4, PACK_WINO4_T = 0, 1, 2, 3, 4, 5
WX_FLOATS = 36 * 64 * N_RRDB_CH          # one network's first-convolution weights in the 6x6 Winograd domain

# The conditioning part of the first convolutions through the Winograd bulk kernel of the inference path (ifl_train_bulk_conv1)
# instead of 32 direct convolutions per level; tests switch it off to compare.
USE_BULK = True
# ... and their weight gradient wrt those 320 channels in the same Winograd domain (with USE_BULK): K-major transforms of ft
# and of the saved output gradients, 36 plain GEMMs over the tiles (torch.bmm = rocBLAS), map back to 3x3 taps
# (ifl_wino4_kmajor / ifl_wino4_dw): 4x fewer multiplications than the direct weight-gradient kernel.
WINO_WGRAD = True
# The coupling networks' other weight gradients (last 3x3, 1x1, z1 taps -- and the conditioning part when WINO_WGRAD is off) as
# one launch per kind, network and level over the gradients the backward saved per step (ifl_conv2d_wgrad_batched) instead
# of one small launch per step.
BATCHED_WGRAD = True
# Samples are independent and the per-step chain is latency-bound at batch 8: the level drivers run the two halves of the
# batch as two concurrent chains (ifl_train_level.groups).
GROUPS = int(os.environ.get("IFL_TRAIN_GROUPS", "2"))
# fFeatures sees only the conditioning: with USE_BULK and BATCHED_WGRAD its ActNorm+ReLU / 1x1 / last convolution run as ONE
# launch each for all 16 steps of a level, before the chain (forward) and after it (backward), and the sequential chain
# only applies its coupling (ifl_train_level.f_batched).
F_BATCHED = os.environ.get("IFL_TRAIN_F_BATCHED", "1") != "0"
WDST_DTYPE = np.dtype([("dst", "<u8"), ("cin", "<i4"), ("zoff", "<i4")])

_bound = False


def _lib():
    global _bound
    L = hip.lib()
    if not _bound:
        L.ifl_pack_table.argtypes = [c_void_p, ctypes.c_int, c_int64, c_void_p]
        L.ifl_pack_table.restype = ctypes.c_int
        L.ifl_wino4_untile.argtypes = [c_void_p, c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void_p]
        L.ifl_wino4_untile.restype = ctypes.c_int
        L.ifl_affine_px_blocks.argtypes = [c_int64]
        L.ifl_affine_px_blocks.restype = c_int64
        L.ifl_train_level_fwd.argtypes = [POINTER(IflTrainLevel), c_void_p]
        L.ifl_train_level_fwd.restype = ctypes.c_int
        L.ifl_train_level_bwd.argtypes = [POINTER(IflTrainLevel), c_void_p, c_void_p, c_void_p, c_void_p]
        L.ifl_train_level_bwd.restype = ctypes.c_int
        L.ifl_train_vg_floats.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.ifl_train_vg_floats.restype = c_int64
        L.ifl_train_bulk_conv1.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, ctypes.c_int, c_void_p, c_int64, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, c_void_p]
        L.ifl_train_bulk_conv1.restype = ctypes.c_int
        L.ifl_wino4_tiles.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.ifl_wino4_tiles.restype = c_int64
        L.ifl_wino4_kmajor.argtypes = [ctypes.c_int, c_void_p, ctypes.c_int, ctypes.c_int, c_int64, c_void_p, c_int64, c_int64,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void_p]
        L.ifl_wino4_kmajor.restype = ctypes.c_int
        L.ifl_wino4_dw.argtypes = [c_void_p, c_int64, c_void_p, ctypes.c_int, c_void_p]
        L.ifl_wino4_dw.restype = ctypes.c_int
        ci = ctypes.c_int
        L.ifl_conv2d_wgrad_batched.argtypes = [c_void_p, ci, ci, ci, c_int64, c_void_p, ci, ci, ci, c_int64, c_void_p, ci, ci, c_int64,
                                               ci, ci, ci, ci, ci, c_void_p]
        L.ifl_conv2d_wgrad_batched.restype = ctypes.c_int
        _bound = True
    return L


def _net_params(seq):
    """The nine parameters of one coupling network, in the order the level function takes them."""
    c1, c2, c3 = seq[0], seq[2], seq[4]
    return [c1.weight, c1.actnorm.bias, c1.actnorm.logs, c2.weight, c2.actnorm.bias, c2.actnorm.logs, c3.weight, c3.bias, c3.logs]


class LevelPlan:
    """Arena, descriptors and re-layout tables of one flow level for one (B, h, w, device)."""

    def __init__(self, steps, B, h, w, C, device, bulk=True, wino_wgrad=True, batched=True, f_batched=True, decode=False):
        L = _lib()
        self.steps, self.B, self.h, self.w, self.C, self.device = list(steps), B, h, w, C, device
        self.decode = bool(decode)            # steps in the order the decode runs them; couplings before the (inverse) affine map
        self.busy = False
        self.bulk = bool(bulk)
        ns = len(self.steps)
        self.cond = [hasattr(s, "affine") for s in self.steps]
        nc = sum(self.cond)
        ch = C // 2
        self.ch = ch
        aff = next((s.affine for s in self.steps if hasattr(s, "affine")), None)
        self.eps = float(aff.affine_eps) if aff is not None else 0.0
        # the encode direction with multReverse (FlowAffineCouplingsAblation.py:164-165,187-207): fAffine on [ch, C) first
        # (its network reads z[:, :ch]), then fFeatures on all channels, both in the "divide" branch
        self.nets = [("fAffine", ch, C - ch, True), ("fFeatures", 0, C, False)]
        self.hstride = [T.pad32(2 * n) for _, _, n, _ in self.nets]
        f32 = dict(dtype=torch.float32, device=device)

        def zeros(*shape):
            return torch.zeros(*shape, **f32)

        self.groups = 2 if (GROUPS == 2 and B >= 2) else 1
        halves = [B // 2, B - B // 2] if self.groups == 2 else [B]
        nblk_a = sum(int(L.ifl_affine_px_blocks(b * h * w)) for b in halves)        # each group's blocks, one after the other
        nblk_an = sum(int(L.ifl_an_relu_blocks(b, h, w)) for b in halves)
        nblk_c = int(L.ifl_coupling_blocks(h, w))
        cin1 = [(32 if z1 else 0) + N_RRDB_CH for _, _, _, z1 in self.nets]
        # ---- saved activations and gradients ----
        self.M, self.c = zeros(ns, C, C), zeros(ns, C)
        self.dM_part = zeros(ns, nblk_a, C * C + C)
        self.z = zeros(ns + 2 * nc + 1, B, h, w, C)                    # z_in of every step, z_a / z_mid of conditional ones, the output
        self.act = zeros(max(nc, 1), 2, 4, B, h + 2, w + 2, 64)        # p1, a1, p2, a2 per (step, network)
        self.hraw = [zeros(max(nc, 1), B, h + 2, w + 2, hs) for hs in self.hstride]
        self.an_part = zeros(max(nc, 1), 2, 2, nblk_an, 128)
        self.an_sum = zeros(max(nc, 1), 2, 2, 128)                     # the partial sums added up (persistent: gradient views below)
        self.cpl_part = [zeros(max(nc, 1), 2, B * nblk_c, 2 * n) for _, _, n, _ in self.nets]      # (d logs3 | d bias3)
        self.cpl_sum = [zeros(max(nc, 1), 2, 2 * n) for _, _, n, _ in self.nets]
        self.ldpart = zeros(max(nc, 1), 2, B, nblk_c)
        # packed weights / dense gradients / parameter-layout gradients: one flat buffer each
        sizes_w, sizes_g, sizes_r = [], [], []
        for i, (_, _, n, z1) in enumerate(self.nets):
            hs = self.hstride[i]
            w1_fwd = (64 * 32 * 9 if z1 else 0) if self.bulk else 64 * cin1[i] * 9        # bulk: only z1's taps run as a direct conv
            sizes_w.append([w1_fwd, 64 * 64, hs * 64 * 9, (32 * 64 * 9) if z1 else 0, 64 * 64, 64 * hs * 9, hs])
            sizes_g.append([64 * cin1[i] * 9, 64 * 64, hs * 64 * 9])
            cin_real = (ch if z1 else 0) + N_RRDB_CH
            sizes_r.append([64 * cin_real * 9, 64 * 64, 2 * n * 64 * 9])
        per_w, per_g, per_r = (sum(sum(x) for x in s) for s in (sizes_w, sizes_g, sizes_r))
        self.wbuf, self.dW, self.graw = zeros(max(nc, 1) * per_w), zeros(max(nc, 1) * per_g), zeros(max(nc, 1) * per_r)
        if self.bulk and nc:
            self.vg = zeros(int(L.ifl_train_vg_floats(B, h, w)))
            self.wx = zeros(nc * 2 * WX_FLOATS + 4096)                   # + slack: the kernel's weight ring reads past the last block
            self.raw = zeros(nc, B, h + 2, w + 2, 128)
        else:
            self.bulk = False
        self.wino_wgrad = bool(wino_wgrad) and self.bulk and nc > 0
        self.batched = bool(batched) and nc > 0
        if self.wino_wgrad or self.batched:
            self.dp1 = zeros(nc * 2, B, h + 2, w + 2, 64)                # gradient wrt p1 of every (step, network), zero rings
        if self.batched:
            self.dp2 = zeros(nc * 2, B, h + 2, w + 2, 64)
            self.dh_all = [zeros(nc, B, h + 2, w + 2, hs) for hs in self.hstride]
            self.z1_all = zeros(nc, B, h + 2, w + 2, 32)
        self.f_batched = bool(f_batched) and self.bulk and self.batched
        if self.f_batched:
            self.anp = zeros(nc, 256)                                    # fFeatures' ActNorm parameters per step (an1 b | l | an2 b | l)
            self.da_all = zeros(nc, B, h + 2, w + 2, 64)
        if self.wino_wgrad:
            K = int(L.ifl_wino4_tiles(B, h, w))
            self.Vt = zeros(36, N_RRDB_CH, K)
            self.dM = zeros(36, nc * 2 * 64, K)
            self.dU = zeros(36, nc * 2 * 64, N_RRDB_CH)
        # ---- scratch ----
        self.z1ring, self.dz1ring = zeros(B, h + 2, w + 2, 32), zeros(B, h + 2, w + 2, 32)
        self.dzs = zeros(4, B, h, w, C)
        self.dh = [zeros(B, h + 2, w + 2, hs) for hs in self.hstride]
        self.d64 = zeros(2, B, h + 2, w + 2, 64)
        self.dz_in = zeros(B, h, w, C)
        self.ft = None                                                  # bound per call (pointer written into the level struct)

        # ---- descriptors ----
        self.c_steps = (IflTrainStep * ns)()
        pack_rows, grad_rows, wdst_rows = [], [], []
        self.dW_first = []                                              # per network: dense gradient pointers of the first conditional step
        self.per_g = per_g
        self.ut_rows, self.ut_tab = [], None                            # the data gradient wrt ft (ensure_ft_grad)
        self.param_slices = []                                          # per parameter of the function: how to find its gradient
        zi, ci = 0, 0
        wo = go = ro = 0
        ptr = lambda t: t.data_ptr()
        self.params = []
        for s, step in enumerate(self.steps):
            S = self.c_steps[s]
            S.M, S.c, S.dM_part = ptr(self.M[s]), ptr(self.c[s]), ptr(self.dM_part[s])
            S.z_in = ptr(self.z[zi])
            S.cond = 1 if self.cond[s] else 0
            if not self.cond[s]:
                S.z_a = ptr(self.z[zi + 1])
                zi += 1
                continue
            S.z_a, S.z_mid, S.z_out = ptr(self.z[zi + 1]), ptr(self.z[zi + 2]), ptr(self.z[zi + 3])
            if self.batched:
                S.z1 = ptr(self.z1_all[ci])
            zi += 3
            for i, (name, c0, n, z1) in enumerate(self.nets):
                N = S.net[i]
                seq = getattr(step.affine, name)
                prm = _net_params(seq)
                self.params += prm
                hs = self.hstride[i]
                w1, an1b, an1l, w2, an2b, an2l, w3, b3, l3 = prm
                assert tuple(w1.shape) == (64, (ch if z1 else 0) + N_RRDB_CH, 3, 3) and tuple(w2.shape) == (64, 64, 1, 1)
                assert tuple(w3.shape) == (2 * n, 64, 3, 3) and b3.numel() == 2 * n and l3.numel() == 2 * n
                assert an1b.numel() == 64 and an2l.numel() == 64
                offs = np.cumsum([0] + sizes_w[i])
                wv = [self.wbuf[wo + offs[k]: wo + offs[k + 1]] for k in range(7)]
                wo += int(offs[-1])
                N.w1 = ptr(wv[0]) if wv[0].numel() else ptr(self.wx)          # (unused for a network whose whole first conv is bulk)
                N.w2, N.w3 = ptr(wv[1]), ptr(wv[2])
                N.w1t = ptr(wv[3]) if z1 else None
                N.w2t, N.w3t, N.bias3 = ptr(wv[4]), ptr(wv[5]), ptr(wv[6])
                N.an1_bias, N.an1_logs, N.an2_bias, N.an2_logs, N.logs3 = ptr(an1b), ptr(an1l), ptr(an2b), ptr(an2l), ptr(l3)
                a = self.act[ci, i]
                N.p1, N.a1, N.p2, N.a2 = ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(a[3])
                N.p1_stride = 64
                if self.bulk and not z1:                                       # p1 = the network's 64 channels of the bulk slab
                    N.p1, N.p1_stride = ptr(self.raw[ci]) + 64 * 4, 128
                N.hraw = ptr(self.hraw[i][ci])
                goffs = np.cumsum([0] + sizes_g[i])
                gv = [self.dW[go + goffs[k]: go + goffs[k + 1]] for k in range(3)]
                go += int(goffs[-1])
                N.dW1, N.dW2, N.dW3 = ptr(gv[0]), ptr(gv[1]), ptr(gv[2])
                N.an1_part, N.an2_part = ptr(self.an_part[ci, i, 0]), ptr(self.an_part[ci, i, 1])
                N.dlogs_part, N.dbias_part = ptr(self.cpl_part[i][ci, 0]), ptr(self.cpl_part[i][ci, 1])
                N.ldpart = ptr(self.ldpart[ci, i])
                if self.wino_wgrad or self.batched:
                    N.dp1 = ptr(self.dp1[ci * 2 + i])
                if self.batched:
                    N.dp2, N.dh = ptr(self.dp2[ci * 2 + i]), ptr(self.dh_all[i][ci])
                    if ci == 0:
                        self.dW_first.append((ptr(gv[0]), ptr(gv[1]), ptr(gv[2])))
                N.c0, N.n, N.hstride, N.has_z1 = c0, n, hs, 1 if z1 else 0
                r0, p0 = (ch, 32) if z1 else (0, 0)
                common1 = dict(cout_real=64, cout_pad=64, taps=9, real0=r0, pad0=p0, real1=N_RRDB_CH, pad1=N_RRDB_CH)
                common2 = dict(cout_real=64, cout_pad=64, taps=1, real0=64, pad0=64, real1=0, pad1=0)
                common3 = dict(cout_real=2 * n, cout_pad=hs, taps=9, real0=64, pad0=64, real1=0, pad1=0)
                if not self.bulk:
                    pack_rows.append(dict(src=ptr(w1), dst=ptr(wv[0]), nelem=wv[0].numel(), mode=PACK_FWD, **common1))
                else:
                    if z1:
                        pack_rows.append(dict(src=ptr(w1), dst=ptr(wv[0]), nelem=wv[0].numel(), mode=PACK_FWD, cout_real=64, cout_pad=64,
                                              taps=9, real0=ch, pad0=32, real1=0, pad1=0, cin_src=ch + N_RRDB_CH))
                    wxv = self.wx[(ci * 2 + i) * WX_FLOATS:(ci * 2 + i + 1) * WX_FLOATS]
                    pack_rows.append(dict(src=ptr(w1), dst=ptr(wxv), nelem=WX_FLOATS, mode=PACK_WINO4, cout_real=64, cout_pad=64, taps=9,
                                          real0=ch if z1 else 0, real1=N_RRDB_CH, cin_src=(ch if z1 else 0) + N_RRDB_CH))
                pack_rows += [dict(src=ptr(w2), dst=ptr(wv[1]), nelem=wv[1].numel(), mode=PACK_FWD, **common2),
                              dict(src=ptr(w3), dst=ptr(wv[2]), nelem=wv[2].numel(), mode=PACK_FWD, **common3),
                              dict(src=ptr(w2), dst=ptr(wv[4]), nelem=wv[4].numel(), mode=PACK_TRANSPOSED, a0=0, acnt=64, **common2),
                              dict(src=ptr(w3), dst=ptr(wv[5]), nelem=wv[5].numel(), mode=PACK_TRANSPOSED, a0=0, acnt=64, **common3),
                              dict(src=ptr(b3), dst=ptr(wv[6]), nelem=hs, mode=PACK_VECTOR, cout_real=2 * n, cout_pad=hs, taps=1)]
                if z1:
                    pack_rows.append(dict(src=ptr(w1), dst=ptr(wv[3]), nelem=wv[3].numel(), mode=PACK_TRANSPOSED, a0=0, acnt=32, **common1))
                elif self.f_batched:
                    for k, prm_ in enumerate((an1b, an1l, an2b, an2l)):
                        pack_rows.append(dict(src=ptr(prm_), dst=ptr(self.anp[ci, k * 64:(k + 1) * 64]), nelem=64, mode=PACK_VECTOR,
                                              cout_real=64, cout_pad=64, taps=1))
                roffs = np.cumsum([0] + sizes_r[i])
                rv = [self.graw[ro + roffs[k]: ro + roffs[k + 1]] for k in range(3)]
                ro += int(roffs[-1])
                grad_rows += [dict(src=ptr(gv[0]), dst=ptr(rv[0]), nelem=rv[0].numel(), mode=PACK_GRAD, **common1),
                              dict(src=ptr(gv[1]), dst=ptr(rv[1]), nelem=rv[1].numel(), mode=PACK_GRAD, **common2),
                              dict(src=ptr(gv[2]), dst=ptr(rv[2]), nelem=rv[2].numel(), mode=PACK_GRAD, **common3)]
                wdst_rows.append((ptr(rv[0]), (ch if z1 else 0) + N_RRDB_CH, ch if z1 else 0))
                self.ut_rows.append(dict(src=ptr(w1), nelem=36 * N_RRDB_CH * 64, mode=PACK_WINO4_T, cout_real=64, cout_pad=64, taps=9,
                                         real0=ch if z1 else 0, real1=N_RRDB_CH, cin_src=(ch if z1 else 0) + N_RRDB_CH,
                                         a0=(ci * 2 + i) * 64, acnt=nc * 2 * 64))
                self.param_slices += [("w", rv[0], w1.shape), ("an", (ci, i, 0, 0), an1b.shape), ("an", (ci, i, 0, 1), an1l.shape),
                                      ("w", rv[1], w2.shape), ("an", (ci, i, 1, 0), an2b.shape), ("an", (ci, i, 1, 1), an2l.shape),
                                      ("w", rv[2], w3.shape), ("cpl", (i, ci, 1), b3.shape), ("cpl", (i, ci, 0), l3.shape)]
            ci += 1
        # the gradient of every parameter as a view of the persistent buffers, in the order of self.params
        self.grad_views = []
        for kind, where, shape in self.param_slices:
            if kind == "w":
                self.grad_views.append(where.view(shape))
            elif kind == "an":
                s_, i_, a_, k_ = where
                self.grad_views.append(self.an_sum[s_, i_, a_, k_ * 64:(k_ + 1) * 64].reshape(shape))
            else:
                i_, s_, k_ = where
                self.grad_views.append(self.cpl_sum[i_][s_, k_].reshape(shape))
        assert all(v.data_ptr() != 0 and v._base is not None for v in self.grad_views)
        assert zi == self.z.shape[0] - 1
        self.z_first, self.z_last = self.z[0], self.z[zi]
        self.param_ptrs = tuple(p.data_ptr() for p in self.params)

        self.pack_tab = self._table(pack_rows)
        self.grad_tab = self._table(grad_rows)

        lv = IflTrainLevel()
        # multReverse (FlowAffineCouplingsAblation.py:164-165): the encode runs the "divide" branch (1), the decode the "multiply" one
        lv.B, lv.h, lv.w, lv.C, lv.ch, lv.nsteps, lv.branch, lv.ft_ch = B, h, w, C, ch, ns, 0 if self.decode else 1, N_RRDB_CH
        lv.decode = 1 if self.decode else 0
        lv.eps = self.eps
        lv.groups = self.groups
        lv.z1ring, lv.dz1ring = ptr(self.z1ring), ptr(self.dz1ring)
        for k in range(4):
            lv.dz[k] = ptr(self.dzs[k])
        for k in range(2):
            lv.dh[k] = ptr(self.dh[k])
            lv.d64[k] = ptr(self.d64[k])
        lv.steps = ctypes.cast(self.c_steps, POINTER(IflTrainStep))
        if self.bulk:
            lv.raw, lv.raw_step = ptr(self.raw), self.raw[0].numel()
        if self.batched:
            lv.skip_step_wgrad = 1
        if self.f_batched:
            lv.f_batched, lv.anp, lv.anp_step, lv.da_all = 1, ptr(self.anp), 256, ptr(self.da_all)
        if self.wino_wgrad:
            lv.skip_ft_wgrad = 1
            arr = np.zeros(len(wdst_rows), dtype=WDST_DTYPE)
            for k, r in enumerate(wdst_rows):
                arr[k] = r
            self.wdst = torch.from_numpy(arr.view(np.uint8)).to(device)
        self.level = lv
        self.nc = nc

    def _table(self, rows):
        if not rows:
            return None, 0, 0
        arr = np.zeros(len(rows), dtype=PACK_DTYPE)
        for k, r in enumerate(rows):
            for key, v in r.items():
                arr[k][key] = v
        return torch.from_numpy(arr.view(np.uint8)).to(self.device), len(rows), int(arr["nelem"].max())

    @property
    def can_ft_grad(self):
        return self.bulk and hasattr(self, "dp1") and self.nc > 0

    def ensure_ft_grad(self):
        """Buffers of the gradient wrt the conditioning features (allocated when an encoder is trained): U' [36][320][64 pairs],
        the transformed output gradients [36][64 pairs][tiles] and the product [36][320][tiles]."""
        if self.ut_tab is None:
            K = int(_lib().ifl_wino4_tiles(self.B, self.h, self.w))
            rows = self.nc * 2 * 64
            f32 = dict(dtype=torch.float32, device=self.device)
            self.Ut = torch.zeros(36, N_RRDB_CH, rows, **f32)
            self.Xd = torch.zeros(36, rows, K, **f32)
            self.Yd = torch.zeros(36, N_RRDB_CH, K, **f32)
            self.ut_tab = self._table([dict(r, dst=self.Ut.data_ptr()) for r in self.ut_rows])

    def ft_grad(self, dp1=None):
        """d loss / d ft [B, 320, h, w] from the saved gradients wrt the first convolutions' outputs (call after the level's
        backward; dp1: planes of the same shape to use instead, e.g. the sum over several passes): sum over all (step, network) pairs of the 'same' convolution with the rotated taps, as 36 GEMMs in the
        Winograd F(4x4,3x3) domain (a quarter of the direct form's multiplications, one launch instead of 32 convolutions
        and 31 adds of 262 MB tensors)."""
        L = _lib()
        self.ensure_ft_grad()
        self.run_table(self.ut_tab)
        rows = self.Xd.shape[1]
        dp1 = self.dp1 if dp1 is None else dp1
        assert dp1.shape == self.dp1.shape and dp1.is_contiguous()
        hip.check(L.ifl_wino4_kmajor(0, hip._ptr(dp1), 64, dp1.shape[0], dp1[0].numel(), hip._ptr(self.Xd), rows, 0,
                                     self.B, self.h, self.w, hip._stream()), "ifl_wino4_kmajor")
        torch.bmm(self.Ut, self.Xd, out=self.Yd)
        dft = torch.empty(self.B, N_RRDB_CH, self.h, self.w, dtype=torch.float32, device=self.device)
        hip.check(L.ifl_wino4_untile(hip._ptr(self.Yd), hip._ptr(dft), N_RRDB_CH, self.B, self.h, self.w, hip._stream()), "ifl_wino4_untile")
        return dft

    def batched_wgrads(self, ft_ring):
        """The coupling networks' weight gradients of all conditional steps, one launch per (kind, network)."""
        L = _lib()
        B, h, w, nc = self.B, self.h, self.w, self.nc
        st = hip._stream()
        p64 = self.dp1[0].numel()

        def go(x, xs, cin, xz, dy, dys, cout, dyz, dW, cin_total, ci0, taps):
            hip.check(L.ifl_conv2d_wgrad_batched(x, xs, 0, cin, xz, dy, dys, 0, cout, dyz, dW, cin_total, ci0, self.per_g, nc, taps,
                                                 B, h, w, st), "ifl_conv2d_wgrad_batched")
        for i, (_, _, _, z1) in enumerate(self.nets):
            hs = self.hstride[i]
            dW1, dW2, dW3 = self.dW_first[i]
            astep = self.act.stride(0)
            go(hip._ptr(self.act[0, i, 3]), 64, 64, astep, hip._ptr(self.dh_all[i]), hs, hs, self.dh_all[i][0].numel(), dW3, 64, 0, 9)
            go(hip._ptr(self.act[0, i, 1]), 64, 64, astep, hip._ptr(self.dp2[i]), 64, 64, 2 * p64, dW2, 64, 0, 1)
            cin_total = (32 if z1 else 0) + N_RRDB_CH
            if z1:
                go(hip._ptr(self.z1_all), 32, 32, self.z1_all[0].numel(), hip._ptr(self.dp1[i]), 64, 64, 2 * p64, dW1, cin_total, 0, 9)
            if not self.wino_wgrad:
                go(hip._ptr(ft_ring), N_RRDB_CH, N_RRDB_CH, 0, hip._ptr(self.dp1[i]), 64, 64, 2 * p64, dW1, cin_total, 1 if z1 else 0, 9)

    def matches(self, params):
        return len(params) == len(self.param_ptrs) and all(p.data_ptr() == q for p, q in zip(params, self.param_ptrs))

    def run_table(self, tab):
        t, n, mx = tab
        if n:
            hip.check(_lib().ifl_pack_table(hip._ptr(t), n, mx, hip._stream()), "ifl_pack_table")


class _Lease:
    """Frees the plan when the backward has run or the graph that holds it is dropped."""

    def __init__(self, plan):
        self.plan = plan

    def release(self):
        if self.plan is not None:
            self.plan.busy = False
            self.plan = None

    def __del__(self):
        self.release()


# ---- one d ft product for several passes over the same conditioning features ------------------------------------------------
# The training step with the LR-consistency term runs four gradient-enabled passes (three encodes, one decode) over the same
# features with the same weights, and d ft is linear in the saved first-convolution gradients: inside `shared_ft_grad()` every
# pass adds its dp1 planes into an accumulator owned by a hub node that sits between the features and the passes; autograd
# runs the hub's backward after all of them, and that is where the ONE Winograd-domain product per level happens.
# The same context shares the z-independent part of the FORWARD: the first convolutions' conditioning part of both networks
# (the Winograd bulk launches) and the whole of fFeatures' network depend on the features and the weights only, so the second,
# third and fourth pass copy them from the first pass's arena (a decode arena lists the steps in reverse) instead of
# computing them again.
_ft_hubs = None
_fwd_donors = None


class shared_ft_grad:
    def __enter__(self):
        global _ft_hubs, _fwd_donors
        self.prev = (_ft_hubs, _fwd_donors)
        _ft_hubs, _fwd_donors = {}, {}
        return self

    def __exit__(self, *exc):
        global _ft_hubs, _fwd_donors
        _ft_hubs, _fwd_donors = self.prev


class _FtHub:
    def __init__(self, ft):
        self.ft, self.acc, self.plan = ft, None, None              # (ft kept alive: the registry is keyed on its id)
        self.out, self.tok = _FtHubFn.apply(self, ft)

    def add(self, plan):
        if self.acc is None:
            self.acc, self.plan = plan.dp1.clone(), plan           # planes in this plan's (step, network) order, with its U'
            return
        assert self.acc.shape == plan.dp1.shape and plan.nc == self.plan.nc
        d = plan.dp1
        if plan.decode != self.plan.decode:                        # a decode plan lists the level's steps in reverse
            d = d.view(plan.nc, 2, *d.shape[1:]).flip(0).reshape(d.shape)
        self.acc.add_(d)

    def finish(self):
        if self.acc is None:
            return None
        dft = self.plan.ft_grad(self.acc)
        self.acc = self.plan = None
        return dft


class _FtHubFn(torch.autograd.Function):
    """Identity on the features plus a one-element token every pass takes as an input and returns a (zero) gradient for: the
    hub's backward is then scheduled after every pass's backward whatever they return for the features themselves."""

    @staticmethod
    def forward(ctx, hub, ft):
        ctx.hub = hub
        ctx.set_materialize_grads(False)
        return ft.view_as(ft), torch.zeros(1, dtype=torch.float32, device=ft.device)

    @staticmethod
    def backward(ctx, g_ft, g_tok):
        dft = ctx.hub.finish()
        if dft is None:
            return None, g_ft
        return None, dft if g_ft is None else dft + g_ft


def _ring_of(ft):
    """The features as the kernels read them (ringed channels-last); inside `shared_ft_grad()` made once per tensor."""
    if _fwd_donors is None:
        return T.to_ring(ft.detach().float())
    key = ("ring", id(ft))
    r = _fwd_donors.get(key)
    if r is None:
        r = _fwd_donors[key] = (ft, T.to_ring(ft.detach().float()))      # (ft kept alive: keyed on its id)
    return r[1]


def _hub_for(ft):
    if _ft_hubs is None:
        return None
    h = _ft_hubs.get(id(ft))
    if h is None:
        h = _ft_hubs[id(ft)] = _FtHub(ft)
    return h


MARK = None            # tools/phase_events.py: a callable(name) invoked at the phase boundaries of the level drivers


def _mark(name):
    if MARK is not None:
        MARK(name)


class _LevelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, ft_ring, ft_src, hub, tok, share_key, z, M_all, c_all, *params):
        # ft_ring: the conditioning features as the kernels read them (ringed channels-last, no gradient); ft_src: the NCHW
        # tensor they were made from when it needs a gradient (a trained encoder), else None; hub, tok: inside
        # `shared_ft_grad()` the hub that computes that gradient for all passes at once, and its token
        ctx.hub = hub
        L = _lib()
        _mark("before level")
        if any(ctx.needs_input_grad[9:]):
            _note_forward(plan.params)
        plan.busy = True
        ctx.lease = _Lease(plan)
        ctx.ft_ring = ft_ring                           # the backward reads it (weight gradients of the first convolutions)
        with torch.cuda.device(plan.device):
            plan.z_first.copy_(z)
            plan.M.copy_(M_all)
            plan.c.copy_(c_all)
            plan.level.ft = ft_ring.data_ptr() if ft_ring is not None else None
            plan.run_table(plan.pack_tab)
            _mark("pack")
            donor = None
            if _fwd_donors is not None and plan.f_batched and share_key is not None:
                donor = _fwd_donors.get(share_key)
                if donor is None or donor is plan or donor.nc != plan.nc or donor.raw.shape != plan.raw.shape:
                    _fwd_donors[share_key] = plan
                    donor = None
            plan.level.f_batched = 1 if plan.f_batched else 0
            if donor is not None:
                flip = (lambda t: t.flip(0)) if donor.decode != plan.decode else (lambda t: t)
                plan.raw.copy_(flip(donor.raw))
                plan.act[:, 1, 1:].copy_(flip(donor.act[:, 1, 1:]))        # fFeatures: a1, p2, a2 (its p1 is the slab)
                plan.hraw[1].copy_(flip(donor.hraw[1]))
                plan.level.f_batched = 2
            elif plan.bulk:
                hip.check(L.ifl_train_bulk_conv1(hip._ptr(ft_ring), hip._ptr(plan.vg), hip._ptr(plan.wx), 2 * WX_FLOATS, plan.nc,
                                                 hip._ptr(plan.raw), plan.raw[0].numel(), plan.B, plan.h, plan.w, hip._stream()),
                          "ifl_train_bulk_conv1")
            _mark("bulk conv1")
            hip.check(L.ifl_train_level_fwd(ctypes.byref(plan.level), hip._stream()), "ifl_train_level_fwd")
            _mark("level fwd")
            ld = plan.ldpart.sum(dim=(0, 1, 3))
        return plan.z_last.clone(), ld

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gz, gld):
        plan = ctx.lease.plan
        assert plan is not None, "the level's arena was released before its backward ran"
        L = _lib()
        f32 = dict(dtype=torch.float32, device=plan.device)
        gz = torch.zeros(plan.z_last.shape, **f32) if gz is None else gz.contiguous().float()
        gld = torch.zeros(plan.B, **f32) if gld is None else gld.contiguous().float()
        C = plan.C
        with torch.cuda.device(plan.device):
            plan.level.ft = ctx.ft_ring.data_ptr() if ctx.ft_ring is not None else None
            _mark("before level bwd")
            plan.dW.zero_()
            hip.check(L.ifl_train_level_bwd(ctypes.byref(plan.level), hip._ptr(gz), hip._ptr(gld), hip._ptr(plan.dz_in), hip._stream()),
                      "ifl_train_level_bwd")
            _mark("level bwd")
            dMc = plan.dM_part.sum(dim=1)                                  # [steps, C*C + C]
            dz = plan.dz_in.clone()
            direct = _direct_grad and all(p.grad is not None and p.grad.dtype == torch.float32 for p in plan.params)
            if direct and SIDE_STREAM:
                # everything below only produces parameter gradients: on a second stream, beside the next level's chain
                main = torch.cuda.current_stream()
                side = _side_stream(plan.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    grads = _LevelFn._weight_phase(plan, ctx.ft_ring, True)
                _pending.add(side)
            else:
                grads = _LevelFn._weight_phase(plan, ctx.ft_ring, direct)
            dft = gtok = None
            if ctx.hub is not None:
                ctx.hub.add(plan)                                          # (the hub's backward makes the product)
                gtok = torch.zeros(1, **f32)
            elif ctx.needs_input_grad[2]:
                dft = plan.ft_grad()
        ctx.lease.release()
        return (None, None, dft, None, gtok, None, dz, dMc[:, :C * C].reshape(-1, C, C), dMc[:, C * C:]) + tuple(grads)

    @staticmethod
    def _weight_phase(plan, ft_ring, direct):
        """Weight gradients of the level's coupling networks from what the chain saved, brought into the parameters' layout;
        `direct`: added into the parameters' .grad here (one multi-tensor add), else returned."""
        L = _lib()
        fork = None
        if plan.wino_wgrad and direct and SIDE_STREAM and WINO_FORK:
            # the Winograd-domain part (two transforms, 36 GEMMs) on a stream of its own beside the batched weight gradients:
            # both are one-round launches at a fraction of the matrix peak, and the last level's weight phase is the step's tail
            cur = torch.cuda.current_stream()
            fork = _side_stream(plan.device, 1)
            fork.wait_stream(cur)
        if plan.batched:
            plan.batched_wgrads(ft_ring)
        plan.run_table(plan.grad_tab)
        if plan.wino_wgrad:
            rows = plan.dM.shape[1]
            ctx_ = torch.cuda.stream(fork) if fork is not None else contextlib.nullcontext()
            with ctx_:
                hip.check(L.ifl_wino4_kmajor(0, hip._ptr(ft_ring), N_RRDB_CH, 1, 0, hip._ptr(plan.Vt), N_RRDB_CH, 0,
                                             plan.B, plan.h, plan.w, hip._stream()), "ifl_wino4_kmajor")
                hip.check(L.ifl_wino4_kmajor(1, hip._ptr(plan.dp1), 64, plan.dp1.shape[0], plan.dp1[0].numel(), hip._ptr(plan.dM), rows, 0,
                                             plan.B, plan.h, plan.w, hip._stream()), "ifl_wino4_kmajor")
                torch.bmm(plan.dM, plan.Vt.transpose(1, 2), out=plan.dU)
            if fork is not None:
                torch.cuda.current_stream().wait_stream(fork)         # (the map back to taps writes where the re-layout table wrote)
            hip.check(L.ifl_wino4_dw(hip._ptr(plan.dU), rows, hip._ptr(plan.wdst), plan.dp1.shape[0], hip._stream()), "ifl_wino4_dw")
        torch.sum(plan.an_part, dim=3, out=plan.an_sum)               # [cond steps, network, actnorm, 128]
        for t, o in zip(plan.cpl_part, plan.cpl_sum):                  # per network slot: [cond steps, (logs3 | bias3), 2n]
            torch.sum(t, dim=2, out=o)
        if direct:
            torch._foreach_add_([p.grad for p in plan.params], plan.grad_views)
            _settle(plan.params)                                       # (on the stream the add was queued on)
            return [None] * len(plan.params)
        _settle(())
        return [g.clone() for g in plan.grad_views]


class _StackGroups(torch.autograd.Function):
    """k stacks of n same-shaped parameters each (inputs: group 0's n tensors, then group 1's, ...).  torch.stack's own
    backward hands every parameter its slice through AccumulateGrad -- one small add launch per parameter, 90 per level;
    inside `accumulate_into_grad()` the slices go into the existing .grad tensors with one multi-tensor add instead."""

    @staticmethod
    def forward(ctx, k, *params):
        n = len(params) // k
        ctx.k, ctx.n, ctx.params = k, n, params
        ctx.set_materialize_grads(False)                  # a stack nobody differentiates leaves its parameters' .grad alone
        if any(ctx.needs_input_grad[1:]):
            _note_forward(params)
        return tuple(torch.stack(params[g * n:(g + 1) * n]) for g in range(k))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        k, n, params = ctx.k, ctx.n, ctx.params
        need = [(p, grads[i // n][i % n]) for i, p in enumerate(params) if ctx.needs_input_grad[1 + i] and grads[i // n] is not None]
        if _direct_grad and all(p.grad is not None and p.grad.dtype == g.dtype for p, g in need):
            if need:
                torch._foreach_add_([p.grad for p, _ in need], [g for _, g in need])
            _settle(params)
            return (None,) * (1 + len(params))
        k = id(params[0]) if params else None
        if k in _contrib:
            _contrib[k] = max(_contrib[k] - 1, 0)
        out = [None] * len(params)
        for i, p in enumerate(params):
            if ctx.needs_input_grad[1 + i] and grads[i // n] is not None:
                out[i] = grads[i // n][i % n]
        return (None,) + tuple(out)


def _affine_maps(steps, C, pixels, decode=False):
    """Per step: M = W diag(exp(logs)), c = M bias (ActNorm2d then InvertibleConv1x1), batched over the level's steps; and the
    level's data-independent log-det  pixels * sum_steps (sum logs + log|det W|)  (FlowActNorms.py:69-85, Permutations.py:76-101).
    decode: the inverse maps, z' = exp(-logs) (W^-1 z) - bias, i.e. M = diag(exp(-logs)) W^-1, c = -bias, with W^-1 =
    U^-1 L^-1 P^T from float64 inverses of the triangular factors as the reference (Permutations.py:103-112); the caller
    subtracts the log-det."""
    dev = steps[0].actnorm.bias.device
    ics = [s.invconv for s in steps]
    if all(ic.LU for ic in ics):
        logs, bias, l_raw, log_s, u_raw = _StackGroups.apply(5, *[s.actnorm.logs for s in steps], *[s.actnorm.bias for s in steps],
                                                             *[ic.l for ic in ics], *[ic.log_s for ic in ics], *[ic.u for ic in ics])
        logs, bias = logs.reshape(len(steps), -1), bias.reshape(len(steps), -1)
        l_mask = torch.tril(torch.ones(C, C, device=dev), -1)
        eye = torch.eye(C, device=dev)
        l = l_raw * l_mask + eye
        sign_s = torch.stack([ic.sign_s for ic in ics])
        u = u_raw * l_mask.t() + torch.diag_embed(sign_s * torch.exp(log_s))
        P = torch.stack([ic.p for ic in ics])
        if decode:
            li = torch.linalg.inv_ex(l.double())[0].float()        # (inv_ex: no error flag read on the host)
            ui = torch.linalg.inv_ex(u.double())[0].float()
            W = torch.matmul(ui, torch.matmul(li, P.transpose(1, 2)))      # p^-1 = p^T exactly
        else:
            W = torch.matmul(P, torch.matmul(l, u))
        dl = log_s.sum()
    else:
        logs = torch.stack([s.actnorm.logs.reshape(-1) for s in steps])
        bias = torch.stack([s.actnorm.bias.reshape(-1) for s in steps])
        from .train_flow import invconv_weight
        ws, dls = zip(*(invconv_weight(ic, decode) for ic in ics))
        W, dl = torch.stack(ws), torch.stack(dls).sum()
    if decode:
        return W * torch.exp(-logs)[:, :, None], -bias, (logs.sum() + dl) * pixels
    M = W * torch.exp(logs)[:, None, :]
    c = torch.matmul(M, bias[:, :, None]).squeeze(-1)
    return M, c, (logs.sum() + dl) * pixels


def usable(fu, feats) -> bool:
    """Whether the fused level path applies to this gradient-enabled encode."""
    kinds = [k for k, _ in fu._kinds]
    if any(k not in ("squeeze", "step0", "stepc") for k in kinds):
        return False
    if any(ft.requires_grad for ft in feats.values()) and not (USE_BULK and (WINO_WGRAD or BATCHED_WGRAD)):
        return False                          # gradients towards the conditioning encoder come from the saved first-convolution gradients
    from .training import _uninitialised_actnorms
    if _uninitialised_actnorms(fu.layers):
        return False                          # data-dependent ActNorm initialisation is still due
    return all(c in (12, 48) for k, c in fu._kinds if k != "squeeze")


def _plan_for(fu, level, steps, B, h, w, C, device, params, decode=False):
    cache = fu.__dict__.setdefault("_train_plans", {})
    key = (level, B, h, w, C, str(device), F_BATCHED, GROUPS, BATCHED_WGRAD, USE_BULK, WINO_WGRAD, bool(decode))
    plans = cache.setdefault(key, [])
    plans[:] = [p for p in plans if p.matches(params)]             # parameters moved (.to(), load): drop stale descriptors
    for p in plans:
        if not p.busy:
            return p
    if len(plans) >= MAX_PLANS_PER_SHAPE:
        return None
    p = LevelPlan(steps, B, h, w, C, device, bulk=USE_BULK, wino_wgrad=WINO_WGRAD, batched=BATCHED_WGRAD, f_batched=F_BATCHED, decode=decode)
    plans.append(p)
    return p


def run_encode(fu, src, feats, ld):
    """FlowUpsamplerNet.encode with gradients on the level drivers.  src: NCHW float32 on the GPU; feats: {level: NCHW
    conditioning map}; ld: [B] log-det so far.  Returns (z NCHW, logdet [B]) or None when no arena is free."""
    from .train_flow import squeeze_cl
    levels, cur = [], None
    for layer, (kind, C) in zip(fu.layers, fu._kinds):
        if kind == "squeeze":
            cur = []
            levels.append((C, cur))
        else:
            cur.append(layer)
    z = src.float().permute(0, 2, 3, 1)
    B = src.shape[0]
    plans = []
    h, w = z.shape[1], z.shape[2]
    for lv, (C, steps) in enumerate(levels, start=1):
        h, w = h // 2, w // 2
        pcache = fu.__dict__.setdefault("_train_params", {})
        if lv not in pcache:                      # the modules (and their Parameter objects) are fixed after construction
            pcache[lv] = [p for s in steps if hasattr(s, "affine") for name in ("fAffine", "fFeatures") for p in _net_params(getattr(s.affine, name))]
        params = pcache[lv]
        plan = _plan_for(fu, lv, steps, B, h, w, C, src.device, params)
        if plan is None:
            return None
        plans.append((plan, params, steps, C, h, w, lv))
    # every level's affine maps first: their backward nodes (the chain rule to the ActNorm / LU parameters, some forty small
    # launches per level) then run after the last level's backward -- beside the weight phase on the side stream -- and not
    # between two levels on the critical path
    maps = [_affine_maps(steps, C, h * w) for _, _, steps, C, h, w, _ in plans]
    for (plan, params, steps, C, h, w, lv), (M, c, ld_const) in zip(plans, maps):
        z = squeeze_cl(z).contiguous()
        ft = feats.get(lv)
        ft_ring = None
        if any(plan.cond):
            assert ft is not None and ft.shape[1] == N_RRDB_CH and tuple(ft.shape[2:]) == (h, w), (lv, None if ft is None else tuple(ft.shape))
            ft_ring = _ring_of(ft)
        ft_src = ft if (ft is not None and ft_ring is not None and ft.requires_grad) else None
        if ft_src is not None and not plan.can_ft_grad:
            return None
        hub = _hub_for(ft) if ft_src is not None else None
        tok = None
        if hub is not None:
            ft_src, tok = hub.out, hub.tok
        share_key = (id(ft), lv) if (ft is not None and _fwd_donors is not None) else None
        z, ld_lv = _LevelFn.apply(plan, ft_ring, ft_src, hub, tok, share_key, z, M, c, *params)
        ld = ld + ld_lv + ld_const
    return z.permute(0, 3, 1, 2).contiguous(), ld


def run_decode(fu, z_nchw, feats, ld):
    """FlowUpsamplerNet.decode (FlowUpsamplerNet.py:271-304) with gradients on the level drivers: the levels from the last to
    the first, each level's steps in reverse with the couplings before the inverse affine map (FlowStep.py:153-196), an
    unsqueeze after each level.  z_nchw: [B, C_last, h, w]; feats: {level: NCHW conditioning map}.  Returns (x NCHW, logdet)
    or None when no arena is free."""
    from .train_flow import unsqueeze_cl
    levels, cur = [], None
    for layer, (kind, C) in zip(fu.layers, fu._kinds):
        if kind == "squeeze":
            cur = []
            levels.append((C, cur))
        else:
            cur.append(layer)
    z = z_nchw.float().permute(0, 2, 3, 1)
    B, h, w = z.shape[0], z.shape[1], z.shape[2]
    assert z.shape[3] == levels[-1][0], (tuple(z.shape), levels[-1][0])
    plans = []
    for lv in range(len(levels), 0, -1):
        C, steps = levels[lv - 1]
        steps = list(reversed(steps))
        pcache = fu.__dict__.setdefault("_train_params_dec", {})
        if lv not in pcache:
            pcache[lv] = [p for s in steps if hasattr(s, "affine") for name in ("fAffine", "fFeatures") for p in _net_params(getattr(s.affine, name))]
        params = pcache[lv]
        plan = _plan_for(fu, lv, steps, B, h, w, C, z.device, params, decode=True)
        if plan is None:
            return None
        plans.append((plan, params, steps, C, h, w, lv))
        h, w = h * 2, w * 2
    maps = [_affine_maps(steps, C, h_ * w_, decode=True) for _, _, steps, C, h_, w_, _ in plans]
    for (plan, params, steps, C, h_, w_, lv), (M, c, ld_const) in zip(plans, maps):
        ft = feats.get(lv)
        ft_ring = None
        if any(plan.cond):
            assert ft is not None and ft.shape[1] == N_RRDB_CH and tuple(ft.shape[2:]) == (h_, w_), (lv, None if ft is None else tuple(ft.shape))
            ft_ring = _ring_of(ft)
        ft_src = ft if (ft is not None and ft_ring is not None and ft.requires_grad) else None
        if ft_src is not None and not plan.can_ft_grad:
            return None
        hub = _hub_for(ft) if ft_src is not None else None
        tok = None
        if hub is not None:
            ft_src, tok = hub.out, hub.tok
        share_key = (id(ft), lv) if (ft is not None and _fwd_donors is not None) else None
        z, ld_lv = _LevelFn.apply(plan, ft_ring, ft_src, hub, tok, share_key, z.contiguous(), M, c, *params)
        ld = ld + ld_lv - ld_const
        z = unsqueeze_cl(z)
    return z.permute(0, 3, 1, 2).contiguous(), ld
