"""The conditioning encoder with gradients as two library calls (csrc/rrdb.hip: ifl_rrdb_train_fwd / ifl_rrdb_train_bwd).

The reference trains the encoder with the flow (`network_G.train_RRDB`, RRDBGlowNet_arch.py:26-33; train.py:55 hands every
parameter that requires grad to Adam).  `RRDBNet._forward_hipconv` evaluates that as ~350 autograd Functions with ATen glue
between them (concatenations, LeakyReLU, residuals): 18 000 launches per step.  Here the forward keeps every dense block's
buffer in a caller-owned workspace, the backward walks the blocks in reverse inside the library (one data-gradient launch
per convolution, LeakyReLU's derivative in its epilogue) and computes the weight gradients of the same convolution of all
69 blocks in one launch.  Only 'fea_up1' and 'fea_up2' -- the maps the shipped flow reads -- are produced; upconv2, HRconv
and conv_last get no gradient from the flow's loss (as in the reference, where their .grad stays None)."""
from __future__ import annotations

import ctypes
from ctypes import c_int, c_int64, c_void_p

import numpy as np
import torch

from . import hip, pack
from . import train_fused as TF

_declared = False


def _lib():
    global _declared
    L = hip.lib()
    if not _declared:
        L.ifl_rrdb_train_ws_bytes.argtypes = [c_int, c_int, c_int, c_int]
        L.ifl_rrdb_train_ws_bytes.restype = c_int64
        L.ifl_rrdb_train_ws_init.argtypes = [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]
        L.ifl_rrdb_train_ws_init.restype = c_int
        L.ifl_rrdb_train_fwd.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
        L.ifl_rrdb_train_fwd.restype = c_int
        L.ifl_rrdb_train_bwd.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
        L.ifl_rrdb_train_bwd.restype = c_int
        _declared = True
    TF._lib()                                   # (ifl_pack_table's signature)
    return L


class EncoderPlan:
    """Packed weights, gradient block and workspace of one (batch, size) of one RRDBNet."""

    def __init__(self, net, nb, block_idxs, B, H, W, device):
        L = _lib()
        self.nb, self.block_idxs, self.B, self.H, self.W, self.device = nb, list(block_idxs), B, H, W, device
        names = pack.rrdb_conv_names(nb)
        mods = dict(net.named_modules())
        self.used = len(names) - 3                                   # conv_first, the blocks, trunk_conv, upconv1
        self.params = []
        n = int(L.ifl_rrdb_param_floats(nb))
        f32 = dict(dtype=torch.float32, device=device)
        self.pk, self.pkt, self.dpar = torch.zeros(n, **f32), torch.zeros(n, **f32), torch.zeros(n, **f32)
        rows, self.views = [], []
        cin, cout, wo, bo = c_int(), c_int(), c_int64(), c_int64()
        for i, name in enumerate(names[:self.used]):
            hip.check(L.ifl_rrdb_conv_desc(nb, i, ctypes.byref(cin), ctypes.byref(cout), ctypes.byref(wo), ctypes.byref(bo)), "ifl_rrdb_conv_desc")
            m = mods[name]
            w, b = m.weight, m.bias
            cr, ci = w.shape[0], w.shape[1]
            assert tuple(w.shape[2:]) == (3, 3) and cr <= cout.value and ci <= cin.value, (name, tuple(w.shape))
            self.params += [w, b]
            common = dict(cout_real=cr, cout_pad=cout.value, taps=9, real0=ci, pad0=cin.value, real1=0, pad1=0)
            nw = cout.value * cin.value * 9
            rows += [dict(src=w.data_ptr(), dst=self.pk.data_ptr() + 4 * wo.value, nelem=nw, mode=TF.PACK_FWD, **common),
                     dict(src=b.data_ptr(), dst=self.pk.data_ptr() + 4 * bo.value, nelem=cout.value, mode=TF.PACK_VECTOR, cout_real=cr,
                          cout_pad=cout.value, taps=1),
                     dict(src=w.data_ptr(), dst=self.pkt.data_ptr() + 4 * wo.value, nelem=nw, mode=TF.PACK_TRANSPOSED, a0=0, acnt=cin.value, **common)]
            gw = self.dpar[wo.value:wo.value + nw].view(cout.value, cin.value, 3, 3)
            if (cr, ci) != (cout.value, cin.value):
                gw = gw[:cr, :ci]
            self.views += [gw, self.dpar[bo.value:bo.value + cr]]
        arr = np.zeros(len(rows), dtype=TF.PACK_DTYPE)
        for k, r in enumerate(rows):
            for key, v in r.items():
                arr[k][key] = v
        self.tab = (torch.from_numpy(arr.view(np.uint8)).to(device), len(rows), int(arr["nelem"].max()))
        nbytes = int(L.ifl_rrdb_train_ws_bytes(B, H, W, nb))
        if nbytes < 0:
            hip.check(nbytes, "ifl_rrdb_train_ws_bytes")
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            hip.check(L.ifl_rrdb_train_ws_init(hip._ptr(self.ws), nbytes, B, H, W, nb, hip._stream()), "ifl_rrdb_train_ws_init")
        self.idx = (c_int * max(len(self.block_idxs), 1))(*self.block_idxs)
        self.ptrs = tuple(p.data_ptr() for p in self.params)
        self.busy = False

    def matches(self):
        return all(p.data_ptr() == q for p, q in zip(self.params, self.ptrs))


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, want2, x, *params):
        L = _lib()
        plan.busy = True
        ctx.lease = TF._Lease(plan)
        B, H, W = plan.B, plan.H, plan.W
        Ct = 64 * (1 + len(plan.block_idxs))
        with torch.cuda.device(plan.device):
            t, n, mx = plan.tab
            hip.check(L.ifl_pack_table(hip._ptr(t), n, mx, hip._stream()), "ifl_pack_table")
            up1 = torch.empty(B, Ct, H // 4, W // 4, dtype=torch.float32, device=plan.device)
            up2 = torch.empty(B, Ct, H // 2, W // 2, dtype=torch.float32, device=plan.device) if want2 else None
            hip.check(L.ifl_rrdb_train_fwd(hip._ptr(plan.pk), plan.nb, plan.idx, len(plan.block_idxs), hip._ptr(plan.ws), hip._ptr(x),
                                           hip._ptr(up1), hip._ptr(up2), B, H, W, hip._stream()), "ifl_rrdb_train_fwd")
        ctx.want2 = want2
        if want2:
            return up1, up2
        dummy = torch.zeros((), device=plan.device)
        ctx.mark_non_differentiable(dummy)
        return up1, dummy

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g1, g2):
        plan = ctx.lease.plan
        assert plan is not None, "the encoder's workspace was released before its backward ran"
        L = _lib()
        g1 = g1.contiguous().float() if g1 is not None else None
        g2 = g2.contiguous().float() if (g2 is not None and ctx.want2) else None
        with torch.cuda.device(plan.device):
            plan.dpar.zero_()
            hip.check(L.ifl_rrdb_train_bwd(hip._ptr(plan.pkt), plan.nb, plan.idx, len(plan.block_idxs), hip._ptr(plan.ws), hip._ptr(g1),
                                           hip._ptr(g2), hip._ptr(plan.dpar), plan.B, plan.H, plan.W, hip._stream()), "ifl_rrdb_train_bwd")
            direct = TF._direct_grad and all(p.grad is not None and p.grad.dtype == torch.float32 for p in plan.params)
            if direct:
                # one multi-tensor add for the contiguous views; conv_first's weight gradient is a strided view of its padded
                # block (a single non-contiguous tensor would send the whole list down the one-launch-per-tensor path)
                dense = [(p.grad, v) for p, v in zip(plan.params, plan.views) if v.is_contiguous() and p.grad.is_contiguous()]
                torch._foreach_add_([a for a, _ in dense], [b for _, b in dense])
                for p, v in zip(plan.params, plan.views):
                    if not (v.is_contiguous() and p.grad.is_contiguous()):
                        p.grad.add_(v)
                if TF._on_ready is not None:
                    TF._on_ready(plan.params, True)                        # one encoder pass per step: these gradients are final
                grads = [None] * len(plan.params)
            else:
                grads = [v.clone() for v in plan.views]
        ctx.lease.release()
        return (None, None, None) + tuple(grads)


def usable(arch, lr, outputs) -> bool:
    from .flow_model import opt_get
    if outputs is None or not set(outputs) <= {"fea_up1", "fea_up2", "last_lr_fea"}:
        return False
    if not (isinstance(lr, torch.Tensor) and lr.is_cuda and lr.dtype == torch.float32 and lr.dim() == 4 and lr.shape[1] == 3):
        return False
    if lr.requires_grad or lr.shape[2] % 4 or lr.shape[3] % 4:
        return False
    # ('fea_up0' / 'fea_up-1' of the options are not among the maps the caller asked for: not produced)
    net = arch.RRDB
    return all(p.requires_grad for p in net.parameters()) and net.conv_first.weight.shape[0] == 64


def encode(arch, lr, outputs):
    """{'fea_up1', 'last_lr_fea', 'fea_up2'} (320-channel NCHW, attached to the encoder's parameters) or None when no
    workspace is free (a second gradient-enabled pass before the first one's backward)."""
    from .flow_model import opt_get
    net = arch.RRDB
    block_idxs = list(opt_get(arch.opt, ['network_G', 'flow', 'stackRRDB', 'blocks']) or [])
    nb = len(net.RRDB_trunk)
    B, _, H, W = lr.shape
    cache = arch.__dict__.setdefault("_rrdb_train_plans", {})
    key = (B, H, W, str(lr.device))
    plan = cache.get(key)
    if plan is not None and not plan.matches():
        plan = None
    if plan is None:
        if len(cache) >= 2:
            cache.pop(next(iter(cache)))
        plan = cache[key] = EncoderPlan(net, nb, block_idxs, B, H, W, lr.device)
    if plan.busy:
        return None
    want2 = "fea_up2" in outputs
    up1, up2 = _EncoderFn.apply(plan, want2, lr.contiguous(), *plan.params)
    return {'last_lr_fea': up1, 'fea_up1': up1, 'fea_up2': up2 if want2 else None, 'fea_up4': None, 'fea_up8': None,
            'fea_up16': None, 'fea_up32': None, 'out': None}
