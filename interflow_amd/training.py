"""The training step of the reference (`interflow/train.py:66-131`, NLL + ib terms) and its data-parallel form.

One process per GPU; every rank holds a replica and takes its own slice of the global batch; the only exchange per
iteration is the gradient all-reduce (RCCL over xGMI on GPUs, gloo in the CPU tests): SURVEY.md section 8(e).

* `GradientAllReducer` -- gradients live as views into a few flat buckets; a bucket is all-reduced (asynchronously, so it
  overlaps the rest of the backward pass) as soon as autograd has accumulated its last gradient; `finish()` waits and
  divides by the world size.  Parameters that never receive a gradient (the reference registers dead modules:
  `degradation_net`, `flow_upsampler.f`, `arch.mean_shift/cov_shift`, RRDB's `HRconv/conv_last/upconv2`)
  simply stay zero in their bucket.
* `broadcast_parameters` -- after the data-dependent ActNorm initialisation (FlowActNorms.py:29-43) of rank 0's first
  batch, so that all replicas start identical.
* `train_step` -- `train.py:72-131`: NLL + ib terms and, when a pair of differently scaled LR images is given, the
  LR-consistency term (two more encodes, latent blend, gradient-enabled decode, antialiased bicubic x1/4 of
  `interflow_amd/resize.py`, L1).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import os

import torch


# the conditioning maps the shipped K=16, L=2 flow reads (FlowUpsamplerNet.py:66-73: level 1 -> 'fea_up2', level 2 -> 'fea_up1');
# the training step asks the frozen encoder for these only (its full-resolution tail feeds nothing the flow uses)
FLOW_FEATURES = ("fea_up1", "fea_up2")


def _drain_for_gloo(dist, t):
    """Rehearsal mode only (shard.rank_device: several ranks on ONE GPU, gloo because RCCL refuses that): gloo's all-reduce of a
    device tensor never completes when it is issued while that device still has work queued (tools/exp_gloo_cuda.py
    reproduces it without this library), so the device is drained first.  Nothing happens under RCCL or on CPU tensors."""
    if t.is_cuda and dist.get_backend() == "gloo":
        torch.cuda.synchronize(t.device)


class GradientAllReducer:
    """Gradients as views into flat buckets, each all-reduced asynchronously as soon as its last gradient exists.

    Two ways a gradient becomes "ready":
      * autograd accumulates it into .grad -> the post-accumulate hook (`_on_grad`): the layer-by-layer path and the few
        parameters outside the level drivers (the prior's, the stacked ActNorm / LU parameters when not in direct mode);
      * the fused level / encoder drivers add a whole level's ~290 gradients into the .grad views themselves with one
        multi-tensor add and hand autograd None, so no hook fires: they call `mark_ready(params)` right behind that add, on
        the stream it ran on (`train_fused.accumulate_into_grad(on_ready=...)`); the collective is issued from that stream,
        i.e. ordered after the add, and runs beside the rest of the backward pass (the next level's chain).
    A bucket whose parameters never all become ready is reduced by `finish()`.  After the FIRST finished step the parameters
    that received nothing (the reference's dead modules) are moved behind the live ones into buckets of
    their own (`rebucket`), so that from the second step on every bucket of live parameters is launched during the backward
    pass; a "dead" parameter that does receive a gradient later is still reduced, by `finish()`."""

    def __init__(self, params: Iterable[torch.nn.Parameter], dist=None, bucket_bytes: int = 32 << 20):
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.bucket_bytes = bucket_bytes
        seen, self.params = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:        # the model registers aliases of the same tensors
                seen.add(id(p))
                self.params.append(p)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self.defer = False            # True: nothing is launched before finish() (train_step's late_check)
        self.launched_early = 0       # buckets launched before finish() in the current step (tests, logs)
        self.launch_log = []          # (bucket index, "early" | "finish") of the current step, in launch order
        self._rebucketed = False
        # buckets in reverse registration order: roughly the order in which backward produces the gradients
        self._build(list(reversed(self.params)))

    def _build(self, ordered, groups=None, cut_after=()):
        """ordered: parameters in bucket order; groups: optional list of lists that must not share a bucket; cut_after: ids of
        parameters behind which a bucket that is at least half full is closed (the end of a set that became ready together)."""
        self.buckets: List[dict] = []
        for plist in (groups if groups is not None else [ordered]):
            cur, size = [], 0
            for p in plist:
                n = p.numel() * p.element_size()
                if cur and (size + n > self.bucket_bytes or cur[0].dtype != p.dtype or cur[0].device != p.device):
                    self._close(cur)
                    cur, size = [], 0
                cur.append(p)
                size += n
                if id(p) in cut_after and 2 * size >= self.bucket_bytes:
                    self._close(cur)
                    cur, size = [], 0
            if cur:
                self._close(cur)
        self._bucket_of, self._offset_of = {}, {}
        for bi, b in enumerate(self.buckets):
            o = 0
            for p in b["params"]:
                self._bucket_of[id(p)] = bi
                self._offset_of[id(p)] = o          # (looked up by every gradient hook: no linear search per call)
                o += p.numel()
        self._ready, self._order, self._group = set(), {}, 0
        self.zero_grad(reattach=True)

    def _close(self, plist):
        flat = torch.zeros(sum(p.numel() for p in plist), dtype=plist[0].dtype, device=plist[0].device)
        self.buckets.append({"params": list(plist), "flat": flat, "pending": 0, "work": None, "launched": False, "index": len(self.buckets)})

    def zero_grad(self, reattach: bool = False):
        """Zero the buckets; attach every parameter's .grad as a view into its bucket (once: autograd accumulates into an
        existing .grad in place, and `_on_grad` brings back a tensor that was replaced; `reattach` forces it again)."""
        if getattr(self, "_pending_rebucket", None) and self.rebucket():
            return                                # (once, at the start of the second step: the new buckets are zeroed and attached)
        self._ready = set()
        self._order, self._group = {}, 0
        self.launched_early = 0
        self.launch_log = []
        for b in self.buckets:
            b["flat"].zero_()
            b["pending"], b["work"], b["launched"] = len(b["params"]), None, False
            if reattach or not b.get("attached") or any(p.grad is None for p in b["params"]):
                o = 0
                for p in b["params"]:
                    p.grad = b["flat"][o:o + p.numel()].view_as(p)
                    o += p.numel()
                b["attached"] = True

    def _launch(self, b, early=False):
        b["launched"] = True
        self.launch_log.append((b["index"], "early" if early else "finish"))
        self.launched_early += int(early)
        if os.environ.get("IFL_DEBUG_REDUCER"):
            print(f"[reducer rank {self.dist.get_rank() if self.dist else 0}] launch bucket {b['index']} n={b['flat'].numel()} pending={b['pending']}", flush=True)
        if self.dist is not None and self.world > 1:
            _drain_for_gloo(self.dist, b["flat"])
            b["work"] = self.dist.all_reduce(b["flat"], async_op=True)

    def _count(self, p):
        if id(p) in self._ready:              # a second contribution to a gradient that was already counted (several passes)
            return None
        self._ready.add(id(p))
        self._order[id(p)] = (self._group, len(self._order))      # when it became ready: the first step's order makes the buckets
        b = self.buckets[self._bucket_of[id(p)]]
        b["pending"] -= 1
        return b

    def _on_grad(self, p):
        b = self.buckets[self._bucket_of[id(p)]]
        if p.grad.data_ptr() != b["flat"].data_ptr() + self._offset(b, p) * p.element_size():
            o = self._offset(b, p)                               # autograd replaced the tensor: copy it back into the bucket
            b["flat"][o:o + p.numel()].copy_(p.grad.reshape(-1))
            p.grad = b["flat"][o:o + p.numel()].view_as(p)
        if self._count(p) is not None and b["pending"] == 0 and not b["launched"] and not self.defer:
            self._launch(b, early=True)

    def mark_ready(self, params, final: bool = True):
        """The gradients of `params` have been added into their .grad views by work queued on the CURRENT stream (the fused
        drivers' multi-tensor add); final=False: more contributions follow in this backward pass (several flow passes over
        the same parameters), so nothing is counted yet.  Buckets that are complete are all-reduced from this stream."""
        if not final:
            return
        self._group += 1
        touched = {}
        for p in params:
            if id(p) not in self._bucket_of:
                continue
            b = self._count(p)
            if b is not None:
                touched[b["index"]] = b
        self._group += 1                      # (gradients that autograd's hooks report in between form a set of their own)
        for b in touched.values():
            if b["pending"] == 0 and not b["launched"] and not self.defer:
                self._launch(b, early=True)

    def _offset(self, b, p):
        return self._offset_of[id(p)]

    def finish(self):
        """Call after backward(): reduce the buckets whose parameters did not all receive a gradient, wait, average."""
        for b in self.buckets:
            if not b["launched"]:
                self._launch(b)
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
            if self.world > 1:
                b["flat"].div_(self.world)
        if not self._rebucketed and not self.defer:
            self._rebucketed = True
            self._pending_rebucket = ([p for p in self.params if id(p) not in self._ready], dict(self._order))

    def rebucket(self):
        """Between steps (after the optimizer has read the gradients): the live parameters in the order in which their gradients
        became ready in the first step -- the prior's, the last flow level's, the first level's, the stacked ActNorm / LU
        parameters, the encoder's -- with a bucket closed where a set that became ready together ends (if it is at least half
        full), and the parameters that received nothing in buckets of their own behind them.  Registration order mixes a
        level's large weights with the small parameters whose gradients arrive last, which held every bucket back until the
        end of the backward pass.  Returns the number of parameters without a gradient."""
        pend = getattr(self, "_pending_rebucket", None)
        self._pending_rebucket = None
        if not pend:
            return 0
        dead, order = pend
        if len(dead) == len(self.params):
            return 0
        dead_ids = {id(p) for p in dead}
        live = sorted((p for p in self.params if id(p) not in dead_ids), key=lambda p: order[id(p)])
        cut = {id(p) for p, q in zip(live, live[1:] + [None]) if q is None or order[id(q)][0] != order[id(p)][0]}
        self._build(None, groups=[live, [p for p in self.params if id(p) in dead_ids]], cut_after=cut)
        return max(len(dead), 1)

    def remove(self):
        for h in self._hooks:
            h.remove()


def broadcast_parameters(model: torch.nn.Module, dist, src: int = 0):
    """Make every replica equal to rank `src` (parameters and buffers)."""
    if dist is None or dist.get_world_size() == 1:
        return
    seen, groups = set(), {}
    for t in list(model.parameters()) + list(model.buffers()):
        if id(t) in seen:
            continue
        seen.add(id(t))
        groups.setdefault((t.dtype, t.device), []).append(t.data)
    # one flat buffer and ONE collective per dtype (≈1 500 tensors, 125 MB: per-tensor broadcasts cost a launch and a
    # rendezvous each -- minutes over gloo, milliseconds each over RCCL)
    for ts in groups.values():
        flat = torch.cat([t.reshape(-1) for t in ts])
        _drain_for_gloo(dist, flat)
        dist.broadcast(flat, src)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view(t.shape))
            off += n


def _uninitialised_actnorms(model):
    """ActNorms in training mode whose data-dependent initialisation is still due (FlowActNorms.py:29-35: `inited` unset
    and the bias all zero; a module with a non-zero bias is marked initialised, as the reference does).  One device
    synchronisation for all candidates, none once every module is marked."""
    from .flow_modules import ActNorm2d
    acts = model.__dict__.get("_ifl_actnorms")                    # the module list is fixed after construction
    if acts is None:
        acts = [m for m in model.modules() if isinstance(m, ActNorm2d)]
        model.__dict__["_ifl_actnorms"] = acts
    cands = [m for m in acts if not m.inited and m.training]
    if not cands:
        return []
    nonzero = torch.stack([(m.bias != 0).any() for m in cands]).cpu().tolist()
    todo = []
    for m, nz in zip(cands, nonzero):
        if nz:
            m.inited = True
        else:
            todo.append(m)
    return todo


def init_actnorm(model, gt, low_res, dist=None, lr_enc=None, src: int = 0, add_gt_noise: bool = True):
    """Data-dependent ActNorm initialisation (FlowActNorms.py:29-43, and inside every flow.Conv2d, flow.py:57-63) as a step
    of its own, BEFORE the first optimizer step: one gradient-enabled forward whose result is dropped (each rank initialises
    from its own batch, as the reference's first training forward does -- with the quantisation noise of flow_model.py:107-110
    unless add_gt_noise=False), then every replica takes rank `src`'s parameters, so that data-parallel replicas start
    identical (only gradients are averaged afterwards).  Returns the number of ActNorms that were initialised; 0 (and no
    forward) when there was nothing to do on any rank.

    The decision is made ONCE: after a call that found nothing to do on any rank, or after an initialisation pass, the model is
    marked (`model._ifl_actnorm_done`) and later calls return at once -- no collective, no host read.  Every rank sets the mark
    in the same call, so the replicas keep taking the same branch.  `inited` is never reset by this package; code that resets
    it by hand must also `del model._ifl_actnorm_done` on every rank."""
    if model.__dict__.get("_ifl_actnorm_done"):
        return 0
    todo = _uninitialised_actnorms(model)
    n = len(todo)
    multi = dist is not None and dist.get_world_size() > 1
    if multi:                                   # every rank must take the same branch: the broadcast below is a collective
        flag = torch.tensor([float(n)], device=gt.device)
        _drain_for_gloo(dist, flag)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        n = int(flag.item())
    if n == 0:
        model.__dict__["_ifl_actnorm_done"] = True
        return 0
    with torch.enable_grad():
        if lr_enc is None:
            lr_enc = model.arch.rrdbPreprocessing(gt)
        model(low_res, gt, lr_enc=lr_enc, add_gt_noise=add_gt_noise, calc_loss=False)
    for m in todo:
        m.inited = True
    if multi:
        broadcast_parameters(model, dist, src)
    model.__dict__["_ifl_actnorm_done"] = True
    return n


class TableAdam(torch.optim.Optimizer):
    """torch.optim.Adam(betas, eps, no weight decay, no amsgrad) with the update of every parameter tensor in ONE launch
    (`ifl_adam_table`) over a device table of 4 096-element chunks, instead of the fused multi-tensor form's 36 tensors per
    launch (22 launches of ~43 us for the flow's 770 tensors, to move 230 MB).  Same arithmetic as torch's fused kernel.
    State per parameter as torch's (`step`, `exp_avg`, `exp_avg_sq`: a state dict loads into torch.optim.Adam and back);
    `step` is one device cell shared by all parameters of a group: a parameter whose .grad is None is left untouched, as
    torch does (the encoder's three convolutions behind 'fea_up4' / 'out' never receive one), but its bias correction
    follows the group's count, which differs from torch only for a parameter that receives gradients in some steps and none
    in others.  `found_inf` (a 0-dim device tensor set as an attribute, the mechanism of
    torch's GradScaler and of train_step(host_sync=False)) skips the update on the device, step counter included.
    Parameters are written through raw pointers; their version counters are bumped by hand so that caches keyed on them
    (packed weights) see the update."""
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        # the group keys of torch.optim.Adam, so that state dicts travel between the two (values other than these defaults
        # are not implemented here)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                                      capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False))
        self._tables = {}

    def _group_table(self, gi, group):
        import ctypes
        import numpy as np
        from . import hip
        key = tuple((p.data_ptr(), p.grad.data_ptr() if p.grad is not None else 0) for p in group["params"])
        ent = self._tables.get(gi)
        if ent is not None and ent["key"] == key:
            return ent
        params = [p for p in group["params"] if p.grad is not None]
        if not params:
            return None
        dev = params[0].device
        rows = []
        steps = []
        step_cache = {}
        for p in params:
            if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous() or p.device != dev:
                raise RuntimeError("TableAdam: float32 contiguous parameters and gradients on one device")
            st = self.state[p]
            if "exp_avg" not in st:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            else:
                t = st["step"]                                  # one host read per distinct cell (all parameters share one)
                k = t.data_ptr() if torch.is_tensor(t) else None
                if k is None or k not in step_cache:
                    step_cache[k] = float(t)
                steps.append(step_cache[k])
            n, CH = p.numel(), hip.ADAM_CHUNK
            for o in range(0, n, CH):
                rows.append((p.data_ptr() + 4 * o, p.grad.data_ptr() + 4 * o, st["exp_avg"].data_ptr() + 4 * o,
                             st["exp_avg_sq"].data_ptr() + 4 * o, min(CH, n - o)))
        if steps and min(steps) != max(steps):
            raise RuntimeError("TableAdam: parameters with different step counts (a state dict of torch.optim.Adam in which some "
                               "parameters skipped steps): one shared step cell cannot represent it")
        if ent is None:
            cells = torch.full((2,), steps[0] if steps else 0.0, dtype=torch.float32, device=dev)
            ent = {"cells": cells, "cur": 0}
        for p in params:
            self.state[p]["step"] = ent["cells"][ent["cur"]:ent["cur"] + 1].reshape(())     # a view of the live cell
        arr = np.array(rows, dtype=np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8")]))
        ent.update(key=key, table=torch.from_numpy(arr.view(np.uint8).copy()).to(dev), nrows=len(rows), params=params)
        self._tables[gi] = ent
        return ent

    @torch.no_grad()
    def step(self, closure=None):
        from . import hip
        assert closure is None
        L = hip.lib()
        found_inf = getattr(self, "found_inf", None)
        if getattr(self, "grad_scale", None) is not None:
            raise RuntimeError("TableAdam: gradient scaling is not supported (fp32 training)")
        for gi, group in enumerate(self.param_groups):
            if not group["params"]:
                continue
            if group.get("weight_decay", 0) or group.get("amsgrad") or group.get("maximize"):
                raise NotImplementedError("TableAdam: weight_decay / amsgrad / maximize")
            ent = self._group_table(gi, group)
            if ent is None:
                continue
            cur = ent["cur"]
            cells = ent["cells"]
            b1, b2 = group["betas"]
            with torch.cuda.device(cells.device):
                hip.check(L.ifl_adam_table(hip._ptr(ent["table"]), ent["nrows"], float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                           cells.data_ptr() + 4 * cur, cells.data_ptr() + 4 * (1 - cur),
                                           hip._ptr(found_inf) if found_inf is not None else None, hip._stream()), "ifl_adam_table")
            ent["cur"] = 1 - cur
            live = cells[1 - cur:2 - cur].reshape(())
            for p in ent["params"]:
                self.state[p]["step"] = live
            torch.autograd.graph.increment_version(ent["params"])

    def state_dict(self):
        """As torch.optim.Adam's: `step` per parameter as a CPU scalar tensor (one device read for the shared cell)."""
        sd = super().state_dict()
        cache = {}
        for st in sd["state"].values():
            t = st.get("step")
            if torch.is_tensor(t):
                k = t.data_ptr()
                if k not in cache:
                    cache[k] = float(t)
                st["step"] = torch.tensor(cache[k], dtype=torch.float32)
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}                                     # moments are new tensors: rebuild the tables (and the step cells)


def make_optimizer(model: torch.nn.Module, lr: float = 1e-4, table: bool = None):
    """train.py:53-56: Adam over the parameters that require grad, MultiStepLR(50000, 75000, 90000; 0.5).
    table (default: environment IFL_TABLE_ADAM, on): `TableAdam`, else torch.optim.Adam(fused=True)."""
    import os
    seen, params = set(), []
    for p in model.parameters():
        if p.requires_grad and id(p) not in seen:
            seen.add(id(p))
            params.append(p)
    on_gpu = bool(params) and all(p.is_cuda and p.dtype == torch.float32 for p in params)
    if table is None:
        table = os.environ.get("IFL_TABLE_ADAM", "1") != "0"
    if table and on_gpu:
        opt = TableAdam(params, lr=lr)
    else:
        # fused: one multi-tensor launch sequence for all parameter tensors instead of the per-op foreach lists
        opt = torch.optim.Adam(params, lr=lr, fused=bool(params) and all(p.is_cuda for p in params))
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[50000, 75000, 90000], gamma=0.5)
    return opt, sched


def consistency_loss(model, gt, lr_enc, low_res, img_lr1, scale1, img_lr2, scale2, interpolate_weight=None,
                     add_gt_noise: bool = True):
    """The LR-consistency term, train.py:78-106: blend the latents of two differently scaled LR images with per-sample
    weights, decode, and compare (L1) the x1/4 antialiased-bicubic reductions of the decoded image and of the blended
    LR images.  interpolate_weight: per-sample weights in [0, 1] (drawn as train.py:81,88 does when None)."""
    from .resize import resize_last2
    B = gt.shape[0]
    if interpolate_weight is None:
        target = torch.distributions.Uniform(float(scale1), float(scale2)).sample([B]).to(gt.device)      # :81
        interpolate_weight = (target - scale1) / (scale2 - scale1)                                        # :88
    z1, _ = model(img_lr1, gt, torch.tensor([scale1] * B), lr_enc=lr_enc, calc_loss=False, add_gt_noise=add_gt_noise)   # :85
    z2, _ = model(img_lr2, gt, torch.tensor([scale2] * B), lr_enc=lr_enc, calc_loss=False, add_gt_noise=add_gt_noise)   # :86
    w1 = (1 - interpolate_weight).reshape(-1, 1, 1, 1)                                  # :89
    w2 = interpolate_weight.reshape(-1, 1, 1, 1)                                        # :90
    interpolated_z = w1 * z1 + w2 * z2                                                  # :92
    interpolated_img, _ = model(interpolated_z, gt, reverse=True, test_input_size=low_res.shape[2], lr_enc=lr_enc,
                                add_gt_noise=add_gt_noise)                              # :94
    synth_lr = torch.zeros_like(low_res) + w1 * img_lr1 + w2 * img_lr2                  # :99-100
    synth_lr = synth_lr + ((torch.rand(synth_lr.shape, device=synth_lr.device) - 0.5) / 256)     # :101
    interp_down = resize_last2(interpolated_img, 1. / 4)                                # :102
    synth_down = resize_last2(synth_lr, 1. / 4)                                         # :103
    return torch.nn.functional.l1_loss(synth_down, interp_down).mean()                  # :104-106


def _finish_late(model, optimizer, reducer, scheduler, step, loss, nll_loss, ib_loss, cons, out, device, host_sync=True):
    """train_step's tail with the skip guards after the backward pass (see `late_check`)."""
    from .train_fused import accumulate_into_grad
    dist = reducer.dist if reducer is not None else None
    if reducer is not None:
        reducer.zero_grad()
        reducer.defer = True                    # the hooks only collect; finish() launches the all-reduces after the decision
    else:
        optimizer.zero_grad(set_to_none=False)  # in place: TableAdam's table is keyed on the gradients' addresses
    try:
        with accumulate_into_grad(reducer.mark_ready if reducer is not None else None):
            loss.backward()                                                             # train.py:129
    finally:
        if reducer is not None:
            reducer.defer = False
    if not host_sync:
        # No host read at all: the guards are evaluated on the device and handed to the fused optimizer as its `found_inf`
        # flag (the mechanism torch's GradScaler uses: the update kernel leaves parameters, moments and step counters
        # untouched when it is set), so the host can go on issuing the next step while this one still runs.
        if not getattr(optimizer, "_step_supports_amp_scaling", False):
            raise RuntimeError("train_step(host_sync=False) needs an optimizer that takes a device-side skip flag (Adam(fused=True))")
        l = loss.detach()
        flag = (~torch.isfinite(l)) | ((l > 30) if step > 100 else torch.zeros_like(l, dtype=torch.bool))   # train.py:113-126
        flag = flag.to(torch.float32).reshape(())
        if dist is not None and dist.get_world_size() > 1:
            _drain_for_gloo(dist, flag)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)       # collective decision, on the device
        if reducer is not None:
            reducer.finish()
        optimizer.found_inf = flag
        try:
            optimizer.step()                                                            # :130 (skipped on the device when flag = 1)
        finally:
            del optimizer.found_inf
        if scheduler is not None:
            scheduler.step()                                  # every iteration: the host does not know about a skip
        out.update(skipped=flag, loss=l, nll=nll_loss.detach(), ib_loss=ib_loss.detach(), consistency_loss=cons.detach())
        return out
    terms = torch.stack([loss.detach(), nll_loss.detach(), ib_loss.detach(), cons.detach()]).tolist()    # the step's one host read
    bad = (terms[0] != terms[0] or abs(terms[0]) == float("inf")) or (step > 100 and terms[0] > 30)     # :113-126
    if dist is not None and dist.get_world_size() > 1:
        flag = torch.tensor([1.0 if bad else 0.0], device=device)
        _drain_for_gloo(dist, flag)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        bad = bool(flag.item() > 0)
    if bad:
        out["skipped"] = True
        if reducer is not None:
            reducer.zero_grad()
        else:
            optimizer.zero_grad()
    else:
        if reducer is not None:
            reducer.finish()
        optimizer.step()                                                                # :130
        if scheduler is not None:
            scheduler.step()                                                            # :131
    out.update(loss=terms[0], nll=terms[1], ib_loss=terms[2], consistency_loss=terms[3])
    return out


class EncoderPrefetcher:
    """The frozen conditioning encoder one batch ahead, on a stream of its own.

    With the encoder frozen its features depend on the data only, so the encoder pass of batch n+1 (345 small dependent
    convolutions, latency-bound at batch 8) can run while batch n goes through the flow and its backward pass -- also
    chains of small kernels -- instead of in front of them: `submit(gt_next)` before `train_step(..., lr_enc=get())`.
    Every step still runs its own encoder pass on the device; only the order of issue changes.  Refuses to work when
    gradients have to reach the encoder (then the features belong to the step's autograd graph)."""

    def __init__(self, model):
        self.model = model
        self.stream = torch.cuda.Stream()
        self.pending = None

    def submit(self, gt):
        assert self.pending is None, "one batch ahead: get() the previous features first"
        arch = self.model.arch
        if any(p.requires_grad for p in arch.RRDB.parameters()):
            raise RuntimeError("EncoderPrefetcher needs a frozen encoder (arch.set_rrdb_training(False))")
        self.stream.wait_stream(torch.cuda.current_stream(gt.device))      # gt, and every earlier use of the encoder's buffers
        with torch.cuda.stream(self.stream), torch.no_grad():
            feats = arch.rrdbPreprocessing(gt, outputs=FLOW_FEATURES)
            done = torch.cuda.Event()
            done.record(self.stream)
        self.pending = (feats, done, gt)

    def get(self):
        feats, done, _gt = self.pending
        self.pending = None
        cur = torch.cuda.current_stream()
        cur.wait_event(done)
        for t in feats.values():
            if isinstance(t, torch.Tensor):
                t.record_stream(cur)                                       # allocated on the side stream, consumed here
        return feats


def train_step(model, optimizer, gt, low_res, labels, ib_lambda: float = 1.0, reducer: Optional[GradientAllReducer] = None,
               scheduler=None, step: int = 0, add_gt_noise: bool = True, consistency_lambda: float = 0.0, interp=None,
               interpolate_weight=None, lr_enc=None, late_check: bool = False, host_sync: bool = True):
    """One iteration of train.py:66-131.  interp = (img_lr1, scale1, img_lr2, scale2) enables the LR-consistency term
    (weight consistency_lambda; train.py's default is 10).  lr_enc: the conditioning features of `gt` when the caller has
    them already (`EncoderPrefetcher`).  late_check: evaluate the reference's skip guards AFTER the backward pass instead of
    before it (train.py:113-126 reads the loss on the host first): the step then has no device synchronisation between the
    forward and the backward pass, so the host issues the backward while the forward still runs; a skipped step has then
    computed gradients that nobody applies, and in data-parallel runs the all-reduce is launched only after the collective
    decision (no overlap with the backward, and never fed with a skipped step's gradients).  host_sync=False (with
    late_check): no host read at all -- the guards are evaluated on the device and passed to the fused optimizer as its
    skip flag, the returned dict holds 0-dim device tensors (read them when a log line is due), and the scheduler is stepped
    every iteration (the reference does not step it on a skipped one).  Returns a dict of python floats (this synchronises, as the
    reference's progress bar does at train.py:133-134) and `skipped`.

    Data parallel (reducer given): uninitialised ActNorms are initialised and broadcast first (`init_actnorm`), and the
    reference's skip guards (train.py:113-126) are decided collectively -- if ANY rank's loss is non-finite or too large,
    every rank drops the update and no gradient is exchanged, so a NaN on one rank can neither reach the other replicas'
    parameters nor make the replicas diverge."""
    dist = reducer.dist if reducer is not None else None
    if model.training:
        init_actnorm(model, gt, low_res, dist, add_gt_noise=add_gt_noise)
    if lr_enc is None:                                                                  # (given: EncoderPrefetcher's features of gt)
        lr_enc = model.arch.rrdbPreprocessing(gt, outputs=FLOW_FEATURES)                # train.py:72 (the maps the L=2 flow reads)
    from .train_fused import shared_ft_grad
    import contextlib
    several = consistency_lambda > 0 and interp is not None                             # four passes over the same features
    with (shared_ft_grad() if several else contextlib.nullcontext()):
        return _train_step_body(model, optimizer, gt, low_res, labels, ib_lambda, reducer, scheduler, step, add_gt_noise,
                                consistency_lambda, interp, interpolate_weight, lr_enc, late_check, host_sync, dist)


def _train_step_body(model, optimizer, gt, low_res, labels, ib_lambda, reducer, scheduler, step, add_gt_noise, consistency_lambda,
                     interp, interpolate_weight, lr_enc, late_check, host_sync, dist):
    from .train_fused import reset_contributions
    reset_contributions()
    if late_check:
        z, logdet = model(low_res, gt, lr_enc=lr_enc, add_gt_noise=add_gt_noise, calc_loss=False)    # :73
        nll, ib_loss = model.loss_terms(z, logdet, int(gt.size(2) * gt.size(3)), labels)             # :74
    else:
        z, nll, logdet, ll_ib = model(low_res, gt, labels, lr_enc=lr_enc, add_gt_noise=add_gt_noise)     # :73
        ib_loss = -torch.mean(ll_ib)                                                    # :74
    nll_loss = torch.mean(nll)                                                          # :75
    if consistency_lambda > 0 and interp is not None:                                   # :78-106
        cons = consistency_loss(model, gt, lr_enc, low_res, *interp, interpolate_weight=interpolate_weight,
                                add_gt_noise=add_gt_noise)
    else:
        cons = torch.zeros((), device=gt.device)                                        # :107-108
    loss = nll_loss + consistency_lambda * cons + ib_lambda * ib_loss                   # :110
    out = {"skipped": False}
    if late_check:
        return _finish_late(model, optimizer, reducer, scheduler, step, loss, nll_loss, ib_loss, cons, out, gt.device, host_sync)
    bad = (not bool(torch.isfinite(loss))) or (step > 100 and float(loss.detach()) > 30)         # :113-126
    if dist is not None and dist.get_world_size() > 1:
        flag = torch.tensor([1.0 if bad else 0.0], device=gt.device)
        _drain_for_gloo(dist, flag)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        bad = bool(flag.item() > 0)
    if bad:
        # The reference calls loss.backward() and continues: those gradients are cleared by the next update's
        # optimizer.zero_grad() before anything reads them, so the backward only frees the graph.  Here the graph is
        # simply dropped, and -- the decision being collective -- no rank feeds an all-reduce this iteration.
        out["skipped"] = True
        if reducer is not None:
            reducer.zero_grad()
        else:
            optimizer.zero_grad()
    else:
        if reducer is not None:
            reducer.zero_grad()
        else:
            optimizer.zero_grad(set_to_none=False)                                      # :128 (in place: see TableAdam)
        from .train_fused import accumulate_into_grad
        with accumulate_into_grad(reducer.mark_ready if reducer is not None else None):
            loss.backward()                                                             # :129
        if reducer is not None:
            reducer.finish()
        optimizer.step()                                                                # :130
        if scheduler is not None:
            scheduler.step()                                                            # :131
    out.update(loss=float(loss.detach()), nll=float(nll_loss.detach()), ib_loss=float(ib_loss.detach()),
               consistency_loss=float(cons.detach()))
    return out
