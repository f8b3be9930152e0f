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
* `train_step` -- `train.py:72-75,110-131` with `consistency_lambda = 0` (the LR-consistency branch needs the bicubic
  resize of utils/resize_right.py, SURVEY.md section 8(f) row 3, not built).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch


class GradientAllReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], dist=None, bucket_bytes: int = 32 << 20):
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        seen, self.params = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:        # the model registers aliases of the same tensors
                seen.add(id(p))
                self.params.append(p)
        # buckets in reverse registration order: roughly the order in which backward produces the gradients
        self.buckets: List[dict] = []
        cur, size = [], 0
        for p in reversed(self.params):
            n = p.numel() * p.element_size()
            if cur and (size + n > bucket_bytes or cur[0].dtype != p.dtype or cur[0].device != p.device):
                self._close(cur)
                cur, size = [], 0
            cur.append(p)
            size += n
        if cur:
            self._close(cur)
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._bucket_of[id(p)] = bi
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self.zero_grad()

    def _close(self, plist):
        flat = torch.zeros(sum(p.numel() for p in plist), dtype=plist[0].dtype, device=plist[0].device)
        self.buckets.append({"params": list(plist), "flat": flat, "pending": 0, "work": None, "launched": False})

    def zero_grad(self):
        """Zero the buckets and (re)attach every parameter's .grad as a view into its bucket."""
        for b in self.buckets:
            b["flat"].zero_()
            b["pending"], b["work"], b["launched"] = len(b["params"]), None, False
            o = 0
            for p in b["params"]:
                p.grad = b["flat"][o:o + p.numel()].view_as(p)
                o += p.numel()

    def _launch(self, b):
        b["launched"] = True
        if self.dist is not None and self.world > 1:
            b["work"] = self.dist.all_reduce(b["flat"], async_op=True)

    def _on_grad(self, p):
        b = self.buckets[self._bucket_of[id(p)]]
        if p.grad.data_ptr() != b["flat"].data_ptr() + self._offset(b, p) * p.element_size():
            o = self._offset(b, p)                               # autograd replaced the tensor: copy it back into the bucket
            b["flat"][o:o + p.numel()].copy_(p.grad.reshape(-1))
            p.grad = b["flat"][o:o + p.numel()].view_as(p)
        b["pending"] -= 1
        if b["pending"] == 0 and not b["launched"]:
            self._launch(b)

    @staticmethod
    def _offset(b, p):
        o = 0
        for q in b["params"]:
            if q is p:
                return o
            o += q.numel()
        raise KeyError

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

    def remove(self):
        for h in self._hooks:
            h.remove()


def broadcast_parameters(model: torch.nn.Module, dist, src: int = 0):
    """Make every replica equal to rank `src` (parameters and buffers)."""
    if dist is None or dist.get_world_size() == 1:
        return
    seen = set()
    for t in list(model.parameters()) + list(model.buffers()):
        if id(t) in seen:
            continue
        seen.add(id(t))
        dist.broadcast(t.data, src)


def make_optimizer(model: torch.nn.Module, lr: float = 1e-4):
    """train.py:53-56: Adam over the parameters that require grad, MultiStepLR(50000, 75000, 90000; 0.5)."""
    seen, params = set(), []
    for p in model.parameters():
        if p.requires_grad and id(p) not in seen:
            seen.add(id(p))
            params.append(p)
    opt = torch.optim.Adam(params, lr=lr)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[50000, 75000, 90000], gamma=0.5)
    return opt, sched


def train_step(model, optimizer, gt, low_res, labels, ib_lambda: float = 1.0, reducer: Optional[GradientAllReducer] = None,
               scheduler=None, step: int = 0, add_gt_noise: bool = True):
    """One iteration of train.py:66-131 with consistency_lambda = 0.  Returns a dict of python floats (this synchronises,
    as the reference's progress bar does at train.py:133-134) and `skipped`."""
    lr_enc = model.arch.rrdbPreprocessing(gt)                                           # train.py:72
    z, nll, logdet, ll_ib = model(low_res, gt, labels, lr_enc=lr_enc, add_gt_noise=add_gt_noise)     # :73
    ib_loss = -torch.mean(ll_ib)                                                        # :74
    nll_loss = torch.mean(nll)                                                          # :75
    loss = nll_loss + ib_lambda * ib_loss                                               # :110
    out = {"skipped": False}
    bad = (not bool(torch.isfinite(loss))) or (step > 100 and float(loss) > 30)         # :113-126
    if bad:
        out["skipped"] = True
        if reducer is not None:
            reducer.zero_grad()
        loss.backward()                                     # the reference backpropagates and drops the update
        if reducer is not None:
            reducer.finish()
            reducer.zero_grad()
    else:
        if reducer is not None:
            reducer.zero_grad()
        else:
            optimizer.zero_grad()                                                       # :128
        loss.backward()                                                                 # :129
        if reducer is not None:
            reducer.finish()
        optimizer.step()                                                                # :130
        if scheduler is not None:
            scheduler.step()                                                            # :131
    out.update(loss=float(loss), nll=float(nll_loss), ib_loss=float(ib_loss))
    return out
