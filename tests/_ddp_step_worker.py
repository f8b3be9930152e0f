"""Worker of tests/test_ddp_gloo.py: WORLD_SIZE gloo ranks on the CPU drive interflow_amd.training.train_step end to end
(ActNorm initialisation + broadcast, collective skip decision, bucketed all-reduce, optimizer step) on a small CPU stand-in
with the model API train_step uses -- the flow itself needs a GPU, the control flow does not.  Every rank has its OWN data."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from interflow_amd import flow_modules, shard, train_flow, training  # noqa: E402


class Stub(torch.nn.Module):
    """`model.arch.rrdbPreprocessing(gt)` and `model(low_res, gt, labels, lr_enc=, add_gt_noise=, calc_loss=)` ->
    (z, nll, logdet, ll_ib), with a data-dependent ActNorm (the holder class and the init code of the real path)."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.conv = torch.nn.Conv2d(3, 4, 1)
        self.an = flow_modules.ActNorm2d(4)
        self.arch = types.SimpleNamespace(rrdbPreprocessing=lambda gt, outputs=None: {"fea": gt})

    def forward(self, low_res, gt, label=None, lr_enc=None, add_gt_noise=True, calc_loss=True):
        h = self.conv(low_res).permute(0, 2, 3, 1)
        y, _ = train_flow.actnorm(self.an, h)
        z = y.permute(0, 3, 1, 2)
        if not calc_loss:
            return z, z.new_zeros(z.shape[0])
        nll = z.square().mean(dim=(1, 2, 3))
        return z, nll, z.new_zeros(z.shape[0]), -0.1 * z.abs().mean(dim=(1, 2, 3))

    def loss_terms(self, z, logdet, pixels, label):                    # (nll per sample, ib loss): train_step's late_check path
        return z.square().mean(dim=(1, 2, 3)), (0.1 * z.abs().mean(dim=(1, 2, 3))).mean()


def main():
    out_path, mode = sys.argv[1], sys.argv[2]
    late = mode.endswith("_late")                                      # guards after the backward pass, all-reduce after the decision
    mode = mode.replace("_late", "")
    rank, _, world = shard.dist_env()
    dist = shard.init_dist("gloo")
    torch.set_num_threads(1)
    model = Stub().train()
    g = torch.Generator().manual_seed(100 + rank)                      # per-rank data
    opt, _ = training.make_optimizer(model, lr=1e-2)
    red = training.GradientAllReducer(model.parameters(), dist, bucket_bytes=64)
    log = []
    snap = None
    for it in range(4):
        x = torch.randn(4, 3, 6, 6, generator=g) * (1.0 + rank) + 0.5 * rank
        step = it
        if mode == "nan" and it == 2 and rank == world - 1:
            x[0, 0, 0, 0] = float("nan")                               # one rank's loss is not finite
        if mode == "large" and it == 2:
            step = 200                                                 # past train.py:121's `step > 100`
            if rank == 0:
                x = x * 1e3                                            # one rank's loss is > 30
        if it == 2:
            snap = [p.detach().clone() for p in model.parameters()]
        out = training.train_step(model, opt, x, x, [2] * 4, reducer=red, step=step, add_gt_noise=False, late_check=late)
        log.append(out["skipped"])
        if it == 2 and mode in ("nan", "large"):
            assert out["skipped"], (rank, out)                         # every rank skips, also the ones whose loss was fine
            for p, q in zip(model.parameters(), snap):
                assert torch.equal(p.detach(), q)                      # the update was dropped
        for p in model.parameters():
            assert torch.isfinite(p).all(), (rank, it)
    assert log == [False, False, mode in ("nan", "large"), False], log
    assert model.an.inited
    # replicas identical: compare with rank 0's parameters
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    ref = flat.clone()
    if dist is not None:
        dist.broadcast(ref, 0)
    assert torch.equal(flat, ref), (rank, float((flat - ref).abs().max()))
    shard.barrier(dist)
    if rank == 0:
        np.savez(out_path, flat=flat.numpy(), bias=model.an.bias.detach().numpy())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
