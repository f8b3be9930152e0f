#!/usr/bin/env python
"""BASELINE.json configs[3]: flow training step (NLL + ib, forward + backward + Adam), batch 8 per GPU (global 64 on 8
GPUs), 160x160 patches, fp32; one process per GPU, gradient all-reduce over RCCL (interflow_amd.training).

  python tools/bench_train.py [--steps 5] [--warmup 2] [--batch 8] [--frozen-rrdb]
  torchrun --nproc-per-node N tools/bench_train.py ...

Prints one JSON line: training patches/s (whole job), ms per step, and the split of one step measured with
torch.cuda.synchronize() between phases on rank 0 (encoder, flow forward, backward incl. all-reduce, optimizer)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from interflow_amd import shard, synth, training  # noqa: E402
from interflow_amd.flow_model import InterpFlow  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--patch", type=int, default=160)
    ap.add_argument("--consistency", action="store_true", help="add the LR-consistency term (train.py's default lambda 10)")
    ap.add_argument("--prefetch-encoder", action="store_true",
                    help="with --frozen-rrdb: the encoder pass of the next batch on its own stream, overlapping this step (training.EncoderPrefetcher)")
    ap.add_argument("--no-host-sync", action="store_true",
                    help="train_step(late_check=True, host_sync=False): guards on the device, no host read inside a step")
    ap.add_argument("--late-check", action="store_true", help="train_step(late_check=True): no host read of the loss before the backward pass")
    ap.add_argument("--frozen-rrdb", action="store_true", help="do not train the conditioning encoder (no gradient wrt ft)")
    args = ap.parse_args()
    if os.environ.get("IFL_WATCHDOG_S"):           # a stuck rank prints every thread's stack and exits (collective mismatches)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["IFL_WATCHDOG_S"]), exit=True)
    rank, local_rank, world = shard.dist_env()
    dev, backend = shard.rank_device()             # cuda:LOCAL_RANK, RCCL
    torch.cuda.set_device(dev)
    dist = shard.init_dist(backend, dev)
    model = InterpFlow([2, 4])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_interpflow_state([2, 4], seed=0).items()})
    model.train().to(dev)
    if args.frozen_rrdb:
        model.arch.set_rrdb_training(False)
    training.broadcast_parameters(model, dist)
    opt, sched = training.make_optimizer(model, lr=1e-4)
    red = training.GradientAllReducer(model.parameters(), dist)
    B, P = args.batch, args.patch
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    gt = torch.rand(B, 3, P, P, device=dev, generator=g)
    lr = torch.rand(B, 3, P, P, device=dev, generator=g)
    labels = torch.tensor([2, 4] * (B // 2) + [2] * (B % 2), device=dev)
    kw = {}
    if args.consistency:
        kw = dict(consistency_lambda=10.0, interp=(torch.rand(B, 3, P, P, device=dev, generator=g), 2,
                                                   torch.rand(B, 3, P, P, device=dev, generator=g), 4))
    if args.late_check:
        kw["late_check"] = True
    if args.no_host_sync:
        kw.update(late_check=True, host_sync=False)
    if args.prefetch_encoder:
        assert args.frozen_rrdb, "--prefetch-encoder needs --frozen-rrdb"
        training.init_actnorm(model, gt, lr, dist)
        pf = training.EncoderPrefetcher(model)
        pf.submit(gt)

        def step_fn():
            feats = pf.get()
            pf.submit(gt)                                   # the next batch's encoder pass (here: the same synthetic batch again)
            return training.train_step(model, opt, gt, lr, labels, reducer=red, scheduler=sched, lr_enc=feats, **kw)
    else:
        step_fn = lambda: training.train_step(model, opt, gt, lr, labels, reducer=red, scheduler=sched, **kw)
    for _ in range(args.warmup):
        out = step_fn()
    shard.barrier(dist, dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step_fn()
    shard.barrier(dist, dev)
    dt = shard.max_over_ranks(time.perf_counter() - t0, dist, dev)

    def sync():
        torch.cuda.synchronize(dev)
        return time.perf_counter()
    if args.prefetch_encoder:
        pf.get()
    t = [sync()]
    lr_enc = model.arch.rrdbPreprocessing(gt); t.append(sync())
    z, nll, ld, ll_ib = model(lr, gt, labels, lr_enc=lr_enc); t.append(sync())
    loss = nll.mean() - ll_ib.mean()
    red.zero_grad(); loss.backward(); red.finish(); t.append(sync())
    opt.step(); t.append(sync())
    out = {k: (float(v) if isinstance(v, torch.Tensor) and k != "skipped" else bool(v) if isinstance(v, torch.Tensor) else v)
           for k, v in out.items()}
    ranks = shard.rank_devices(dist, dev)
    if rank == 0:
        print(json.dumps({"workload": f"training step (NLL + ib; forward, backward, Adam), batch {B} of {P}x{P} per GPU, fp32, "
                                      f"{'encoder frozen' if args.frozen_rrdb else 'encoder trained'}{', with LR-consistency term' if args.consistency else ''}"
                                      f"{', encoder pass of the next batch overlapped on a second stream' if args.prefetch_encoder else ''}"
                                      f"{', loss guards after the backward pass' if args.late_check else ''}"
                                      f"{', loss guards on the device (no host read inside a step)' if args.no_host_sync else ''}",
                          "n_gpus": world, "world_size": dist.get_world_size() if dist is not None else 1,
                          "ranks": ranks, "global_batch": B * world, "steps": args.steps, "ms_per_step": dt / args.steps * 1e3,
                          "patches_per_s": B * world * args.steps / dt, "last": out,
                          "split_ms": {"encoder_fwd": (t[1] - t[0]) * 1e3, "flow_fwd_loss": (t[2] - t[1]) * 1e3,
                                       "backward_allreduce": (t[3] - t[2]) * 1e3, "adam": (t[4] - t[3]) * 1e3}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
