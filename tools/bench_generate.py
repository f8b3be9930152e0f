#!/usr/bin/env python
"""BASELINE.json configs[2]: x2->x4 latent-interpolation LR generation over synthetic 2K images, tiled (160x160,
border tiles 128), 19 weights per tile (generate_images.py:141-143), units sharded over ranks (no collective).

  python tools/bench_generate.py [--images 1] [--size 2048] [--tile 160] [--batch 32] [--naive-tiles 32]
  torchrun --nproc-per-node N tools/bench_generate.py ...

Prints one JSON line: tiles/s and images/s of the harness (interflow_amd/generate.py) and, on a small sample of tiles,
of the reference's call pattern (one weighted_interpolate_image call per weight, generate_images.py:79-81) through the
same HIP library with conditioning reuse off, i.e. three flow passes and three encoder passes per weight.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from interflow_amd import generate, shard, synth  # noqa: E402
from interflow_amd.flow_model import InterpFlow  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=1)
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--tile", type=int, default=160)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--naive-tiles", type=int, default=32)
    args = ap.parse_args()
    rank, local_rank, world = shard.dist_env()
    dev, backend = shard.rank_device()             # cuda:LOCAL_RANK, RCCL
    torch.cuda.set_device(dev)
    dist = shard.init_dist(backend, dev)
    model = InterpFlow([2, 4])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_interpflow_state([2, 4], seed=0).items()})
    model.eval().to(dev)
    g = torch.Generator(device=dev).manual_seed(7)
    N, S = args.images, args.size
    gt, la, lb = (torch.rand(N, 3, S, S, device=dev, generator=g) for _ in range(3))
    weights = generate.default_weights(0)
    # warm-up: the same job with two weights, so that every (batch, tile shape) the timed run uses has been seen
    # (MIOpen picks its conv algorithms for the PyTorch-ROCm RRDB encoder on first use: seconds per new shape)
    generate.generate(model, gt, la, lb, weights[:2], tile=args.tile, batch=args.batch, rank=rank, world=world)
    shard.barrier(dist, dev)
    t0 = time.perf_counter()
    units, res = generate.generate(model, gt, la, lb, weights, tile=args.tile, batch=args.batch, rank=rank, world=world)
    shard.barrier(dist, dev)
    dt = shard.max_over_ranks(time.perf_counter() - t0, dist, dev)
    n_tiles = len(shard.tile_origins(S, S, args.tile)) * N
    out = {"workload": f"{N} synthetic {S}x{S} GT/LR2/LR4 triples, {args.tile}x{args.tile} tiles (border tiles keep the "
                       f"remainder), {len(weights)} weights per tile, uint8 NHWC outputs on the device",
           "n_gpus": world, "tiles": n_tiles, "seconds": dt, "tiles_per_s": n_tiles / dt,
           "outputs_per_s": n_tiles * len(weights) / dt, "images_per_s": N / dt}
    # the same job with the results handed to a host-side writer (generate.HostHandoff: asynchronous copies into pinned
    # buffers, a writer thread; the sink here checksums every array, standing in for the PNG encoder)
    got = {"units": 0, "bytes": 0, "sum": 0}

    def sink(u, a):
        got["units"] += 1
        got["bytes"] += a.nbytes
        got["sum"] += int(a[:, ::16, ::16].sum())
    generate.generate(model, gt, la, lb, weights[:2], tile=args.tile, batch=args.batch, rank=rank, world=world, sink=sink)   # pins the ring
    generate.generate(model, gt, la, lb, weights, tile=args.tile, batch=min(args.batch, 8), rank=rank, world=world, sink=sink)
    got.update(units=0, bytes=0, sum=0)
    shard.barrier(dist, dev)
    t0 = time.perf_counter()
    generate.generate(model, gt, la, lb, weights, tile=args.tile, batch=args.batch, rank=rank, world=world, sink=sink)
    shard.barrier(dist, dev)
    dth = shard.max_over_ranks(time.perf_counter() - t0, dist, dev)
    out["host_handoff"] = {"seconds": dth, "tiles_per_s": n_tiles / dth, "images_per_s": N / dth,
                           "rank0_units": got["units"], "rank0_host_bytes": got["bytes"],
                           "note": "results delivered to a host-side sink through pinned buffers on a copy stream"}
    if rank == 0 and args.naive_tiles:
        T, nb = args.tile, args.naive_tiles
        cg = torch.stack([gt[0, :, 0:T, i * T:(i + 1) * T] for i in range(nb)]) if nb * T <= S else gt[:1, :, :T, :T].repeat(nb, 1, 1, 1)
        ca, cb = cg.flip(1), cg.flip(2)
        real = model.flow_upsampler.prepare_cond
        model.flow_upsampler.prepare_cond = lambda r, *a, **k: r          # reference pattern: nothing reused
        with torch.inference_mode():
            model.weighted_interpolate_image(ca, cb, 0.5, cond_gt=cg)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for a in weights:
                model.weighted_interpolate_image(ca, cb, float(a), cond_gt=cg)
            torch.cuda.synchronize()
            dn = time.perf_counter() - t1
        model.flow_upsampler.prepare_cond = real
        out["reference_call_pattern_tiles_per_s"] = nb / dn
        out["speedup_vs_reference_call_pattern"] = out["tiles_per_s"] / world / (nb / dn)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
