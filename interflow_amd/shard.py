"""Sharding of independent units (patches, tiles, images) over one process per GPU.

Every patch/tile is independent in both directions of the flow: ranks take contiguous
slices of the unit list, weights are replicated, and the data path has NO collective.  `torch.distributed` (RCCL on
GPUs, gloo in the CPU tests) is used only for barriers, the max-over-ranks clock of the benchmarks and, when a caller
wants the results in one place, an object gather on the host side.

The reference has no counterpart (single GPU chosen by CUDA_VISIBLE_DEVICES, train.py:30, generate_images.py:132).
"""
from __future__ import annotations

import os
from typing import Callable, List, Sequence, Tuple

import torch


def dist_env() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) as torchrun exports them; (0, 0, 1) for a plain launch."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def rank_device():
    """(device, backend) of this rank for the benchmarks: cuda:LOCAL_RANK and RCCL ("nccl").  IFL_REHEARSE_ONE_GPU=1 puts
    every rank on cuda:0 with gloo for the collectives: a rehearsal of the multi-rank control flow (launch, barriers, the
    clock, the gradient exchange) on a one-GPU box -- RCCL refuses two ranks on one device; its figures mean nothing."""
    _, local_rank, _ = dist_env()
    if os.environ.get("IFL_REHEARSE_ONE_GPU") == "1":
        return torch.device("cuda", 0), "gloo"
    return torch.device("cuda", local_rank), "nccl"


def init_dist(backend: str, device=None):
    """Join the process group when WORLD_SIZE > 1 (MASTER_ADDR defaults to 127.0.0.1: one node).  Returns the
    torch.distributed module or None for a single process."""
    rank, _, world = dist_env()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def partition(n_units: int, world: int, rank: int) -> range:
    """Contiguous, balanced slice of range(n_units) for `rank`: sizes differ by at most one, every unit has exactly
    one owner, and the concatenation over ranks is range(n_units) in order."""
    assert world >= 1 and 0 <= rank < world and n_units >= 0
    return range(n_units * rank // world, n_units * (rank + 1) // world)


def tile_origins(H: int, W: int, tile: int) -> List[Tuple[int, int, int, int]]:
    """Cut an H x W image into independent tiles (y, x, h, w), row-major.  Border tiles keep their remainder size; all
    sizes must stay multiples of 4 (two squeezes, flow.py:172; dataset.py:139-147 crops images to multiples of 8)."""
    assert tile % 4 == 0 and tile > 0, tile
    assert H % 4 == 0 and W % 4 == 0, "{}".format((H, W, 4))
    return [(y, x, min(tile, H - y), min(tile, W - x)) for y in range(0, H, tile) for x in range(0, W, tile)]


def barrier(dist, device=None):
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value: float, dist, device="cpu") -> float:
    """The slowest rank's value (the benchmark's clock)."""
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def rank_devices(dist, device) -> list:
    """[{rank, device, name}] of every rank (benchmarks record it, so a scaling run shows which GPUs it really used)."""
    dev = torch.device(device)
    me = {"rank": dist_env()[0], "device": str(dev),
          "name": torch.cuda.get_device_name(dev) if dev.type == "cuda" else "cpu"}
    if dist is None:
        return [me]
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, me)
    return sorted(parts, key=lambda d: d["rank"])


def run_sharded(units: Sequence, process: Callable[[Sequence], Sequence], batch: int, rank: int, world: int):
    """Run `process` over this rank's slice of `units` in batches of at most `batch`; returns [(unit index, result)].
    `process(list of units) -> list of results`, one per unit."""
    own = partition(len(units), world, rank)
    out = []
    for s in range(own.start, own.stop, max(batch, 1)):
        idx = list(range(s, min(s + batch, own.stop)))
        res = process([units[i] for i in idx])
        assert len(res) == len(idx)
        out.extend(zip(idx, res))
    return out


def gather_results(local, dist, n_units: int):
    """Host-side gather of run_sharded() results to every rank, ordered by unit index (tests / single writer)."""
    if dist is None:
        parts = [local]
    else:
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, local)
    merged = dict(kv for part in parts for kv in part)
    assert sorted(merged) == list(range(n_units)), "units lost or duplicated by the sharding"
    return [merged[i] for i in range(n_units)]
