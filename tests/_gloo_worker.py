"""Worker of tests/test_shard.py: one of WORLD_SIZE gloo ranks on the CPU.  Each rank runs the CPU oracle (the checker,
standing in for the GPU path, which needs no collective) on its slice of the units and the results are gathered on the
host side; also exercises the benchmark's barrier and max-over-ranks clock."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from interflow_amd import shard, synth  # noqa: E402
from oracle import interflow_oracle as O  # noqa: E402


def make_units(n):
    x = synth.synth_images(31, n, 8, 12)
    f = synth.synth_features(32, n, 8, 12)
    return [(x[i:i + 1], {k: v[i:i + 1] for k, v in f.items()}) for i in range(n)]


def process_factory(sd):
    def process(units):
        out = []
        for x, feats in units:
            feats = {k: torch.from_numpy(v) for k, v in feats.items()}
            z, ld = O.flow_encode(torch.from_numpy(x), feats, sd, "", torch.zeros(1))
            xr, _ = O.flow_decode(z, feats, sd, "", torch.zeros(1))
            out.append((z.numpy(), ld.numpy(), xr.numpy()))
        return out
    return process


def main():
    out_path, n_units = sys.argv[1], int(sys.argv[2])
    rank, _, world = shard.dist_env()
    dist = shard.init_dist("gloo")
    torch.set_num_threads(1)
    sd = O.to_torch(synth.synth_flow_state(seed=5, K=1, extra=1))
    units = make_units(n_units)
    shard.barrier(dist)
    local = shard.run_sharded(units, process_factory(sd), batch=2, rank=rank, world=world)
    assert [i for i, _ in local] == list(shard.partition(n_units, world, rank))
    slowest = shard.max_over_ranks(float(rank + 1), dist)
    assert slowest == float(world), slowest
    merged = shard.gather_results(local, dist, n_units)
    shard.barrier(dist)
    if rank == 0:
        np.savez(out_path, **{f"{k}{i}": m[j] for i, m in enumerate(merged) for j, k in enumerate(("z", "ld", "xr"))})
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
