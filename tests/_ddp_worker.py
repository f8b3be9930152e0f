"""Worker of tests/test_ddp_gloo.py: one of WORLD_SIZE gloo ranks on the CPU running interflow_amd.training's
data-parallel machinery (bucketed asynchronous gradient all-reduce, parameter broadcast, the optimizer step) on a small
CPU module -- the flow itself needs a GPU, the exchange logic does not."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from interflow_amd import shard, training  # noqa: E402


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = torch.nn.Conv2d(3, 8, 3, padding=1)
        self.b = torch.nn.Conv2d(8, 4, 1)
        self.dead = torch.nn.Linear(5, 5)                 # registered, never used (like degradation_net)
        self.alias = self.a                               # the model registers aliases of the same modules
        self.scale = torch.nn.Parameter(torch.ones(4, 1, 1))

    def forward(self, x):
        return (self.b(torch.relu(self.a(x))) * self.scale).square().mean(dim=(1, 2, 3))


def main():
    out_path = sys.argv[1]
    rank, _, world = shard.dist_env()
    dist = shard.init_dist("gloo")
    torch.set_num_threads(1)
    net = Net()
    if rank != 0:                                         # replicas that start different ...
        with torch.no_grad():
            for p in net.parameters():
                p.add_(0.1 * rank)
    training.broadcast_parameters(net, dist)              # ... are made equal to rank 0
    x = torch.randn(8, 3, 6, 6, generator=torch.Generator().manual_seed(1))
    own = shard.partition(8, world, rank)
    opt, sched = training.make_optimizer(net, lr=1e-2)
    red = training.GradientAllReducer(net.parameters(), dist, bucket_bytes=512)        # several small buckets
    assert len(red.buckets) > 2 and len(red.params) == 7
    for it in range(3):
        red.zero_grad()
        loss = net(x[own.start:own.stop]).mean()
        loss.backward()
        red.finish()
        if it == 0:
            grads = {n: p.grad.clone().numpy() for n, p in net.named_parameters()}
        opt.step()
    shard.barrier(dist)
    if rank == 0:
        np.savez(out_path, **{"g_" + k: v for k, v in grads.items()},
                 **{"p_" + n: p.detach().numpy() for n, p in net.named_parameters()})
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
