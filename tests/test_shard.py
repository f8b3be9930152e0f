"""Multi-process path on the CPU (gloo, world_size 2) and the sharding helpers."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from interflow_amd import shard

HERE = os.path.dirname(os.path.abspath(__file__))


def test_partition_properties():
    for n in (0, 1, 5, 32, 169, 1000):
        for world in (1, 2, 3, 8):
            parts = [shard.partition(n, world, r) for r in range(world)]
            assert [i for p in parts for i in p] == list(range(n))            # every unit once, in order
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(AssertionError):
        shard.partition(4, 2, 2)


def test_tile_origins():
    t = shard.tile_origins(2048, 2048, 160)
    assert len(t) == 13 * 13 and t[0] == (0, 0, 160, 160) and t[-1] == (1920, 1920, 128, 128)
    cover = np.zeros((2048, 2048), np.int32)
    for y, x, h, w in t:
        cover[y:y + h, x:x + w] += 1
        assert h % 4 == 0 and w % 4 == 0
    assert (cover == 1).all()
    assert shard.tile_origins(160, 320, 160) == [(0, 0, 160, 160), (0, 160, 160, 160)]
    with pytest.raises(AssertionError):
        shard.tile_origins(162, 160, 160)                                       # flow.py:172


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_world(world, out, n_units):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py"), out, str(n_units)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=240)
        assert p.returncode == 0, o.decode(errors="replace")[-3000:]


def test_two_gloo_ranks_equal_one_process(tmp_path):
    """world_size 2 over gloo: the ranks' slices, gathered, equal the single-process run bit for bit (no unit lost,
    duplicated or reordered; no cross-sample coupling), and the max-over-ranks clock sees the slowest rank."""
    n = 5
    _run_world(1, str(tmp_path / "w1.npz"), n)
    _run_world(2, str(tmp_path / "w2.npz"), n)
    a, b = np.load(tmp_path / "w1.npz"), np.load(tmp_path / "w2.npz")
    assert sorted(a.files) == sorted(b.files) and len(a.files) == 3 * n
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


def test_host_handoff_ring_order_backpressure_and_errors():
    """generate.HostHandoff with host tensors (the GPU test covers the pinned / copy-stream leg): every unit reaches the sink
    once, in order, with its own data; a slow sink with one slot throttles the producer instead of losing data; the sink's
    exception surfaces in the producer."""
    import threading
    import time

    import torch
    from interflow_amd.generate import HostHandoff
    got = []
    with HostHandoff(lambda u, a: got.append((u, a.copy())), None, slots=2) as hh:
        for b in range(5):
            n = 1 + b % 3
            hh.put([(b, j) for j in range(n)], torch.full((n, 2, 4, 4, 3), b, dtype=torch.uint8) + torch.arange(n, dtype=torch.uint8).view(n, 1, 1, 1, 1))
    assert [u for u, _ in got] == [(b, j) for b in range(5) for j in range(1 + b % 3)]
    assert all(a.shape == (2, 4, 4, 3) and (a == u[0] + u[1]).all() for u, a in got)
    assert hh.batches == 5 and hh.bytes == sum((1 + b % 3) * 96 for b in range(5))
    # one slot, slow sink: put() waits for the slot; nothing is overwritten while the sink still reads it
    seen, inside = [], threading.Event()

    def slow(u, a):
        inside.set()
        time.sleep(0.05)
        seen.append(int(a[0, 0, 0, 0]))
    t0 = time.time()
    with HostHandoff(slow, None, slots=1) as hh:
        for b in range(4):
            hh.put([b], torch.full((1, 1, 2, 2, 3), b, dtype=torch.uint8))
    assert seen == [0, 1, 2, 3] and time.time() - t0 >= 0.19
    # sink failure
    def bad(u, a):
        raise OSError("disk full")
    hh = HostHandoff(bad, None, slots=2)
    hh.put([0], torch.zeros(1, 1, 2, 2, 3, dtype=torch.uint8))
    with pytest.raises(OSError):
        hh.close()
    with pytest.raises(ValueError):
        HostHandoff(bad, None, slots=0)
