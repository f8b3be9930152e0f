"""Data-parallel training machinery over gloo (world_size 2) on the CPU: the bucketed gradient all-reduce and the
parameter broadcast give every rank the gradients / parameters of the single-process run on the whole batch."""
import os
import subprocess
import sys

import numpy as np

from tests.test_shard import _free_port

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(world, out, worker="_ddp_worker.py", *extra):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, worker), out, *extra], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=240)
        assert p.returncode == 0, o.decode(errors="replace")[-3000:]


def test_gradient_allreduce_and_broadcast_two_ranks(tmp_path):
    _run(1, str(tmp_path / "w1.npz"))
    _run(2, str(tmp_path / "w2.npz"))
    a, b = np.load(tmp_path / "w1.npz"), np.load(tmp_path / "w2.npz")
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        np.testing.assert_allclose(b[k], a[k], rtol=2e-5, atol=1e-7, err_msg=k)
    assert np.all(a["g_dead.weight"] == 0) and np.all(b["g_dead.weight"] == 0)      # never-used parameters stay zero


import pytest


@pytest.mark.parametrize("mode", ["plain", "nan", "large", "plain_late", "nan_late", "large_late"])
def test_train_step_two_ranks_init_broadcast_and_collective_skip(tmp_path, mode):
    """train_step with per-rank data on 2 gloo ranks: the data-dependent ActNorm initialisation is broadcast before the
    first optimizer step, a non-finite / too large loss on ONE rank makes EVERY rank drop the update (train.py:113-126),
    parameters stay finite and bit-identical across the replicas (asserted inside the workers).  `_late`: the same
    with `late_check=True` (guards after the backward pass, gradients all-reduced only after the collective decision)."""
    out = str(tmp_path / f"s_{mode}.npz")
    _run(2, out, "_ddp_step_worker.py", mode)
    r = np.load(out)
    assert np.isfinite(r["flat"]).all() and np.any(r["bias"] != 0)
