"""ResizeRight (antialiased bicubic x1/4, train.py:99-100): the oracle's restatement and the product's matrix form against
outputs of the reference itself (tests/golden/train_seed0.npz, made by tests/golden/make_golden_train.py).  CPU only."""
import os

import numpy as np
import torch

from interflow_amd import resize as R, synth
from oracle import interflow_oracle as O


def test_resize_against_reference_outputs(golden_dir):
    g = np.load(os.path.join(golden_dir, "train_seed0.npz"))
    for i in range(3):
        shape = tuple(int(v) for v in g[f"resize{i}_shape"])
        x = torch.from_numpy(synth.synth_images(300 + i, shape[0], shape[2], shape[3])[:, :shape[1]].copy())
        want = g[f"resize{i}_out"]
        assert np.array_equal(O.resize_right(x, 0.25).numpy(), want)                    # same operations, same order
        np.testing.assert_allclose(R.resize_last2(x, 0.25).numpy(), want, rtol=0, atol=5e-7)
    A = R.resize_matrix(160, 0.25)
    assert A.shape == (40, 160)
    assert torch.allclose(A[5:-5].sum(1), torch.ones(30), atol=1e-6)                     # interior rows are normalised
    assert float(A[0].sum()) < 1.0                                                       # border rows lose the zero-padded taps
