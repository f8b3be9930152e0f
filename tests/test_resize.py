"""ResizeRight (antialiased bicubic x1/4, train.py:99-100): the oracle's restatement and the product's tap tables against
outputs of the reference itself (tests/golden/train_seed0.npz, made by tests/golden/make_golden_train.py) on the CPU; the HIP
kernel (forward and adjoint) against them on the GPU."""
import os

import numpy as np
import pytest
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
        Ah, Aw = R.resize_matrix(shape[2], 0.25), R.resize_matrix(shape[3], 0.25)        # the map the kernel's tables encode
        np.testing.assert_allclose(torch.matmul(torch.matmul(Ah, x), Aw.t()).numpy(), want, rtol=0, atol=5e-7)
    A = R.resize_matrix(160, 0.25)
    assert A.shape == (40, 160)
    assert torch.allclose(A[5:-5].sum(1), torch.ones(30), atol=1e-6)                     # interior rows are normalised
    assert float(A[0].sum()) < 1.0                                                       # border rows lose the zero-padded taps


def test_tap_tables_encode_the_matrix_and_its_transpose():
    for n in (160, 36, 17):
        A = R.resize_matrix(n, 0.25)
        (fi, fw), (bi, bw) = R.resize_taps(n, 0.25)
        assert fi.dtype == torch.int32 and fi.shape == fw.shape and fi.shape[0] == A.shape[0] and fi.shape[1] <= 16
        assert bi.shape[0] == n and int(fi.min()) >= 0 and int(fi.max()) < n and int(bi.max()) < A.shape[0]
        B = torch.zeros_like(A)
        B.index_put_((torch.arange(A.shape[0])[:, None].expand_as(fi), fi.long()), fw, accumulate=True)
        Bt = torch.zeros(n, A.shape[0])
        Bt.index_put_((torch.arange(n)[:, None].expand_as(bi), bi.long()), bw, accumulate=True)
        assert torch.equal(B, A) and torch.equal(Bt, A.t())


@pytest.mark.gpu
def test_resize_kernel_forward_and_adjoint_on_gpu(golden_dir):
    """ifl_resize_sep through resize.resize_last2: the reference's outputs (atol 5e-7), the gradient against autograd of the
    dense-matrix form, odd sizes, and no CPU evaluation."""
    from interflow_amd import hip
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "train_seed0.npz"))
    for i in range(3):
        shape = tuple(int(v) for v in g[f"resize{i}_shape"])
        x = torch.from_numpy(synth.synth_images(300 + i, shape[0], shape[2], shape[3])[:, :shape[1]].copy())
        np.testing.assert_allclose(R.resize_last2(x.to(dev), 0.25).cpu().numpy(), g[f"resize{i}_out"], rtol=0, atol=5e-7)
    for shape in ((2, 3, 160, 160), (1, 2, 36, 20), (3, 1, 17, 52)):
        x = torch.randn(*shape, generator=torch.Generator().manual_seed(shape[2]))
        gy_shape = (*shape[:2], -(-shape[2] // 4), -(-shape[3] // 4))
        gy = torch.randn(*gy_shape, generator=torch.Generator().manual_seed(7))
        xr = x.clone().requires_grad_()
        Ah, Aw = R.resize_matrix(shape[2], 0.25), R.resize_matrix(shape[3], 0.25)
        yr = torch.matmul(torch.matmul(Ah, xr), Aw.t())
        yr.backward(gy)
        xd = x.to(dev).requires_grad_()
        y = R.resize_last2(xd, 0.25)
        assert tuple(y.shape) == gy_shape
        np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), rtol=0, atol=2e-6)
        y.backward(gy.to(dev))
        np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=0, atol=2e-6)
    with pytest.raises(hip.InterflowHipError):
        R.resize_last2(torch.zeros(1, 3, 16, 16), 0.25)
