"""Training path on the GPU: the differentiable HIP convolution and the gradients of the flow against CPU autograd of the
oracle (plain torch CPU ops).  Tolerance: rel 1e-4 of the tensor's max magnitude unless stated."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from interflow_amd import train_ops as T

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.isfinite(a).all()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("cins,cout,k,B,h,w", [((6, 320), 64, 3, 2, 20, 36), ((64,), 64, 1, 2, 20, 36), ((64,), 24, 3, 1, 5, 7),
                                               ((64,), 96, 3, 2, 9, 40), ((24, 320), 64, 3, 1, 11, 13), ((320,), 64, 3, 3, 80, 80)])
def test_conv2d_ring_forward_and_gradients(cins, cout, k, B, h, w):
    g = torch.Generator().manual_seed(cout * 7 + h)
    xs = [torch.randn(B, c, h, w, generator=g) for c in cins]
    wt = torch.randn(cout, sum(cins), k, k, generator=g) * 0.05
    bias = torch.randn(cout, generator=g)
    gy = torch.randn(B, cout, h, w, generator=g)
    # reference: torch CPU autograd
    xr = [x.clone().requires_grad_() for x in xs]
    wr, br = wt.clone().requires_grad_(), bias.clone().requires_grad_()
    yr = F.conv2d(torch.cat(xr, 1), wr, br, 1, k // 2)
    yr.backward(gy)
    # HIP
    xd = [x.to(dev()).requires_grad_() for x in xs]
    wd, bd = wt.to(dev()).requires_grad_(), bias.to(dev()).requires_grad_()
    y = T.from_ring(T.conv2d_ring([T.to_ring(x) for x in xd], cins, wd, bd), cout)
    assert rel(y, yr) < 1e-5
    y.backward(gy.to(dev()))
    assert rel(wd.grad, wr.grad) < 1e-4 and rel(bd.grad, br.grad) < 1e-5
    for a, b in zip(xd, xr):
        assert rel(a.grad, b.grad) < 1e-5
    # without bias, and with a source that needs no gradient
    xd2 = [x.to(dev()) for x in xs]
    wd2 = wt.to(dev()).requires_grad_()
    y2 = T.from_ring(T.conv2d_ring([T.to_ring(x) for x in xd2], cins, wd2, None), cout)
    assert rel(y2, yr - bias.reshape(1, -1, 1, 1)) < 1e-5
    y2.backward(gy.to(dev()))
    assert rel(wd2.grad, wr.grad) < 1e-4
