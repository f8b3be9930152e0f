"""ifl_gaussian_logp_static (the entry point that allocates nothing) against ifl_gaussian_logp: same kernels, so the results
must be bit-equal (integer/bit-exact bar), on two calls in a row, the second with a smaller input, plus the size limit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(hip, dev, B, C, H, W, seed, with_cov):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(B, C, H, W, generator=g).to(dev)
    mean = cov_inv = None
    ld = 0.0
    if with_cov:
        a = torch.randn(C, C, generator=g, dtype=torch.float64) * 0.2 + torch.eye(C, dtype=torch.float64)
        cov = a @ a.T
        cov_inv = torch.inverse(cov).float().to(dev).contiguous()
        ld = float(torch.slogdet(cov)[1])
        mean = torch.full((C,), 0.05).to(dev)
    want = hip.gaussian_logp(z, mean, cov_inv, ld)
    out = torch.full((B,), float("nan"), device=dev)
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated(dev)
    got = hip.gaussian_logp_static(z, out, mean, cov_inv, ld)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated(dev) == before          # torch's allocator was not touched either
    assert torch.isfinite(got).all()
    assert torch.equal(got, want), (got - want).abs().max()


def test_static_scratch_logp_matches_and_survives_a_smaller_second_call():
    from interflow_amd import hip
    dev = torch.device("cuda:0")
    _case(hip, dev, 8, 48, 40, 40, 1, True)       # 7 blocks per sample
    _case(hip, dev, 3, 48, 12, 20, 2, True)       # smaller: 1 block per sample; stale partials of the first call stay unread
    _case(hip, dev, 5, 12, 80, 80, 3, False)      # standard normal, C = 12
    _case(hip, dev, 2, 7, 9, 5, 4, True)          # generic-C kernel, ragged plane


def test_static_scratch_logp_rejects_sizes_beyond_its_array():
    from interflow_amd import hip
    dev = torch.device("cuda:0")
    B, HW = 65, 512 * 512                          # 65 * 1024 blocks > 65 536 floats
    assert B * hip.lib().ifl_logp_scratch_floats(HW) > hip.LOGP_STATIC_MAX_FLOATS
    z = torch.zeros(B, 1, 512, 512, device=dev)
    out = torch.zeros(B, device=dev)
    with pytest.raises(NotImplementedError, match="static scratch"):
        hip.gaussian_logp_static(z, out)
    hip.gaussian_logp_static(z[:64], out[:64])     # 64 * 1024 just fits
    torch.cuda.synchronize()
    assert torch.allclose(out[:64], torch.full((64,), -0.5 * 1.8378770664093453 * HW, device=dev), rtol=1e-6)
