"""CPU-only checks of the host side: state-dict contract, C-ABI exports, parameter packing layout, error paths."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from interflow_amd import hip, pack, synth
from interflow_amd.flow_model import InterpFlow

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model():
    return InterpFlow([2, 4])


def test_state_dict_contract(golden_dir, model):
    keys = json.load(open(os.path.join(golden_dir, "state_keys.json")))
    sd = model.state_dict()
    assert set(sd) == set(keys)
    for k, shp in keys.items():
        assert list(sd[k].shape) == shp, k
    sd_np = synth.synth_interpflow_state([2, 4], seed=0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    # aliases share storage, as in the reference (flow_model.py:53-55)
    assert model.flow_upsampler is model.arch.flowUpsamplerNet and model.rrdb is model.arch.RRDB
    assert model.classes == [2, 4] and model.C == 48


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "interflow_hip.h")).read()
    declared = set(re.findall(r"\b(ifl_[a-z0-9_]+)\s*\(", header))
    assert declared == set(hip.EXPORTS), declared ^ set(hip.EXPORTS)
    L = hip.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.ifl_version() == 1


def test_param_block_layout():
    L = hip.lib()
    for C in (12, 48):
        offs = [L.ifl_step_field_off(C, f) for f in range(14)]
        assert offs == sorted(offs) and all(o % 64 == 0 for o in offs)
        assert L.ifl_step_param_floats(C, 1) > offs[-1]
        assert L.ifl_step_param_floats(C, 0) == offs[2]          # only the two ActNorm+1x1 blocks
    assert L.ifl_ws_bytes(2, 30, 32) == hip.IFL_EINVAL           # not a multiple of 4 (flow.py:172)
    assert L.ifl_ws_bytes(2, 32, 32) > 0


def test_pack_flow_shapes_and_invconv():
    sd = synth.synth_flow_state(seed=3, K=1, extra=1)
    kinds = synth.layer_plan(K=1, extra=1)
    params, layers = pack.pack_flow(sd, kinds)
    assert len(layers) == len(kinds) == 6
    assert [l.kind for l in layers] == [0, 1, 2, 0, 1, 2]
    assert np.isfinite(params).all()
    w, winv, sls = pack.invconv_matrices(sd, "layers.1.invconv.")
    np.testing.assert_allclose(w @ winv, np.eye(12), atol=2e-6)
    assert abs(sls - np.log(abs(np.linalg.det(w.astype(np.float64))))) < 1e-5
    # conv1 packing is a pure permutation (plus zero padding of the z1 channels)
    wt = sd["layers.2.affine.fAffine.0.weight"]
    p = pack._pack_conv1(wt, 6, 8)
    assert p.size == 64 * 9 * 328 and np.isclose(np.abs(p).sum(), np.abs(wt).sum(), rtol=1e-6)
    np.testing.assert_array_equal(np.sort(p[p != 0]), np.sort(wt.reshape(-1)[wt.reshape(-1) != 0]))
    rows = pack._row_map(12, 1)
    assert sorted(rows[rows >= 0].tolist()) == list(range(24))    # every fFeatures row placed exactly once
    rows = pack._row_map(48, 0)
    assert sorted(rows[rows >= 0].tolist()) == list(range(48))


def test_training_mode_with_uninitialised_actnorm_is_rejected():
    sd = synth.synth_flow_state(seed=3, K=1, extra=1)
    sd["layers.1.actnorm.bias"] = np.zeros_like(sd["layers.1.actnorm.bias"])
    with pytest.raises(NotImplementedError):
        pack.pack_flow(sd, synth.layer_plan(K=1, extra=1), training=True)


def test_cpu_tensors_are_rejected(model):
    x = torch.zeros(1, 3, 8, 8)
    with pytest.raises(hip.InterflowHipError):
        hip.squeeze2d(x)
    feats = {k: torch.from_numpy(v) for k, v in synth.synth_features(0, 1, 8, 8).items()}
    with pytest.raises(hip.InterflowHipError):
        model(x, None, lr_enc=feats, add_gt_noise=False, calc_loss=False)
    with pytest.raises(NotImplementedError):
        model(None, x, reverse=True)                              # flow_model.py:85-86


def test_winograd_weight_transform_reproduces_the_direct_convolution():
    """pack._pack_conv1_wino: U = G g G^T in the kernel's layout.  Unpacked and pushed through the data and output
    transforms of conv1_ft_wino_kernel (restated here in numpy, including the clamped window of an odd-sized plane), it
    must give the direct 3x3 correlation (flow.py:25-64, padding "same")."""
    from interflow_amd import pack
    rng = np.random.default_rng(0)
    zoff = 6
    wt = rng.standard_normal((64, zoff + 320, 3, 3)).astype(np.float32)
    blob = pack._pack_conv1_wino(wt, zoff)
    assert blob.dtype == np.float32 and blob.size == 16 * 64 * 320
    u = blob.reshape(4, 10, 4, 4, 2, 2, 32, 4)                     # jc cc kq i mt h o e
    U = np.zeros((64, 320, 4, 4), np.float32)
    for jc in range(4):
        for i in range(4):
            U[:, :, i, jc] = u[jc, :, :, i].transpose(2, 4, 0, 1, 3, 5).reshape(64, 320)
    H, W = 5, 3
    x = np.zeros((H + 2, W + 2, 320), np.float32)
    x[1:-1, 1:-1] = rng.standard_normal((H, W, 320))
    g = wt[:, zoff:].astype(np.float64)
    ref = np.zeros((H, W, 64))
    for ky in range(3):
        for kx in range(3):
            ref += np.einsum("yxc,oc->yxo", x[ky:ky + H, kx:kx + W].astype(np.float64), g[:, :, ky, kx])
    TH, TW = (H + 1) // 2, (W + 1) // 2
    out = np.zeros((2 * TH, 2 * TW, 64), np.float32)
    for tr in range(TH):
        for tc in range(TW):
            d = np.stack([np.stack([x[min(2 * tr + r, H + 1), min(2 * tc + c, W + 1)] for c in range(4)]) for r in range(4)])
            R = np.stack([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]])
            V = np.stack([R[:, 0] - R[:, 2], R[:, 1] + R[:, 2], R[:, 2] - R[:, 1], R[:, 1] - R[:, 3]], axis=1)
            M = np.einsum("ocij,ijc->ijo", U, V)
            T = np.stack([M[0] + M[1] + M[2], M[1] - M[2] - M[3]])
            out[2 * tr:2 * tr + 2, 2 * tc:2 * tc + 2] = np.stack([T[:, 0] + T[:, 1] + T[:, 2], T[:, 1] - T[:, 2] - T[:, 3]], axis=1)
    assert np.abs(out[:H, :W] - ref).max() <= 1e-5 * np.abs(ref).max()


def test_encoder_has_no_eager_cpu_path():
    """RRDBNet.forward on a CPU tensor raises instead of evaluating with eager PyTorch ops (no fallback)."""
    import pytest
    import torch
    from interflow_amd import hip
    from interflow_amd.flow_model import RRDBNet
    net = RRDBNet(3, 3, 64, 1, opt={"network_G": {"flow": {"stackRRDB": {"blocks": [0]}}}})
    with pytest.raises(hip.InterflowHipError):
        net(torch.zeros(1, 3, 8, 8))
    with pytest.raises(hip.InterflowHipError):
        net(torch.zeros(1, 3, 8, 8, dtype=torch.float64))


@pytest.mark.parametrize("correct_splits,channels", [(False, [12, 24, 96]), (True, [12, 24, 48])])
def test_multi_level_configurations_build_with_split2d(correct_splits, channels):
    """L=3 with Split2d layers (FlowUpsamplerNet.py:149-167): the layer list, the channel counts and the state-dict keys of the
    synthesiser (which the reference loads with strict=True: tests/golden/make_golden_general.py)."""
    from interflow_amd.flow_modules import FlowUpsamplerNet, Split2d
    from tests.golden.make_golden_general import general_opt, K, L, EXTRA, SEED
    fu = FlowUpsamplerNet((160, 160, 3), 64, K, flow_coupling='CondAffineSeparatedAndCond', LU_decomposed=True,
                          opt=general_opt(correct_splits))
    assert fu._general and fu.C == channels[-1]
    assert [c for k, c in fu._kinds if k == "squeeze"] == channels
    assert sum(isinstance(m, Split2d) for m in fu.layers) == (2 if correct_splits else 1)
    assert [k for k, _ in fu._kinds] == [k for k, _ in synth.layer_plan(L, K, EXTRA, split={"correct_splits": correct_splits})]
    sd_np = synth.synth_flow_state(SEED, L=L, K=K, extra=EXTRA, split={"correct_splits": correct_splits})
    fu.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    with pytest.raises(hip.InterflowHipError):
        fu(gt=torch.zeros(1, 3, 32, 32), rrdbResults={k: torch.zeros(1, 320, 32 >> i, 32 >> i) for i, k in
                                                        enumerate(("fea_up4", "fea_up2", "fea_up1", "fea_up0"))},
           reverse=False, y_onehot=torch.zeros(1), test_input_size=32)
    opt = general_opt(False)
    opt['network_G']['flow']['split']['conditional'] = True
    with pytest.raises(NotImplementedError):
        FlowUpsamplerNet((160, 160, 3), 64, K, flow_coupling='CondAffineSeparatedAndCond', LU_decomposed=True, opt=opt)
    assert not InterpFlow([2, 4]).flow_upsampler._general        # the shipped configuration stays on the fused driver


def test_unchanged_conditioning_is_recognised_by_object_version_and_layout(model):
    """hip.set_cross_pass_overlap: a pass may start its bulk work early only if every conditioning tensor is the very object the
    previous pass loaded, unwritten since (version counter), at the same address, float32 and contiguous (host logic of
    FlowUpsamplerNet._cond_unchanged / _remember_cond; the GPU side is tests/test_gpu_parity.py)."""
    fu = model.flow_upsampler
    saved = fu._last_cond
    try:
        a, b = torch.zeros(1, 4, 6, 6), torch.zeros(1, 4, 3, 3)
        feats = {1: a, 2: b}
        fu._last_cond = None
        assert not fu._cond_unchanged(feats)                      # nothing loaded yet
        fu._remember_cond(feats)
        assert fu._cond_unchanged(feats)
        assert fu._cond_unchanged({1: a, 2: b})                   # another dict of the same tensors
        assert not fu._cond_unchanged({1: a})                     # another set of levels
        assert not fu._cond_unchanged({1: a, 2: b.clone()})       # equal values, another object
        a.add_(1.0)                                               # written in place
        assert not fu._cond_unchanged(feats)
        fu._remember_cond(feats)
        assert fu._cond_unchanged(feats)
        assert not fu._cond_unchanged({1: a.double(), 2: b})
        nc = torch.zeros(1, 4, 6, 12)[..., ::2]
        fu._remember_cond({1: nc, 2: b})
        assert not fu._cond_unchanged({1: nc, 2: b})              # needs a contiguous copy: never early
        with torch.inference_mode():
            inf = torch.zeros(1, 4, 6, 6)
        fu._remember_cond({1: inf, 2: b})
        assert not fu._cond_unchanged({1: inf, 2: b})             # no version counter: writes could not be seen
        c = torch.zeros(1, 4, 6, 6)
        fu._remember_cond({1: c, 2: b})
        del c                                                     # the remembered tensor is gone (weak reference)
        assert not fu._cond_unchanged({1: torch.zeros(1, 4, 6, 6), 2: b})
        prev = hip.set_cross_pass_overlap(False)
        assert hip.set_cross_pass_overlap() is False and hip.set_cross_pass_overlap(prev) is False
        assert hip.set_cross_pass_overlap() == prev
    finally:
        fu._last_cond = saved
