"""Golden fixture for the training path, produced by running the REFERENCE itself on CPU (authoring container only).

Like make_golden.py: nothing of the reference is copied; it is imported (placeholder modules for the unused natsort / cv2 /
torchvision imports of utils/util.py, `.cuda()` made identity), loaded with the build's synthesised weights and fed seeded
inputs.  Stored (tests/golden/train_seed0.npz):
  * resize: `utils.resize_right.resize(x, scale_factors=1/4, interp_method=cubic, antialiasing=True)` (train.py:99-100) on
    three shapes;
  * one training step's loss terms as train.py:72-110 computes them -- NLL, ib and the LR-consistency term with the random
    draws pinned (interpolation weights given, the quantisation / dequantisation noise set to zero) -- and, after
    `loss.backward()`, per-parameter gradient checksums (sum and sum of squares in float64) for every parameter that
    received a gradient, plus a few small gradient tensors in full.

usage: python tests/golden/make_golden_train.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/interflow"
from interflow_amd import synth  # noqa: E402

FULL = ("flow_upsampler.layers.1.actnorm.logs", "flow_upsampler.layers.3.invconv.l", "flow_upsampler.layers.37.affine.fAffine.4.bias",
        "flow_upsampler.layers.25.affine.fFeatures.2.weight", "mean_shift_0", "cov_shift_1", "arch.RRDB.conv_first.bias",
        "arch.RRDB.RRDB_trunk.22.RDB3.conv5.bias")


def main():
    for name in ("natsort", "cv2", "torchvision", "torchvision.utils"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision.utils"].make_grid = None
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.path.insert(0, REF)
    from models.flow_model import InterpFlow
    from utils.resize_right import resize
    from utils.interp_methods import cubic
    torch.Tensor.cuda = lambda self, *a, **k: self
    out = {}
    for i, shape in enumerate(((2, 3, 32, 40), (1, 3, 160, 160), (1, 2, 20, 36))):
        x = torch.from_numpy(synth.synth_images(300 + i, shape[0], shape[2], shape[3])[:, :shape[1]].copy())
        out[f"resize{i}_shape"] = np.array(shape)
        out[f"resize{i}_out"] = resize(input=x, scale_factors=1. / 4, interp_method=cubic, antialiasing=True).numpy()

    scales = [2, 4]
    sd_np = synth.synth_interpflow_state(scales, seed=0)
    model = InterpFlow(scales)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    model.train()
    B, H = 2, 32
    gt, low_res, lr1, lr2 = (torch.from_numpy(synth.synth_images(s, B, H, H)) for s in (201, 202, 203, 204))
    labels = torch.tensor([2, 4])
    w = torch.tensor([0.3, 0.8])                                       # interpolate_weight (train.py:88), pinned
    real_rand = torch.rand
    torch.rand = lambda *a, **k: torch.full(a[0] if isinstance(a[0], (tuple, list, torch.Size)) else a, 0.5)   # zero noise
    try:
        lr_enc = model.arch.rrdbPreprocessing(gt)                                                   # train.py:72
        z, nll, logdet, ll_ib = model(low_res, gt, labels, lr_enc=lr_enc)                           # :73
        ib_loss, nll_loss = -torch.mean(ll_ib), torch.mean(nll)                                     # :74-75
        z1, _ = model(lr1, gt, torch.tensor([2] * B), lr_enc=lr_enc, calc_loss=False)               # :85
        z2, _ = model(lr2, gt, torch.tensor([4] * B), lr_enc=lr_enc, calc_loss=False)               # :86
        w1, w2 = (1 - w).reshape(-1, 1, 1, 1), w.reshape(-1, 1, 1, 1)                               # :89-90
        interp_img, _ = model(w1 * z1 + w2 * z2, gt, reverse=True, test_input_size=H, lr_enc=lr_enc)    # :92-94
        synth_lr = torch.zeros_like(low_res) + w1 * lr1 + w2 * lr2                                  # :99-100
        synth_lr = synth_lr + ((torch.rand(synth_lr.shape) - 0.5) / 256)                            # :101
        interp_down = resize(input=interp_img, scale_factors=1. / 4, interp_method=cubic, antialiasing=True)
        synth_down = resize(input=synth_lr, scale_factors=1. / 4, interp_method=cubic, antialiasing=True)
        cons = torch.nn.L1Loss()(synth_down, interp_down).mean()                                    # :104-106
    finally:
        torch.rand = real_rand
    loss = nll_loss + 10 * cons + 1 * ib_loss                                                       # :110, default lambdas
    loss.backward()
    out.update(nll=nll.detach().numpy(), ll_ib=ll_ib.detach().numpy(), cons=cons.detach().numpy(), loss=loss.detach().numpy(),
               interp_img=interp_img.detach().numpy(), w=w.numpy())
    names, sums, sqs = [], [], []
    seen = set()
    for name, p in model.named_parameters():
        if p.grad is None or id(p) in seen:
            continue
        seen.add(id(p))
        names.append(name)
        sums.append(p.grad.double().sum().item())
        sqs.append((p.grad.double() ** 2).sum().item())
    out["grad_names"] = np.array(names)
    out["grad_sum"], out["grad_sq"] = np.array(sums), np.array(sqs)
    named = dict(model.named_parameters())
    alias = {n.replace("arch.flowUpsamplerNet.", "flow_upsampler."): n for n in named}
    for k in FULL:
        out["g_" + k] = named[alias.get(k, k)].grad.numpy()
    out["digest"] = np.array(synth.state_digest(sd_np))
    path = os.path.join(HERE, "train_seed0.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "params with grad:", len(names), "loss", float(loss), "cons", float(cons))


if __name__ == "__main__":
    main()
