"""Degradation generation by latent interpolation: the loop of the reference's `generate_images.py:45-110`.

For every (GT, LR_a, LR_b) triple and every interpolation weight the reference calls
`model.weighted_interpolate_image(LR_a, LR_b, w, cond_gt=GT)` (`generate_images.py:79-81`), i.e. per weight: three
RRDB encoder passes, two flow encodes and one flow decode, although only the decode depends on the weight
(SURVEY.md section 3.1).  Here, per batch of images or tiles:

    lr_enc = rrdbPreprocessing(GT)                      once
    prepare_cond(lr_enc)                                once   (the conditioning-only 93 % of the flow's FLOPs)
    z1, z2 = encode(LR_a), encode(LR_b)                 once
    for w in weights:  decode((1-w) z1 + w z2) -> clip -> uint8 NHWC      (generate_images.py:89-91)

The outputs equal the reference's call by call (same kernels as `InterpFlow.weighted_interpolate_image`).
Units (whole images, or independent tiles: BASELINE.json configs[2]) are sharded over ranks with
`interflow_amd.shard` -- no collective on the data path.  PNG writing stays with the caller (out of scope).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

from . import hip, shard


def default_weights(divide_mode: int = 0) -> np.ndarray:
    """generate_images.py:141-150"""
    return np.linspace(0.0, 1.0, 21)[1:-1] if divide_mode == 0 else np.linspace(0.0, 1.0, 11)[1:-1]


@torch.inference_mode()
def interpolate_batch(model, gt, lr_a, lr_b, weights: Sequence[float], add_noise: bool = False, quantize: bool = True):
    """All interpolation weights for one batch [B,3,H,W] of images/tiles.  Returns a list (one entry per weight) of
    uint8 NHWC tensors (quantize=True) or float32 NCHW tensors."""
    B, _, H, W = lr_a.shape
    fu = model.flow_upsampler
    lr_enc = fu.prepare_cond(model.arch.rrdbPreprocessing(gt), B, H, W)
    z1, _ = model(lr_a, gt, lr_enc=lr_enc, add_gt_noise=add_noise, calc_loss=False)
    z2, _ = model(lr_b, gt, lr_enc=lr_enc, add_gt_noise=add_noise, calc_loss=False)
    outs = []
    for a in weights:
        z = hip.lerp(z1, z2, 1 - a, a)                                        # flow_model.py:198
        x = model(z, gt, reverse=True, add_gt_noise=add_noise, lr_enc=lr_enc, test_input_size=H, calc_loss=False)[0]
        outs.append(hip.quantize_u8_nhwc(x) if quantize else x)
    return outs


@torch.inference_mode()
def generate(model, gt, lr_a, lr_b, weights: Optional[Sequence[float]] = None, tile: Optional[int] = None,
             batch: int = 32, rank: int = 0, world: int = 1, add_noise: bool = False):
    """gt, lr_a, lr_b: [N,3,H,W] float32 device tensors (aligned HR / LR images of one size, as RealSR-v2).
    tile=None: whole images, as the reference does.  tile=T: every image is cut into independent T x T tiles
    (border tiles keep the remainder; a tile's output differs from the whole-image output near its border, since
    each tile is zero-padded on its own).
    Returns (units, results): units = [(image, y, x, h, w)] owned by this rank, results[i] = uint8 array
    [n_weights, h, w, 3] of unit i, on the device."""
    weights = default_weights() if weights is None else list(weights)
    N, _, H, W = gt.shape
    tiles = [(0, 0, H, W)] if tile is None else shard.tile_origins(H, W, tile)
    units = [(n, y, x, h, w) for n in range(N) for (y, x, h, w) in tiles]
    own = [units[i] for i in shard.partition(len(units), world, rank)]
    results: List[Optional[torch.Tensor]] = [None] * len(own)
    by_shape = {}
    for i, u in enumerate(own):                       # batches need one tile size
        by_shape.setdefault((u[3], u[4]), []).append(i)
    for (h, w), idx in by_shape.items():
        for s in range(0, len(idx), batch):
            sel = idx[s:s + batch]

            def crop(t):
                return torch.stack([t[own[i][0], :, own[i][1]:own[i][1] + h, own[i][2]:own[i][2] + w] for i in sel])
            outs = interpolate_batch(model, crop(gt), crop(lr_a), crop(lr_b), weights, add_noise)
            stacked = torch.stack(outs, dim=1)        # [b, n_weights, h, w, 3]
            for j, i in enumerate(sel):
                results[i] = stacked[j]
    return own, results


def assemble(units, results, N, H, W, n_weights, device):
    """Paste unit results back into [n_weights, N, H, W, 3] uint8 images (single writer / tests)."""
    out = torch.zeros(n_weights, N, H, W, 3, dtype=torch.uint8, device=device)
    for (n, y, x, h, w), r in zip(units, results):
        out[:, n, y:y + h, x:x + w] = r.to(device)
    return out
