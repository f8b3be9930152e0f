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
`interflow_amd.shard` -- no collective on the data path.  PNG writing stays with the caller (out of scope); the hand-off to
it is `HostHandoff`: the quantised results of a batch go to the host asynchronously (a copy stream, a ring of pinned
buffers) and a writer thread passes each unit's array to the caller's `sink` -- the reference's `ImageSaver` threads fed
from a `Queue` (`generate_images.py:24-43,83-101`), without its blocking `.cpu()` inside the generation loop.
"""
from __future__ import annotations

import queue
import threading
from typing import Callable, List, Optional, Sequence

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


_PINNED = {}          # (device index, slot) -> pinned uint8 buffer, kept between hand-offs (pinning host memory costs milliseconds per MB)


class HostHandoff:
    """Device -> host hand-off of finished results to a writer (generate_images.py:83-101).

    `put(units, stacked)` enqueues the copy of a batch's results (`stacked[j]` belongs to `units[j]`) into one of `slots`
    pinned host buffers on a copy stream of its own, ordered after the producing stream by an event, and returns at once;
    a writer thread waits for the copy (the wait releases the GIL) and calls `sink(unit, array)` for every unit, `array` a
    numpy view of the pinned buffer that is valid during the call only.  A slot is reused once the writer is through with
    it, so `put` blocks -- back-pressure -- when the writer falls `slots` batches behind.  `close()` drains the ring, joins
    the thread and re-raises the first exception of `sink`.  Host tensors (tests) are copied synchronously."""

    def __init__(self, sink: Callable, device=None, slots: int = 3):
        if slots < 1:
            raise ValueError(f"slots = {slots}")
        self.sink = sink
        self.device = torch.device(device) if device is not None else None
        self.cuda = self.device is not None and self.device.type == "cuda"
        if self.cuda and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.copy_stream = torch.cuda.Stream(self.device) if self.cuda else None
        self.bufs: List[Optional[torch.Tensor]] = [None] * slots
        self.free: "queue.Queue[int]" = queue.Queue()
        self.ready: "queue.Queue" = queue.Queue()
        for i in range(slots):
            self.free.put(i)
        self.error: Optional[BaseException] = None
        self.batches = 0
        self.bytes = 0
        self.thread = threading.Thread(target=self._writer, name="interflow-handoff", daemon=True)
        self.thread.start()

    def _writer(self):
        while True:
            item = self.ready.get()
            if item is None:
                return
            slot, units, host, done = item
            try:
                if done is not None:
                    done.synchronize()
                if self.error is None:
                    arr = host.numpy()
                    for j, u in enumerate(units):
                        self.sink(u, arr[j])
            except BaseException as e:           # noqa: BLE001 -- handed to the producer by put() / close()
                if self.error is None:
                    self.error = e
            finally:
                self.free.put(slot)

    def put(self, units, stacked: torch.Tensor):
        if self.error is not None:
            raise self.error
        assert stacked.shape[0] == len(units), (tuple(stacked.shape), len(units))
        stacked = stacked.contiguous()
        slot = self.free.get()                    # back-pressure: waits for the writer when every slot is in flight
        n = stacked.numel() * stacked.element_size()
        buf = self.bufs[slot]
        if buf is None or buf.numel() < n:
            key = (self.device.index if self.cuda else -1, slot)
            buf = _PINNED.get(key) if self.cuda else None
            if buf is None or buf.numel() < n:
                buf = torch.empty(max(n, 1), dtype=torch.uint8, pin_memory=self.cuda)
                if self.cuda:
                    _PINNED[key] = buf
            self.bufs[slot] = buf
        host = buf[:n].view(stacked.dtype).view(stacked.shape)
        done = None
        if stacked.is_cuda:
            with torch.cuda.device(stacked.device):
                produced = torch.cuda.Event()
                produced.record()                 # on the producer's current stream
                self.copy_stream.wait_event(produced)
                with torch.cuda.stream(self.copy_stream):
                    host.copy_(stacked, non_blocking=True)
                    done = torch.cuda.Event()
                    done.record()
                stacked.record_stream(self.copy_stream)
        else:
            host.copy_(stacked)
        self.batches += 1
        self.bytes += n
        self.ready.put((slot, list(units), host, done))

    def close(self):
        self.ready.put(None)
        self.thread.join()
        if self.error is not None:
            raise self.error

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is None:
            self.close()
        else:                                     # do not mask the producer's exception
            self.ready.put(None)
            self.thread.join()
        return False


@torch.inference_mode()
def generate(model, gt, lr_a, lr_b, weights: Optional[Sequence[float]] = None, tile: Optional[int] = None,
             batch: int = 32, rank: int = 0, world: int = 1, add_noise: bool = False,
             sink: Optional[Callable] = None, host_slots: int = 3):
    """gt, lr_a, lr_b: [N,3,H,W] float32 device tensors (aligned HR / LR images of one size, as RealSR-v2).
    tile=None: whole images, as the reference does.  tile=T: every image is cut into independent T x T tiles
    (border tiles keep the remainder; a tile's output differs from the whole-image output near its border, since
    each tile is zero-padded on its own).
    Returns (units, results): units = [(image, y, x, h, w)] owned by this rank, results[i] = uint8 array
    [n_weights, h, w, 3] of unit i, on the device.
    sink: a callable `sink(unit, array)` -- then every unit's uint8 [n_weights, h, w, 3] numpy array is handed to it on a
    writer thread through `HostHandoff` (asynchronous copies into `host_slots` pinned buffers, overlapping the next
    batch's kernels) instead of being kept on the device, and `results` is None."""
    weights = default_weights() if weights is None else list(weights)
    N, _, H, W = gt.shape
    tiles = [(0, 0, H, W)] if tile is None else shard.tile_origins(H, W, tile)
    units = [(n, y, x, h, w) for n in range(N) for (y, x, h, w) in tiles]
    own = [units[i] for i in shard.partition(len(units), world, rank)]
    results: List[Optional[torch.Tensor]] = [None] * len(own)
    by_shape = {}
    for i, u in enumerate(own):                       # batches need one tile size
        by_shape.setdefault((u[3], u[4]), []).append(i)
    handoff = HostHandoff(sink, gt.device, host_slots) if sink is not None else None
    try:
        for (h, w), idx in by_shape.items():
            for s in range(0, len(idx), batch):
                sel = idx[s:s + batch]

                def crop(t):
                    return torch.stack([t[own[i][0], :, own[i][1]:own[i][1] + h, own[i][2]:own[i][2] + w] for i in sel])
                outs = interpolate_batch(model, crop(gt), crop(lr_a), crop(lr_b), weights, add_noise)
                stacked = torch.stack(outs, dim=1)        # [b, n_weights, h, w, 3]
                if handoff is not None:
                    handoff.put([own[i] for i in sel], stacked)
                    continue
                for j, i in enumerate(sel):
                    results[i] = stacked[j]
    except BaseException:
        if handoff is not None:
            handoff.__exit__(RuntimeError, None, None)
        raise
    if handoff is not None:
        handoff.close()
        return own, None
    return own, results


def assemble(units, results, N, H, W, n_weights, device):
    """Paste unit results back into [n_weights, N, H, W, 3] uint8 images (single writer / tests)."""
    out = torch.zeros(n_weights, N, H, W, 3, dtype=torch.uint8, device=device)
    for (n, y, x, h, w), r in zip(units, results):
        out[:, n, y:y + h, x:x + w] = r.to(device)
    return out
