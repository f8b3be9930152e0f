"""Experiment: how long does the host take to issue one encode / decode of the bench's workload, and is it ahead of the device?
Prints per call the host time of the call and, right after it returns, whether the device still has work queued."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interflow_amd import synth  # noqa: E402
from interflow_amd.flow_model import InterpFlow  # noqa: E402

dev = torch.device("cuda", 0)
model = InterpFlow([2, 4])
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_interpflow_state([2, 4], seed=0).items()}, strict=True)
model.eval().to(dev)
B, P = 32, 160
x = torch.from_numpy(synth.synth_images(1000, B, P, P)).to(dev)
feats = {k: torch.from_numpy(v).to(dev) for k, v in synth.synth_features(2000, B, P, P, keys=("fea_up2", "fea_up1")).items()}
feats["fea_up4"] = torch.zeros(1, device=dev)
with torch.no_grad():
    for _ in range(3):
        z, ld = model(x, None, lr_enc=feats, add_gt_noise=False, calc_loss=False)
        xr, _ = model(z, x, reverse=True, lr_enc=feats, add_gt_noise=False)
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for it in range(6):
        ev = torch.cuda.Event()
        t0 = time.perf_counter()
        z, ld = model(x, None, lr_enc=feats, add_gt_noise=False, calc_loss=False)
        t1 = time.perf_counter()
        ev.record()
        busy1 = not ev.query()
        xr, _ = model(z, x, reverse=True, lr_enc=feats, add_gt_noise=False)
        t2 = time.perf_counter()
        ev2 = torch.cuda.Event(); ev2.record()
        busy2 = not ev2.query()
        print(f"iter {it}: encode call {1e3 * (t1 - t0):.2f} ms host (device still busy after: {busy1}), decode call {1e3 * (t2 - t1):.2f} ms host (busy after: {busy2})", flush=True)
    torch.cuda.synchronize()
    print(f"6 steps: {1e3 * (time.perf_counter() - t_all) / 6:.2f} ms per step")

# device time between the end of a pass and the end of an explicit conditioning load issued right after it (kernels of the
# load: ~0.43 ms at this size); more than that is the device idling between passes
from interflow_amd import hip  # noqa: E402
fu = model.flow_upsampler
with torch.no_grad():
    lv = fu._level_features(feats, P, None, False)
    ws = hip.Workspace.get(B, P, P, dev)
    evs = []
    for it in range(6):
        z, ld = model(x, None, lr_enc=feats, add_gt_noise=False, calc_loss=False)
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        torch.zeros_like(x[:, 0, 0, 0])
        fu._load_cond(ws, lv, B, P, P)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        xr, _ = model(z, x, reverse=True, lr_enc=feats, add_gt_noise=False)
        e2 = torch.cuda.Event(enable_timing=True); e2.record()
        evs.append((e0, e1, e2))
    torch.cuda.synchronize()
    for e0, e1, e2 in evs:
        print(f"end of encode -> conditioning loaded: {e0.elapsed_time(e1):.3f} ms; decode pass after it: {e1.elapsed_time(e2):.3f} ms")
