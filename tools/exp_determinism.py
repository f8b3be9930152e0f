"""Race screen for the fused path: 40 encode + decode passes of batch 32 under the default multi-stream schedule must be
bit-identical (fixed-order reductions, no atomics on the inference path)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interflow_amd import synth
from interflow_amd.flow_model import InterpFlow
dev = torch.device("cuda", 0)
sd = synth.synth_interpflow_state([2, 4], seed=0)
m = InterpFlow([2, 4]); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True); m.eval().to(dev)
g = torch.Generator(device=dev).manual_seed(5)
B = 32
x = torch.rand(B, 3, 160, 160, device=dev, generator=g)
feats = {k: torch.rand(B, 320, 160 // d, 160 // d, device=dev, generator=g) * 2 - 1 for k, d in (("fea_up4", 1), ("fea_up2", 2), ("fea_up1", 4))}
ref = None; bad = 0
with torch.no_grad():
    for i in range(40):
        z, ld = m(x, None, lr_enc=feats, add_gt_noise=False, calc_loss=False)
        xr, _ = m(z, x, reverse=True, lr_enc=feats, add_gt_noise=False)
        cur = (z.clone(), ld.clone(), xr.clone())
        if ref is None: ref = cur
        elif not all(torch.equal(a, b) for a, b in zip(ref, cur)): bad += 1
print("runs differing from the first:", bad, "finite:", bool(torch.isfinite(z).all()))
