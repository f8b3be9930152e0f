"""Experiment helper: time of fFeatures' last conv (conv3w F mode) and of the bulk conv1 launches inside ifl_cond_prepare
(16 steps per launch, batch 32 of 160x160), for variant builds of conv3w.hip (IFL_LIB_PATH)."""
import sys
import torch
sys.path.insert(0, ".")
from interflow_amd import hip, synth
from interflow_amd.flow_model import InterpFlow

dev = torch.device("cuda", 0)
sd = synth.synth_interpflow_state([1.0, 4.0], seed=0)
model = InterpFlow([1.0, 4.0])
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
model.eval().to(dev)
B, P = 32, 160
gen = torch.Generator(device=dev).manual_seed(1)
feats = {k: torch.rand(B, 320, P // d, P // d, device=dev, generator=gen) - 0.5 for k, d in (("fea_up4", 1), ("fea_up2", 2), ("fea_up1", 4))}
fu = model.flow_upsampler
with torch.no_grad():
    for _ in range(2):
        fu.prepare_cond(feats, B, P, P)
    torch.cuda.synchronize()
    hip.prof_enable([hip.K_CONV3_F, hip.K_CONV1_FT_L1])
    for _ in range(5):
        fu.prepare_cond(feats, B, P, P)
    torch.cuda.synchronize()
    ms, n = hip.prof_read(hip.K_CONV3_F)
    ms1, n1 = hip.prof_read(hip.K_CONV1_FT_L1)
    print("conv3f: %d launches (level 1 and level 2 alternate), %.3f ms per prepare; conv1 L1 %.3f ms per launch" % (n, ms / 5, ms1 / max(n1, 1)))
