"""Experiment helper: time of the level-1 bf16 bulk launches at 512x512, batch 4 (ifl_cond_prepare with the bf16 flag), for
variant builds of conv1.hip (IFL_LIB_PATH)."""
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
B, P = 4, 512
gen = torch.Generator(device=dev).manual_seed(1)
feats = {k: torch.rand(B, 320, P // d, P // d, device=dev, generator=gen) - 0.5 for k, d in (("fea_up4", 1), ("fea_up2", 2), ("fea_up1", 4))}
fu = model.flow_upsampler
fu.conv_dtype = "bf16"
with torch.no_grad():
    for _ in range(2):
        fu.prepare_cond(feats, B, P, P)
    torch.cuda.synchronize()
    hip.prof_enable([hip.K_CONV1_FT_L1, hip.K_CONV1_FT_L2])
    for _ in range(5):
        fu.prepare_cond(feats, B, P, P)
    torch.cuda.synchronize()
    for name, k in (("L1", hip.K_CONV1_FT_L1), ("L2", hip.K_CONV1_FT_L2)):
        ms, n = hip.prof_read(k)
        print("%s: %d launches, %.3f ms each" % (name, n, ms / max(n, 1)))
