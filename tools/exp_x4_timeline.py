"""Experiment helper (needs tools/build_exp.sh and IFL_LIB_PATH=gpurun_exp/libinterflow_hip_exp.so): per-wave timestamps of the
Winograd F(4x4,3x3) bulk kernel.  Runs ifl_cond_prepare alone (phase 1 only) and prints phase durations per wave role."""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from interflow_amd import hip, synth
from interflow_amd.flow_model import InterpFlow

import os
SCALES, PATCH, B = [1.0, 4.0], 160, int(os.environ.get("EXP_B", "32"))
dev = torch.device("cuda", 0)
sd = synth.synth_interpflow_state(SCALES, seed=0)
model = InterpFlow(SCALES)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
model.eval().to(dev)
gen = torch.Generator(device=dev).manual_seed(1)
feats = {k: torch.rand(B, 320, PATCH // d, PATCH // d, device=dev, generator=gen) for k, d in (("fea_up4", 1), ("fea_up2", 2), ("fea_up1", 4))}
with torch.no_grad():
    for _ in range(3):
        model.flow_upsampler.prepare_cond(feats, B, PATCH, PATCH)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        model.flow_upsampler.prepare_cond(feats, B, PATCH, PATCH)
    e1.record(); torch.cuda.synchronize()
    print("prepare_cond: %.3f ms per call" % (e0.elapsed_time(e1) / 10))
L = hip.lib()
if not hasattr(L, "ifl_dbgx_read"):
    sys.exit(0)
nwg = 8192
buf = np.zeros(nwg * 8 * 8, np.uint64)
L.ifl_dbgx_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = L.ifl_dbgx_read(buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
t = buf.reshape(nwg, 8, 8).astype(np.int64)
LVL = int(os.environ.get("EXP_LVL", "1"))           # the library must be built with -DIFL_EXP_TIME_LVL=2 for level 2
tiles = B * (400 if LVL == 1 else 100)
nwg = min(nwg, 32 * ((tiles + 31) // 32))          # workgroups of one launch of ifl_cond_prepare at that level: 32 (network, step) pairs x groups
t = t[:nwg]
ns = lambda a: a * 10.0          # 100 MHz ticks -> ns
net = (np.arange(nwg) & 1)       # combo = (id & 7) + ...: net = combo & 1 = id & 1 when ncombo % 8 == 0
for nname, sel in (("fAffine", net == 0), ("fFeatures", net == 1)):
    for w in range(8):
        d = t[sel, w, :]
        loop_ns = np.median(ns(d[:, 2] - d[:, 1]))
        clk = np.median(d[:, 6])
        print("%-9s wave %d  ns: prologue %6.0f  loop %7.0f  out-transform %6.0f  finish %6.0f  total %7.0f | loop clk %8.0f (%.2f GHz)  barrier-wait clk %7.0f (%.0f%%)  staging clk %7.0f (%.0f%%)" % (
            nname, w, np.median(ns(d[:, 1] - d[:, 0])), loop_ns, np.median(ns(d[:, 3] - d[:, 2])), np.median(ns(d[:, 4] - d[:, 3])),
            np.median(ns(d[:, 4] - d[:, 0])), clk, clk / loop_ns, np.median(d[:, 5]), 100 * np.median(d[:, 5] / d[:, 6]),
            np.median(d[:, 7]), 100 * np.median(d[:, 7] / d[:, 6])))
