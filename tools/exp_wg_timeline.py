"""Experiment helper (needs a library built with -DIFL_EXP_TIME): per-workgroup timestamps of the Winograd bulk kernel.
Runs ifl_cond_prepare alone (phase 1 only) and prints phase durations and the gap between consecutive workgroups on a CU."""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from interflow_amd import hip, synth
from interflow_amd.flow_model import InterpFlow

SCALES, PATCH, B = [1.0, 4.0], 160, 32
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
L = hip.lib()
n = 8 * 65536
buf = np.zeros(n, np.uint64)
L.ifl_dbg_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = L.ifl_dbg_read(buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
nwg = 12800
t = buf[:nwg * 16].reshape(nwg, 2, 8).astype(np.int64)
ids = t[:, 0, 7]
xcc = (ids >> 32) & 0xF
hw = ids & 0xFFFFFFFF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
t0 = t[:, :, 0].min()
ns = lambda a: (a * 10.0)          # 100 MHz ticks -> ns
for w, name in ((0, "fAffine wave"), (1, "fFeatures wave")):
    d = t[:, w, :6]
    print(name, "median ns: prologue %.0f  loop %.0f  wait-barrier %.0f  exchange %.0f  finish %.0f  total %.0f" % tuple(
        np.median(ns(x)) for x in (d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2], d[:, 4] - d[:, 3], d[:, 5] - d[:, 4], d[:, 5] - d[:, 0])))
d = t[:, 1, :]
if (d[:, 6] > 0).all():
    print("fFeatures finish split, median ns: ActNorm1 %.0f  1x1 MFMA %.0f  ActNorm2+stores %.0f" % (
        np.median(ns(d[:, 6] - d[:, 4])), np.median(ns(d[:, 7] - d[:, 6])), np.median(ns(d[:, 5] - d[:, 7]))))
start = t[:, :, 0].min(axis=1); end = t[:, :, 5].max(axis=1)
gaps, per_cu = [], []
for c in np.unique(cuid):
    idx = np.where(cuid == c)[0]
    o = idx[np.argsort(start[idx])]
    per_cu.append(len(o))
    gaps.extend(ns(start[o][1:] - end[o][:-1]))
gaps = np.array(gaps)
print("distinct CUs %d, WGs per CU min/median/max %d/%d/%d" % (len(per_cu), min(per_cu), int(np.median(per_cu)), max(per_cu)))
print("gap between WGs on a CU, ns: median %.0f mean %.0f p90 %.0f min %.0f" % (np.median(gaps), gaps.mean(), np.percentile(gaps, 90), gaps.min()))
print("kernel span us %.1f ; sum over a CU of WG totals (median) us %.1f" % (ns(end.max() - start.min()) / 1e3, np.median([ns((end - start)[cuid == c].sum()) for c in np.unique(cuid)]) / 1e3))
