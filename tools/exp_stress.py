"""Error table on the stress fixture: every conv1 algorithm, z / x against the reference and against the exact answer."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from interflow_amd import hip, synth
from interflow_amd.flow_model import InterpFlow
from tests._stress import stress_fixture, stress_state

def errs(got, want):
    got = got.detach().cpu().double().numpy(); want = np.asarray(want, dtype=np.float64)
    e = np.abs(got - want); rms = np.sqrt((want ** 2).mean())
    return e.max() / np.abs(want).max(), (e / np.maximum(np.abs(want), rms)).max()

g = stress_fixture(os.path.join(ROOT, "tests", "golden"))
dev = torch.device("cuda:0")
m = InterpFlow([2, 4]); m.load_state_dict({k: torch.from_numpy(v) for k, v in stress_state(g).items()}); m = m.eval().to(dev)
with torch.no_grad():
    for tag, B, H, xs, fs in (("s32", 2, 32, 31, 32), ("s160", 2, 160, 33, 34)):
        x = torch.from_numpy(synth.synth_images(xs, B, H, H)).to(dev)
        feats = {k: torch.from_numpy(v).to(dev) for k, v in synth.synth_features(fs, B, H, H, rms=float(g["ft_rms"])).items()}
        print(tag, "reference x_rec vs true x:", errs(torch.from_numpy(g[f"{tag}_xr"]), x.cpu().numpy()))
        for algo in (0, 1, 2):
            hip.set_conv1_algo(algo)
            z, ld = m(x, None, lr_enc=feats, add_gt_noise=False, calc_loss=False)
            xr, ldr = m(torch.from_numpy(g[f"{tag}_z"]).to(dev), x, reverse=True, lr_enc=feats, add_gt_noise=False, test_input_size=H)
            xo, _ = m(z, x, reverse=True, lr_enc=feats, add_gt_noise=False, test_input_size=H)
            print(f"  algo {algo}: z {errs(z, g[f'{tag}_z'])} ld {errs(ld, g[f'{tag}_ld'])} x(ref z) vs ref x_rec {errs(xr, g[f'{tag}_xr'])} "
                  f"vs true {errs(xr, x.cpu().numpy())} own round trip {errs(xo, x.cpu().numpy())}")
