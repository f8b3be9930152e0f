import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from interflow_amd import hip, synth
from interflow_amd.flow_model import InterpFlow
from tests._stress import stress_fixture, stress_state, _errs
for seed in (1, 2):
    g = stress_fixture("tests/golden", seed=seed)
    dev = torch.device("cuda:0")
    m = InterpFlow([2, 4]); m.load_state_dict({k: torch.from_numpy(v) for k, v in stress_state(g).items()}); m = m.eval().to(dev)
    cases = [("s160", 1, 160, 53, 54)] if seed == 2 else [("s32", 2, 32, 31, 32), ("s160", 2, 160, 33, 34)]
    with torch.no_grad():
        for tag, B, H, xs, fs in cases:
            x = torch.from_numpy(synth.synth_images(xs, B, H, H)).to(dev)
            feats = {k: torch.from_numpy(v).to(dev) for k, v in synth.synth_features(fs, B, H, H, rms=float(g["ft_rms"])).items()}
            n = _errs(g[f"{tag}_xr"], g[f"{tag}_xr64"])
            print(f"seed {seed} {tag}: reference's own x(ref z) vs float64 {n[0]:.2e} {n[1]:.2e}; own round trip {_errs(g[f'{tag}_xr'], x.cpu().numpy())}")
            for algo in ("default", 2, 1, 0):
                if algo == "default":
                    hip.set_conv1_algo(2); hip.set_conv1_tail(int(os.environ.get("EXP_TAIL", "4")), 1)
                else:
                    hip.set_conv1_algo(algo); hip.set_conv1_tail(0)
                xr, _ = m(torch.from_numpy(g[f"{tag}_z"]).to(dev), x, reverse=True, lr_enc=feats, add_gt_noise=False, test_input_size=H)
                e = _errs(xr.cpu().numpy(), g[f"{tag}_xr64"])
                err = np.abs(xr.cpu().numpy().astype(np.float64) - g[f"{tag}_xr64"])
                print(f"   algo {algo!s:8s}: vs float64 {e[0]:.2e} {e[1]:.2e}  = {e[0]/n[0]:.2f}x / {e[1]/n[1]:.2f}x the reference's; rms err {np.sqrt((err**2).mean()):.2e} (reference rms {np.sqrt(((g[f'{tag}_xr'].astype(np.float64)-g[f'{tag}_xr64'])**2).mean()):.2e})")
