"""Golden fixture for multi-level flows with Split2d layers (FlowUpsamplerNet.py:94-105,149-167,243-269,292-311; SURVEY.md
section 8(f) row 4), produced by running the REFERENCE's FlowUpsamplerNet on CPU with the shipped `opt` changed to L=3 -- the
configurations a general SRFlow / DeFlow checkpoint has.  Same conventions as make_golden.py: the reference is imported,
never copied; the weights come from the build's synthesiser (interflow_amd/synth.py, `load_state_dict(strict=True)`), so only
seeds, inputs and outputs are stored.

Two configurations (K=2 conditional steps and one no-coupling step per level):
  a  split.correct_splits=False: one Split2d after level 1; channels 12 -> 6 | 24 | 96 (no split after level 2)
  b  split.correct_splits=True : Split2d after levels 1 and 2; channels 12 -> 6 | 24 -> 12 | 48
Each: encode with an `epses` list (the reference returns [eps..., z]), decode from that list, encode without a list (plain z),
with test_input_size given; labels 0/1 mixed (Split2d's two class priors).

usage: python tests/golden/make_golden_general.py   ->  tests/golden/general_seed5.npz
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, t, synth  # noqa: E402

SEED, K, EXTRA, L = 5, 2, 1, 3


def general_opt(correct_splits):
    return {'scale': 4,
            'network_G': {'in_nc': 3, 'out_nc': 3, 'nf': 64, 'nb': 23, 'upscale': 4, 'train_RRDB': True, 'scale': 4,
                          'flow': {'K': K, 'L': L, 'noInitialInj': True, 'LU_decomposed': True,
                                   'coupling': 'CondAffineSeparatedAndCond', 'additionalFlowNoAffine': EXTRA,
                                   'split': {'enable': True, 'correct_splits': correct_splits}, 'fea_up0': True,
                                   'stackRRDB': {'blocks': [1, 8, 15, 22], 'concat': True},
                                   'CondAffineSeparatedAndCond': {'eps': 0.001, 'multReverse': True}}}}


def main():
    torch.manual_seed(0)
    np.random.seed(0)
    import_reference()
    from models.modules.FlowUpsamplerNet import FlowUpsamplerNet
    out = {}
    B, H, W = 3, 32, 48
    y = np.array([0, 1, 1], np.float32)
    xin = synth.synth_images(31, B, H, W)
    feats_np = synth.synth_features(32, B, H, W, keys=("fea_up4", "fea_up2", "fea_up1", "fea_up0"))
    out["y"] = y
    with torch.no_grad():
        for tag, cs in (("a", False), ("b", True)):
            opt = general_opt(cs)
            fu = FlowUpsamplerNet((160, 160, 3), 64, K, flow_coupling='CondAffineSeparatedAndCond', LU_decomposed=True,
                                  opt=copy.deepcopy(opt))
            sd = synth.synth_flow_state(SEED, L=L, K=K, extra=EXTRA, zeros_std=0.05, split={"correct_splits": cs})
            fu.load_state_dict({k: t(v) for k, v in sd.items()}, strict=True)
            fu.eval()
            feats = {k: t(v) for k, v in feats_np.items()}
            ld0 = torch.zeros(B)
            epses, ld = fu(gt=t(xin), rrdbResults=feats, epses=[], logdet=ld0, reverse=False, y_onehot=t(y), test_input_size=H)
            for i, e in enumerate(epses):
                out[f"{tag}.eps{i}"] = e.numpy()
            out[f"{tag}.n_eps"] = np.array(len(epses))
            out[f"{tag}.ld"] = ld.numpy()
            xr, ldr = fu(z=None, rrdbResults=feats, epses=epses, logdet=ld, reverse=True, y_onehot=t(y), test_input_size=H)
            out[f"{tag}.xr"], out[f"{tag}.ldr"] = xr.numpy(), ldr.numpy()
            z, ld2 = fu(gt=t(xin), rrdbResults=feats, logdet=ld0, reverse=False, y_onehot=t(y), test_input_size=H)
            out[f"{tag}.z"], out[f"{tag}.ld2"] = z.numpy(), ld2.numpy()
            out[f"{tag}.digest"] = np.array(synth.state_digest(sd))
            print(tag, [tuple(e.shape) for e in epses], "ld", ld.numpy(), "|x - xr|", float((xr - t(xin)).abs().max()),
                  "ldr", ldr.numpy(), "z std", float(z.std()))
    np.savez_compressed(os.path.join(HERE, "general_seed5.npz"), **out)


if __name__ == "__main__":
    main()
