"""Golden fixture for the non-LU InvertibleConv1x1 (Permutations.py:78-86), produced by running
the REFERENCE's FlowStep (flow_coupling='noCoupling', LU_decomposed=False) on CPU.  Same conventions as make_golden.py: the
reference is imported, never copied; parameters and inputs are seeded here and stored with the outputs.

usage: python tests/golden/make_golden_nonlu.py   ->  tests/golden/nonlu_seed0.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, t  # noqa: E402


def main():
    import_reference()
    from models.modules.FlowStep import FlowStep
    out = {}
    rng = np.random.default_rng(77)
    with torch.no_grad():
        for C, hw in ((12, (5, 7)), (48, (3, 4))):
            step = FlowStep(in_channels=C, hidden_channels=64, flow_permutation="invconv", flow_coupling="noCoupling",
                            LU_decomposed=False)
            q, _ = np.linalg.qr(rng.standard_normal((C, C)))
            w = (q * np.exp(rng.standard_normal(C) * 0.2)[None, :]).astype(np.float32)     # well conditioned, |det| != 1
            prm = {"actnorm.bias": (rng.standard_normal((1, C, 1, 1)) * 0.3).astype(np.float32),
                   "actnorm.logs": (rng.standard_normal((1, C, 1, 1)) * 0.2).astype(np.float32),
                   "invconv.weight": w}
            step.load_state_dict({k: t(v) for k, v in prm.items()}, strict=True)
            step.actnorm.inited = True
            step.eval()
            z = rng.standard_normal((2, C, *hw)).astype(np.float32)
            l0 = rng.standard_normal(2).astype(np.float32)
            zf, lf = step(t(z), logdet=t(l0), reverse=False)
            zr, lr = step(zf, logdet=lf, reverse=True)
            for k, v in prm.items():
                out[f"c{C}.{k}"] = v
            out[f"c{C}.z"] = z; out[f"c{C}.ld"] = l0
            out[f"c{C}.zf"] = zf.numpy(); out[f"c{C}.lf"] = lf.numpy()
            out[f"c{C}.zr"] = zr.numpy(); out[f"c{C}.lr"] = lr.numpy()
            print(C, float(np.abs(zr.numpy() - z).max()), lf.numpy(), lr.numpy() - l0)
    np.savez_compressed(os.path.join(HERE, "nonlu_seed0.npz"), **out)


if __name__ == "__main__":
    main()
