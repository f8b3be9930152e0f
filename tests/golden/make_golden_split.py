"""Golden fixture for Split2d (Split.py:13-125), produced by running the REFERENCE's module on
CPU.  Same conventions as make_golden.py: the reference is imported, never copied; parameters and inputs are seeded here
and stored with the outputs.

usage: python tests/golden/make_golden_split.py   ->  tests/golden/split_seed0.npz
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
    from models.modules.Split import Split2d
    out = {}
    rng = np.random.default_rng(99)
    with torch.no_grad():
        for tag, C, cond, logs_eps, hw in (("a", 48, 320, 0.0, (3, 5)), ("b", 24, 0, 0.1, (4, 4))):
            m = Split2d(num_channels=C, logs_eps=logs_eps, cond_channels=cond, consume_ratio=0.5)
            cons = C // 2
            prm = {"conv.weight": (rng.standard_normal((2 * cons, C - cons + cond, 3, 3)) * (0.004 if cond else 0.02)).astype(np.float32),
                   "conv.bias": (rng.standard_normal(2 * cons) * 0.1).astype(np.float32),
                   "conv.logs": (rng.standard_normal((2 * cons, 1, 1)) * 0.1).astype(np.float32),
                   "I": np.eye(cons, dtype=np.float32),
                   "mean_shift": (rng.standard_normal(cons) * 0.2).astype(np.float32),
                   "cov_shift": (np.eye(cons) * 0.7 + rng.standard_normal((cons, cons)) * 0.05).astype(np.float32)}
            m.load_state_dict({k: t(v) for k, v in prm.items()}, strict=True)
            m.eval()
            B = 4
            x = rng.standard_normal((B, C, *hw)).astype(np.float32)
            ft = rng.standard_normal((B, cond, *hw)).astype(np.float32) if cond else None
            y = np.array([0, 1, 1, 0], np.float32)
            l0 = rng.standard_normal(B).astype(np.float32)
            z1, ld, eps = m(t(x), logdet=t(l0), reverse=False, ft=None if ft is None else t(ft), y_onehot=t(y))
            zr, ldr = m(z1, logdet=ld, reverse=True, eps=eps, ft=None if ft is None else t(ft), y_onehot=t(y))
            for k, v in prm.items():
                out[f"{tag}.{k}"] = v
            out[f"{tag}.x"] = x; out[f"{tag}.y"] = y; out[f"{tag}.ld"] = l0; out[f"{tag}.logs_eps"] = np.float32(logs_eps)
            if ft is not None:
                out[f"{tag}.ft"] = ft
            out[f"{tag}.z1"] = z1.numpy(); out[f"{tag}.ld_f"] = ld.numpy(); out[f"{tag}.eps"] = eps.numpy()
            out[f"{tag}.zr"] = zr.numpy(); out[f"{tag}.ld_r"] = ldr.numpy()
            print(tag, float(np.abs(zr.numpy() - x).max()), ld.numpy(), ldr.numpy() - l0)
    np.savez_compressed(os.path.join(HERE, "split_seed0.npz"), **out)


if __name__ == "__main__":
    main()
