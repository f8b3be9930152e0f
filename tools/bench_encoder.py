#!/usr/bin/env python
"""Conditioning encoder (rrdbPreprocessing) timing: HIP path (ifl_rrdb_encode) vs the PyTorch-ROCm modules, B x 160x160."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from interflow_amd import synth  # noqa: E402
from interflow_amd.flow_model import InterpFlow  # noqa: E402

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda:0")
    model = InterpFlow([2, 4])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_interpflow_state([2, 4], seed=0).items()})
    model.eval().to(dev)
    x = torch.rand(B, 3, 160, 160, device=dev)
    def timeit(fn, reps=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    with torch.no_grad():
        t_hip = timeit(lambda: model.arch.rrdbPreprocessing(x))
        model.arch._needs_autograd = lambda: True            # the differentiable path (HIP convs + ATen glue), forward only
        t_torch = timeit(lambda: model.arch.rrdbPreprocessing(x))
    gflop = 57.5 * B
    print(json.dumps({"batch": B, "hip_ms": t_hip * 1e3, "hip_tflops": gflop / t_hip / 1e3, "train_path_fwd_ms": t_torch * 1e3,
                      "train_path_fwd_tflops": gflop / t_torch / 1e3, "algorithmic_gflop_per_image": 57.5}))

if __name__ == "__main__":
    main()
