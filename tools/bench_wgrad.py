#!/usr/bin/env python
"""Times ifl_conv2d_wgrad / ifl_conv2d alone on the shapes of the training step (HIP events on the launch stream).
  python tools/bench_wgrad.py [--batch 8]        (IFL_LIB_PATH selects an experiment build)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interflow_amd import hip  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = hip.lib()
    B = args.batch
    for (h, w, cin, cout, taps) in ((80, 80, 320, 64, 9), (40, 40, 320, 64, 9), (80, 80, 32, 64, 9), (40, 40, 64, 64, 9), (80, 80, 64, 32, 9),
                                    (40, 40, 64, 96, 9), (80, 80, 64, 64, 1)):
        x = torch.randn(B, h + 2, w + 2, cin, device=dev)
        dy = torch.randn(B, h + 2, w + 2, cout, device=dev)
        dW = torch.zeros(cout, cin, taps, device=dev)

        def run():
            hip.check(L.ifl_conv2d_wgrad(hip._ptr(x), cin, 0, cin, hip._ptr(dy), cout, 0, cout, hip._ptr(dW), cin, 0, taps, B, h, w, hip._stream()), "wgrad")
        us = timed(run)
        fl = 2.0 * B * h * w * cin * cout * taps
        print(f"wgrad B={B} {h}x{w} cin={cin} cout={cout} taps={taps}: {us:8.1f} us  {fl / us * 1e-6:6.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
