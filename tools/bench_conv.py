#!/usr/bin/env python
"""Times ifl_conv2d alone on the shapes of the training step's coupling networks (HIP events on the launch stream).
  python tools/bench_conv.py        (IFL_LIB_PATH selects an experiment build)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interflow_amd import hip  # noqa: E402
from tools.bench_wgrad import timed  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    L = hip.lib()
    # (images, h, w, cin, cout, taps): one sample group (4 images) of the two flow levels, and the whole batch
    for (B, h, w, cin, cout, taps) in ((4, 40, 40, 64, 64, 9), (4, 40, 40, 32, 64, 9), (4, 40, 40, 64, 32, 9), (4, 40, 40, 64, 64, 1),
                                       (4, 80, 80, 64, 64, 9), (4, 80, 80, 32, 64, 9), (4, 80, 80, 64, 32, 9), (4, 80, 80, 64, 64, 1),
                                       (8, 40, 40, 64, 64, 9),
                                       # the conditioning encoder's dense blocks at batch 8
                                       (8, 40, 40, 64, 32, 9), (8, 40, 40, 96, 32, 9), (8, 40, 40, 128, 32, 9), (8, 40, 40, 160, 32, 9), (8, 40, 40, 192, 64, 9),
                                       (1, 40, 40, 64, 32, 9), (1, 40, 40, 192, 64, 9)):
        x = torch.randn(B, h + 2, w + 2, cin, device=dev)
        y = torch.zeros(B, h + 2, w + 2, cout, device=dev)
        wp = torch.randn(cout * cin * taps, device=dev)
        d = hip.IflConvDesc()
        d.nsrc = 1
        d.src[0], d.src_stride[0], d.src_off[0], d.src_ch[0] = x.data_ptr(), cin, 0, cin
        d.w, d.bias, d.out = wp.data_ptr(), None, y.data_ptr()
        d.out_stride, d.out_off, d.cout, d.taps, d.act = cout, 0, cout, taps, 0
        d.alpha = d.beta = 1.0

        def run():
            hip.check(L.ifl_conv2d(ctypes.byref(d), B, h, w, hip._stream()), "conv2d")
        us = timed(run, 50)
        fl = 2.0 * B * h * w * cin * cout * taps
        print(f"conv2d B={B} {h}x{w} cin={cin} cout={cout} taps={taps}: {us:8.1f} us  {fl / us * 1e-6:6.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
