#!/usr/bin/env python
"""Times ifl_wino4_kmajor (both modes) on the training step's shapes.  python tools/bench_kmajor.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interflow_amd import hip, train_fused
from tools.bench_wgrad import timed
L = train_fused._lib()
dev = torch.device("cuda:0")
for (B, h, w, C, planes) in ((8, 80, 80, 64, 32), (8, 40, 40, 64, 32), (8, 80, 80, 320, 1)):
    K = int(L.ifl_wino4_tiles(B, h, w))
    src = torch.randn(planes, B, h + 2, w + 2, C, device=dev)
    out = torch.empty(36, planes * C, K, device=dev)
    for mode in (0, 1):
        if mode == 1 and C != 64:
            continue
        us = timed(lambda: hip.check(L.ifl_wino4_kmajor(mode, hip._ptr(src), C, planes, src[0].numel(), hip._ptr(out), planes * C, 0, B, h, w, hip._stream()), "kmajor"))
        gb = (out.numel() + src.numel()) * 4 / 1e9
        print(f"kmajor mode {mode} B={B} {h}x{w} C={C} planes={planes}: {us:8.1f} us  {gb / us * 1e6 / 1e3:5.2f} TB/s (read once + write)", flush=True)
