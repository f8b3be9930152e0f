#!/usr/bin/env python
"""Per-kernel roofline table (BASELINE.json north_star: "rocprof roofline fractions reported for every kernel").

Times every kernel of the path alone on one stream with HIP events, at the benchmark's size (batch 32 of 160x160
patches) and, for the bandwidth-bound kernels, at a size past the 256 MiB Infinity Cache, and prices it against its
bound: algorithmic bytes / duration vs 8 TB/s HBM, or algorithmic FLOP / duration vs 157.3 TFLOP/s fp32 MFMA
(MI355X_MICROARCH.md).  Algorithmic = what the reference's operator must move or compute, not what the kernel executes.

  python tools/kernel_roofline.py [--out gpurun_out/kernel_roofline.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from interflow_amd import hip, synth  # noqa: E402
from interflow_amd.flow_model import InterpFlow  # noqa: E402

HBM, MFMA = 8000.0, 157.3      # GB/s, TFLOP/s
HBM_ACHIEVABLE = 6290.0        # GB/s: float4 copy measured on MI355X (MI355X_MICROARCH.md, "HBM3E peak BW")


def ev_time(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3          # seconds per call


def raw_time(call, reps=200, warm=10):
    """Back-to-back launches through the C ABI on preallocated buffers (no Python-side allocation between launches, so the
    GPU, not the host, paces the loop even for kernels of a few microseconds)."""
    for _ in range(warm):
        call()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        call()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def lib_time(kid, fn, reps=5, warm=2):
    """Mean duration of the launches of library kernel id `kid` inside fn() (event pairs inside the library)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    hip.prof_enable([kid])
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms, n = hip.prof_read(kid)
    hip.prof_enable([])
    return ms * 1e-3 / max(n, 1), n // reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "kernel_roofline.json"))
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    hip.set_schedule(aux_stream=0, groups=1)                 # kernels strictly one after another
    model = InterpFlow([2, 4])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_interpflow_state([2, 4], seed=0).items()})
    model.eval().to(dev)
    fu = model.flow_upsampler
    rows = []

    def add(kernel, what, bound, work, secs, size):
        if bound == "hbm":
            ach, peak, unit = work / secs / 1e9, HBM, "GB/s"
        else:
            ach, peak, unit = work / secs / 1e12, MFMA, "TFLOP/s"
        rows.append({"kernel": kernel, "replaces": what, "bound": bound, "size": size, "us": secs * 1e6,
                     "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                     "frac_of_achievable": ach / HBM_ACHIEVABLE if bound == "hbm" else None})

    g = torch.Generator(device=dev).manual_seed(3)
    with torch.inference_mode():
        for B in (32, 1024):
            size = f"B={B} 160x160"
            x = torch.rand(B, 3, 160, 160, device=dev, generator=g)
            n = x.numel()
            L, st, P = hip.lib(), hip._stream(), hip._ptr
            y = hip.squeeze2d(x)
            s = raw_time(lambda: L.ifl_squeeze2d(P(x), P(y), B, 3, 160, 160, 2, st))
            add("squeeze2_vec_kernel", "squeeze2d flow.py:163-176", "hbm", 2 * 4 * n, s, size)
            xo = torch.empty_like(x)
            s = raw_time(lambda: L.ifl_unsqueeze2d(P(y), P(xo), B, 12, 80, 80, 2, st))
            add("squeeze2_vec_kernel (reverse)", "unsqueeze2d flow.py:179-193", "hbm", 2 * 4 * n, s, size)
            q8 = hip.quantize_u8_nhwc(x)
            s = raw_time(lambda: L.ifl_quantize_u8_nhwc(P(x), P(q8), B, 3, 160, 160, st))
            add("quantize_u8_kernel", "clip*255->uint8 NHWC generate_images.py:89-91", "hbm", 5 * n, s, size)
            z = torch.randn(B, 48, 40, 40, device=dev, generator=g)
            z2 = torch.randn(B, 48, 40, 40, device=dev, generator=g)
            zo, wa, wb = torch.empty_like(z), torch.full((B,), 0.3, device=dev), torch.full((B,), 0.7, device=dev)
            s = raw_time(lambda: L.ifl_lerp(P(z), P(z2), P(wa), P(wb), P(zo), B, 48 * 40 * 40, st))
            add("lerp_kernel", "latent interpolation flow_model.py:198", "hbm", 3 * 4 * z.numel(), s, size)
            scr = torch.empty(B * L.ifl_logp_scratch_floats(1600), device=dev)
            lo = torch.empty(B, device=dev)
            s = raw_time(lambda: L.ifl_gaussian_logp(P(z), None, None, 0.0, P(scr), P(lo), B, 48, 40, 40, st))
            add("logp_kernel + ld_finalize (standard normal)", "Gaussian.logp flow.py:158-161", "hbm", 4 * z.numel(), s, size)
            cov = model.cov_shift_1 @ model.cov_shift_1.T
            ci, mu = torch.inverse(cov).contiguous(), (model.mean_shift_1 * torch.ones(48, device=dev)).contiguous()
            s = raw_time(lambda: L.ifl_gaussian_logp(P(z), P(mu), P(ci), 0.0, P(scr), P(lo), B, 48, 40, 40, st))
            add("logp48_mfma_kernel + ld_finalize (full covariance)", "Gaussian.logp flow.py:146-157", "hbm", 4 * z.numel(), s, size)
            if B > 256:
                continue
            # one FlowStep without coupling: the fused ActNorm + 1x1 conv kernel, timed inside the library
            for li, C, hw in ((1, 12, 80), (20, 48, 40)):
                step = fu.layers[li]
                zz = torch.randn(B, C, hw, hw, device=dev, generator=g)
                s, _ = lib_time(hip.K_ANIC, lambda: step(zz, None, reverse=False))
                add(f"anic_kernel<{C}>", "ActNorm2d + InvertibleConv1x1 FlowActNorms.py:87-92, Permutations.py:114-118",
                    "hbm", 2 * 4 * zz.numel(), s, f"B={B} {C}x{hw}x{hw}")
        # the coupling-network kernels inside a whole encode / decode at the benchmark size
        B = 32
        x = torch.rand(B, 3, 160, 160, device=dev, generator=g)
        feats = {k: (torch.rand(B, 320, 160 // d, 160 // d, device=dev, generator=g) * 2 - 1) * float(np.sqrt(3.0))
                 for k, d in (("fea_up4", 1), ("fea_up2", 2), ("fea_up1", 4))}
        z, _ = model(x, None, lr_enc=feats, add_gt_noise=False, calc_loss=False)
        enc = lambda: model(x, None, lr_enc=feats, add_gt_noise=False, calc_loss=False)
        dec = lambda: model(z, x, reverse=True, lr_enc=feats, add_gt_noise=False)
        both = lambda: (enc(), dec())
        px = {1: 6400, 2: 1600}
        size = "B=32 160x160, one pass"
        for algo, macs, kn, note in ((hip.CONV1_WINOGRAD4, 2 * 64 * 9 * 320 // 4 + 64 * 64, "conv1_ft_wino4_kernel",
                                      "Winograd F(4x4,3x3), default; MFMA multiply-adds issued"),
                                     (hip.CONV1_WINOGRAD, 2 * 64 * 4 * 320 + 64 * 64, "conv1_ft_wino_kernel",
                                      "Winograd F(2x2,3x3); MFMA multiply-adds issued"),
                                     (hip.CONV1_DIRECT, 2 * 64 * 9 * 320 + 64 * 64, "conv1_ft_kernel", "direct implicit GEMM")):
            hip.set_conv1_algo(algo)
            for kid, lvl in ((hip.K_CONV1_FT_L1, 1), (hip.K_CONV1_FT_L2, 2)):
                s, n = lib_time(kid, enc)
                add(f"{kn}<{lvl}>", "first 3x3 conv of fFeatures+fAffine over ft, fFeatures 1x1 (flow.py:25-64); " + note,
                    "mfma", 2.0 * macs * px[lvl] * B * 16 / n, s, f"B=32, level {lvl}, {16 // n} steps per launch")
        hip.set_conv1_algo(hip.CONV1_WINOGRAD4)
        s, n = lib_time(hip.K_CONV1_FINISH, enc)
        flop = 2.0 * sum((64 * 9 * (C // 2) + 64 * 64) * px[l] for l, C in ((1, 12), (2, 48))) * B * 16 / n
        add("conv1_finish_kernel", "fAffine z1 taps + ActNorm/ReLU/1x1 (Ablation.py:228-233)", "mfma", flop, s,
            "B=32, mean of the two levels")
        s, n = lib_time(hip.K_CONV3_F, enc)
        flop = 2.0 * sum(64 * 9 * 2 * C * px[l] for l, C in ((1, 12), (2, 48))) * B * 16 / n
        add("conv3w_kernel<F> (Winograd F(2x2,3x3); direct-form FLOPs)", "fFeatures' Conv2dZeros 3x3 + sigmoid -> shift/scale slabs, sum log scale (flow.py:67-82, "
            "Ablation.py:219-240)", "mfma", flop, s, "B=32, mean of the two levels")
        for kid, fn, name, launches in ((hip.K_CONV3_ENC, enc, "conv3w_kernel<ENC> (encode: fAffine conv + both couplings; direct-form FLOPs)", 1),
                                        (hip.K_CONV3_DEC, dec, "fapply_kernel + conv3w_kernel<DEC> (decode, per step; direct-form FLOPs)", 2)):
            s, n = lib_time(kid, fn)
            flop = 2.0 * sum(64 * 9 * C * px[l] for l, C in ((1, 12), (2, 48))) * B * 16 * launches / n
            add(name, "fAffine's Conv2dZeros 3x3 + sigmoid + affine coupling(s) + sum log scale (flow.py:67-82, Ablation.py:167-240)",
                "mfma", flop, s * launches, "B=32, mean of the two levels")
        s = ev_time(both, reps=5, warm=2)
        add("whole path (encode + decode), one stream", "InterpFlow.forward both directions", "mfma", 213.1e9 * B, s, "B=32")
    hip.set_schedule(aux_stream=1, groups=2)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out, "w"), indent=1)
    md = ["| kernel | replaces | size | µs | achieved | peak | frac | frac of achievable (6.29 TB/s) | bound |", "|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        md.append("| `%s` | %s | %s | %.1f | %.1f %s | %.1f | %.3f | %s | %s |" % (
            r["kernel"], r["replaces"], r["size"], r["us"], r["achieved"], r["unit"], r["peak"], r["frac"],
            ("%.3f" % r["frac_of_achievable"]) if r["frac_of_achievable"] is not None else "", r["bound"]))
    open(os.path.splitext(args.out)[0] + ".md", "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
