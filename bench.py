#!/usr/bin/env python
"""bench.py -- flow fwd+inv patches/sec (160x160, fp32) on N MI355X  (BASELINE.json metric).

A step = one pass of the hot path over one batch: `InterpFlow.forward` encode (x -> z, logdet) followed by the
decode (z -> x) of BASELINE.json configs[1]: batch 32 of synthetic 160x160 RGB patches per GPU, fp32,
conditioning features (`lr_enc`) precomputed, resident in HBM and shared by the two passes, `add_gt_noise=False`
(SURVEY.md section 8(d)).  Nothing is cached between steps: every step converts the conditioning maps to the
kernels' layout again and runs all 38 layers in both directions.

Multi-GPU (`torchrun --nproc-per-node N bench.py --gpus N`): one process per GPU, each with its own batch of 32
(weak scaling, no data-path collective); the barrier and the max-over-ranks time use RCCL.

Prints ONE JSON line on rank 0 with the throughput, the roofline of the dominant kernel (the coupling networks'
first conv, timed live with HIP events on the launch stream) and, at N=1, the CPU oracle timed on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from interflow_amd import hip, shard, synth  # noqa: E402
from interflow_amd.flow_model import InterpFlow  # noqa: E402

PATCH = 160
BATCH = 32
PEAK_F32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
SCALES = [2, 4]


def conv1_macs_per_pixel(C):
    """Algorithmic MACs per pixel of one conditional step's first two conv layers, both networks
    (SURVEY.md section 8(a) row a11): 3x3 (C/2+320 -> 64), 3x3 (320 -> 64), two 1x1 (64 -> 64)."""
    return 64 * 9 * (C // 2 + 320) + 64 * 9 * 320 + 2 * 64 * 64


def cpu_baseline(sd_np, seconds_budget=20.0):
    """The CPU oracle (port of the reference, pinned to it by tests/golden) on BASELINE.json configs[0]:
    4 patches, fp32, encode + decode with precomputed features, all host threads torch chooses."""
    from oracle import interflow_oracle as O
    sd = O.to_torch(sd_np)
    B = 4
    x = torch.from_numpy(synth.synth_images(1, B, PATCH, PATCH))
    feats = {k: torch.from_numpy(v) for k, v in synth.synth_features(2, B, PATCH, PATCH).items()}
    threads = min(16, os.cpu_count() or 1)         # the GPU box gives one GPU's job a 16-CPU share (nproc shows the whole machine)
    torch.set_num_threads(threads)

    def one():
        z, ld = O.flow_encode(x, feats, sd, "flow_upsampler.", torch.zeros(B))
        O.flow_decode(z, feats, sd, "flow_upsampler.", torch.zeros(B))
    with torch.no_grad():
        one()                                   # warm-up
        times = []
        t_end = time.perf_counter() + seconds_budget
        while len(times) < 10 and (time.perf_counter() < t_end or len(times) < 2):
            t0 = time.perf_counter()
            one()
            times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": B / med, "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": f"{len(times)} timed encode+decode passes of 4 patches 160x160 (BASELINE configs[0]) after 1 warm-up; median {med:.3f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)                 # 100 steps = 2.3 s of timed region
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH, help="patches per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", action="store_true",
                    help="after the timed region also measure the bf16 variant of the dominant convolution (never `value`)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary fields measured AFTER the timed region by default: `cond_reuse` (encode + decode with "
                         "one shared conditioning preparation), `serial_schedule` (the steps on one stream: the dominant kernel "
                         "alone) and `kernels_ms_per_step` (every kernel of one step timed)")
    ap.add_argument("--no-cross-pass-overlap", action="store_true",
                    help="every pass's bulk convolutions behind the whole previous pass (the plain stream order) instead of behind "
                         "its bulk convolutions only (hip.set_cross_pass_overlap: passes in a row on unchanged conditioning tensors)")
    ap.add_argument("--no-hires", action="store_true", help="skip the secondary `config4_512_bf16` measurement (512x512 patches, bf16 convolutions)")
    ap.add_argument("--breakdown", action="store_true", help="(kept for old command lines: the breakdown is on by default)")
    args = ap.parse_args()

    rank, local_rank, world = shard.dist_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torchrun --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback of the product path)")
    dev, backend = shard.rank_device()             # cuda:LOCAL_RANK, RCCL
    torch.cuda.set_device(dev)
    if args.no_cross_pass_overlap:
        hip.set_cross_pass_overlap(False)
    dist = shard.init_dist(backend, dev)            # RCCL; only the barrier and the max-over-ranks clock use it

    sd_np = synth.synth_interpflow_state(SCALES, seed=0)
    model = InterpFlow(SCALES)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    model.eval().to(dev)

    B = args.batch
    # the generators of the parity fixtures (interflow_amd/synth.py: U[0,1) images, zero-mean uniform features of rms 1), one seed
    # per rank; 'fea_up4' is only looked up by the decode (FlowUpsamplerNet.py:279-280), never read: a placeholder stands for it
    x = torch.from_numpy(synth.synth_images(1000 + rank, B, PATCH, PATCH)).to(dev)
    feats = {k: torch.from_numpy(v).to(dev)
             for k, v in synth.synth_features(2000 + rank, B, PATCH, PATCH, keys=("fea_up2", "fea_up1")).items()}
    feats["fea_up4"] = torch.zeros(1, device=dev)

    def step():
        z, ld = model(x, None, lr_enc=feats, add_gt_noise=False, calc_loss=False)
        xr, ldr = model(z, x, reverse=True, lr_enc=feats, add_gt_noise=False)
        return z, ld, xr, ldr

    def barrier():
        shard.barrier(dist, dev)

    prof, serial = {}, None
    with torch.no_grad():
        for _ in range(args.warmup):
            out = step()
        barrier()
        # only the dominant kernel is timed inside the timed region: an event pair around each of its launches (4 per
        # step by default), recorded on the stream the kernel is launched on (the library's auxiliary stream)
        hip.prof_enable([hip.K_CONV1_FT_L1])
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        barrier()
        elapsed = time.perf_counter() - t0
        dom_ms, dom_n = hip.prof_read(hip.K_CONV1_FT_L1)
        hip.prof_enable([])
        z, ld, xr, ldr = out
        err = float((xr - x).abs().max())
        if not os.environ.get("IFL_BENCH_TIMING_ONLY"):      # kernel timing experiments with deliberately wrong arithmetic
            assert np.isfinite(err) and err < 1e-3, f"decode(encode(x)) differs from x by {err}"
        secondary = not args.no_secondary
        sec_steps = min(args.steps, 20)
        if secondary:
            # Secondary figure (never `value`): the conditioning-only part of the coupling networks computed ONCE per
            # step (ifl_cond_prepare) and used by both the encode and the decode of that step, as
            # InterpFlow.weighted_interpolate_image / generate.py do.  Outputs are bit-identical.
            fu = model.flow_upsampler

            def step_reuse():
                prep = fu.prepare_cond(feats, B, PATCH, PATCH)
                z, ld = model(x, None, lr_enc=prep, add_gt_noise=False, calc_loss=False)
                return model(z, x, reverse=True, lr_enc=prep, add_gt_noise=False)
            step_reuse()
            barrier()
            t2 = time.perf_counter()
            for _ in range(sec_steps):
                xr2, _ = step_reuse()
            barrier()
            reuse_elapsed = shard.max_over_ranks(time.perf_counter() - t2, dist, dev)
            assert torch.equal(xr2, xr)
        bf16 = None
        if args.extras:
            # Secondary figure (never `value`): BASELINE.json configs[4]'s arithmetic on this workload -- the conditioning
            # part of the coupling networks' first conv and fFeatures' 1x1 on the bf16 matrix cores, fp32 elsewhere.
            model.flow_upsampler.conv_dtype = "bf16"
            step(); step()
            barrier()
            hip.prof_enable([hip.K_CONV1_FT_L1])
            t3 = time.perf_counter()
            for _ in range(args.steps):
                ob = step()
            barrier()
            bf_elapsed = shard.max_over_ranks(time.perf_counter() - t3, dist, dev)
            b_ms, b_n = hip.prof_read(hip.K_CONV1_FT_L1)
            hip.prof_enable([])
            model.flow_upsampler.conv_dtype = "fp32"
            bf16 = {"elapsed": bf_elapsed, "dom_ms": b_ms, "dom_n": b_n,
                    "round_trip_err": float((ob[2] - x).abs().max()),
                    "z_rel_err_vs_fp32": float((ob[0] - z).abs().max() / z.abs().max())}
        hires = None
        if secondary and not args.no_hires:
            # Secondary figure (never `value`): BASELINE.json configs[4] -- 512x512 patches, the coupling networks' z-independent
            # convolutions on the bf16 matrix cores (conv_dtype "bf16": fp32 ActNorm / log-det / couplings / 1x1 maps).
            HB, HP, hsteps = 4, 512, 5
            gen = torch.Generator(device=dev).manual_seed(3000 + rank)
            hx = torch.rand(HB, 3, HP, HP, device=dev, generator=gen)
            hf = {"fea_up2": (torch.rand(HB, 320, HP // 2, HP // 2, device=dev, generator=gen) - 0.5) * 12 ** 0.5,
                  "fea_up1": (torch.rand(HB, 320, HP // 4, HP // 4, device=dev, generator=gen) - 0.5) * 12 ** 0.5,
                  "fea_up4": torch.zeros(1, device=dev)}

            def hstep():
                hz, _ = model(hx, None, lr_enc=hf, add_gt_noise=False, calc_loss=False)
                return model(hz, hx, reverse=True, lr_enc=hf, add_gt_noise=False, test_input_size=HP)
            model.flow_upsampler.conv_dtype = "bf16"
            try:
                hstep(); hstep()
                barrier()
                hip.prof_enable([hip.K_CONV1_FT_L1])
                t4 = time.perf_counter()
                for _ in range(hsteps):
                    hxr, _ = hstep()
                barrier()
                h_elapsed = shard.max_over_ranks(time.perf_counter() - t4, dist, dev)
                h_ms, h_n = hip.prof_read(hip.K_CONV1_FT_L1)
                hip.prof_enable([])
            finally:
                model.flow_upsampler.conv_dtype = "fp32"
            hires = {"elapsed": h_elapsed, "dom_ms": h_ms, "dom_n": h_n, "B": HB, "P": HP, "steps": hsteps,
                     "round_trip_err": float((hxr - hx).abs().max())}
            del hx, hf, hxr
            hip.Workspace.clear()
        if secondary:
            # (1) the same steps with everything on ONE stream: kernels strictly one after another, so the dominant
            #     kernel's duration is free of time-sharing with the per-step chain; (2) one step with every kernel timed
            hip.set_schedule(aux_stream=0, groups=1)
            step(); torch.cuda.synchronize()
            hip.prof_enable([hip.K_CONV1_FT_L1])
            t1 = time.perf_counter()
            for _ in range(sec_steps):
                step()
            torch.cuda.synchronize()
            serial_elapsed = time.perf_counter() - t1
            s_ms, s_n = hip.prof_read(hip.K_CONV1_FT_L1)
            hip.prof_enable([])
            serial = {"ms_per_step": serial_elapsed / sec_steps * 1e3, "dominant_avg_launch_ms": s_ms / max(s_n, 1), "steps": sec_steps}
            names = (("conv1_ft_l1", hip.K_CONV1_FT_L1), ("conv1_ft_l2", hip.K_CONV1_FT_L2),
                     ("conv3_f", hip.K_CONV3_F), ("conv1_finish", hip.K_CONV1_FINISH),
                     ("conv3_enc", hip.K_CONV3_ENC), ("conv3_dec", hip.K_CONV3_DEC), ("anic", hip.K_ANIC))
            hip.prof_enable([k for _, k in names])
            step()
            torch.cuda.synchronize()
            prof = {name: hip.prof_read(kid) for name, kid in names}
            hip.prof_enable([])
            hip.set_schedule(aux_stream=1, groups=2)

    elapsed = shard.max_over_ranks(elapsed, dist, dev)
    ranks = shard.rank_devices(dist, dev)

    if rank == 0:
        patches = world * B * args.steps
        value = patches / elapsed
        # dominant kernel: the level-1 (C=12, 80x80 planes) instantiation of the bulk kernel: the conditioning part of both
        # coupling networks' first conv (+ fFeatures' 1x1).  Work per conditional step and pixel in the direct form
        # SURVEY.md section 8(a) row a11 counts (without the z1 taps, fAffine's 1x1 and the last conv):
        # 2 x 64*9*320 + 64*64 MACs.  The Winograd F(4x4,3x3) form (default) issues 36 multiplications per 4x4 outputs, input
        # and output channel instead of 144, i.e. 2 x 64*2.25*320 + 64*64 MACs per pixel on the matrix cores (F(2x2,3x3): 16
        # per 2x2 outputs, 2 x 64*4*320 + 64*64); the data / output transforms run on the vector ALUs and are not counted.
        algo = hip.set_conv1_algo(-1)
        wino = algo != hip.CONV1_DIRECT
        kname = {hip.CONV1_DIRECT: "conv1_ft_kernel<1>", hip.CONV1_WINOGRAD: "conv1_ft_wino_kernel<1>",
                 hip.CONV1_WINOGRAD4: "conv1_ft_wino4_kernel<1>"}[algo]
        ms1, n1 = dom_ms, dom_n
        px1 = (PATCH // 2) ** 2
        macs_px_direct = 2 * 64 * 9 * 320 + 64 * 64
        macs_px = {hip.CONV1_DIRECT: macs_px_direct, hip.CONV1_WINOGRAD: 2 * 64 * 4 * 320 + 64 * 64,
                   hip.CONV1_WINOGRAD4: 2 * 64 * 9 * 320 // 4 + 64 * 64}[algo]
        per_mac = 2.0 * px1 * B * 16 * 2 * args.steps                   # 16 conditional steps, encode + decode
        flops_total, flops_total_direct = per_mac * macs_px, per_mac * macs_px_direct
        flops_launch = flops_total / max(n1, 1)
        avg_ms = ms1 / max(n1, 1)
        achieved = flops_launch / (avg_ms * 1e-3) / 1e12 if n1 else None
        direct_equiv = flops_total_direct / max(n1, 1) / (avg_ms * 1e-3) / 1e12 if n1 else None
        traffic, traffic_note = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and B == BATCH:
            tj = json.load(open(tpath)).get(kname)
            if tj:
                traffic, traffic_note = tj["bytes_per_launch"], tj["note"]
        res = {
            "metric": "flow fwd+inv patches/sec (160x160, fp32)",
            "value": value, "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"InterFlow flow encode+decode, batch {B} of 160x160 RGB patches per GPU, fp32, "
                                   "K=16 L=2 shipped config, lr_enc precomputed (synthetic 320-channel features), "
                                   "add_gt_noise=False (BASELINE.json configs[1])",
                       "cross_pass_overlap": bool(hip.set_cross_pass_overlap()),
                       "schedule_note": "every pass loads and converts the conditioning maps and runs all of both coupling networks "
                                        "(no PreparedCond); with cross_pass_overlap a pass whose conditioning tensors are the previous "
                                        "pass's, unwritten since, enqueues that work on the library's bulk stream behind the previous "
                                        "pass's bulk work instead of behind its whole chain (IFL_FLOW_COND_EARLY; bit-identical "
                                        "outputs; --no-cross-pass-overlap measures the plain order)",
                       "batch_per_gpu": B, "patch": PATCH, "sharding": f"dp{world} (independent batches, no collective)",
                       "world_size": dist.get_world_size() if dist is not None else 1, "ranks": ranks},
            "roofline": {"bound": "mfma",
                         "kernel": kname + " (both coupling networks' first 3x3 conv over the 320 conditioning "
                                   "channels + fFeatures' 1x1, level 1, %d conditional steps per launch)" % round(32 * args.steps / max(n1, 1)),
                         "algorithm": (("Winograd F(4x4,3x3)" if algo == hip.CONV1_WINOGRAD4 else "Winograd F(2x2,3x3)") +
                                       " minimal filtering in fp32: achieved/frac count the multiply-adds the kernel issues on "
                                       "the matrix cores (" + ("36 per 4x4 outputs instead of the direct form's 144" if algo == hip.CONV1_WINOGRAD4
                                                               else "16 per 2x2 outputs instead of the direct form's 36") +
                                       "); direct_equivalent_tflops prices the same launch at the direct-form FLOPs of "
                                       "SURVEY.md section 8(d) and may exceed the hardware peak") if wino else "direct implicit GEMM",
                         "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": (achieved / PEAK_F32_MFMA_TFLOPS) if achieved else None,
                         "direct_equivalent_tflops": direct_equiv,
                         "direct_flops_per_launch": flops_total_direct / max(n1, 1),
                         "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_note": traffic_note,
                         "avg_launch_ms": avg_ms, "launches": n1, "flops_per_launch": flops_launch,
                         "method": "HIP event pair around every launch of this kernel inside the timed region, on the "
                                   "stream it is launched on; the light per-step chain runs concurrently on other "
                                   "streams, so the duration includes sharing the CUs with it (--breakdown re-times it alone)"},
            "path_tflops": 213.1e9 * B * args.steps / elapsed / 1e12,     # at the reference's 213.1 GFLOP per patch fwd+inv

            "roundtrip_max_abs_err": err,
        }
        if secondary:
            res["cond_reuse"] = {"value": world * B * sec_steps / reuse_elapsed, "unit": "patches/s",
                                 "ms_per_step": reuse_elapsed / sec_steps * 1e3, "steps": sec_steps,
                                 "note": "secondary, NOT the metric: same encode+decode, but the z-independent first conv of "
                                         "the coupling networks is computed once per step for both passes "
                                         "(ifl_cond_prepare); 0.54 of the reference's FLOPs are executed"}
        if bf16:
            fl = flops_total_direct / max(bf16["dom_n"], 1)           # the bf16 variant uses the direct form
            res["bf16_conv1"] = {"value": world * B * args.steps / bf16["elapsed"], "unit": "patches/s",
                                 "ms_per_step": bf16["elapsed"] / args.steps * 1e3, "dtype": "bf16 operands / f32 accumulate in the "
                                 "coupling networks' z-independent convolutions (conv1_ft, fFeatures' 1x1 and last conv), f32 elsewhere",
                                 "conv1_ft_bf16_kernel<1>": {"avg_launch_ms": bf16["dom_ms"] / max(bf16["dom_n"], 1),
                                                             "achieved_tflops": fl / (bf16["dom_ms"] / max(bf16["dom_n"], 1) * 1e-3) / 1e12,
                                                             "peak_tflops": 2500.0},
                                 "z_rel_err_vs_fp32": bf16["z_rel_err_vs_fp32"], "roundtrip_max_abs_err": bf16["round_trip_err"],
                                 "note": "secondary, NOT the metric (BASELINE.json configs[4] arithmetic on the configs[1] workload)"}
        if hires:
            hfl = 2.0 * macs_px_direct * (hires["P"] // 2) ** 2 * hires["B"] * 16 * 2 * hires["steps"] / max(hires["dom_n"], 1)
            h_avg = hires["dom_ms"] / max(hires["dom_n"], 1)
            res["config4_512_bf16"] = {
                "value": world * hires["B"] * hires["steps"] / hires["elapsed"], "unit": "512x512 patches/s",
                "patches_160_equivalent_per_s": world * hires["B"] * hires["steps"] / hires["elapsed"] * (hires["P"] / PATCH) ** 2,
                "ms_per_step": hires["elapsed"] / hires["steps"] * 1e3, "steps": hires["steps"], "batch_per_gpu": hires["B"],
                "dtype": "bf16 operands / f32 accumulate in the coupling networks' z-independent convolutions (conv1_ft, fFeatures' "
                         "1x1 and last conv); f32 ActNorm, log-det, couplings, invertible 1x1 maps and fAffine's z-dependent chain",
                "roofline": {"bound": "mfma", "kernel": "conv1_ft_bf16 (level 1, direct implicit GEMM, v_mfma_f32_32x32x16_bf16)",
                             "avg_launch_ms": h_avg, "launches": hires["dom_n"], "achieved": hfl / (h_avg * 1e-3) / 1e12,
                             "peak": 2500.0, "unit": "TFLOP/s", "frac": hfl / (h_avg * 1e-3) / 1e12 / 2500.0},
                "roundtrip_max_abs_err": hires["round_trip_err"],
                "note": "secondary, NOT the metric: BASELINE.json configs[4] (high-res 512x512 patches, bf16 coupling-NN convs with "
                        "fp32 ActNorm/logdet) per GPU; encode + decode, measured after the timed region"}
        if serial:
            res["serial_schedule"] = serial
            # (the one-stream schedule launches 4 first-level steps at a time, the timed region 8: price each at its own launch count)
            serial_flops_launch = flops_total / args.steps * serial["steps"] / max(s_n, 1)
            res["serial_schedule"]["dominant_steps_per_launch"] = round(32 * serial["steps"] / max(s_n, 1))
            res["serial_schedule"]["dominant_tflops_alone"] = serial_flops_launch / (serial["dominant_avg_launch_ms"] * 1e-3) / 1e12
            res["serial_schedule"]["dominant_frac_alone"] = res["serial_schedule"]["dominant_tflops_alone"] / PEAK_F32_MFMA_TFLOPS
            res["kernels_ms_per_step"] = {k: v[0] for k, v in prof.items()}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(sd_np)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
