#!/bin/bash
# GPU session: bf16 parity tests, then the bench's config4_512_bf16 line (and the batch-32 bf16 line with --extras)
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 200 --timeout-method thread -k "bf16" 2>&1 | tail -2 || exit 1
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 10 --extras > gpurun_out/b512.json 2> gpurun_out/b512.err || { tail -5 gpurun_out/b512.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/b512.json").read().strip().splitlines()[-1])
h=d["config4_512_bf16"]; r=h["roofline"]; e=d.get("bf16_conv1", {})
print("fp32 %.1f | 512 bf16: %.1f patches/s %.2f ms/step, L1 bulk %.3f ms = %.0f TF (%.3f) | B32 bf16: %.1f patches/s, %.0f TF" % (d["value"], h["value"], h["ms_per_step"], r["avg_launch_ms"], r["achieved"], r["frac"], e.get("value", 0), e.get("conv1_ft_bf16_kernel<1>", {}).get("achieved_tflops", 0)))
PY
