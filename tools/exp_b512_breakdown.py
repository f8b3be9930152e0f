"""Experiment helper: per-kernel sums of one encode + decode at 512x512, batch 4, bf16 convolutions (config4), one stream."""
import sys, time
import torch
sys.path.insert(0, ".")
from interflow_amd import hip, synth
from interflow_amd.flow_model import InterpFlow

dev = torch.device("cuda", 0)
model = InterpFlow([1.0, 4.0])
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_interpflow_state([1.0, 4.0], seed=0).items()}, strict=True)
model.eval().to(dev)
B, P = 4, 512
gen = torch.Generator(device=dev).manual_seed(1)
x = torch.rand(B, 3, P, P, device=dev, generator=gen)
hf = {"fea_up2": torch.rand(B, 320, P // 2, P // 2, device=dev, generator=gen) - 0.5,
      "fea_up1": torch.rand(B, 320, P // 4, P // 4, device=dev, generator=gen) - 0.5, "fea_up4": torch.zeros(1, device=dev)}
model.flow_upsampler.conv_dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"


def step():
    z, _ = model(x, None, lr_enc=hf, add_gt_noise=False, calc_loss=False)
    return model(z, x, reverse=True, lr_enc=hf, add_gt_noise=False, test_input_size=P)


names = (("conv1_ft_l1", hip.K_CONV1_FT_L1), ("conv1_ft_l2", hip.K_CONV1_FT_L2), ("conv3_f", hip.K_CONV3_F),
         ("conv1_finish", hip.K_CONV1_FINISH), ("conv3_enc", hip.K_CONV3_ENC), ("conv3_dec", hip.K_CONV3_DEC), ("anic", hip.K_ANIC))
with torch.no_grad():
    step(); step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    print("overlapped: %.2f ms per step" % ((time.perf_counter() - t0) / 5 * 1e3))
    hip.set_schedule(aux_stream=0, groups=1)
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    print("one stream: %.2f ms per step" % ((time.perf_counter() - t0) / 5 * 1e3))
    hip.prof_enable([k for _, k in names])
    step(); torch.cuda.synchronize()
    print({n: round(hip.prof_read(k)[0], 3) for n, k in names})
