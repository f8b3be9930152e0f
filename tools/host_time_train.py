"""Host-side issue time of the phases of one training step (no device synchronisation inside a step): shows whether the step
is bound by the host issuing launches or by the device (the wait at the end).  python tools/host_time_train.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interflow_amd import synth, training, train_fused
from interflow_amd.flow_model import InterpFlow
dev = torch.device("cuda:0")
model = InterpFlow([2, 4])
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_interpflow_state([2, 4], seed=0).items()})
model.train().to(dev); model.arch.set_rrdb_training(False)
opt, sched = training.make_optimizer(model)
red = training.GradientAllReducer(model.parameters(), None)
B = 8
gt = torch.rand(B, 3, 160, 160, device=dev); lr = torch.rand(B, 3, 160, 160, device=dev)
labels = torch.tensor([2, 4] * 4, device=dev)
for _ in range(4):
    training.train_step(model, opt, gt, lr, labels, reducer=red, late_check=True)
feats = model.arch.rrdbPreprocessing(gt)
acc = {}
def tick(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t - t0); return t
N = 10
for it in range(N):
    torch.cuda.synchronize()
    t = time.perf_counter()
    f2 = model.arch.rrdbPreprocessing(gt); t = tick("encoder issue", t)
    torch.cuda.synchronize(); t = time.perf_counter()
    z, logdet = model(lr, gt, lr_enc=feats, add_gt_noise=True, calc_loss=False); t = tick("flow fwd issue", t)
    nll, ib = model.loss_terms(z, logdet, 160 * 160, labels); loss = nll.mean() + ib; t = tick("loss issue", t)
    red.zero_grad(); t = tick("zero_grad", t)
    with train_fused.accumulate_into_grad():
        loss.backward()
    t = tick("backward issue", t)
    red.finish(); opt.step(); t = tick("optimizer issue", t)
    torch.cuda.synchronize(); t = tick("final sync wait", t)
for k, v in acc.items():
    print(f"{k:20s} {v / N * 1e3:7.2f} ms")
