"""Device-side duration of the phases of one training step, free running (no host read inside a step, as train_step(...,
late_check=True, host_sync=False) with a prefetched encoder): HIP events recorded on the main stream at the phase boundaries
and at the points where the backward pass enters and leaves the flow (tensor hooks).  python tools/phase_events.py"""
import os, sys, torch
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
N = 12
marks = []
def ev(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
train_fused.MARK = ev
torch.cuda.synchronize()
for it in range(N):
    ev("start")
    lrq = lr.clone().requires_grad_(True)                      # (a leaf so that the flow's input gradient marks the end of its backward)
    z, logdet = model(lrq, gt, lr_enc=feats, add_gt_noise=True, calc_loss=False); ev("flow fwd")
    z.register_hook(lambda g: ev("loss bwd"))
    lrq.register_hook(lambda g: ev("flow bwd"))
    nll, ib = model.loss_terms(z, logdet, 160 * 160, labels); loss = nll.mean() + ib; ev("loss fwd")
    red.zero_grad()
    with train_fused.accumulate_into_grad():
        loss.backward()
    ev("bwd tail+join")
    red.finish(); opt.step(); ev("optimizer")
torch.cuda.synchronize()
acc, order = {}, []
per = len(marks) // N
for it in range(2, N):
    m = marks[it * per:(it + 1) * per]
    for (n0, e0), (n1, e1) in zip(m, m[1:]):
        k = (m.index((n1, e1)), n1)
        if k not in acc: order.append(k)
        acc[k] = acc.get(k, 0.0) + e0.elapsed_time(e1)
tot = 0.0
for k in order:
    print(f"{k[1]:16s} {acc[k] / (N - 2):7.2f} ms"); tot += acc[k] / (N - 2)
print(f"{'sum':16s} {tot:7.2f} ms")
