"""ATen operator counts per phase of one training step (torch.profiler, host side): which glue launches the small
element-wise kernels that show up in the rocprof list.  python tools/op_count_train.py"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
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
for _ in range(3):
    training.train_step(model, opt, gt, lr, labels, reducer=red, late_check=True)
phases = collections.OrderedDict()
def run(name, fn):
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        out = fn()
    torch.cuda.synchronize()
    cnt = collections.Counter()
    for e in prof.key_averages():
        if e.key.startswith("aten::") and e.key.split("::")[1] in ("fill_", "zero_", "copy_", "add_", "add", "mul", "zeros", "clone", "sum", "stack", "cat", "contiguous", "empty", "sub", "exp", "where", "_foreach_add_"):
            cnt[e.key] = e.count
    phases[name] = cnt
    return out
feats = run("encoder", lambda: model.arch.rrdbPreprocessing(gt))
z, logdet = run("flow fwd", lambda: model(lr, gt, lr_enc=feats, add_gt_noise=True, calc_loss=False))
def lossf():
    nll, ib = model.loss_terms(z, logdet, 160 * 160, labels); return nll.mean() + ib
loss = run("loss", lossf)
run("zero_grad", red.zero_grad)
def bwd():
    with train_fused.accumulate_into_grad():
        loss.backward()
run("backward", bwd)
run("optimizer", lambda: (red.finish(), opt.step()))
keys = sorted({k for c in phases.values() for k in c})
print("op".ljust(22) + "".join(n.rjust(11) for n in phases))
for k in keys:
    print(k.ljust(22) + "".join(str(c.get(k, 0)).rjust(11) for c in phases.values()))
