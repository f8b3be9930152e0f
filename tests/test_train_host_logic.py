"""Host-side logic of the training path that needs no GPU: the stacked-parameter function, the batched class log-densities and
the hub node that defers the gradient towards the conditioning features until every pass over them has run its backward."""
import torch

from interflow_amd import train_flow, train_fused


def test_stack_groups_is_torch_stack_with_a_grouped_gradient_add():
    g = torch.Generator().manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(1, 5, 1, 1, generator=g)) for _ in range(3)]
    b = [torch.nn.Parameter(torch.randn(4, 4, generator=g)) for _ in range(3)]
    wa, wb = torch.randn(3, 1, 5, 1, 1, generator=g), torch.randn(3, 4, 4, generator=g)
    sa, sb = train_fused._StackGroups.apply(2, *a, *b)
    assert torch.equal(sa, torch.stack(a)) and torch.equal(sb, torch.stack(b))
    ((sa * wa).sum() + (sb * wb).sum()).backward()
    for i in range(3):
        assert torch.equal(a[i].grad, wa[i]) and torch.equal(b[i].grad, wb[i])
    # inside accumulate_into_grad(): added into the existing .grad tensors, nothing handed to autograd
    for p in a + b:
        p.grad = torch.full_like(p, 0.5)
    ptrs = [p.grad.data_ptr() for p in a + b]
    sa, sb = train_fused._StackGroups.apply(2, *a, *b)
    with train_fused.accumulate_into_grad():
        ((sa * wa).sum() + (sb * wb).sum()).backward()
    for i in range(3):
        assert torch.equal(a[i].grad, wa[i] + 0.5) and torch.equal(b[i].grad, wb[i] + 0.5)
    assert ptrs == [p.grad.data_ptr() for p in a + b], "the .grad tensors must be updated in place (bucket views)"
    # a group that receives no gradient
    sa, sb = train_fused._StackGroups.apply(2, *a, *b)
    for p in a + b:
        p.grad = None
    (sa * wa).sum().backward()
    assert all(p.grad is not None for p in a) and all(p.grad is None for p in b)


def test_gaussian_logp_classes_matches_the_per_class_form():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 6, 4, 5, generator=g, requires_grad=True)
    A = [(torch.randn(6, 6, generator=g) + 2 * torch.eye(6)).requires_grad_() for _ in range(3)]
    m = [torch.randn(6, generator=g, requires_grad=True) for _ in range(3)]
    covs = [a @ a.T for a in A]
    ref = torch.stack([train_flow.gaussian_logp(m[i], covs[i], x) for i in range(3)], dim=1)
    got = train_flow.gaussian_logp_classes(torch.stack(m), torch.stack(covs), x)
    assert got.shape == ref.shape == (3, 3)
    assert float((got - ref).detach().abs().max()) <= 1e-5 * float(ref.detach().abs().max())
    w = torch.randn(3, 3, generator=g)
    g1 = torch.autograd.grad((ref * w).sum(), [x] + A + m, retain_graph=True)
    g2 = torch.autograd.grad((got * w).sum(), [x] + A + m)
    for u, v in zip(g1, g2):
        assert float((u - v).abs().max()) <= 1e-5 * float(v.abs().max()) + 1e-7


class _FakePlan:
    """What _FtHub needs of a LevelPlan: the saved planes, their (step, network) order and the product."""

    def __init__(self, nc, decode, dp1):
        self.nc, self.decode, self.dp1, self.calls = nc, decode, dp1, 0

    def ft_grad(self, acc):
        self.calls += 1
        return acc.sum(dim=(1, 2)).reshape(self.nc, 2).t().contiguous()          # [2, nc]: keeps the plane order visible


class _FakePass(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ft, tok, hub, plan, log):
        ctx.hub, ctx.plan, ctx.log = hub, plan, log
        return ft.sum().reshape(1)

    @staticmethod
    def backward(ctx, g):
        ctx.log.append("pass")
        ctx.hub.add(ctx.plan)
        return None, torch.zeros(1), None, None, None


def test_ft_hub_makes_one_product_after_every_pass():
    nc = 3
    ft = torch.randn(2, 3, requires_grad=True)
    log = []
    planes = [torch.arange(nc * 2 * 4, dtype=torch.float32).reshape(nc * 2, 2, 2) * (k + 1) for k in range(3)]
    plans = [_FakePlan(nc, False, planes[0]), _FakePlan(nc, True, planes[1]), _FakePlan(nc, False, planes[2])]
    with train_fused.shared_ft_grad():
        hub = train_fused._hub_for(ft)
        assert train_fused._hub_for(ft) is hub
        outs = [_FakePass.apply(hub.out, hub.tok, hub, p, log) for p in plans]
    assert train_fused._ft_hubs is None and train_fused._hub_for(ft) is None           # outside the context: one product per pass

    real_finish = hub.finish

    def finish():
        log.append("hub")
        return real_finish()
    hub.finish = finish
    # the product has the shape of whatever the plan returns; here [2, nc] -- give ft that shape's gradient through a custom check
    grads = []
    orig = train_fused._FtHubFn.backward

    def spy(ctx, g_ft, g_tok):
        out = orig(ctx, g_ft, g_tok)
        grads.append(out[1])
        return None, torch.zeros_like(ft)
    train_fused._FtHubFn.backward = staticmethod(spy)
    try:
        (outs[0] + 2 * outs[1] + outs[2]).sum().backward()
    finally:
        train_fused._FtHubFn.backward = staticmethod(orig)
    assert log == ["pass", "pass", "pass", "hub"], log                                  # the hub's backward after all three
    assert sum(p.calls for p in plans) == 1 and len(grads) == 1
    # planes summed in the FIRST contributor's order; the decode plan's (step) order is reversed
    first = next(p for p in plans if p.calls == 1)
    want = torch.zeros_like(planes[0])
    for p in plans:
        d = p.dp1
        if p.decode != first.decode:
            d = d.view(nc, 2, 2, 2).flip(0).reshape(d.shape)
        want = want + d
    assert torch.equal(grads[0], first.ft_grad(want))


def test_reducer_mark_ready_contributions_and_rebucketing():
    """training.GradientAllReducer on the CPU, world size 1 (nothing is exchanged; the launch log is what is checked):
    gradients added by a fused driver become ready through mark_ready (no autograd hook fires for them), a parameter that is
    written by several passes is final only with the last contribution, and after the first step the parameters that received
    nothing move into buckets of their own, so that every bucket of live parameters is launched before finish()."""
    from interflow_amd import training
    g = torch.Generator().manual_seed(3)
    live = [torch.nn.Parameter(torch.randn(64, generator=g)) for _ in range(6)]
    dead = [torch.nn.Parameter(torch.randn(64, generator=g)) for _ in range(2)]
    params = [live[0], dead[0], live[1], live[2], dead[1], live[3], live[4], live[5]]
    red = training.GradientAllReducer(params, None, bucket_bytes=3 * 64 * 4)         # three parameters per bucket
    assert len(red.buckets) == 3

    def step():
        red.zero_grad()
        # live[0..2] through autograd (hooks); live[3..5] the way the level drivers do it: two passes add into .grad directly
        loss = sum((p * p).sum() for p in live[:3])
        train_fused.reset_contributions()
        train_fused._note_forward(live[3:])
        train_fused._note_forward(live[3:])
        with train_fused.accumulate_into_grad(red.mark_ready):
            loss.backward()
            for k in range(2):
                torch._foreach_add_([p.grad for p in live[3:]], [torch.ones(64) for _ in range(3)])
                before = red.launched_early
                train_fused._settle(live[3:])
                if k == 0:
                    assert red.launched_early == before, "a gradient that another pass still writes must not be exchanged"
        early = red.launched_early
        red.finish()
        return early

    early1 = step()
    assert early1 < len(red.buckets)                      # the dead parameters hold their buckets back in the first step
    assert all(torch.equal(p.grad, 2 * p.detach()) for p in live[:3]) and all(torch.equal(p.grad, torch.full((64,), 2.0)) for p in live[3:])
    early2 = step()                                       # zero_grad re-buckets: live parameters first, dead ones behind
    assert [len(b["params"]) for b in red.buckets] == [3, 3, 2]
    assert {id(p) for p in red.buckets[2]["params"]} == {id(p) for p in dead}
    assert early2 == 2 and red.launch_log[-1] == (2, "finish")
    assert all(torch.equal(p.grad, 2 * p.detach()) for p in live[:3]) and all(torch.equal(p.grad, torch.full((64,), 2.0)) for p in live[3:])
    assert all(float(p.grad.abs().sum()) == 0 for p in dead)
    # gradients are views into the new buckets
    for b in red.buckets:
        o = 0
        for p in b["params"]:
            assert p.grad.data_ptr() == b["flat"].data_ptr() + 4 * o
            o += p.numel()
