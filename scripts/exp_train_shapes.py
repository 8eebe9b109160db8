"""Round 4: the one-launch training layer calls against round 3's launch per time step (training.STEP_LAUNCHES) over odd shapes --
one step, one row block, ragged last row block, unshared gates, no BatchNorm, the full-band size: spike agreement, BatchNorm
buffers and gradients side by side (the products differ in their rounding order -- fp32 MFMA against a scalar fma chain -- so a
membrane within ~1e-6 of the threshold may flip; this prints, it does not assert)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spiking_fullsubnet_amd import training
from spiking_fullsubnet_amd import modeling_spiking_fullsubnet as M
DEV = "cuda:0"
cases = [  # I, H, L, R, T, shared, bn
    (8, 16, 1, 2, 1, True, True), (8, 16, 2, 3, 2, True, True), (12, 32, 2, 17, 3, False, True), (12, 32, 2, 64, 9, True, False),
    (38, 224, 2, 600, 12, True, True), (64, 320, 2, 64, 10, True, True), (64, 320, 2, 70, 6, False, True), (20, 48, 3, 130, 7, True, True),
    (38, 224, 2, 1700, 4, True, True),
]
for I, H, L, R, T, shared, bn in cases:
    torch.manual_seed(I + H + R)
    st = M.StackedGSU(I, H, L, shared, bn).to(DEV).train()
    tw = copy.deepcopy(st)
    x = torch.randn(T, R, I, device=DEV)
    cot = torch.randn(T, R, H, device=DEV)
    res = []
    for stack, step in ((st, False), (tw, True)):
        training.STEP_LAUNCHES = step
        xi = x.clone().requires_grad_(True)
        t0 = time.perf_counter()
        outs = training.gsn_stack(xi, stack, True)
        (outs[-1] * cot).sum().backward()
        training.check_pending()
        torch.cuda.synchronize()
        res.append((outs, xi.grad, [p.grad for p in stack.parameters()], [b.clone() for b in stack.buffers()], time.perf_counter() - t0))
    training.STEP_LAUNCHES = False
    (oa, ga, pa, ba, ta), (ob, gb, pb, bb, tb) = res
    agree = min(float((u == v).float().mean()) for u, v in zip(oa[1:], ob[1:]))
    def rel(u, v):
        return float((u - v).norm() / (v.norm() + 1e-30))
    print(f"I={I} H={H} L={L} R={R} T={T} shared={int(shared)} bn={int(bn)}: spike agreement {agree:.6f}  d_x rel {rel(ga, gb):.2e}  "
          f"param grads max rel {max(rel(u, v) for u, v in zip(pa, pb)):.2e}  buffers max rel {max([rel(u.float(), v.float()) for u, v in zip(ba, bb)] or [0]):.2e}  "
          f"finite {all(bool(torch.isfinite(g_).all()) for g_ in [ga] + pa)}  one launch {ta*1e3:.1f} ms / per step {tb*1e3:.1f} ms", flush=True)
