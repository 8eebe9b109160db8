"""Training step of the live baseline_m model (forward in .train() mode + backward): this package's path (HIP training-step kernels +
library GEMMs) against the same step with the cell loop written as plain ATen operations per time step (the reference's own
structure: ~12 small ops per cell step, efficient_spiking_neuron.py:132-153).  python scripts/exp_train.py [B] [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
from spiking_fullsubnet_amd import training
DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 751
kw = rw.LIVE_M; sd = rw.live_state_dict(kw, 21)
m = pkg.SpikingFullSubNet(**kw); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True); m = m.to(DEV).train()
wave = torch.from_numpy(rw.synth_wave(B, T, 3)).to(DEV)


class Tri(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u):
        ctx.save_for_backward(u)
        return (u >= 0).float()

    @staticmethod
    def backward(ctx, g):
        (u,) = ctx.saved_tensors
        return g * torch.clamp(1 - u.abs(), min=0)


def aten_stack(x, stack, train):  # per-step ATen ops (what the reference's GSULayer / GSUCell do)
    outs = [x]
    cur = x
    for layer in stack.layers:
        cell = layer.cell
        R, H = cur.shape[1], cell.hidden_size
        h = torch.zeros(R, H, device=cur.device); c = torch.zeros(R, H, device=cur.device)
        wi = cell.weight_ih.repeat(2, 1) if cell.shared_weights else cell.weight_ih
        wh = cell.weight_hh.repeat(2, 1) if cell.shared_weights else cell.weight_hh
        ys = []
        for t in range(cur.shape[0]):
            gates = torch.mm(cur[t], wi.t()) + cell.bias_ih + torch.mm(h, wh.t())
            f, g = gates.chunk(2, 1)
            f = torch.sigmoid(f)
            c = f * c + (1 - f) * g
            if cell.use_bn:
                c = cell.batchnorm(c)
            h = Tri.apply(c)
            ys.append(h)
        cur = torch.stack(ys)
        outs.append(cur)
    return outs


def step(fn):
    for p in m.parameters():
        p.grad = None
    orig = training.gsn_stack
    training.gsn_stack = fn
    try:
        out = m(wave)
        loss = out[0].pow(2).mean() + out[1].mean()
        loss.backward()
    finally:
        training.gsn_stack = orig
    return float(loss.detach())


for name, fn, reps in (("HIP training-step kernels", training.gsn_stack, 3), ("per-step ATen ops", aten_stack, 1)):
    l0 = step(fn); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step(fn)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {dt*1e3:.1f} ms per training step (forward + backward), B={B}, T={T}, loss {l0:.6f}", flush=True)
