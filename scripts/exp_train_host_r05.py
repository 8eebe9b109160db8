"""Round 5: host time against device time of a training step (B = 64, T = 1000): wall time per step with a synchronisation after
EVERY step (a trainer that reads the loss each step) and without (three steps in flight), by chunk count of the pipelined stacks."""
import sys, os, time, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import refweights as rw
import spiking_fullsubnet_amd as pkg
from spiking_fullsubnet_amd import training
dev = torch.device("cuda")
wave = torch.from_numpy(rw.synth_wave(64, 1000, 0)).to(dev)
kw = rw.LIVE_M
m = pkg.SpikingFullSubNet(**kw); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in rw.live_state_dict(kw, 21).items()}); m = m.to(dev).train()
def step():
    for p_ in m.parameters(): p_.grad = None
    o = m(wave); (o[0].pow(2).mean() + o[1].mean()).backward()
for K in (20, 10, 5, 1):
    training.STACK_CHUNKS = K
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6): step()
    torch.cuda.synchronize(); free = (time.perf_counter() - t0) / 6
    t0 = time.perf_counter(); host = 0.0
    for _ in range(6):
        h0 = time.perf_counter(); step(); host += time.perf_counter() - h0
        torch.cuda.synchronize()
    synced = (time.perf_counter() - t0) / 6
    print(f"chunks {K}: {free*1e3:.1f} ms per step free-running, {synced*1e3:.1f} ms with a sync per step, host enqueue {host/6*1e3:.1f} ms", flush=True)
