"""Round 6: a training step of the live baseline_m model eager vs replayed from a HIP graph (training.GraphedTrainStep).
Prints host enqueue time, wall time per step, and the largest gradient difference between the two on the same batch."""
import os, sys, time, json, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
from spiking_fullsubnet_amd import training as tr

B, T = int(os.environ.get("B", 64)), int(os.environ.get("T", 1000))
dev = torch.device("cuda:0")
kw = rw.LIVE_M
sd = rw.live_state_dict(kw, seed=3)
model = pkg.SpikingFullSubNet(**kw)
if True:
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
model = model.to(dev).train()
waves = [torch.from_numpy(rw.synth_wave(B, T, seed=s)).to(dev) for s in (7, 8)]
loss_fn = lambda out: out[0].pow(2).mean() + out[1].mean()
state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}

def restore():
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(state0[k])

def eager(w):
    for p in model.parameters():
        p.grad = None
    loss = loss_fn(model(w)); loss.backward(); return loss

res = {}
eager(waves[0]); torch.cuda.synchronize()
t0 = time.perf_counter(); enq = 0.0
for i in range(4):
    a = time.perf_counter(); eager(waves[i & 1]); enq += time.perf_counter() - a
torch.cuda.synchronize()
res["eager_ms"] = (time.perf_counter() - t0) / 4 * 1e3
res["eager_host_enqueue_ms"] = enq / 4 * 1e3
restore()
l_e = float(eager(waves[1])); torch.cuda.synchronize()
g_e = [p.grad.clone() for p in model.parameters()]
bn_e = {k: v.clone() for k, v in model.state_dict().items() if "running" in k}
restore()
print("eager done", res, flush=True)
t0 = time.perf_counter()
gs = tr.GraphedTrainStep(model, waves[0], loss_fn)
res["capture_s"] = time.perf_counter() - t0
res["layer_calls_captured"] = gs.layer_calls_captured
print("captured", res, flush=True)
l_g = gs(waves[1]); torch.cuda.synchronize()
res["loss_eager"], res["loss_graph"] = l_e, float(l_g)
res["grad_max_abs_diff"] = max(float((a - p.grad).abs().max()) for a, p in zip(g_e, model.parameters()))
res["grad_max_abs"] = max(float(a.abs().max()) for a in g_e)
res["bn_stats_max_abs_diff"] = max(float((bn_e[k] - v).abs().max()) for k, v in model.state_dict().items() if "running" in k)
for _ in range(2):
    gs(waves[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(6):
    gs(waves[i & 1])
torch.cuda.synchronize()
res["graph_ms"] = (time.perf_counter() - t0) / 6 * 1e3
res["mem_GB"] = torch.cuda.max_memory_allocated() / 2**30
print(json.dumps(res))
