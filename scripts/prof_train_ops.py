import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
from torch.profiler import profile, ProfilerActivity
kw = rw.LIVE_M; sd = rw.live_state_dict(kw, 21)
m = pkg.SpikingFullSubNet(**kw); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True); m = m.to("cuda:0").train()
w = torch.from_numpy(rw.synth_wave(64, 1000, seed=7)).to("cuda:0")
def step():
    for p in m.parameters(): p.grad = None
    out = m(w); (out[0].pow(2).mean() + out[1].mean()).backward()
step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=40, max_shapes_column_width=70))
