import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
dev = torch.device("cuda", 0)
kw = rw.LIVE_M
model = pkg.SpikingFullSubNet(**kw)
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in rw.live_state_dict(kw, 21).items()})
model = model.eval().to(dev)
eng = model.engine()
stft = model._stft(torch.from_numpy(rw.synth_wave(64, 1000, 0)).to(dev)).contiguous()
def run(pipe, chunk, n=4):
    eng.pipeline_chunk = chunk
    eng.forward_stft(stft, pipeline=pipe); torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = 0
    for _ in range(n):
        a = time.perf_counter(); eng.forward_stft(stft, pipeline=pipe); th += time.perf_counter() - a
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, th / n * 1e3
print("sequential          %.2f ms (host enqueue %.2f)" % run(False, 128))
for c in (500, 250, 125):
    print("pipelined chunk=%3d  %.2f ms (host enqueue %.2f)" % ((c,) + run(True, c)))
