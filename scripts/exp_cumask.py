"""Diagnostic: how does hipExtStreamCreateWithCUMask map mask bits to compute units on MI355X?
Times a bandwidth-bound torch kernel (big copy) and the sequential forward on streams with different masks."""
import sys, os, time, ctypes
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
wave = torch.from_numpy(rw.synth_wave(64, 1000, 0)).to(dev)
stft = model._stft(wave).contiguous()
a = torch.empty(256 * 1024 * 1024 // 4, device=dev); b = torch.empty_like(a)
n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
print("CUs", n_cu)

def masks():
    yield "all", list(range(n_cu))
    yield "first128", list(range(128))
    yield "even", list(range(0, n_cu, 2))
    yield "first32", list(range(32))
    yield "stride8_32", list(range(0, n_cu, 8))
    yield "first8", list(range(8))
    yield "words0", [i for i in range(n_cu) if (i // 32) == 0]
    yield "mod32lt16", [i for i in range(n_cu) if (i % 32) < 16]

for name, cus in masks():
    s = eng._masked_stream(cus)
    with torch.cuda.stream(s):
        for _ in range(2): b.copy_(a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): b.copy_(a)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        eng.forward_stft(stft, pipeline=False); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2): eng.forward_stft(stft, pipeline=False)
        torch.cuda.synchronize()
        fw = (time.perf_counter() - t0) / 2
    print(f"{name:12s} n={len(cus):3d} copy {2*a.numel()*4/dt/1e9:8.1f} GB/s   forward {fw*1e3:7.2f} ms")
