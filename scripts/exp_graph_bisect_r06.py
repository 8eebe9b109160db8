"""Which part of a captured training step breaks hipStreamEndCapture?  Each variant in a child process."""
import os, sys, subprocess, json
V = os.environ.get("VARIANT")
if V is None:
    for v in ("stft", "stft_istft", "fb_fwd", "fwd", "fwd_bwd_nopin", "fwd_bwd_pin", "layer_fwd_bwd"):
        r = subprocess.run([sys.executable, __file__], env=dict(os.environ, VARIANT=v), capture_output=True, text=True)
        print(v, "rc", r.returncode, (r.stdout.strip().splitlines() or ["-"])[-1][:200], flush=True)
        if r.returncode != 0:
            print("   ", "\n    ".join(l[:200] for l in r.stderr.strip().splitlines()[-6:]), flush=True)
    sys.exit(0)
import faulthandler; faulthandler.enable()
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
from spiking_fullsubnet_amd import training as tr
B, T = 8, 200
dev = torch.device("cuda:0")
kw = rw.LIVE_M
model = pkg.SpikingFullSubNet(**kw)
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in rw.live_state_dict(kw, seed=3).items()}, strict=True)
model = model.to(dev).train()
wave = torch.from_numpy(rw.synth_wave(B, T, seed=7)).to(dev)
window = torch.hann_window(512, device=dev)
x_fb = torch.randn(T, B, 64, device=dev)

def body():
    if V == "stft":
        return torch.stft(wave, 512, 128, 512, window=window, return_complex=True, pad_mode="constant").abs().sum()
    if V == "stft_istft":
        c = torch.stft(wave, 512, 128, 512, window=window, return_complex=True, pad_mode="constant")
        return tr._istft(c, 512, 128, 512, window, wave.shape[1]).sum()
    if V == "fb_fwd":
        with torch.no_grad():
            return tr.gsn_stack(x_fb, model.fb_model.sequence_model, True)[-1].sum()
    if V == "layer_fwd_bwd":
        xx = x_fb.clone().requires_grad_()
        out = tr.gsn_stack(xx, model.fb_model.sequence_model, True)[-1]
        l = out.sum(); l.backward(); return l
    if V == "fwd":
        with torch.no_grad():
            out = model(wave)
        return out[0].sum()
    for p in model.parameters():
        p.grad = None
    out = model(wave)
    l = out[0].pow(2).mean() + out[1].mean()
    l.backward()
    return l

nograd_variants = ("fb_fwd", "fwd")
tr._capture_errs = []
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        tr._capture_errs = None
        if V in nograd_variants:
            try:
                body()
            except Exception as e:
                print("eager warmup raised", repr(e)[:200])
        else:
            body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
tr.check_pending()
for p in model.parameters():
    p.grad = None
pin = torch.zeros((1,), dtype=torch.int32, pin_memory=True)
g = torch.cuda.CUDAGraph()
tr._capture_errs = []
with torch.cuda.graph(g):
    r = body()
    if V == "fwd_bwd_pin" and tr._capture_errs:
        pin.copy_(torch.stack(tr._capture_errs).max().reshape(1), non_blocking=True)
n = len(tr._capture_errs); tr._capture_errs = None
print("captured", n, flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed ok", float(r), n, flush=True)
