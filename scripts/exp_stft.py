"""How long do the two edges of the path (torch.stft / torch.istft = rocFFT + elementwise kernels) take at B=64, T=1000?"""
import time, torch
dev = torch.device("cuda")
B, T = 64, 1000
wave = 0.05 * torch.randn(B, (T - 1) * 128, device=dev)
win = torch.hann_window(512, device=dev)
def stft(y): return torch.stft(y, 512, 128, 512, window=win, return_complex=True, pad_mode="constant")
def istft(s): return torch.istft(s, 512, 128, 512, window=win, length=wave.shape[-1])
X = stft(wave)
for name, fn, arg in (("stft", stft, wave), ("istft", istft, X)):
    for _ in range(3): fn(arg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn(arg)
    torch.cuda.synchronize(); print(name, "%.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3), tuple(fn(arg).shape))
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiking_fullsubnet_amd import spectral
for name, fn, arg in (("hip stft", lambda y: spectral.stft(y, 512, 128), wave), ("hip istft", lambda s: spectral.istft(s, 512, 128, length=wave.shape[-1]), X)):
    for _ in range(3): fn(arg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn(arg)
    torch.cuda.synchronize(); print(name, "%.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3), tuple(fn(arg).shape))
