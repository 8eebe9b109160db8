"""Round 6: chunk lengths of the strict forward's overlapped schedule (Engine.overlap_fracs) with the one-launch sub-band epilogue.
B = 64, T = 1000, live baseline_m; ms per forward, three interleaved rounds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg

B, T = int(os.environ.get("B", 64)), int(os.environ.get("T", 1000))
dev = torch.device("cuda:0")
kw = rw.LIVE_M
sd = rw.live_state_dict(kw, 21)
m = pkg.SpikingFullSubNet(**kw)
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
m = m.eval().to(dev)
x = m._stft(torch.from_numpy(rw.synth_wave(B, T, seed=0)).to(dev)).contiguous()
eng = m.engine()
CASES = [None, (0.24, 0.38, 0.38), (0.20, 0.40, 0.40), (0.24, 0.42, 0.34), (0.24, 0.44, 0.32), (0.28, 0.40, 0.32), (0.22, 0.30, 0.28, 0.20),
         (0.24, 0.28, 0.26, 0.22), (0.20, 0.30, 0.30, 0.20), (0.30, 0.40, 0.30), (0.24, 0.46, 0.30), (0.18, 0.28, 0.28, 0.26)]
def strict(lean, n=8):
    for _ in range(2): eng.forward_stft(x, pipeline=False, want_layers=not lean, want_counts=lean)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): eng.forward_stft(x, pipeline=False, want_layers=not lean, want_counts=lean)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
res = {c: [] for c in CASES}
for rnd in range(3):
    for c in CASES:
        eng.overlap_fracs = list(c) if c else None
        eng.overlap_chunks = len(c) if c else 3
        res[c].append((strict(False), strict(True)))
for c in CASES:
    a = np.array(res[c])
    print(f"{str(c):34s} api {a[:,0].min():.3f}-{a[:,0].max():.3f} (mean {a[:,0].mean():.3f})  lean {a[:,1].min():.3f}-{a[:,1].max():.3f} (mean {a[:,1].mean():.3f})", flush=True)
eng.check_stack_errors()
