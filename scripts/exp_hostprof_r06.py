"""Round 6: where the host's ~0.33 ms per enqueued forward go (timed region's configuration), cProfile."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
B, T = 64, 1000
dev = torch.device("cuda:0")
kw = rw.LIVE_M
m = pkg.SpikingFullSubNet(**kw)
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in rw.live_state_dict(kw, 21).items()}, strict=True)
m = m.to(dev).eval()
eng = m.engine()
eng.overlap_chunks = 0
eng.rows_per_wg = (8, 16)
eng.stack_rows_fb_auto = 8
xs = [m._stft(torch.from_numpy(rw.synth_wave(B, T, seed=i)).to(dev)).contiguous() for i in range(12)]
lanes = [torch.cuda.Stream(device=dev) for _ in range(12)]
want = os.environ.get("LEAN", "0") == "0"
def fwd(k):
    with torch.cuda.stream(lanes[k]):
        return eng.forward_stft(xs[k], want_layers=want, want_counts=not want, pipeline=False)
for k in range(24):
    fwd(k % 12)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(48):
    fwd(i % 12)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"enqueue {t_enq / 48 * 1e3:.3f} ms per forward, wall {t_all / 48 * 1e3:.3f} ms per forward ({B * T * 48 / t_all / 1e6:.1f} M frames/s)")
pr = cProfile.Profile()
pr.enable()
for i in range(48):
    fwd(i % 12)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
print(s.getvalue()[:5000])
