"""Host time to enqueue one forward (no synchronisation inside the loop) vs the device time per forward in the timed region."""
import os, sys, time, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import refweights as rw
import spiking_fullsubnet_amd as pkg
dev = torch.device("cuda")
kw = rw.LIVE_M
m = pkg.SpikingFullSubNet(**kw); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in rw.live_state_dict(kw, 21).items()}); m = m.eval().to(dev)
eng = m.engine(); eng.rows_per_wg = (4, 16)
stft = m._stft(torch.from_numpy(rw.synth_wave(64, 1000, 0)).to(dev))
lanes = [torch.cuda.Stream(device=dev) for _ in range(12)]
for s_ in lanes:
    with torch.cuda.stream(s_): eng.forward_stft(stft)
torch.cuda.synchronize()
ts = []
t_all = time.perf_counter()
for i in range(96):
    t0 = time.perf_counter()
    with torch.cuda.stream(lanes[i % 12]): eng.forward_stft(stft)
    ts.append(time.perf_counter() - t0)
t_enq = time.perf_counter() - t_all
torch.cuda.synchronize()
t_tot = time.perf_counter() - t_all
ts = np.array(ts) * 1e3
print("enqueue per forward: median %.3f ms, mean %.3f ms, max %.3f ms; all 96 enqueued in %.1f ms, finished in %.1f ms (%.3f ms per forward)" % (np.median(ts), ts.mean(), ts.max(), t_enq * 1e3, t_tot * 1e3, t_tot * 1e3 / 96))
