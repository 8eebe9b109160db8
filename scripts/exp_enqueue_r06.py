"""Round 6: host time to enqueue one forward in the timed region's configuration (no overlap chunks, rows per workgroup (8, 16))."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
B, T = 64, 1000
dev = torch.device("cuda:0")
kw = rw.LIVE_M
m = pkg.SpikingFullSubNet(**kw)
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in rw.live_state_dict(kw, seed=3).items()}, strict=True)
m = m.to(dev).eval()
eng = m.engine()
eng.overlap_chunks = 0
eng.rows_per_wg = (8, 16)
rng = np.random.default_rng(0)
xs = [torch.from_numpy((0.05 * (rng.standard_normal((B, 257, T)) + 1j * rng.standard_normal((B, 257, T)))).astype(np.complex64)).to(dev) for _ in range(12)]
lanes = [torch.cuda.Stream(device=dev) for _ in range(12)]
for want in (True, False):
    def fwd(k):
        with torch.cuda.stream(lanes[k]):
            return eng.forward_stft(xs[k], want_layers=want, want_counts=not want, pipeline=False)
    for k in range(12):
        fwd(k)
    torch.cuda.synchronize()
    per = []
    t0 = time.perf_counter()
    for i in range(48):
        a = time.perf_counter(); fwd(i % 12); per.append(time.perf_counter() - a)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    per = np.array(per) * 1e3
    print(json.dumps(dict(want_layers=want, enqueue_ms_first12=[round(float(v), 3) for v in per[:12]], enqueue_ms_median=round(float(np.median(per)), 3),
                          enqueue_total_ms=round(t_enq * 1e3, 2), wall_ms=round(t_all * 1e3, 2), launches=dict(eng.launches) if hasattr(eng, "launches") else None)))
