import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
DEV = "cuda:0"
kw = rw.LIVE_M; sd = rw.live_state_dict(kw, 21)
m = pkg.SpikingFullSubNet(**kw); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True); m = m.eval().to(DEV)
stft = m._stft(torch.from_numpy(rw.synth_wave(64, 1000, 3)).to(DEV)); eng = m.engine()
for rp in (4, 8, 16):
    eng.stack_rows_fb_auto = rp
    for ov in (0, 3):
        eng.overlap_chunks = ov
        for _ in range(3): eng.forward_stft(stft)
        eng.timers, eng.timer_tags = {}, {"stack:fb", "stack:sb"}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): eng.forward_stft(stft)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
        ts = eng.timer_summary(); eng.timers = None
        print(f"full-band stack rows/wg {rp:2d}, chunks {ov}: forward {dt*1e3:.3f} ms; stack:fb {ts.get('stack:fb')}, stack:sb {ts.get('stack:sb')}", flush=True)
