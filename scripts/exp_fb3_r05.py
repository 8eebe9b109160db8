"""Round 5: the full-band stack with IO-specialised waves (sfsn_scan3w_dev.h) against round 2's body (SFSN_STACK_FB_V2=1): the stack alone
as one whole-sequence launch and the strict forward (three chunks), full-band rows per workgroup 4 and 8; 12 forwards each, medians."""
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
DEV = "cuda:0"
kw = rw.LIVE_M; sd = rw.live_state_dict(kw, 21)
m = pkg.SpikingFullSubNet(**kw); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True); m = m.eval().to(DEV)
stft = m._stft(torch.from_numpy(rw.synth_wave(64, 1000, 3)).to(DEV)); eng = m.engine()
out = []
for rp in (4, 8):
    eng.stack_rows_fb_auto = rp
    for ov in (0, 3):
        eng.overlap_chunks = ov
        for _ in range(3): eng.forward_stft(stft)
        ts_ = []
        eng.timers, eng.timer_tags = ({}, {"stack:fb", "stack:sb"}) if ov == 0 else (None, None)
        for _ in range(12):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.forward_stft(stft)
            torch.cuda.synchronize(); ts_.append(time.perf_counter() - t0)
        s = eng.timer_summary() if ov == 0 else {}
        eng.timers = None
        out.append(f"rows {rp} chunks {ov}: forward median {np.median(ts_)*1e3:.3f} ms (min {np.min(ts_)*1e3:.3f})" + (f"; stack:fb {s['stack:fb']['mean_ms']:.3f} ms, stack:sb {s['stack:sb']['mean_ms']:.3f}" if s else ""))
print(os.environ.get("SFSN_LIB_PATH", "default build") + (" FB_V2" if os.environ.get("SFSN_STACK_FB_V2") else ""))
print("\n".join("   " + o for o in out))
eng.check_stack_errors()
