"""Timing experiment (library built with -DSFSN_TIMING_EXPERIMENTS): sub-band scan launch time at 16 rows per workgroup with
the normal and the doubled (SFSN_SCAN_DEBUG_OUT=1027) matrix work per step, B=64 and B=256."""
import os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import refweights as rw
import spiking_fullsubnet_amd as pkg
dev = torch.device("cuda")
kw = rw.LIVE_M
sd = rw.live_state_dict(kw, 21)
m = pkg.SpikingFullSubNet(**kw); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.eval().to(dev)
eng = m.engine()
for B in (64, 256):
    stft = m._stft(torch.from_numpy(rw.synth_wave(B, 1000, 0)).to(dev))
    eng.rows_per_wg = (16, 16)
    eng.timers, eng.timer_tags = {}, {"scan:sb", "scan:fb"}
    for _ in range(3): eng.forward_stft(stft)
    print("B", B, "DEBUG_OUT", os.environ.get("SFSN_SCAN_DEBUG_OUT"), {k: round(v["min_ms"], 4) for k, v in eng.timer_summary().items()})
    eng.timers = None; eng._ws.clear(); torch.cuda.empty_cache()
