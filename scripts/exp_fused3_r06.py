"""Round 6: the 16-row fused-input scan with IO waves (scan3j_role) against round 2's body (SFSN_FUSED_V2=1): the timed region's
geometry (8, 16), one forward at a time, whole-sequence launches, HIP-event time of the scan groups."""
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
eng.rows_per_wg, eng.stack_rows_fb_auto, eng.overlap_chunks = (8, 16), 8, 0
for rnd in range(2):
    for v2 in ("1", ""):
        if v2: os.environ["SFSN_FUSED_V2"] = v2
        else: os.environ.pop("SFSN_FUSED_V2", None)
        for lean in (False, True):
            eng.timers, eng.timer_tags = {}, {"scanf:sb", "scanx:sb", "scan:sb", "stack:fb"}
            for _ in range(4): eng.forward_stft(x, pipeline=False, want_layers=not lean, want_counts=lean)
            s = eng.timer_summary(); eng.timers = None
            print(f"round {rnd} body={'round 2' if v2 else 'IO waves'} lean={lean}:", {k: round(v['mean_ms'], 4) for k, v in s.items()}, flush=True)
eng.check_stack_errors()
