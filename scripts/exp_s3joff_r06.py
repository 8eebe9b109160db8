"""Round 6: scan3j_role's OFF form (the IO waves compute the input terms of tiles 12 / 13) against the plain form (SFSN_S3J_OFF=0): the timed
region's geometry (8, 16), one forward at a time, whole-sequence launches, HIP-event time of the fused sub-band scan."""
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
ref = None
for rnd in range(3):
    for off in ("0", "2"):
        os.environ["SFSN_S3J_OFF"] = off
        for lean in (False, True):
            eng.timers, eng.timer_tags = {}, {"scanf:sb", "scanx:sb"}
            for _ in range(4): res = eng.forward_stft(x, pipeline=False, want_layers=not lean, want_counts=lean)
            s = eng.timer_summary(); eng.timers = None
            if not lean:
                if ref is None: ref = res["enh_mag"].clone()
                same = bool(torch.equal(ref, res["enh_mag"]))
            print(f"round {rnd} OFF={off} lean={lean}:", {k: round(v['mean_ms'], 4) for k, v in s.items()}, "" if lean else f"enh_mag identical: {same}", flush=True)
eng.check_stack_errors()
