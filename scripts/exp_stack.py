"""First-contact / timing script for the layer-pipelined stack scan: bit-identity against the per-layer path and
single-stream timings.  Run on the GPU box: python scripts/exp_stack.py [B T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg

DEV = "cuda:0"
def build(front, kw, sd):
    cls = pkg.SpikingFullSubNet if front == "live" else pkg.Separator
    m = cls(**kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.eval().to(DEV)

def same(a, b):
    ok = torch.equal(torch.view_as_real(a["enh_stft"]), torch.view_as_real(b["enh_stft"]))
    bad = []
    for i, (x, y) in enumerate(zip(a["fb_all"] + sum(a["sb_all"], []), b["fb_all"] + sum(b["sb_all"], []))):
        if not torch.equal(x, y):
            bad.append((i, int((x != y).sum().item()), tuple(x.shape)))
    return ok and not bad, bad

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T = int(sys.argv[2]) if len(sys.argv) > 2 else 70
cases = [("live", rw.LIVE_M, 5), ("frozen", rw.FROZEN_S, 6), ("live", rw.LIVE_TINY, 11)]
if os.environ.get("ONLY_M"): cases = cases[:1]
for front, kw, seed in cases:
    sd = rw.live_state_dict(kw, seed) if front == "live" else rw.frozen_state_dict(kw, seed)
    model = build(front, kw, sd)
    stft = model._stft(torch.from_numpy(rw.synth_wave(B, T, seed)).to(DEV))
    eng = model.engine()
    eng.stack_scan = False
    ref = eng.forward_stft(stft); torch.cuda.synchronize()
    eng.stack_scan = True
    for rp in ((4, 8), (8, 16), (16, 4)):
        eng.stack_rows_per_wg = {"fb": rp[0], "sb": rp[1]}
        out = eng.forward_stft(stft); eng.check_stack_errors()
        ok, bad = same(ref, out)
        print(front, kw.get("fb_hidden_size"), "rpw", rp, "bit-identical" if ok else f"MISMATCH {bad[:6]}", "launches", eng.launches, flush=True)
    eng.stack_rows_per_wg = {"fb": 4, "sb": 8}
    rps = [tuple(int(v) for v in a.split(",")) for a in os.environ.get("RPS", "4,8").split(";")]
    for mode in (False, True, "narrow"):
        eng.stack_wide = mode != "narrow"
        mode = bool(mode)
        for rp in (rps if mode else [(0, 0)]):
            eng.stack_scan = mode
            if mode: eng.stack_rows_per_wg = {"fb": rp[0], "sb": rp[1]}
            for _ in range(3): eng.forward_stft(stft)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): eng.forward_stft(stft)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            eng.timers, eng.timer_tags = {}, {"scan:sb", "scan:fb", "stack:sb", "stack:fb", "scanx:sb", "scanf:sb"}
            for _ in range(3): eng.forward_stft(stft)
            tm = eng.timer_summary(); eng.timers = None
            print(f"   stack={mode} wide={eng.stack_wide} rpw={rp}: {dt*1e3:.3f} ms per forward (B={B}, T={T}); scans: " +
                  ", ".join(f"{k} {v['min_ms']:.3f} ms x{v['n']//3}" for k, v in tm.items()), flush=True)
