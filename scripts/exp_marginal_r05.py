#!/usr/bin/env python3
"""Round 5: what does each kernel class COST the twelve-lane region?  (the marginal ledger)

The per-workgroup stamps (scripts/exp_wgtimes_r05.py) say the scan workgroups hold 196 of the 422 CU-ms a step has; the rest is the
time-parallel kernels (features, input products, projections, deep filter) plus whatever stays idle -- their workgroups share
compute units, so stamps cannot price them.  This script records every C-ABI call of one forward per lane (a recording proxy around
the ctypes library; every entry point takes the stream last, and the region's geometry runs a lane's forward on one stream) and
replays the twelve lanes' call lists round-robin, 36 forwards, with one class of calls left out at a time: the drop in ms per forward
x 256 CUs is that class's marginal CU-cost inside the region.  (Results of a replay with calls left out are garbage -- timing only.)"""
import ctypes, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import refweights as rw
import spiking_fullsubnet_amd as pkg

B, T, LANES, STEPS = 64, 1000, int(os.environ.get('LANES', 12)), int(os.environ.get('STEPS', 36))
dev = torch.device("cuda", 0)
kw = rw.LIVE_M
sd = rw.live_state_dict(kw, 21)
model = pkg.SpikingFullSubNet(**kw)
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
model = model.eval().to(dev)
eng = model.engine()
eng.rows_per_wg, eng.stack_rows_fb_auto, eng.overlap_chunks = (8, 16), 8, 0
want_layers = os.environ.get("LAYERS", "1") != "0"
inputs = [model._stft(torch.from_numpy(rw.synth_wave(B, T, seed=i)).to(dev)).contiguous() for i in range(LANES)]
lanes = [torch.cuda.Stream(device=dev) for _ in range(LANES)]
real = eng.lib


class Recorder:
    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        fn = getattr(real, name)
        if not name.startswith("sfsn_") or name in ("sfsn_stack_scratch_bytes", "sfsn_strerror"):
            return fn

        def wrapped(*a):
            self.calls.append((name, a))
            return fn(*a)
        return wrapped


keep, lists = [], []
for k in range(LANES):
    for _ in range(2):  # (the second forward of a lane is the one recorded: its workspaces exist)
        rec = Recorder()
        eng.lib = rec
        with torch.cuda.stream(lanes[k]):
            keep.append(eng.forward_stft(inputs[k], want_layers=want_layers, pipeline=False))
        eng.lib = real
    lists.append(rec.calls)
torch.cuda.synchronize()
eng.check_stack_errors()
names = sorted({n for n, _ in lists[0]})
print("calls per forward:", {n: sum(1 for m, _ in lists[0] if m == n) for n in names}, flush=True)


def klass(n):
    if "scan" in n:
        return "scans"
    if "features" in n:
        return "features"
    if "input_proj" in n:
        return "input products (layer 0)"
    if "spike_proj" in n:
        return "projections (spike products)"
    if "deepfilter" in n:
        return "deep filter"
    return "other"


def replay(skip=(), only=None, steps=STEPS, warm=12):
    def go(n):
        for i in range(n):
            for name, a in lists[i % LANES]:
                c = klass(name)
                if c in skip or (only is not None and c not in only):
                    continue
                rc = getattr(real, name)(*a)
                assert rc == 0, (name, rc)
    go(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(steps)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


classes = ["scans", "features", "input products (layer 0)", "projections (spike products)", "deep filter"]
full = min(replay() for _ in range(2))
out = dict(B=B, T=T, lanes=LANES, steps=STEPS, layer_outputs="fp32 spike tensors" if want_layers else "none", ms_per_forward_all=round(full, 4),
           cu_ms_available=round(256 * full, 1), classes=[])
print(f"all calls: {full:.4f} ms per forward = {B * T / full / 1e3:.2f} M frames/s  ({256 * full:.1f} CU-ms per forward)")
for c in (classes if not os.environ.get("QUICK") else classes[:1]):
    w = min(replay(skip=(c,)) for _ in range(2))
    o = min(replay(only=(c,)) for _ in range(2))
    out["classes"].append(dict(klass=c, ms_without=round(w, 4), marginal_ms=round(full - w, 4), marginal_cu_ms=round(256 * (full - w), 1),
                               ms_alone_in_12_lanes=round(o, 4), cu_ms_alone_in_12_lanes=round(256 * o, 1)))
    print("  without %-30s %.4f ms  -> marginal %.4f ms = %5.1f CU-ms | only this class on the twelve lanes: %.4f ms = %5.1f CU-ms" % (
        c, w, full - w, 256 * (full - w), o, 256 * o), flush=True)
tp = [c for c in classes if c != "scans"]
o = min(replay(only=tuple(tp)) for _ in range(2))
out["time_parallel_only_ms"] = round(o, 4)
print(f"  only the time-parallel kernels: {o:.4f} ms = {256 * o:.1f} CU-ms; only the scans: {out['classes'][0]['ms_alone_in_12_lanes']} ms")
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "marginal_r05%s.json" % ("" if want_layers else "_nolayers")), "w"), indent=1)
