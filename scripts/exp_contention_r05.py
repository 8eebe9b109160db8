#!/usr/bin/env python3
"""Round 5: why do the scan kernels execute 1.4-2x more cycles per step inside the twelve-lane region than alone (the clock is within 5 %)?

Replays ONE recorded launch of each scan kernel of the region's geometry (the fused layer-2 scan, the fused-x layer-1 scan, the plain
16-row scan, the full-band stack) under controlled company:
  alone          the launch by itself, four times
  x4 same        four copies of the SAME launch on four streams (208 / 128 / 80 / 96 workgroups resident: every busy CU runs the same
                 code -- the instruction caches are warm with one kernel; HBM sees 4x the kernel's own traffic)
  + copy         the launch beside a stream of 256 MiB device-to-device copies (HBM contention without any other kernel's code or CUs
                 beyond the copy kernel's)
  + other scans  the launch beside the OTHER three scan kernels (different code on neighbouring CUs, moderate traffic)
Durations from HIP events on the launch's own stream."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import refweights as rw
import spiking_fullsubnet_amd as pkg

B, T = 64, 1000
dev = torch.device("cuda", 0)
kw = rw.LIVE_M
sd = rw.live_state_dict(kw, 21)
model = pkg.SpikingFullSubNet(**kw)
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
model = model.eval().to(dev)
eng = model.engine()
stft = model._stft(torch.from_numpy(rw.synth_wave(B, T, seed=0)).to(dev)).contiguous()
eng.rows_per_wg = (8, 16)
eng.stack_rows_fb_auto = 8
eng.overlap_chunks = 0
want_layers = os.environ.get("LAYERS", "1") != "0"

# record the scan launches of one forward (their argument lists), then replay them
rec = {}
for name in ("_stage_scan_fused", "_stage_scan_fused_x", "_stage_scan", "_stage_stack"):
    orig = getattr(eng, name)

    def wrap(*a, _orig=orig, _name=name, **k):
        rec.setdefault(_name, (a, k))
        return _orig(*a, **k)
    setattr(eng, name, wrap)
res = eng.forward_stft(stft, want_layers=want_layers, pipeline=False)
torch.cuda.synchronize()
for name in list(rec):
    delattr(eng, name)  # back to the class's methods
print("recorded:", {k: len(v[0]) for k, v in rec.items()}, flush=True)

streams = [torch.cuda.Stream(device=dev) for _ in range(6)]


def launch(name, stream):
    a, k = rec[name]
    a = list(a)
    # the stream handle is a positional argument: find the ctypes void pointer that was the recording stream and swap it
    import ctypes
    h = eng._handle(stream)
    idx = {"_stage_scan_fused": 7, "_stage_scan_fused_x": 7, "_stage_scan": 9, "_stage_stack": 4}[name]
    a[idx] = h
    with torch.cuda.stream(stream):
        getattr(eng, name)(*a, **k)


def timed(name, company=None, reps=4):
    """mean duration (ms) of `name` on streams[0]; company(start) starts the other work first"""
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        stop = company() if company else None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(streams[0])
        launch(name, streams[0])
        e1.record(streams[0])
        streams[0].synchronize()
        if stop:
            stop()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    return float(np.mean(out)), float(np.min(out))


src = torch.empty((1 << 26,), dtype=torch.float32, device=dev)  # 256 MiB
dst = torch.empty_like(src)


def copies():
    with torch.cuda.stream(streams[5]):
        for _ in range(40):  # ~40 x 0.1 ms
            dst.copy_(src, non_blocking=True)
    return None


names = [n for n in ("_stage_scan_fused", "_stage_scan_fused_x", "_stage_scan", "_stage_stack") if n in rec]
print(f"layer outputs: {'fp32 spike tensors' if want_layers else 'none'}")
for n in names:
    alone = timed(n)

    def same(n=n):
        for s_ in streams[1:4]:
            launch(n, s_)
    x4 = timed(n, same)
    cp = timed(n, copies)

    def others(n=n):
        for s_, o in zip(streams[1:4], [m for m in names if m != n]):
            launch(o, s_)
    ot = timed(n, others)
    print("%-22s alone %.3f ms (min %.3f) | x4 same kernel %.3f (x%.2f) | beside D2D copies %.3f (x%.2f) | beside the other three scans %.3f (x%.2f)" % (
        n, alone[0], alone[1], x4[0], x4[0] / alone[0], cp[0], cp[0] / alone[0], ot[0], ot[0] / alone[0]), flush=True)
eng.check_stack_errors()
