#!/usr/bin/env python3
"""Round 5: why does the full-band stack launch take 1.7-1.8 us per frame beside the sub-band pair launch (the strict forward's
critical chain) when it takes 1.1-1.3 alone?

Records the stack launches of one strict forward (three chunks), then replays the full-band launch of a middle chunk (a) alone and
(b) beside two back-to-back replays of the sub-band pair launch of the same chunk, for round 2's body (SFSN_STACK_FB3=0) and the
IO-wave kernel (SFSN_STACK_FB3=1).  With an EXPERIMENTS build (scripts/build_exp_lib.sh) the IO-wave kernel's waves account their
stalls (S3_PB_* in sfsn_scan3_dev.h): cycles at the step barrier, in counted vmcnt waits, in hand-off polls -- per role and wave class."""
import ctypes, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
from spiking_fullsubnet_amd import _lib
_exp = os.path.join(ROOT, "spiking_fullsubnet_amd", "csrc_exp", "libsfsn_hip.so")
if os.path.exists(_exp) and not os.environ.get("NO_EXP") and not os.environ.get("SFSN_LIB_PATH"):
    _lib.LIB_PATH = _exp

B, T = int(os.environ.get("B", 64)), 1000
dev = torch.device("cuda", 0)
L = _lib.lib()
probe = hasattr(L, "sfsn_debug_wg_times")
if probe:
    L.sfsn_debug_wg_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sfsn_debug_wg_times.restype = None
    L.sfsn_debug_wg_log.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sfsn_debug_wg_log.restype = ctypes.c_int
kw = rw.LIVE_M
sd = rw.live_state_dict(kw, 21)
model = pkg.SpikingFullSubNet(**kw)
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
model = model.eval().to(dev)
eng = model.engine()
stft = model._stft(torch.from_numpy(rw.synth_wave(B, T, seed=0)).to(dev)).contiguous()
want_layers = os.environ.get("LAYERS", "1") != "0"
for _ in range(2):
    eng.forward_stft(stft, want_layers=want_layers)
torch.cuda.synchronize()

rec = []
orig = eng._stage_stack


def wrap(*a, **k):
    rec.append((a, k))
    return orig(*a, **k)


eng._stage_stack = wrap
eng.forward_stft(stft, want_layers=want_layers)
torch.cuda.synchronize()
del eng._stage_stack
eng.check_stack_errors()
fb = [r for r in rec if r[0][5] == "fb"]
sb = [r for r in rec if r[0][5] == "sb"]
print("recorded stack launches:", [(r[0][5], r[0][2], r[0][3]) for r in rec], flush=True)
fbc, sbc = fb[1], sb[1]
NT = fbc[0][3]
sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def launch(call, stream):
    a, k = call
    a = list(a)
    a[4] = eng._handle(stream)
    with torch.cuda.stream(stream):
        eng._stage_stack(*a, **k)


def timed(beside, reps=6):
    fbt, sbt = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        if beside:
            e[2].record(sB)
            launch(sbc, sB)
            e[3].record(sB)
            launch(sbc, sB)
        e[0].record(sA)
        launch(fbc, sA)
        e[1].record(sA)
        torch.cuda.synchronize()
        fbt.append(e[0].elapsed_time(e[1]))
        if beside:
            sbt.append(e[2].elapsed_time(e[3]))
    return float(np.median(fbt)), (float(np.median(sbt)) if sbt else None)


def pair_alone(reps=6):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(sB)
        launch(sbc, sB)
        e1.record(sB)
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    return float(np.median(out))


def probes(beside):
    """one probed replay of the IO-wave kernel: {role: {wave class: per-frame cycles [barrier, vmcnt, polls, all]}}"""
    cap = 4096
    buf = torch.zeros((cap, 2), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    L.sfsn_debug_wg_times(buf.data_ptr(), cap)
    if beside:
        launch(sbc, sB)
        launch(sbc, sB)
    launch(fbc, sA)
    torch.cuda.synchronize()
    log = (ctypes.c_int * (3 * 64))()
    n = L.sfsn_debug_wg_log(log, 64)
    L.sfsn_debug_wg_times(None, 0)
    st = buf.cpu().numpy().reshape(-1)
    for r in range(n):
        kind, base, nb = log[3 * r], log[3 * r + 1], log[3 * r + 2]
        if kind != 5:
            continue
        blocks = nb // 25
        w = st[2 * base: 2 * base + 50 * blocks]
        stamps = w[:2 * blocks].reshape(blocks, 2)
        pb = w[2 * blocks:].reshape(blocks, 12, 4).astype(np.float64) / NT
        res = (stamps[:, 1] - stamps[:, 0]) / 100.0
        roles = [("layer-1 scan (publishes)", 0, 16), ("PROJ", 16, 24), ("layer-2 scan (gated)", 24, 40)] if blocks == 40 else \
                [("layer-1 scan (publishes)", 0, 8), ("PROJ", 8, 16), ("layer-2 scan (gated)", 16, 24)]
        for name, b0, b1 in roles:
            live = np.nonzero(stamps[b0:b1, 0] > 0)[0]  # (padding blocks never stamp)
            b0, b1 = b0 + int(live.min()), b0 + int(live.max()) + 1
            print(f"    {name} ({b1 - b0} workgroups): resident {res[b0:b1].mean():.0f} us per workgroup = {res[b0:b1].mean() / NT:.3f} us per frame")
            for cls, ws in ((("waves with tiles", slice(0, 5)),) if name == "PROJ" else
                            (("compute waves", slice(0, 10)), ("loader wave", slice(10, 11)), ("storer wave", slice(11, 12)))):
                m = pb[b0:b1, ws].mean(axis=(0, 1))
                mn = pb[b0:b1, ws, 0].min()
                print(f"       {cls:14s} per frame: at the barrier {m[0]:6.0f} clk (min over waves {mn:5.0f}), vmcnt waits {m[1]:6.0f}, hand-off polls {m[2]:6.0f}, all {m[3]:6.0f}")


def probes_pair(beside):
    """one probed replay of the pair launch (alone / beside the full-band launch): per role and wave class, cycles per frame"""
    cap = 16384
    buf = torch.zeros((cap, 2), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    L.sfsn_debug_wg_times(buf.data_ptr(), cap)
    if beside:
        launch(fbc, sA)
    launch(sbc, sB)
    torch.cuda.synchronize()
    log = (ctypes.c_int * (3 * 64))()
    n = L.sfsn_debug_wg_log(log, 64)
    L.sfsn_debug_wg_times(None, 0)
    st = buf.cpu().numpy().reshape(-1)
    for r in range(n):
        kind, base, nb = log[3 * r], log[3 * r + 1], log[3 * r + 2]
        if kind != 6:
            continue
        blocks = nb // 33
        w = st[2 * base: 2 * base + 66 * blocks]
        stamps = w[:2 * blocks].reshape(blocks, 2)
        pb = w[2 * blocks:].reshape(blocks, 16, 4).astype(np.float64) / NT
        res = (stamps[:, 1] - stamps[:, 0]) / 100.0
        Rs = [B * u for u in (8, 3, 2)]
        roles, b0 = [], 0
        for l in (1, 2):
            for g, R in enumerate(Rs):
                nbk = (R + 7) // 8
                roles.append((f"layer {l} group {g} ({'in-scan x product' if l == 1 and g == 0 else 'input term from memory' if l == 1 else 'in-scan spike product'})", b0, b0 + nbk))
                b0 = (b0 + nbk + 7) & ~7
        for name, b0, b1 in roles:
            print(f"    {name} ({b1 - b0} workgroups): resident {res[b0:b1].mean():.0f} us = {res[b0:b1].mean() / NT:.3f} us per frame")
            for cls, ws in (("compute waves", slice(0, 14)), ("loader wave", slice(14, 15)), ("storer wave", slice(15, 16))):
                m = pb[b0:b1, ws].mean(axis=(0, 1))
                mn = pb[b0:b1, ws, 0].mean(axis=0).min()
                print(f"       {cls:14s} per frame: at the barrier {m[0]:6.0f} clk (least-waiting wave {mn:5.0f}), vmcnt waits {m[1]:6.0f}, polls / bf16 split {m[2]:6.0f}, all {m[3]:6.0f}")


pa = pair_alone()
print(f"B={B}: chunk of {NT} frames; pair launch alone {1e3 * pa:.0f} us = {1e3 * pa / NT:.3f} us per frame", flush=True)
for fb3 in ("0", "1"):
    os.environ["SFSN_STACK_FB3"] = fb3
    for _ in (0,):
        timed(False, 2)
        a, _ = timed(False)
        b, s = timed(True)
        print(f"full-band stack, {'IO-wave kernel' if fb3 == '1' else 'round-2 bodies'}: alone {1e3 * a:.0f} us = {1e3 * a / NT:.3f} us per frame | "
              f"beside the pair launch {1e3 * b:.0f} us = {1e3 * b / NT:.3f} us per frame (the pair launch beside it: {1e3 * s:.0f} us = {1e3 * s / NT:.3f})", flush=True)
        if probe and fb3 == "1":
            for beside in (False, True):
                print(f"  stall ledger of the IO-wave kernel, {'beside the pair launch' if beside else 'alone'}:")
                probes(beside)
if probe:
    for beside in (False, True):
        print(f"stall ledger of the sub-band pair launch, {'beside the full-band launch' if beside else 'alone'}:")
        probes_pair(beside)
eng.check_stack_errors()
