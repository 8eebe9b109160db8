#!/usr/bin/env python3
"""Round 5, review item 1: what is bench.py's timed region (twelve forwards in flight) short of?

Measures the SHADER CLOCK (scripts/micro/clock_probe.hip: `s_memtime` against the 100 MHz `s_memrealtime`, one resident wave per
workgroup, launched before the work under test) while
  idle     nothing else runs,
  alone    one forward at a time in the region's launch geometry (full-band stack at 8, sub-band scans at 16 rows per workgroup,
           whole-sequence launches): every scan kernel alone on the chip,
  strict   one forward at a time in the engine's default schedule (pair launch + full-band stack on two streams),
  region   `--inflight` forwards in flight on as many streams (bench.py's timed region),
and the HIP-event duration of every scan launch group in `alone` and in `region`, so that

  cycles per step = duration x clock / T

can be compared: equal cycles, lower clock -> the region is clock (power) bound and only fewer instructions help; more cycles ->
the scans wait longer for something (memory latency under load, LDS-DMA starved) inside the region.

Writes gpurun_out/diag_r05/clock.json and prints a digest.  scripts/ledger_r05.py does the CU-time ledger from kernel traces.
"""
import ctypes, faulthandler, json, os, sys, time
faulthandler.dump_traceback_later(int(os.environ.get("DIAG_WATCHDOG_S", "240")), exit=True)  # a hang prints where, then exits
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import refweights as rw
import spiking_fullsubnet_amd as pkg

B, T = int(os.environ.get("B", 64)), int(os.environ.get("T", 1000))
LANES = int(os.environ.get("LANES", 12))
STEPS = int(os.environ.get("STEPS", 36))
OUT = os.path.join(ROOT, "gpurun_out", "diag_r05")
os.makedirs(OUT, exist_ok=True)
dev = torch.device("cuda", 0)
so = os.path.join(ROOT, "scripts", "micro", "clock_probe.so")
if not os.path.exists(so):
    os.system(f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o {so} {so[:-3]}.hip")
probe = ctypes.CDLL(so)
probe.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

kw = rw.LIVE_M
sd = rw.live_state_dict(kw, 21)
model = pkg.SpikingFullSubNet(**kw)
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
model = model.eval().to(dev)
eng = model.engine()


def make_input(lane):
    wave = torch.from_numpy(rw.synth_wave(B, T, seed=lane)).to(dev)
    return model._stft(wave).contiguous()


inputs = [make_input(i) for i in range(LANES)]
lanes = [torch.cuda.Stream(device=dev) for _ in range(LANES)]
pstream = torch.cuda.Stream(device=dev)
WGS, INTERVAL = 4, 5000  # 4 probe workgroups (the strict forward needs 244-248 of the 256 CUs for its two resident launches), one sample per 50 us


def set_geometry(g):
    eng.rows_per_wg = g
    eng.stack_rows_fb_auto = g[0] if g[0] in (4, 8, 16) else 4


def say(msg):
    print(f"[{time.perf_counter():9.3f}] {msg}", file=sys.stderr, flush=True)


def run_with_probe(name, fn, est_ms, want_layers=True):
    say("phase " + name)
    n = int(est_ms * 1e3 / 50) + 40
    buf = torch.zeros((WGS, n, 4), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    rc = probe.clock_probe_launch(buf.data_ptr(), WGS, n, INTERVAL, pstream.cuda_stream)
    assert rc == 0, rc
    time.sleep(0.0005)  # the probe's workgroups take their slots first
    t0 = time.perf_counter()
    fn()
    for s_ in lanes:
        s_.synchronize()
    torch.cuda.current_stream(dev).synchronize()
    wall = time.perf_counter() - t0
    torch.cuda.synchronize()
    a = buf.cpu().numpy().astype(np.uint64)
    mt, rt = a[:, :, 0].astype(np.int64), a[:, :, 1].astype(np.int64)
    chain_m = (a[:, :, 2] >> np.uint64(32)).astype(np.int64)
    chain_r = (a[:, :, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    xcc = ((a[:, 0, 3] >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
    hw = (a[:, 0, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    # samples taken while the work was in flight: real time within [start + 0.2 ms, start + wall]
    start = rt[:, 0].min()
    busy_hi = start + int(wall * 1e8)
    res = dict(phase=name, wall_ms=round(1e3 * wall, 3), probe_workgroups=WGS, sample_interval_us=INTERVAL / 100, per_wg=[])
    allmhz = []
    for w in range(WGS):
        d_m, d_r = np.diff(mt[w]), np.diff(rt[w])
        ok = (rt[w, 1:] > start + 20000) & (rt[w, 1:] < busy_hi) & (d_r > 0)
        if ok.sum() < 3:
            continue
        mhz = 100.0 * d_m[ok] / d_r[ok]
        allmhz.append(mhz)
        cm = chain_m[w, 1:][ok]
        cr = chain_r[w, 1:][ok]
        res["per_wg"].append(dict(xcc=int(xcc[w]), hw_id=int(hw[w]), cu=int((hw[w] >> 8) & 0xF), se=int((hw[w] >> 13) & 0x7), samples=int(ok.sum()),
                                  mhz_median=round(float(np.median(mhz)), 1), mhz_p10=round(float(np.percentile(mhz, 10)), 1),
                                  mhz_p90=round(float(np.percentile(mhz, 90)), 1),
                                  chain_cycles_median=int(np.median(cm)), chain_ticks100MHz_median=float(np.median(cr))))
    if allmhz:
        m = np.concatenate(allmhz)
        res.update(mhz_median=round(float(np.median(m)), 1), mhz_mean=round(float(m.mean()), 1), mhz_p10=round(float(np.percentile(m, 10)), 1),
                   mhz_p90=round(float(np.percentile(m, 90)), 1), mhz_min=round(float(m.min()), 1), mhz_max=round(float(m.max()), 1))
    return res


def fwd(x, want_layers=True):
    return eng.forward_stft(x, want_layers=want_layers, pipeline=False)


results = []
SCAN_TAGS = {"scan:sb", "scan:fb", "stack:sb", "stack:fb", "scanx:sb", "scanf:sb"}

# warm-up (allocations, LDS attributes).  The strict schedule (pair launch + full-band stack: workgroups that wait for each other inside a
# launch) runs ALONE on the chip, one forward at a time on the main stream -- never on the lanes: a dozen such launches side by side
# cannot be resident together (README, limits).  The lanes are warmed in the region's geometry only.
say("inputs made; warm-up of the strict schedule")
set_geometry((0, 0))
eng.overlap_chunks = 3
for _ in range(2):
    fwd(inputs[0])
    fwd(inputs[0], False)
torch.cuda.synchronize()
say("warm-up of the lanes")
set_geometry((8, 16))
eng.overlap_chunks = 0
for want in (True, False):
    for s_, x_ in zip(lanes, inputs):
        with torch.cuda.stream(s_):
            fwd(x_, want)
    fwd(inputs[0], want)
    torch.cuda.synchronize()

# ---- idle
results.append(run_with_probe("idle", lambda: time.sleep(0.01), 12))

for want_layers in (True, False):
    sfx = "" if want_layers else "_nolayers"
    # ---- alone: the region's geometry, one forward at a time, whole-sequence launches, scan groups timed
    set_geometry((8, 16))
    eng.overlap_chunks = 0
    eng.timers, eng.timer_tags = {}, SCAN_TAGS
    r = run_with_probe("alone" + sfx, lambda: [fwd(inputs[0], want_layers) for _ in range(8)], 8 * 6.5)
    r["scan_groups_ms"] = {k: round(v["mean_ms"], 4) for k, v in eng.timer_summary().items()}
    r["ms_per_forward"] = round(r["wall_ms"] / 8, 4)
    eng.timers = None
    results.append(r)

    # ---- strict: the engine's default schedule for a forward alone
    set_geometry((0, 0))
    eng.overlap_chunks = 3
    say("phase strict" + sfx)
    # (no probe here: its stream would be a third one beside the two resident-workgroup launches of the schedule -- README, limits)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(12):
        fwd(inputs[0], want_layers)
    torch.cuda.synchronize()
    results.append(dict(phase="strict" + sfx, wall_ms=round(1e3 * (time.perf_counter() - t0), 3), ms_per_forward=round(1e3 * (time.perf_counter() - t0) / 12, 4)))

    # ---- region: LANES forwards in flight (bench.py's timed region), scan groups timed on their lane streams
    set_geometry((8, 16))
    eng.overlap_chunks = 0

    def region(timed):
        def go():
            for i in range(STEPS):
                k = i % LANES
                with torch.cuda.stream(lanes[k]):
                    fwd(inputs[k], want_layers)
        return go
    eng.timers, eng.timer_tags = None, None
    r0 = run_with_probe("region_untimed" + sfx, region(False), STEPS * 1.9)
    r0["ms_per_step"] = round(r0["wall_ms"] / STEPS, 4)
    r0["Mframes_per_s"] = round(B * T * STEPS / r0["wall_ms"] / 1e3, 2)
    results.append(r0)
    eng.timers, eng.timer_tags = {}, SCAN_TAGS
    r = run_with_probe("region" + sfx, region(True), STEPS * 1.9)
    r["scan_groups_ms"] = {k: round(v["mean_ms"], 4) for k, v in eng.timer_summary().items()}
    r["ms_per_step"] = round(r["wall_ms"] / STEPS, 4)
    r["Mframes_per_s"] = round(B * T * STEPS / r["wall_ms"] / 1e3, 2)
    eng.timers = None
    results.append(r)
eng.check_stack_errors()

by = {r["phase"]: r for r in results}
digest = []
for r in results:
    digest.append("%-22s wall %8.2f ms  clock median %7.1f MHz (p10 %7.1f, p90 %7.1f, min %7.1f)%s" % (
        r["phase"], r["wall_ms"], r.get("mhz_median", 0), r.get("mhz_p10", 0), r.get("mhz_p90", 0), r.get("mhz_min", 0),
        ("  %.2f M frames/s" % r["Mframes_per_s"]) if "Mframes_per_s" in r else (("  %.3f ms/forward" % r["ms_per_forward"]) if "ms_per_forward" in r else "")))
for sfx in ("", "_nolayers"):
    a, g = by.get("alone" + sfx), by.get("region" + sfx)
    if a and g and "mhz_median" in a and "mhz_median" in g:
        digest.append(f"--- cycles per step{sfx}: duration x median clock / T   (alone | in the region | ratio of cycles | ratio of time)")
        for k in sorted(a["scan_groups_ms"]):
            if k in g["scan_groups_ms"]:
                ca = a["scan_groups_ms"][k] * 1e-3 * a["mhz_median"] * 1e6 / T
                cg = g["scan_groups_ms"][k] * 1e-3 * g["mhz_median"] * 1e6 / T
                digest.append("  %-10s %7.0f clk (%.3f ms) | %7.0f clk (%.3f ms) | x%.3f | x%.3f" % (
                    k, ca, a["scan_groups_ms"][k], cg, g["scan_groups_ms"][k], cg / ca, g["scan_groups_ms"][k] / a["scan_groups_ms"][k]))
json.dump(dict(B=B, T=T, lanes=LANES, steps=STEPS, device=torch.cuda.get_device_name(0), results=results, digest=digest),
          open(os.path.join(OUT, "clock.json"), "w"), indent=1)
print("\n".join(digest))
