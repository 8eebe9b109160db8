#!/usr/bin/env python3
"""Round 5: the EXACT CU-time of the scan launches inside bench.py's timed region, from per-workgroup residency stamps
(`make -C spiking_fullsubnet_amd/csrc EXTRA=-DSFSN_EXPERIMENTS`: every scan workgroup of the region's geometry stamps its first and
last instruction on the 100 MHz clock).  A trace's kernel duration runs from the first workgroup's start to the last one's end; a
scan workgroup needs a whole compute unit, so inside the region the workgroups of a launch start as units fall free.  Per kernel:
residency per workgroup (= time per step x T, the figure that can be compared with the kernel alone on the chip), the spread of the
starts, and the CU-ms per forward that result -- against the CU-ms the chip has per step."""
import ctypes, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
from spiking_fullsubnet_amd import _lib
_exp = os.path.join(ROOT, "spiking_fullsubnet_amd", "csrc_exp", "libsfsn_hip.so")  # (an EXPERIMENTS build beside the product's: scripts/build_exp_lib.sh)
if os.path.exists(_exp):
    _lib.LIB_PATH = _exp

B, T, LANES, STEPS, WARM = 64, 1000, int(os.environ.get("LANES", 12)), 36, 12
dev = torch.device("cuda", 0)
L = _lib.lib()
assert hasattr(L, "sfsn_debug_wg_times"), "needs an EXPERIMENTS build of the library (make EXTRA=-DSFSN_EXPERIMENTS)"
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
inputs = [model._stft(torch.from_numpy(rw.synth_wave(B, T, seed=i)).to(dev)).contiguous() for i in range(LANES)]
lanes = [torch.cuda.Stream(device=dev) for _ in range(LANES)]
eng.rows_per_wg, eng.stack_rows_fb_auto, eng.overlap_chunks = (8, 16), 8, 0
want_layers = os.environ.get("LAYERS", "1") != "0"
NAMES = {1: "gsn_scan_kernel (plain 16-row scan, groups 1-2 layer 1)", 2: "gsn_scan_fused_kernel (layer 2)", 3: "gsn_scan_fusedx_kernel (group 0 layer 1)",
         4: "gsn_stack_kernel (full-band stack, 8 rows)", 6: "gsn_stack_wide_kernel (sub-band layers side by side)"}


def run(n_lanes, steps, warm):
    for i in range(warm):
        with torch.cuda.stream(lanes[i % n_lanes]):
            eng.forward_stft(inputs[i % n_lanes], want_layers=want_layers, pipeline=False)
    torch.cuda.synchronize()
    cap = 400 * steps
    buf = torch.zeros((cap, 2), dtype=torch.int64, device=dev)
    L.sfsn_debug_wg_times(buf.data_ptr(), cap)
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(lanes[i % n_lanes]):
            eng.forward_stft(inputs[i % n_lanes], want_layers=want_layers, pipeline=False)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    log = (ctypes.c_int * (3 * 16384))()
    n = L.sfsn_debug_wg_log(log, 16384)
    L.sfsn_debug_wg_times(None, 0)  # (also clears the log)
    st = buf.cpu().numpy()
    out = {}
    for r in range(n):
        kind, base, nb = log[3 * r], log[3 * r + 1], log[3 * r + 2]
        if kind == 6:  # the wide (pair) launch: stamps of nb / 33 workgroups, then per-wave stall counters
            nb = nb // 33
        if kind == 5:  # the IO-wave full-band stack: its slice holds the stamps of nb / 25 workgroups, then the per-wave stall counters
            kind, nb = 4, nb // 25
        s, e = st[base:base + nb, 0], st[base:base + nb, 1]
        ok = (s > 0) & (e > 0)  # (padding blocks of a stack launch never stamp)
        s, e = s[ok], e[ok]
        d = out.setdefault(kind, dict(res=[], spread=[], span=[], wgs=[], slot_delay=[[] for _ in range(8)], slot_n=[0] * 8))
        idx = np.nonzero(ok)[0]
        for j, ss in zip(idx, s):  # workgroup j of the launch sits on XCD (j mod 8) (the dispatcher deals workgroups round-robin over the eight XCDs)
            d["slot_delay"][j % 8].append((ss - s.min()) / 100.0)
            d["slot_n"][j % 8] += 1
        d["res"].append((e - s) / 100.0)                 # us per workgroup
        d["spread"].append((s.max() - s.min()) / 100.0)  # us between the first and the last start
        d["span"].append((e.max() - s.min()) / 100.0)    # us: what a trace calls the kernel's duration
        d["wgs"].append(int(ok.sum()))
    rep = dict(lanes=n_lanes, steps=steps, ms_per_step=round(1e3 * wall / steps, 4), Mframes_per_s=round(B * T * steps / wall / 1e6, 2), kernels=[])
    tot = 0.0
    for kind, d in sorted(out.items()):
        res = np.concatenate(d["res"])
        cu_ms = float(res.sum()) / 1e3 / steps
        tot += cu_ms
        rep["kernels"].append(dict(kernel=NAMES.get(kind, str(kind)), launches=len(d["span"]), workgroups=int(np.median(d["wgs"])),
                                   residency_ms_median=round(float(np.median(res)) / 1e3, 4), residency_ms_p10=round(float(np.percentile(res, 10)) / 1e3, 4),
                                   residency_ms_p90=round(float(np.percentile(res, 90)) / 1e3, 4), us_per_step_median=round(float(np.median(res)) / T, 4),
                                   start_spread_ms_median=round(float(np.median(d["spread"])) / 1e3, 4), start_spread_ms_p90=round(float(np.percentile(d["spread"], 90)) / 1e3, 4),
                                   launch_span_ms_median=round(float(np.median(d["span"])) / 1e3, 4), cu_ms_per_forward=round(cu_ms, 1),
                                   workgroups_per_xcd_slot=[round(n_ / len(d["span"]), 2) for n_ in d["slot_n"]],
                                   mean_start_delay_ms_per_xcd_slot=[round(float(np.mean(v)) / 1e3, 3) if v else None for v in d["slot_delay"]]))
    rep["scan_cu_ms_per_forward"] = round(tot, 1)
    rep["cu_ms_available_per_forward"] = round(256 * 1e3 * wall / steps, 1)
    return rep


reports = [run(1, 6, 3), run(LANES, STEPS, WARM)]
eng.check_stack_errors()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(dict(B=B, T=T, layer_outputs="fp32 spike tensors" if want_layers else "none", reports=reports),
          open(os.path.join(ROOT, "gpurun_out", "wgtimes_r05%s.json" % ("" if want_layers else "_nolayers")), "w"), indent=1)
for rep in reports:
    print(f"--- {rep['lanes']} forward(s) in flight: {rep['ms_per_step']} ms per step = {rep['Mframes_per_s']} M frames/s; scans {rep['scan_cu_ms_per_forward']} CU-ms of "
          f"{rep['cu_ms_available_per_forward']} per forward")
    for k in rep["kernels"]:
        print("  %-58s wgs %3d  residency %.3f ms (p10 %.3f, p90 %.3f) = %.3f us/step | starts spread %.3f ms (p90 %.3f) | span %.3f ms | %.1f CU-ms/fwd" % (
            k["kernel"], k["workgroups"], k["residency_ms_median"], k["residency_ms_p10"], k["residency_ms_p90"], k["us_per_step_median"],
            k["start_spread_ms_median"], k["start_spread_ms_p90"], k["launch_span_ms_median"], k["cu_ms_per_forward"]))
        print("      workgroups per XCD slot (blockIdx mod 8):", k["workgroups_per_xcd_slot"], " mean start delay (ms):", k["mean_start_delay_ms_per_xcd_slot"])
