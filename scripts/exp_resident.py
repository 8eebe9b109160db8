"""Resident streaming launch (sfsn_stream_hop_resident): watchdog exit, stop latency, per-hop latency against one launch per hop."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import refweights as rw
from spiking_fullsubnet_amd import SpikingFullSubNet  # noqa

def build(kw, seed):
    from test_hip_parity import build_module
    return build_module("live", kw, rw.live_state_dict(kw, seed))

import os, ctypes
model = build(rw.LIVE_M, 5)

def stage_table(sess):
    L, desc = sess.eng.lib, sess._hop["desc"]
    out = (ctypes.c_int * 80)()
    ns = L.sfsn_hop_stages(ctypes.byref(desc), out, 20)
    raw = sess._hop["scratch"].cpu().numpy().view(np.uint8)
    nwg = out[4 * (ns - 1) + 2] + out[4 * (ns - 1) + 3]
    st = raw[64:64 + nwg * 8 * 64].view(np.uint64).reshape(nwg, 8, 8).astype(np.int64)
    t0 = st[:, :, 0][st[:, :, 0] > 0].min()
    us = (st - t0) / 100.0
    names = ["entry", "setup", "rec", "fb_proj", "input", "computed", "df", "exit"]
    print("stage            wgs  " + "  ".join("%9s" % n for n in names))
    for i in range(ns):
        seq, layer, wg0, n = out[4 * i:4 * i + 4]
        u = us[wg0:wg0 + n].reshape(-1, 8)
        u = u[st[wg0:wg0 + n].reshape(-1, 8)[:, 7] > 0]
        lab = "seq%d L%d" % (seq, layer)
        print("%-16s %3d  " % (lab, n) + "  ".join(("%4.1f-%4.1f" % (u[:, j].min(), u[:, j].max()) if len(u) and u[:, j].min() > -1e5 else "    -    ") for j in range(8)))

if os.environ.get("SFSN_HOP_DEBUG"):
    w = torch.from_numpy(rw.synth_wave(1, 41, 5))
    for resident in (False, True):
        sess = model.streaming(batch=1, waveform=True, host_io=True, resident=resident, idle_ms=200)
        for c in range(40):
            sess.step_wave_host(w[:, 128 * c:128 * (c + 1)])
        sess.close()
        torch.cuda.synchronize()
        print("resident" if resident else "one launch per hop")
        stage_table(sess)
    sys.exit(0)

B = 1
wave = torch.from_numpy(rw.synth_wave(B, 4001, 5))
for resident in (False, True, False, True):
    sess = model.streaming(batch=B, waveform=True, host_io=True, resident=resident, idle_ms=200)
    lat = []
    for c in range(3000):
        x = wave[:, 128 * c:128 * (c + 1)]
        t0 = time.perf_counter()
        sess.step_wave_host(x)
        lat.append(time.perf_counter() - t0)
    lat = np.sort(np.asarray(lat[500:])) * 1e6
    print("resident", resident, "p50 %.1f p99 %.1f min %.1f" % (lat[len(lat) // 2], lat[int(len(lat) * .99)], lat[0]), flush=True)
    if resident:
        r = sess._res
        for dt in (0.05, 0.1, 0.2, 0.4, 0.8):
            time.sleep(dt)
            print("  after +%.2f s idle: stream.query() =" % dt, (r["stream"].query(), int(sess._hop["host"]["bell_np"][1])), flush=True)
        t0 = time.perf_counter()
        sess.close()
        print("  close took %.1f us" % ((time.perf_counter() - t0) * 1e6))
    sess.check_errors()

# the watchdog early in a session (as tests/test_hip_parity.py::test_waveform_streaming_resident_launch does)
sess = model.streaming(batch=B, waveform=True, host_io=True, resident=True, idle_ms=200)
for c in range(30):
    if c == 17:
        for dt in (0.1, 0.2, 0.4, 0.8):
            time.sleep(dt)
            print("  hop 17, after +%.2f s idle: stream.query() =" % dt, (sess._res["stream"].query(), int(sess._hop["host"]["bell_np"][1])), flush=True)
    sess.step_wave_host(wave[:, 128 * c:128 * (c + 1)])
sess.close()
sess.check_errors()
print("ok")
