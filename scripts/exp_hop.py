"""One-launch streaming hop (sfsn_stream_hop) against the graph-replay session and the offline forward: equality + latency.
Run on the MI355X box: python scripts/exp_hop.py [tiny|m] [B] [hop]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import refweights as rw
from test_hip_parity import build_module

DEV = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "m"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hop = int(sys.argv[3]) if len(sys.argv) > 3 else 1
kw = {"m": rw.LIVE_M, "tiny": rw.LIVE_TINY, "2spk": rw.LIVE_TINY_2SPK}[name]
model = build_module("live", kw, rw.live_state_dict(kw, 5))
T = 48 * hop
wave = torch.from_numpy(rw.synth_wave(B, T, 5)).to(DEV)
stft = torch.stft(wave, kw["n_fft"], kw["hop_length"], kw["win_length"], window=torch.hann_window(kw["win_length"], device=DEV),
                  return_complex=True, pad_mode="constant")[..., :T].contiguous()
off = model.engine().forward_stft(stft, want_layers=False)
for mode in (True, False):
    sess = model.streaming(batch=B, hop=hop, one_launch=mode)
    outs, mags = [], []
    for t0 in range(0, T, hop):
        e, m = sess.step(stft[..., t0:t0 + hop].contiguous())
        outs.append(e); mags.append(m)
    if mode: sess.check_errors()
    e, m = torch.cat(outs, -1), torch.cat(mags, -1)
    eq = torch.equal(torch.view_as_real(e), torch.view_as_real(off["enh_stft"])) and torch.equal(m, off["enh_mag"])
    print("one_launch" if mode else "graph     ", "equal to offline:", eq)
    if not eq:
        d = (torch.view_as_real(e) - torch.view_as_real(off["enh_stft"])).abs().amax(dim=(0, 1, 4))  # [F, T]
        bad_t = (d.amax(0) > 0).nonzero().flatten()
        bad_f = (d.amax(1) > 0).nonzero().flatten()
        print("  first bad frame", bad_t[:5].tolist(), "bad bins", bad_f[:8].tolist(), "...", len(bad_f), "max", float(d.max()))
    frames = [stft[..., (i * hop) % T:(i * hop) % T + hop].contiguous() for i in range(T // hop)]
    lat = []
    for i in range(1200):
        x = frames[i % len(frames)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sess.step(x, copy=False)
        torch.cuda.synchronize()
        if i >= 200: lat.append(time.perf_counter() - t0)
    lat = np.sort(np.array(lat)) * 1e6
    print("  latency p50 %.1f us  min %.1f  p99 %.1f" % (lat[len(lat) // 2], lat[0], lat[int(len(lat) * .99)]))
    if mode: sess.check_errors()

if os.environ.get("SFSN_HOP_DEBUG"):
    import ctypes
    sess = model.streaming(batch=B, hop=hop, one_launch=True)
    L, desc = sess.eng.lib, sess._hop["desc"]
    out = (ctypes.c_int * 80)()
    ns = L.sfsn_hop_stages(ctypes.byref(desc), out, 20)
    for i in range(6):
        sess.step(frames[i], copy=False)
    torch.cuda.synchronize()
    raw = sess._hop["scratch"].cpu().numpy().view(np.uint8)
    nwg = out[4 * (ns - 1) + 2] + out[4 * (ns - 1) + 3]
    nb = 64
    st = raw[nb:nb + nwg * 8 * 64].view(np.uint64).reshape(nwg, 8, 8).astype(np.int64)
    t0 = st[:, :, 0].min()
    us = (st - t0) / 100.0
    names = ["entry", "setup", "rec", "fb_proj", "input", "computed", "df", "exit"]
    print("stage            wgs  " + "  ".join("%9s" % n for n in names) + "   (us since the first wave's entry: min..max over the stage's waves)")
    for i in range(ns):
        seq, layer, wg0, n = out[4 * i:4 * i + 4]
        u = us[wg0:wg0 + n].reshape(-1, 8)
        u = u[st[wg0:wg0 + n].reshape(-1, 8)[:, 7] > 0]  # waves that ran
        lab = ("fb" if seq == 0 else "sb%d" % (seq - 1)) + (" L%d" % layer if layer >= 0 else " proj")
        print("%-16s %3d  " % (lab, n) + "  ".join(("%4.1f-%4.1f" % (u[:, j].min(), u[:, j].max()) if u[:, j].min() > -1 else "    -    ") for j in range(8)))

# where the host time goes: enqueue alone, enqueue + blocking sync, enqueue + spinning on an event
sess = model.streaming(batch=B, hop=hop, one_launch=True)
ev = torch.cuda.Event()
def timeit(fn, n=1500, skip=300):
    ts = []
    for i in range(n):
        x = frames[i % len(frames)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(x)
        ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    ts = np.sort(np.array(ts[skip:])) * 1e6
    return ts[len(ts) // 2]
def f_enq(x): sess.step(x, copy=False)
def f_sync(x): sess.step(x, copy=False); torch.cuda.synchronize()
def f_ssync(x): sess.step(x, copy=False); torch.cuda.current_stream().synchronize()
def f_spin(x):
    sess.step(x, copy=False); ev.record()
    while not ev.query(): pass
print("host: enqueue %.1f us | + device sync %.1f | + stream sync %.1f | + event spin %.1f" % (timeit(f_enq), timeit(f_sync), timeit(f_ssync), timeit(f_spin)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(2000): sess.step(frames[i % len(frames)], copy=False)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
