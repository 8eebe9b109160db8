"""Round 3: the scan with IO-specialised waves (sfsn_scan3_dev.h) against round 2's body (SFSN_SCAN_V2=1), one layer per launch:
bit-equality of every output (fp32 spikes, int8 spikes, final h / c) and time per step.
python scripts/exp_scan3.py [H] [T]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spiking_fullsubnet_amd import _lib
from spiking_fullsubnet_amd._lib import ScanSegment, check
from spiking_fullsubnet_amd.engine import pack_w3


def run(H, T, Rs, rpw, reps=5, seed=0, f32=True):
    L = _lib.lib(); dev = "cuda:0"
    HP = (H + 63) // 64 * 64
    rng = np.random.default_rng(seed)
    keep = []
    def dv(a): t = torch.from_numpy(np.ascontiguousarray(a)).to(dev); keep.append(t); return t
    ns = len(Rs)
    segs = (ScanSegment * ns)()
    outs = []
    for i, R in enumerate(Rs):
        w = (rng.standard_normal((H, H)) / np.sqrt(H)).astype(np.float32)
        pk, dq = pack_w3(w); pk, dq = dv(pk), dv(dq)
        sg = segs[i]
        sg.zin = dv((rng.standard_normal((T, R, H)) * 0.5).astype(np.float32)).data_ptr()
        sg.w_hh, sg.w_dq = pk.data_ptr(), dq.data_ptr()
        sg.bias = dv((rng.standard_normal(2 * H) * 0.1).astype(np.float32)).data_ptr()
        sg.bn_alpha = dv((1.0 + 0.2 * rng.standard_normal(H)).astype(np.float32)).data_ptr()
        sg.bn_beta = dv((0.3 * rng.standard_normal(H)).astype(np.float32)).data_ptr()
        h0 = (rng.random((R, H)) < 0.3).astype(np.float32); c0 = (0.5 * rng.standard_normal((R, H))).astype(np.float32)
        hs, cs = dv(h0.copy()), dv(c0.copy())
        sg.h_state, sg.c_state = hs.data_ptr(), cs.data_ptr()
        spk = torch.zeros((T, R, H), dtype=torch.float32, device=dev) if f32 else None
        s8 = torch.zeros((T, R, HP), dtype=torch.int8, device=dev)
        sg.spikes_f32 = spk.data_ptr() if f32 else None; sg.spikes_i8 = s8.data_ptr(); sg.membrane = None; sg.R = R
        outs.append((spk, s8, hs, cs, dv(h0), dv(c0)))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def go(): check(L.sfsn_gsn_layer_scan(segs, ns, T, H, 1, rpw, st), "scan")
    def reset():
        for spk, s8, hs, cs, h0, c0 in outs:
            hs.copy_(h0); cs.copy_(c0); s8.zero_()
            if spk is not None: spk.zero_()
    res = {}
    for mode in ("v2", "v3"):
        if mode == "v2": os.environ["SFSN_SCAN_V2"] = "1"
        else: os.environ.pop("SFSN_SCAN_V2", None)
        reset(); go(); torch.cuda.synchronize()
        snap = [tuple(None if x is None else x.clone() for x in o[:4]) for o in outs]
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(); go(); b.record()
        torch.cuda.synchronize()
        res[mode] = (min(a.elapsed_time(b) for a, b in ev), snap)
    bad = []
    for i in range(ns):
        for k, name in enumerate(("spikes_f32", "spikes_i8", "h_state", "c_state")):
            a, b = res["v2"][1][i][k], res["v3"][1][i][k]
            if a is None: continue
            if not torch.equal(a, b):
                d = (a != b)
                first = d.nonzero()[0].tolist()
                bad.append(f"seg{i}.{name}: {int(d.sum())} differ, first at {first}")
    rate = float((res["v3"][1][0][1] != 0).float().mean().item())
    print(f"H={H} T={T} rows={Rs} rpw={rpw} f32={int(f32)}: v2 {1e3*res['v2'][0]/T:.3f} us/step  v3 {1e3*res['v3'][0]/T:.3f} us/step  "
          f"spike rate {rate:.2f}  {'BIT-IDENTICAL' if not bad else 'MISMATCH ' + '; '.join(bad)}", flush=True)
    return not bad


if __name__ == "__main__":
    a = sys.argv[1:]
    H = int(a[0]) if a else 224
    T = int(a[1]) if len(a) > 1 else 1000
    ok = True
    for rpw in (4, 8, 16):
        ok &= run(H, T, [512, 192, 128], rpw)
    ok &= run(H, T, [512, 192, 128], 8, f32=False)
    ok &= run(160, 300, [37, 5, 100], 4)
    ok &= run(160, 300, [37, 5, 100], 8)
    ok &= run(160, 300, [37, 5, 100], 16)
    ok &= run(64, 200, [7], 8)
    ok &= run(32, 100, [3, 18], 4)
    ok &= run(128, 50, [16], 16, f32=False)
    ok &= run(224, 3, [9], 8)
    ok &= run(224, 1, [9], 4)
    print("ALL BIT-IDENTICAL" if ok else "SOME MISMATCH")
