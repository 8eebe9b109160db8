"""Round 4: one forward alone (B x T, live baseline_m) with the sub-band stack as one pair launch (FUSED3 roles) against round 3's
per-layer schedule, by number of full-band / sub-band overlap chunks; every returned tensor compared with the per-layer forward.
usage: python scripts/exp_forward_r04.py [B] [T]   (CHUNKS=0,2,3,4  FIRST=-1  PAIRS=0,1  AHEAD=1  FRACS="a,b,c;...")"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
kw = rw.LIVE_M; sd = rw.live_state_dict(kw, 21)
m = pkg.SpikingFullSubNet(**kw); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True); m = m.eval().to(DEV)
stft = m._stft(torch.from_numpy(rw.synth_wave(B, T, 3)).to(DEV)); eng = m.engine()
eng.overlap_chunks, eng.pair_scan = 0, False
ref = eng.forward_stft(stft); torch.cuda.synchronize()
chunks = [int(v) for v in os.environ.get("CHUNKS", "0,2,3,4").split(",")]
firsts = [int(v) for v in os.environ.get("FIRST", "-1").split(",")]
aheads = [int(v) for v in os.environ.get("AHEAD", "1").split(",")]
for pair, ahead in [(bool(int(p_)), bool(a_)) for p_ in os.environ.get("PAIRS", "0,1").split(",") for a_ in aheads]:
    for n in chunks:
        for first in (firsts if n > 1 else [-1]):
            eng.pair_scan, eng.overlap_chunks, eng.overlap_first, eng.overlap_prep_ahead = pair, n, first, ahead
            eng.launches = {}
            out = eng.forward_stft(stft); eng.check_stack_errors()
            la = dict(eng.launches)
            ok = torch.equal(torch.view_as_real(ref["enh_stft"]), torch.view_as_real(out["enh_stft"])) and all(
                torch.equal(x, y) for x, y in zip(ref["fb_all"] + sum(ref["sb_all"], []), out["fb_all"] + sum(out["sb_all"], [])))
            for _ in range(3): eng.forward_stft(stft)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(8): eng.forward_stft(stft)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
            eng.check_stack_errors()
            print(f"pair={int(pair)} ahead={int(ahead)} overlap_chunks={n} first={first}: {'bit-identical' if ok else 'MISMATCH'}  {dt*1e3:.3f} ms per forward (B={B}, T={T}) launches {la}", flush=True)

if os.environ.get("FRACS"):
    eng.pair_scan, eng.overlap_chunks, eng.overlap_first = True, 3, -1
    for spec in os.environ["FRACS"].split(";"):
        eng.overlap_fracs = [float(v) for v in spec.split(",")]
        out = eng.forward_stft(stft); eng.check_stack_errors()
        ok = torch.equal(torch.view_as_real(ref["enh_stft"]), torch.view_as_real(out["enh_stft"]))
        for _ in range(3): eng.forward_stft(stft)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): eng.forward_stft(stft)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
        print(f"chunk fractions {spec}: {'bit-identical' if ok else 'MISMATCH'}  {dt*1e3:.3f} ms per forward, chunks={out['n_chunks']}", flush=True)
    eng.overlap_fracs = None
