"""fb/sb chunk overlap of one forward: bit identity and time per forward by number of chunks."""
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
eng.overlap_chunks = 0
ref = eng.forward_stft(stft); torch.cuda.synchronize()
firsts = [int(v) for v in os.environ.get("FIRST", "0").split(",")]  # frames of the first chunk (0 = equal chunks, -1 = the default 0.24 T)
for first, n in [(f, n) for f in firsts for n in ((0, 2, 3, 4, 5, 6, 8, 12) if f == 0 and not os.environ.get('ONLY3') else (3,))]:
    eng.overlap_chunks, eng.overlap_first = n, first
    out = eng.forward_stft(stft); eng.check_stack_errors()
    ok = torch.equal(torch.view_as_real(ref["enh_stft"]), torch.view_as_real(out["enh_stft"])) and all(
        torch.equal(x, y) for x, y in zip(ref["fb_all"] + sum(ref["sb_all"], []), out["fb_all"] + sum(out["sb_all"], [])))
    for _ in range(3): eng.forward_stft(stft)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): eng.forward_stft(stft)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    print(f"first={first} overlap_chunks={n}: {'bit-identical' if ok else 'MISMATCH'}  {dt*1e3:.3f} ms per forward (B={B}, T={T}), chunks={out['n_chunks']}", flush=True)
