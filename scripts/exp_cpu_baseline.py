"""bench.py's cpu_baseline leg alone, for a few (worker processes, threads per worker) layouts (no GPU needed)."""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import refweights as rw
src = open(os.path.join(ROOT, "bench.py")).read()
ns = {"ROOT": ROOT}
exec("import os, sys, time, json, numpy as np, torch\n" + src[src.index("def cpu_baseline"):src.index("\n\n\nif __name__")], ns)
kw = rw.LIVE_M; sd = rw.live_state_dict(kw, 21)
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 1000
wave = torch.from_numpy(rw.synth_wave(B, T, 0))
stft = torch.stft(wave, 512, 128, 512, window=torch.hann_window(512), return_complex=True, pad_mode="constant")
n = os.cpu_count()
for w, t in ((None, None), (16, 1), (32, 1), (8, 2)):
    r = ns["cpu_baseline"](kw, sd, stft, workers=w, threads=t)
    print(w, t, r["value"], r["single_core_value"], r["all_cores_over_one_core"], flush=True)
