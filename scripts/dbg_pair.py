"""debug: first differences of the FUSED3 role against sfsn_spike_proj + sfsn_gsn_layer_scan"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_stack_scan as ts
from test_hip_parity import run_scan
from spiking_fullsubnet_amd import _lib
hip = _lib.lib()
I, H, nl, Rs, T, rpw = int(os.environ.get("I", "38")), int(os.environ.get("H", "224")), 2, [int(os.environ.get("R", "16"))], int(os.environ.get("T", "12")), 8
rng = np.random.default_rng(1)
cells = ts._cells(rng, I, H, nl)
zin0 = [rng.standard_normal((T, R, H)).astype(np.float32) * 0.5 for R in Rs]
got = ts.run_stack(hip, zin0, cells, T, H, rpw, wide=False)
sd, alpha, beta, _ = cells[1]
zin = ts._spike_proj(hip, got[0][0][1], sd["weight_ih"], H)
spk, _, s8, hT, cT = run_scan(hip, zin, sd["weight_hh"], sd["bias_ih"], alpha, beta, True, want_mem=False)
d = got[1][0][0] != spk
print("frames with diffs:", np.nonzero(d.any(axis=(1, 2)))[0][:20])
for t in range(min(T, 6)):
    rows = np.nonzero(d[t].any(axis=1))[0]
    print(f"t={t}: {int(d[t].sum())} diffs, rows {rows[:16]}")
    if len(rows):
        r = rows[0]
        cols = np.nonzero(d[t, r])[0]
        print("   row", r, "neurons", cols[:40], " (neuron % 16:", sorted(set(cols % 16)), " tiles:", sorted(set(cols // 16)), ")")
