import sys, os, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_stack_scan as ts
from oracle import Oracle
from spiking_fullsubnet_amd import _lib
hip = _lib.lib()
I, H, nl, Rs, T, rpw = 38, 160, 2, [64], 64, 8
rng = np.random.default_rng(H * 100 + nl * 10 + len(Rs))
cells = ts._cells(rng, I, H, nl)
o = Oracle("f32")
xs = [rng.standard_normal((T, R, I)).astype(np.float32) for R in Rs]
zin0 = [o.linear(x, cells[0][0]["weight_ih"]) for x in xs]
# monkeypatch run_stack to return zin of layer 1: replicate quickly
import types
orig_empty = torch.empty
zbufs = []
def my_empty(*a, **k):
    t = orig_empty(*a, **k)
    if len(a) and a[0] == (T, 64, H): zbufs.append(t)
    return t
torch.empty = my_empty
for trial in range(3):
    zbufs.clear()
    got = ts.run_stack(hip, zin0, cells, T, H, rpw, wide=True)
    # zbufs: [spk L0, zin L1, spk L1] order of creation: l=0: spk; l=1: z then spk
    z1 = zbufs[1].cpu().numpy()
    sd = cells[1][0]
    zref = ts._spike_proj(hip, got[0][0][1], sd["weight_ih"], H) + sd["bias_ih"][:H]
    bad = np.argwhere(z1 != zref.astype(np.float32))
    print("trial", trial, "zin mismatches", len(bad), bad[:5].tolist() if len(bad) else "")
    from test_hip_parity import run_scan
    spk, _, s8, hT, cT = run_scan(hip, zref - sd["bias_ih"][:H], sd["weight_hh"], sd["bias_ih"], cells[1][1], cells[1][2], True, want_mem=False)
    b2 = np.argwhere(got[1][0][0] != spk)
    print("   L1 spike mismatches vs per-layer", len(b2), b2[:5].tolist() if len(b2) else "")
for (t, r, cidx) in bad[:12]:
    print(t, r, cidx, z1[t, r, cidx], zref[t, r, cidx])
rows = sorted(set(int(b[1]) for b in bad)); print("rows with mismatches", rows[:40])
cols = sorted(set(int(b[2]) for b in bad)); print("cols", cols[:60])
fr = sorted(set(int(b[0]) for b in bad)); print("frames", fr[:20], len(fr))
