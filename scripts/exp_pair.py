"""Round 4: the sub-band stack as ONE launch with the layers side by side at 8 rows per workgroup -- layer 1 = the IO-wave scan
(publishing), layer 2 = the FUSED3 role (input product inside the scan, sfsn_scan3i_dev.h) -- against the per-layer launches
(scan3 at 4 rows + sfsn_spike_proj) and the 8-wave FUSED roles (SFSN_STACK_FUSED8=1).  Times per launch through the C ABI with
HIP events; B = 64 geometry (rows 512 + 192 + 128), H = 224, T = 1000.   usage: python scripts/exp_pair.py [T] [lag]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from spiking_fullsubnet_amd import _lib  # noqa: E402
from spiking_fullsubnet_amd._lib import FusedInput, ScanSegment, check  # noqa: E402
from spiking_fullsubnet_amd.engine import fold_batchnorm, pack_w3  # noqa: E402
import refweights as rw  # noqa: E402

DEV = "cuda:0"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
LAG = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H, I, Rs, nl = 224, 38, [512, 192, 128], 2
HP = 256
L = _lib.lib()
rng = np.random.default_rng(3)


def t_(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def p_(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


cells = []
for l in range(nl):
    sd = {}
    rw._cell(rng, "", I if l == 0 else H, H, True, True, sd)
    alpha, beta = fold_batchnorm(sd["batchnorm.weight"], sd["batchnorm.bias"], sd["batchnorm.running_mean"], sd["batchnorm.running_var"])
    pk, dq = pack_w3(sd["weight_hh"])
    d = dict(pk=t_(pk), dq=t_(dq), bias=t_(sd["bias_ih"]), al=t_(alpha), be=t_(beta))
    if l > 0:
        pki, dqi = pack_w3(sd["weight_ih"])
        d.update(pki=t_(pki), dqi=t_(dqi))
    cells.append(d)
zin0 = [torch.randn((T, R, H), device=DEV) * 0.5 for R in Rs]
zin1 = [torch.empty((T, R, H), device=DEV) for R in Rs]
spk = [[torch.empty((T, R, H), device=DEV) for R in Rs] for _ in range(nl)]
s8 = [[torch.zeros((T, R, HP), dtype=torch.int8, device=DEV) for R in Rs] for _ in range(nl)]
hs = [[torch.zeros((R, H), device=DEV) for R in Rs] for _ in range(nl)]
cs = [[torch.zeros((R, H), device=DEV) for R in Rs] for _ in range(nl)]
ns = len(Rs)


def zero_states():
    for l in range(nl):
        for i in range(ns):
            hs[l][i].zero_(); cs[l][i].zero_()


def fill(segs, fin, wide):
    for l in range(nl):
        for i, R in enumerate(Rs):
            s, c = segs[l * ns + i], cells[l]
            s.zin = p_(zin0[i]) if l == 0 else (p_(zin1[i]) if wide else None)
            s.w_hh, s.w_dq, s.bias, s.bn_alpha, s.bn_beta = p_(c["pk"]), p_(c["dq"]), p_(c["bias"]), p_(c["al"]), p_(c["be"])
            s.h_state, s.c_state, s.spikes_f32, s.spikes_i8, s.membrane, s.R = p_(hs[l][i]), p_(cs[l][i]), p_(spk[l][i]), p_(s8[l][i]), None, R
            if l > 0:
                fin[l * ns + i].spikes_in = s8[l - 1][i].data_ptr()
                fin[l * ns + i].w_ih, fin[l * ns + i].w_ih_dq = c["pki"].data_ptr(), c["dqi"].data_ptr()


nb = L.sfsn_stack_scratch_bytes(nl, ns, sum(Rs))
scratch = torch.zeros((nb // 4,), dtype=torch.int32, device=DEV)


xg0 = torch.randn((T, Rs[0], I), device=DEV)
w_ih0 = torch.randn((H, I), device=DEV) / I ** 0.5


def stack(rpw, wide=False, x_groups=()):
    segs, fin = (ScanSegment * (nl * ns))(), (FusedInput * (nl * ns))()
    fill(segs, fin, wide)
    rp = (ctypes.c_int * nl)(*rpw)
    if x_groups:
        from spiking_fullsubnet_amd._lib import FusedX
        fx = (FusedX * ns)()
        for i in x_groups:
            fx[i].x, fx[i].w_ih, fx[i].I = xg0.data_ptr(), w_ih0.data_ptr(), I
            segs[i].zin = None
        check(L.sfsn_gsn_stack_scan_x(segs, fin, fx, nl, ns, T, H, rp, LAG, p_(scratch), nb, None), "stack_x")
    else:
        check(L.sfsn_gsn_stack_scan(segs, fin, nl, ns, T, H, rp, LAG, p_(scratch), nb, None), "stack")


def per_layer(rpw):
    for l in range(nl):
        segs = (ScanSegment * ns)()
        for i, R in enumerate(Rs):
            s, c = segs[i], cells[l]
            if l > 0:
                check(L.sfsn_spike_proj(p_(s8[0][i]), p_(c["pki"]), p_(c["dqi"]), p_(c["bias"]), p_(zin1[i]), T * R, H, H, H, None), "proj")
            s.zin = p_(zin0[i]) if l == 0 else p_(zin1[i])
            s.w_hh, s.w_dq, s.bias, s.bn_alpha, s.bn_beta = p_(c["pk"]), p_(c["dq"]), p_(c["bias"]), p_(c["al"]), p_(c["be"])
            s.h_state, s.c_state, s.spikes_f32, s.spikes_i8, s.membrane, s.R = p_(hs[l][i]), p_(cs[l][i]), p_(spk[l][i]), p_(s8[l][i]), None, R
        check(L.sfsn_gsn_layer_scan(segs, ns, T, H, 1, rpw, None), "scan")


def timeit(name, fn, reps=5):
    zero_states(); fn(); torch.cuda.synchronize()
    assert int(scratch[0].item()) == 0, "hand-off wait expired"
    best = 1e9
    for _ in range(reps):
        zero_states()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    algo = 2 * 14592 * 64 * T  # bytes: both layers, API-faithful
    print(f"{name:58s} {best:8.3f} ms  {best * 1e3 / T:6.3f} us/frame  roofline {algo / (best * 1e-3) / 8e12:.3f}", flush=True)
    return [[x.clone() for x in spk[l]] for l in range(nl)]


ref = timeit("per-layer: scan3 4 rows + spike_proj", lambda: per_layer(4))
timeit("per-layer: scan3 8 rows + spike_proj", lambda: per_layer(8))
a = timeit("pair launch: layer 1 scan3 8 rows | layer 2 FUSED3", lambda: stack((8, 8)))
for l in range(nl):
    for i in range(ns):
        assert torch.equal(ref[l][i], a[l][i]), (l, i)
print("pair launch == per-layer launches, bit for bit")
timeit("pair launch, group 0's layer-1 input product inside (FUSEDX3)", lambda: stack((8, 8), x_groups=(0,)))
timeit("pair launch: layer 1 scan3 4 rows | layer 2 FUSED3", lambda: stack((4, 8)))
timeit("pair launch: layer 1 scan3 16 rows | layer 2 FUSED3", lambda: stack((16, 8)))
os.environ["SFSN_STACK_FUSED8"] = "1"
timeit("narrow stack (8-wave FUSED roles), 8 rows", lambda: stack((8, 8)))
del os.environ["SFSN_STACK_FUSED8"]
timeit("wide stack (PROJ roles), 8 rows", lambda: stack((8, 8), wide=True))
