"""GPU tests of the layer-pipelined stack scan (sfsn_gsn_stack_scan): every layer of a stack in one launch, layer l+1
trailing layer l by a few frames through in-launch hand-offs.  Checked (a) through the C ABI against the CPU oracle's
stacked layers and, bit for bit, against the per-layer entry points it replaces; (b) through the drop-in modules against
the per-layer schedule, bit for bit, for every tensor the reference's forward() returns; (c) under uneven load (other
kernels competing for the CUs while producers and consumers hand frames over); (d) on its argument checks."""
import ctypes
import os

import numpy as np
import pytest
import torch

import parity
import refweights as rw
from oracle import Oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


@pytest.fixture(scope="module")
def hip():
    from spiking_fullsubnet_amd import _lib
    L = _lib.lib()
    assert L.sfsn_device_count() >= 1
    return L


def _cells(rng, I, H, nl):
    """nl stacked cells (shared gates, BatchNorm) as the reference initialises them + randomised BN statistics."""
    from spiking_fullsubnet_amd.engine import fold_batchnorm
    out = []
    for l in range(nl):
        sd = {}
        rw._cell(rng, "", I if l == 0 else H, H, True, True, sd)
        alpha, beta = fold_batchnorm(sd["batchnorm.weight"], sd["batchnorm.bias"], sd["batchnorm.running_mean"], sd["batchnorm.running_var"])
        out.append((sd, alpha, beta, (sd["batchnorm.weight"], sd["batchnorm.bias"], sd["batchnorm.running_mean"], sd["batchnorm.running_var"])))
    return out


def run_stack(hip, zin0s, cells, T, H, rpw, lag=4, want_f32=True, h0=None, c0=None, wide=True, xs=None):
    """zin0s: per segment [T, R, H] = x . W_ih^T of layer 0 (bias added here).  Returns per layer / segment fp32 spikes,
    int8 spikes and the final states.  wide: give the layers >= 1 an input-term buffer (H <= 256: selects the 16-wave flavour
    with PROJ workgroups; without it the 8-wave fused-input roles run; H > 256 always needs the buffer)."""
    from spiking_fullsubnet_amd._lib import FusedInput, ScanSegment, check
    from spiking_fullsubnet_amd.engine import pack_w3
    nl, ns = len(cells), len(zin0s)
    HP = (H + 63) // 64 * 64
    segs, fin, keep = (ScanSegment * (nl * ns))(), (FusedInput * (nl * ns))(), []
    out = [[None] * ns for _ in range(nl)]
    for l, (sd, alpha, beta, _) in enumerate(cells):
        pk, dq = pack_w3(sd["weight_hh"])
        pk, dq, bias, al, be = _t(pk), _t(dq), _t(sd["bias_ih"]), _t(alpha), _t(beta)
        keep += [pk, dq, bias, al, be]
        if l > 0:
            pki, dqi = pack_w3(sd["weight_ih"])
            pki, dqi = _t(pki), _t(dqi)
            keep += [pki, dqi]
        for i, z0 in enumerate(zin0s):
            R = z0.shape[1]
            s = segs[l * ns + i]
            z = _t((z0 + sd["bias_ih"][:H]).astype(np.float32)) if l == 0 else (torch.empty((T, R, H), device=DEV) if (H > 256 or wide) else None)
            h = _t(np.zeros((R, H), np.float32) if h0 is None else h0[l][i])
            c = _t(np.zeros((R, H), np.float32) if c0 is None else c0[l][i])
            spk = torch.empty((T, R, H), device=DEV) if want_f32 else None
            s8 = torch.zeros((T, R, HP), dtype=torch.int8, device=DEV)
            keep += [z, h, c]
            s.zin, s.w_hh, s.w_dq, s.bias, s.bn_alpha, s.bn_beta = _p(z), _p(pk), _p(dq), _p(bias), _p(al), _p(be)
            s.h_state, s.c_state, s.spikes_f32, s.spikes_i8, s.membrane, s.R = _p(h), _p(c), _p(spk), _p(s8), None, R
            if l > 0:
                fin[l * ns + i].spikes_in = out[l - 1][i][1].data_ptr()
                fin[l * ns + i].w_ih, fin[l * ns + i].w_ih_dq = pki.data_ptr(), dqi.data_ptr()
            out[l][i] = (spk, s8, h, c)
    nb = hip.sfsn_stack_scratch_bytes(nl, ns, sum(z.shape[1] for z in zin0s))
    scratch = torch.zeros((nb // 4,), dtype=torch.int32, device=DEV)
    rp = (ctypes.c_int * nl)(*([rpw] * nl))
    if xs is not None:
        # xs[i] = layer-0 input [T, R, I] of segment i (or None: that segment's layer 0 reads zin0s[i]): sfsn_gsn_stack_scan_x
        from spiking_fullsubnet_amd._lib import FusedX
        fx = (FusedX * ns)()
        w0 = _t(cells[0][0]["weight_ih"])
        keep.append(w0)
        for i, x in enumerate(xs):
            if x is not None:
                tx = _t(x)
                keep.append(tx)
                fx[i].x, fx[i].w_ih, fx[i].I = tx.data_ptr(), w0.data_ptr(), x.shape[2]
                segs[i].zin = None
        check(hip.sfsn_gsn_stack_scan_x(segs, fin, fx, nl, ns, T, H, rp, lag, _p(scratch), nb, None), "sfsn_gsn_stack_scan_x")
    else:
        check(hip.sfsn_gsn_stack_scan(segs, fin, nl, ns, T, H, rp, lag, _p(scratch), nb, None), "sfsn_gsn_stack_scan")
    torch.cuda.synchronize()
    assert int(scratch[0].item()) == 0, "a hand-off wait expired"
    return [[tuple(None if x is None else x.cpu().numpy() for x in out[l][i]) for i in range(ns)] for l in range(nl)]


STACKS = [  # I, H, layers, rows per segment, T, rows per workgroup
    (38, 224, 2, [40, 9, 17], 50, 8), (38, 224, 3, [33], 41, 4), (64, 320, 2, [21], 37, 4), (64, 320, 3, [35, 6], 30, 8), (40, 320, 2, [130], 9, 8),
    (12, 32, 3, [5, 20], 33, 16), (38, 160, 2, [64], 64, 8), (30, 256, 2, [18, 3], 26, 16), (64, 240, 4, [7], 29, 4),
    (20, 96, 2, [1], 19, 8),
    # round 4: layers >= 1 at 8 rows per workgroup without an input-term buffer run the FUSED3 role (input product inside the
    # IO-wave scan, sfsn_scan3i_dev.h): every k-step form (H mod 64 in (0, 32]: the 32-wide tail step; otherwise full steps), three
    # layers (a FUSED3 role that publishes), T = 1 / 2 / odd, ragged row tiles
    (38, 224, 3, [33, 8], 41, 8), (16, 192, 2, [19], 23, 8), (24, 128, 2, [9], 31, 8), (12, 48, 2, [5, 20], 33, 8),
    (12, 64, 3, [11], 7, 8), (38, 224, 2, [8], 1, 8), (38, 224, 2, [13], 2, 8), (10, 16, 2, [3], 9, 8), (38, 208, 2, [24], 21, 8),
    (38, 176, 2, [10], 12, 8), (38, 144, 4, [17], 15, 8),
]


@pytest.mark.parametrize("H,nl,Rs,T,rpw,want_f32", [
    (320, 2, [21], 37, 4, True), (320, 3, [35, 6], 30, 8, True), (320, 2, [64], 90, 4, False), (320, 2, [64], 61, 8, False),
    (272, 2, [13], 25, 4, True), (288, 2, [9, 4], 19, 8, True), (304, 3, [11], 1, 4, True), (304, 2, [7], 2, 8, True),
    (320, 2, [130], 14, 8, True),  # nine 16-row blocks: the PROJ role is NOT split by columns (21 / 35 / 6 / 64 rows: four / two parts)
])
def test_full_band_stack_with_io_waves_equals_round_2_bodies(hip, H, nl, Rs, T, rpw, want_f32):
    """Round 5: 256 < H <= 320 stacks at 4 / 8 rows per workgroup run scan3w_role (sfsn_scan3w_dev.h: ten compute waves x two tiles +
    loader + storer, 768 threads, PROJ roles on twelve waves) where the library chooses it (8 rows; 4 rows from 512 frames on).  Forced
    on and off here (SFSN_STACK_FB3, read per call): every output of every layer -- fp32 / int8 spikes, final h and c -- bit for bit equal to round 2's bodies, which the test below holds to the oracle; every
    tile count 17..20, one / two / odd numbers of frames, ragged row blocks, three layers (a scan role that is gated AND publishes)."""
    rng = np.random.default_rng(H + nl + T)
    cells = _cells(rng, 40, H, nl)
    o = Oracle("f32")
    zin0 = [o.linear(rng.standard_normal((T, R, 40)).astype(np.float32), cells[0][0]["weight_ih"]) for R in Rs]
    h0 = [[(rng.random((R, H)) > 0.5).astype(np.float32) for R in Rs] for _ in range(nl)]
    c0 = [[rng.standard_normal((R, H)).astype(np.float32) for R in Rs] for _ in range(nl)]
    res = {}
    for flag in ("1", "0"):
        os.environ["SFSN_STACK_FB3"] = flag
        try:
            res[flag] = run_stack(hip, zin0, cells, T, H, rpw, want_f32=want_f32, h0=h0, c0=c0)
        finally:
            del os.environ["SFSN_STACK_FB3"]
    for l in range(nl):
        for i in range(len(Rs)):
            for a, b, what in zip(res["1"][l][i], res["0"][l][i], ("fp32 spikes", "int8 spikes", "h", "c")):
                assert (a is None) == (b is None)
                if a is not None:
                    np.testing.assert_array_equal(a, b, err_msg=f"layer {l} segment {i}: {what}")
            assert res["1"][l][i][1].any()


@pytest.mark.parametrize("Rs,T,rpw,force", [([64], 150, 8, None), ([64], 40, 8, "1"), ([35, 6], 33, 8, "1"), ([64], 136, 4, None)])
def test_full_band_stack_with_io_waves_vs_oracle(hip, Rs, T, rpw, force):
    """gsn_stack_fb_kernel (scan3w_role + the column-split PROJ role) held to the ORACLE directly, outside the full-size case (round-5
    review): H = 320, two layers, 8 rows per workgroup at 64 rows = the full-band stack's geometry in bench.py's timed region; once as
    the library selects it by itself (>= 128 frames) and once forced on a short launch; the causal parity rule per layer on the
    oracle's own spikes (NEURON:56-61), int8 copies equal to the fp32 spikes."""
    H, nl, I = 320, 2, 64
    rng = np.random.default_rng(4200 + T + len(Rs))
    cells = _cells(rng, I, H, nl)
    o = Oracle("f32")
    xs = [rng.standard_normal((T, R, I)).astype(np.float32) for R in Rs]
    zin0 = [o.linear(x, cells[0][0]["weight_ih"]) for x in xs]
    if force is not None:
        os.environ["SFSN_STACK_FB3"] = force
    try:
        got = run_stack(hip, zin0, cells, T, H, rpw)
    finally:
        os.environ.pop("SFSN_STACK_FB3", None)
    for i, x in enumerate(xs):
        inp, valid = x, np.full(x.shape[1], T)
        for l, (sd, alpha, beta, bnp) in enumerate(cells):
            ref_spk, ref_mem, _, _ = o.gsn_layer(inp, sd["weight_ih"], sd["weight_hh"], sd["bias_ih"], bn=bnp, shared=True)
            valid, st = parity.check_chain(got[l][i][0], ref_spk, np.abs(ref_mem) < parity.TAU, valid, f"fb stack layer {l}")
            assert st["spike_agreement"] > 0.999, st
            np.testing.assert_array_equal(got[l][i][1][:, :, :H], got[l][i][0].astype(np.int8))
            inp = ref_spk
        assert (valid > 0).all()


@pytest.mark.parametrize("wide", [True, False], ids=["wide", "narrow"])
@pytest.mark.parametrize("I,H,nl,Rs,T,rpw", STACKS)
def test_stack_scan_vs_oracle_and_per_layer_calls(hip, I, H, nl, Rs, T, rpw, wide):
    """Every layer's spikes against the oracle's StackedGSU restatement (causal rule: exact until a first flip inside the
    don't-care band) and, bit for bit, against sfsn_spike_proj + sfsn_gsn_layer_scan run layer by layer."""
    from test_hip_parity import run_scan
    rng = np.random.default_rng(H * 100 + nl * 10 + len(Rs))
    cells = _cells(rng, I, H, nl)
    o = Oracle("f32")
    xs = [rng.standard_normal((T, R, I)).astype(np.float32) for R in Rs]
    zin0 = [o.linear(x, cells[0][0]["weight_ih"]) for x in xs]
    got = run_stack(hip, zin0, cells, T, H, rpw, wide=wide)
    for i, x in enumerate(xs):
        # oracle, layer by layer on ITS OWN spikes (NEURON:56-61)
        inp, valid = x, np.full(x.shape[1], T)
        for l, (sd, alpha, beta, bnp) in enumerate(cells):
            ref_spk, ref_mem, _, _ = o.gsn_layer(inp, sd["weight_ih"], sd["weight_hh"], sd["bias_ih"], bn=bnp, shared=True)
            valid, st = parity.check_chain(got[l][i][0], ref_spk, np.abs(ref_mem) < parity.TAU, valid, f"stack H={H} layer {l}")
            assert st["spike_agreement"] > 0.999, st
            np.testing.assert_array_equal(got[l][i][1][:, :, :H], got[l][i][0].astype(np.int8))
            assert not got[l][i][1][:, :, H:].any()
            inp = ref_spk
        # the per-layer entry points on the stack's OWN spikes: bit identity, layer by layer
        for l, (sd, alpha, beta, _) in enumerate(cells):
            zin = zin0[i] if l == 0 else _spike_proj(hip, got[l - 1][i][1], sd["weight_ih"], H)  # (run_scan adds the bias)
            spk, _, s8, hT, cT = run_scan(hip, zin, sd["weight_hh"], sd["bias_ih"], alpha, beta, True, want_mem=False)
            np.testing.assert_array_equal(got[l][i][0], spk)
            np.testing.assert_array_equal(got[l][i][2], hT)
            np.testing.assert_array_equal(got[l][i][3], cT)


def test_fused3_role_equals_the_eight_wave_fused_role(hip, monkeypatch):
    """A/B of the two ways a stack without input-term buffers runs its layers >= 1 at 8 rows per workgroup: the FUSED3 role
    (round 4: 16 waves, input product batched over two frames) against the 8-wave FUSED role it replaced (SFSN_STACK_FUSED8=1),
    full-size sub-band geometry: every word equal."""
    rng = np.random.default_rng(17)
    I, H, nl, Rs, T = 38, 224, 2, [512, 192, 128], 96
    cells = _cells(rng, I, H, nl)
    zin0 = [rng.standard_normal((T, R, H)).astype(np.float32) * 0.5 for R in Rs]
    new = run_stack(hip, zin0, cells, T, H, 8, lag=8, wide=False)
    monkeypatch.setenv("SFSN_STACK_FUSED8", "1")
    old = run_stack(hip, zin0, cells, T, H, 8, lag=8, wide=False)
    monkeypatch.delenv("SFSN_STACK_FUSED8")
    for l in range(nl):
        for i in range(len(Rs)):
            for k in range(4):
                np.testing.assert_array_equal(new[l][i][k], old[l][i][k])


def _input_proj(hip, x, w):
    """sfsn_input_proj_f32(bias = NULL): x [T, R, I] -> [T, R, H] (numpy)."""
    from spiking_fullsubnet_amd._lib import check
    T, R, I = x.shape
    H = w.shape[0]
    tx, tw = _t(x), _t(w)
    z = torch.empty((T * R, H), device=DEV)
    check(hip.sfsn_input_proj_f32(_p(tx), _p(tw), None, _p(z), T * R, I, H, H, None), "input_proj")
    torch.cuda.synchronize()
    return z.cpu().numpy().reshape(T, R, H)


FUSEDX3 = [  # I, H, layers, rows per segment (multiples of 8), which segments take x, T
    (38, 224, 2, [40, 16, 24], [True, True, True], 45), (38, 224, 2, [64, 8], [True, False], 33), (64, 160, 2, [32], [True], 30),
    (30, 96, 3, [16, 8], [True, True], 21), (38, 224, 1, [24], [True], 17), (12, 32, 2, [8], [True], 9), (38, 224, 2, [64], [True], 1),
    (38, 224, 2, [32], [True], 2), (2, 48, 2, [8], [True], 11), (62, 192, 2, [24], [True], 14),
]  # (T x R >= 64 everywhere: below that sfsn_input_proj_f32 itself takes its fp32-MFMA form, whose roundings differ in the last bit)


@pytest.mark.parametrize("I,H,nl,Rs,use_x,T", FUSEDX3)
def test_fusedx3_layer0_role_equals_input_proj_plus_scan(hip, I, H, nl, Rs, use_x, T):
    """Round 4: layer 0 with its real-valued input product INSIDE the 8-row IO-wave scan (sfsn_gsn_stack_scan_x, the bf16 3-way
    split batched over two frames) against sfsn_input_proj_f32 + the per-layer scan, bit for bit -- every layer of the stack (the
    layers above consume its spikes inside the same launch), final states, segments with and without x in one launch, one and two
    k-chunks of the product, T = 1 / 2 / odd."""
    from test_hip_parity import run_scan
    rng = np.random.default_rng(I * 1000 + H + T)
    cells = _cells(rng, I, H, nl)
    xs = [rng.standard_normal((T, R, I)).astype(np.float32) for R in Rs]
    zin0 = [_input_proj(hip, x, cells[0][0]["weight_ih"]) for x in xs]
    got = run_stack(hip, zin0, cells, T, H, 8, wide=False, xs=[x if u else None for x, u in zip(xs, use_x)])
    for i in range(len(Rs)):
        for l, (sd, alpha, beta, _) in enumerate(cells):
            zin = zin0[i] if l == 0 else _spike_proj(hip, got[l - 1][i][1], sd["weight_ih"], H)
            spk, _, s8, hT, cT = run_scan(hip, zin, sd["weight_hh"], sd["bias_ih"], alpha, beta, True, want_mem=False)
            np.testing.assert_array_equal(got[l][i][0], spk)
            np.testing.assert_array_equal(got[l][i][1], s8)
            np.testing.assert_array_equal(got[l][i][2], hT)
            np.testing.assert_array_equal(got[l][i][3], cT)


def test_fusedx3_refuses_what_it_does_not_cover(hip):
    from spiking_fullsubnet_amd import _lib
    rng = np.random.default_rng(4)
    cells = _cells(rng, 38, 224, 2)
    T = 6
    for R, I, rpw in ((12, 38, 8), (16, 37, 8), (16, 38, 4), (16, 66, 8)):  # ragged block, odd I, other rows per workgroup, I > 64
        cells_ = cells if I == 38 else _cells(rng, I, 224, 2)
        x = rng.standard_normal((T, R, I)).astype(np.float32)
        with pytest.raises(NotImplementedError):
            run_stack(hip, [np.zeros((T, R, 224), np.float32)], cells_, T, 224, rpw, wide=False, xs=[x])


def _spike_proj(hip, s8, w, H):
    """sfsn_spike_proj(bias = NULL) of int8 spikes [T, R, HP] -> [T, R, H] (numpy)."""
    from spiking_fullsubnet_amd._lib import check
    from spiking_fullsubnet_amd.engine import pack_w3
    T, R, HP = s8.shape
    pk, dq = pack_w3(w)
    ts, tp, td = _t(s8), _t(pk), _t(dq)
    y = torch.empty((T * R, H), device=DEV)
    check(hip.sfsn_spike_proj(_p(ts), _p(tp), _p(td), None, _p(y), T * R, H, H, H, None), "spike_proj")
    torch.cuda.synchronize()
    return y.cpu().numpy().reshape(T, R, H)


def test_stack_scan_state_carry_and_int8_only_outputs(hip):
    """Two half-length launches carrying (h, c) of every layer == one launch; the int8-only variant (no fp32 spike tensors)
    writes the same spikes."""
    rng = np.random.default_rng(5)
    I, H, nl, Rs, T = 38, 224, 2, [24, 10], 36
    cells = _cells(rng, I, H, nl)
    o = Oracle("f32")
    zin0 = [o.linear(rng.standard_normal((T, R, I)).astype(np.float32), cells[0][0]["weight_ih"]) for R in Rs]
    full = run_stack(hip, zin0, cells, T, H, 8)
    a = run_stack(hip, [z[:15] for z in zin0], cells, 15, H, 8)
    b = run_stack(hip, [z[15:] for z in zin0], cells, T - 15, H, 8, h0=[[a[l][i][2] for i in range(2)] for l in range(nl)],
                  c0=[[a[l][i][3] for i in range(2)] for l in range(nl)])
    lean = run_stack(hip, zin0, cells, T, H, 8, want_f32=False)
    for l in range(nl):
        for i in range(2):
            np.testing.assert_array_equal(np.concatenate([a[l][i][0], b[l][i][0]]), full[l][i][0])
            np.testing.assert_array_equal(b[l][i][3], full[l][i][3])
            np.testing.assert_array_equal(lean[l][i][1], full[l][i][1])


def test_stack_scan_hand_off_under_uneven_load(hip):
    """The in-launch hand-off (write-through stores, progress counters, sc1 loads) with the chip busy with OTHER work, so
    that producers and consumers are dispatched late, unevenly and on different XCDs: full-size sub-band geometry, several
    repetitions, every word of every layer compared with an undisturbed run."""
    rng = np.random.default_rng(11)
    I, H, nl, Rs, T = 38, 224, 2, [512, 192, 128], 300
    cells = _cells(rng, I, H, nl)
    zin0 = [rng.standard_normal((T, R, H)).astype(np.float32) * 0.5 for R in Rs]
    ref = run_stack(hip, zin0, cells, T, H, 8, lag=16)
    noise_a = torch.randn((1 << 25,), device=DEV)
    side = torch.cuda.Stream(device=DEV)
    for rep in range(4):
        with torch.cuda.stream(side):
            for _ in range(6):  # bandwidth hogs + compute hogs that occupy CUs in waves
                noise_b = noise_a * 1.0001 + 0.5
                noise_c = torch.mm(noise_a[: 2048 * 2048].view(2048, 2048), noise_b[: 2048 * 2048].view(2048, 2048))
        got = run_stack(hip, zin0, cells, T, H, 8, lag=(1, 4, 16, 64)[rep])
        torch.cuda.synchronize()
        for l in range(nl):
            for i in range(len(Rs)):
                np.testing.assert_array_equal(got[l][i][0], ref[l][i][0])
                np.testing.assert_array_equal(got[l][i][1], ref[l][i][1])


def test_stack_scan_rejects_what_it_does_not_cover(hip):
    from spiking_fullsubnet_amd import _lib
    from spiking_fullsubnet_amd._lib import FusedInput, ScanSegment
    rng = np.random.default_rng(2)
    I, H, nl, R, T = 12, 32, 2, 4, 8
    cells = _cells(rng, I, H, nl)
    zin0 = [rng.standard_normal((T, R, H)).astype(np.float32)]
    run_stack(hip, zin0, cells, T, H, 4)  # the valid call
    segs, fin = (ScanSegment * 2)(), (FusedInput * 2)()
    scratch = torch.zeros((4096,), dtype=torch.int32, device=DEV)
    rp = (ctypes.c_int * 2)(4, 4)
    assert hip.sfsn_gsn_stack_scan(None, fin, 2, 1, T, H, rp, 4, _p(scratch), 16384, None) == _lib.SFSN_EINVAL
    assert hip.sfsn_gsn_stack_scan(segs, fin, 2, 1, T, 30, rp, 4, _p(scratch), 16384, None) == _lib.SFSN_EUNSUPPORTED  # H % 16
    assert hip.sfsn_gsn_stack_scan(segs, fin, 2, 1, T, H, rp, 4, None, 16384, None) == _lib.SFSN_EINVAL            # no scratch
    assert hip.sfsn_gsn_stack_scan(segs, fin, 2, 1, T, H, rp, 4, _p(scratch), 16384, None) == _lib.SFSN_EINVAL     # NULL segments
    bad = (ctypes.c_int * 2)(4, 5)
    assert hip.sfsn_gsn_stack_scan(segs, fin, 2, 1, T, H, bad, 4, _p(scratch), 16384, None) == _lib.SFSN_EINVAL    # rows per workgroup
    assert hip.sfsn_stack_scratch_bytes(0, 1, 4) == 0


def _build(front, kw, sd):
    import spiking_fullsubnet_amd as pkg
    cls = pkg.SpikingFullSubNet if front == "live" else pkg.Separator
    m = cls(**kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.eval().to(DEV)


@pytest.mark.parametrize("front,kw,seed", [("live", rw.LIVE_M, 5), ("frozen", rw.FROZEN_S, 6), ("frozen", rw.FROZEN_L, 9),
                                           ("live", rw.LIVE_TINY, 11), ("live", rw.LIVE_TINY_2SPK, 8)])
def test_modules_with_stack_scan_are_bit_identical_to_the_per_layer_schedule(front, kw, seed):
    """Every tensor the reference's forward() returns, stack scan (all rows-per-workgroup choices, chunked with state
    carry, int8-only outputs) against the per-layer schedule: hidden sizes 320 / 240 (PROJ + gated scan roles), 224 / 160 /
    256 / 48 (fused roles), 3 and 4 sub-band groups, two speakers."""
    sd = rw.live_state_dict(kw, seed) if front == "live" else rw.frozen_state_dict(kw, seed)
    model = _build(front, kw, sd)
    stft = model._stft(torch.from_numpy(rw.synth_wave(5, 70, seed)).to(DEV))  # odd batch: ragged row tiles
    eng = model.engine()
    eng.stack_scan = False
    ref = eng.forward_stft(stft)
    torch.cuda.synchronize()
    n0 = eng.launches.get("stack", 0)
    outs = []
    eng.stack_scan = True
    for rp, chunk in (((4, 8), 0), ((8, 16), 0), ((16, 4), 0), ((4, 8), 32)):
        eng.stack_rows_per_wg, eng.seq_chunk = {"fb": rp[0], "sb": rp[1]}, chunk
        outs.append(eng.forward_stft(stft))
        eng.check_stack_errors()
    eng.seq_chunk = 0
    eng.stack_scan = "auto"
    outs.append(eng.forward_stft(stft))
    eng.stack_rows_fb_auto = 8  # the full-band stack's geometry in bench.py's timed region
    outs.append(eng.forward_stft(stft))
    eng.stack_rows_fb_auto = 4
    lean = eng.forward_stft(stft, want_layers=False, want_counts=True)
    eng.check_stack_errors()
    assert eng.launches.get("stack", 0) >= n0 + 8
    for b in outs + [lean]:
        assert torch.equal(torch.view_as_real(ref["enh_stft"]), torch.view_as_real(b["enh_stft"]))
        assert torch.equal(ref["enh_mag"], b["enh_mag"])
        for x, y in zip(ref["fb_all"] + sum(ref["sb_all"], []), b["fb_all"] + sum(b["sb_all"], [])):
            if torch.is_tensor(y):
                assert torch.equal(x, y)
            else:
                assert int(y.count.item()) == int((x > 0).sum().item())


def test_full_size_stack_scan_equals_the_per_layer_forward():
    """BASELINE configs[2] sizes (B=64, T=1000, live baseline_m): the default schedule (full-band stack in one pipelined
    launch), the all-stacks schedule and the per-layer schedule return the same tensors, bit for bit."""
    kw, seed = rw.LIVE_M, 21
    model = _build("live", kw, rw.live_state_dict(kw, seed))
    eng = model.engine()
    stft = model._stft(torch.from_numpy(rw.synth_wave(64, 1000, 3)).to(DEV))
    eng.stack_scan = False
    ref = eng.forward_stft(stft)
    torch.cuda.synchronize()
    for mode in ("auto", True):
        eng.stack_scan = mode
        b = eng.forward_stft(stft)
        eng.check_stack_errors()
        assert torch.equal(torch.view_as_real(ref["enh_stft"]), torch.view_as_real(b["enh_stft"]))
        for x, y in zip(ref["fb_all"] + sum(ref["sb_all"], []), b["fb_all"] + sum(b["sb_all"], [])):
            assert torch.equal(x, y)
    eng.stack_scan = "auto"


@pytest.mark.parametrize("front,kw,seed", [("live", rw.LIVE_M, 5), ("live", rw.LIVE_TINY_2SPK, 8), ("frozen", rw.FROZEN_S, 6)])
def test_full_band_sub_band_chunk_overlap_is_bit_identical(front, kw, seed):
    """`overlap_chunks`: the sequence in chunks, the full-band model on one stream and the sub-band models one chunk behind on a
    second one (states carried through the ABI): every returned tensor equals the one-stream forward, bit for bit.  The frozen
    front-end's utterance-level Laplace means need the whole full-band output: it silently stays on one stream."""
    sd = rw.live_state_dict(kw, seed) if front == "live" else rw.frozen_state_dict(kw, seed)
    model = _build(front, kw, sd)
    stft = model._stft(torch.from_numpy(rw.synth_wave(3, 400, seed)).to(DEV))
    eng = model.engine()
    eng.overlap_chunks = 0
    ref = eng.forward_stft(stft)
    torch.cuda.synchronize()
    for n in (2, 4):
        eng.overlap_chunks = n
        b = eng.forward_stft(stft)
        eng.check_stack_errors()
        assert b["overlapped"] == (front == "live") and b["n_chunks"] == (n if front == "live" else 1)
        assert torch.equal(torch.view_as_real(ref["enh_stft"]), torch.view_as_real(b["enh_stft"]))
        assert torch.equal(ref["enh_mag"], b["enh_mag"])
        for x, y in zip(ref["fb_all"] + sum(ref["sb_all"], []), b["fb_all"] + sum(b["sb_all"], [])):
            assert torch.equal(x, y)
