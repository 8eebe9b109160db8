"""GPU parity tests: the HIP path (through the C ABI / the drop-in modules) against the CPU oracle on the
same seeded inputs, against the committed golden fixtures made by the reference, and -- at BASELINE.json's
full sizes -- through size-independent properties (state carry across chunks, batch independence,
run-to-run bit stability).  Tolerances and the causal comparison rule live in tests/parity.py."""
import ctypes
import os
import time

import numpy as np
import pytest
import torch

import parity
import refweights as rw
from oracle import Oracle
from oracle import model as omodel

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


def load(name):
    return dict(np.load(os.path.join(G, name), allow_pickle=False))


def _t(a, dtype=None):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV) if dtype is None else torch.from_numpy(np.ascontiguousarray(a)).to(DEV, dtype)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


@pytest.fixture(scope="module")
def hip():
    from spiking_fullsubnet_amd import _lib
    L = _lib.lib()
    assert L.sfsn_device_count() >= 1
    return L


def run_scan(hip, zin, w_hh, bias, alpha, beta, shared, h0=None, c0=None, want_mem=True, split=False, want_spk=True):
    """x.W_ih^T [T,R,G*H] numpy -> spikes, membrane, spikes_i8, hT, cT (numpy) through sfsn_gsn_layer_scan.
    The ABI's input term includes bias_ih (forget-gate bias when shared, both gate biases otherwise): added here."""
    from spiking_fullsubnet_amd._lib import ScanSegment, check
    from spiking_fullsubnet_amd.engine import pack_w3
    T, R, GH = zin.shape
    H = bias.shape[0] // 2
    zin = (zin + bias[:GH]).astype(np.float32)
    HP = (H + 63) // 64 * 64
    pk, dq = pack_w3(w_hh)
    t = dict(zin=_t(zin), pk=_t(pk), dq=_t(dq), bias=_t(bias), alpha=_t(alpha), beta=_t(beta),
             h=_t(np.zeros((R, H), np.float32) if h0 is None else h0), c=_t(np.zeros((R, H), np.float32) if c0 is None else c0),
             spk=torch.empty((T, R, H), device=DEV) if want_spk else None, s8=torch.zeros((T, R, HP), dtype=torch.int8, device=DEV),
             mem=torch.empty((T, R, H), device=DEV) if want_mem else None)
    seg = (ScanSegment * 1)()
    s = seg[0]
    s.zin, s.w_hh, s.w_dq, s.bias, s.bn_alpha, s.bn_beta = _p(t["zin"]), _p(t["pk"]), _p(t["dq"]), _p(t["bias"]), _p(t["alpha"]), _p(t["beta"])
    s.h_state, s.c_state, s.spikes_f32, s.spikes_i8, s.membrane, s.R = _p(t["h"]), _p(t["c"]), _p(t["spk"]), _p(t["s8"]), _p(t["mem"]), R
    if split:  # the tiles of a row block split over several workgroups (sfsn_gsn_layer_scan_split)
        scr = torch.zeros((hip.sfsn_scan_split_scratch_bytes(R, H) // 4,), dtype=torch.int32, device=DEV)
        check(hip.sfsn_gsn_layer_scan_split(seg, 1, T, H, int(shared), _p(scr), scr.numel() * 4, None), "scan_split")
        torch.cuda.synchronize()
        assert int(scr[0]) == 0, "a wait of the split scan expired"
    else:
        check(hip.sfsn_gsn_layer_scan(seg, 1, T, H, int(shared), 0, None), "scan")
    torch.cuda.synchronize()
    return (t["spk"].cpu().numpy() if want_spk else None, t["mem"].cpu().numpy() if want_mem else None, t["s8"].cpu().numpy(), t["h"].cpu().numpy(),
            t["c"].cpu().numpy())


def make_layer(rng, I, H, shared, bn):
    sd = {}
    rw._cell(rng, "", I, H, shared, bn, sd)
    from spiking_fullsubnet_amd.engine import fold_batchnorm
    if bn:
        alpha, beta = fold_batchnorm(sd["batchnorm.weight"], sd["batchnorm.bias"], sd["batchnorm.running_mean"], sd["batchnorm.running_var"])
        bnp = (sd["batchnorm.weight"], sd["batchnorm.bias"], sd["batchnorm.running_mean"], sd["batchnorm.running_var"])
    else:
        alpha, beta, bnp = np.ones(H, np.float32), np.zeros(H, np.float32), None
    return sd, alpha, beta, bnp


SCAN_SHAPES = [  # I, H, R, T, shared, bn
    (12, 32, 5, 40, True, True), (9, 16, 4, 30, False, False), (20, 48, 19, 25, False, True), (38, 160, 33, 30, True, True),
    (38, 224, 16, 40, True, True), (64, 240, 7, 30, True, True), (30, 256, 18, 20, True, False), (64, 320, 35, 30, True, True),
    (94, 128, 16, 24, True, True), (24, 192, 9, 16, False, True), (64, 320, 21, 14, False, True), (20, 272, 5, 10, False, False),
]


@pytest.mark.parametrize("I,H,R,T,shared,bn", SCAN_SHAPES)
def test_scan_free_running_vs_oracle(hip, I, H, R, T, shared, bn):
    """sfsn_gsn_layer_scan vs the oracle's gsn_layer on identical zin-equivalent inputs (causal rule)."""
    rng = np.random.default_rng(I * 1000 + H)
    sd, alpha, beta, bnp = make_layer(rng, I, H, shared, bn)
    x = rng.standard_normal((T, R, I)).astype(np.float32)
    h0 = (rng.random((R, H)) > 0.5).astype(np.float32)
    c0 = rng.standard_normal((R, H)).astype(np.float32)
    o = Oracle("f32")
    zin = o.linear(x, sd["weight_ih"])  # x . W_ih^T, correctly rounded
    ref_spk, ref_mem, ref_h, ref_c = o.gsn_layer(x, sd["weight_ih"], sd["weight_hh"], sd["bias_ih"], bn=bnp, shared=shared, h0=h0, c0=c0)
    spk, mem, s8, hT, cT = run_scan(hip, zin, sd["weight_hh"], sd["bias_ih"], alpha, beta, shared, h0, c0)
    t_valid, st = parity.check_chain(spk, ref_spk, np.abs(ref_mem) < parity.TAU, np.full(R, T), f"scan H={H}", mem, ref_mem)
    assert st["spike_agreement"] > 0.999, st
    np.testing.assert_array_equal(s8[:, :, :H], spk.astype(np.int8))  # int8 copy == fp32 spikes
    assert not s8[:, :, H:].any()
    ok = t_valid == T
    np.testing.assert_array_equal(hT[ok], ref_h[ok])
    np.testing.assert_array_equal(hT, spk[-1])
    np.testing.assert_allclose(cT[ok], ref_c[ok], atol=parity.MEM_ATOL, rtol=parity.MEM_RTOL)


@pytest.mark.gpu
@pytest.mark.parametrize("H,R,T", [(224, 13, 21), (224, 4, 1), (224, 33, 2), (160, 9, 17), (96, 6, 11), (208, 16, 9)])
def test_separate_gate_weights_scan_with_io_waves_equals_round_2_body(hip, H, R, T, monkeypatch):
    """Round 6: sfsn_gsn_layer_scan with shared = 0 at 4 rows per workgroup and at most 14 tiles runs scan3g_role (both gates of a tile
    in one compute wave, digit plane 0 of both in LDS; baseline_xl's sub-band layers: baseline_xl.toml:61,64, NEURON:137-139).  Bit
    for bit round 2's body (SFSN_SCAN_V2=1) -- fp32 / int8 spikes, final h and c -- and, through test_scan_free_running_vs_oracle's
    unshared shapes, the oracle."""
    rng = np.random.default_rng(H * 3 + R)
    sd, alpha, beta, bnp = make_layer(rng, 30, H, False, True)
    x = rng.standard_normal((T, R, 30)).astype(np.float32)
    zin = Oracle("f32").linear(x, sd["weight_ih"])
    h0 = (rng.random((R, H)) > 0.5).astype(np.float32)
    c0 = rng.standard_normal((R, H)).astype(np.float32)
    new = run_scan(hip, zin, sd["weight_hh"], sd["bias_ih"], alpha, beta, False, h0, c0, want_mem=False)
    monkeypatch.setenv("SFSN_SCAN_V2", "1")
    old = run_scan(hip, zin, sd["weight_hh"], sd["bias_ih"], alpha, beta, False, h0, c0, want_mem=False)
    monkeypatch.delenv("SFSN_SCAN_V2")
    for a, b, nm in zip(new, old, ("spikes", "membrane", "spikes_i8", "h", "c")):
        if a is not None:
            np.testing.assert_array_equal(a, b, err_msg=nm)
    assert new[2].any()
    lean = run_scan(hip, zin, sd["weight_hh"], sd["bias_ih"], alpha, beta, False, h0, c0, want_mem=False, want_spk=False)
    np.testing.assert_array_equal(lean[2], new[2])
    np.testing.assert_array_equal(lean[4], new[4])


@pytest.mark.gpu
@pytest.mark.parametrize("H,R,T", [(320, 64, 60), (320, 21, 33), (272, 40, 25), (320, 130, 17)])
def test_split_scan_for_large_separate_gate_weights_equals_the_streamed_scan(hip, H, R, T):
    """sfsn_gsn_layer_scan_split (round 5: separate gate weights that do not fit one CU -- baseline_xl's full-band model): the tiles
    of a 16-row block over several workgroups with resident weights and a tagged spike exchange per step give the bits of
    sfsn_gsn_layer_scan's streamed-weights kernel -- fp32 spikes, membranes, int8 spikes, final state -- for ragged row counts, a
    hidden size that does not fill the last split, a sequence continued in a second call from the first call's state, and without the
    fp32 outputs; shapes one compute unit serves are refused."""
    from spiking_fullsubnet_amd import _lib
    rng = np.random.default_rng(H + R)
    I = 24
    sd, alpha, beta, bnp = make_layer(rng, I, H, False, True)
    x = rng.standard_normal((T, R, I)).astype(np.float32)
    zin = Oracle("f32").linear(x, sd["weight_ih"])
    h0 = (rng.random((R, H)) > 0.5).astype(np.float32)
    c0 = rng.standard_normal((R, H)).astype(np.float32)
    a = run_scan(hip, zin, sd["weight_hh"], sd["bias_ih"], alpha, beta, False, h0, c0)
    b = run_scan(hip, zin, sd["weight_hh"], sd["bias_ih"], alpha, beta, False, h0, c0, split=True)
    for u, v, nm in zip(a, b, ("spikes", "membrane", "spikes_i8", "h", "c")):
        np.testing.assert_array_equal(u, v, err_msg=nm)
    # two calls, the second from the first one's state; no fp32 outputs
    t1 = T // 2
    c1 = run_scan(hip, zin[:t1], sd["weight_hh"], sd["bias_ih"], alpha, beta, False, h0, c0, want_mem=False, split=True, want_spk=False)
    c2 = run_scan(hip, zin[t1:], sd["weight_hh"], sd["bias_ih"], alpha, beta, False, c1[3], c1[4], want_mem=False, split=True, want_spk=False)
    np.testing.assert_array_equal(np.concatenate([c1[2], c2[2]]), a[2])
    np.testing.assert_array_equal(c2[3], a[3])
    np.testing.assert_array_equal(c2[4], a[4])


@pytest.mark.gpu
def test_counted_wait_jump_table_lands_on_every_entry(hip):
    """wait_vmcnt_n (sfsn_scan_dev.h): `s_waitcnt vmcnt(n)` for a run-time n is a computed jump into a table of 64 entries; a wrong
    offset would be a wrong wait count -- a silent race on the LDS ring, not a crash (round-5 advisor finding).  The probe form of the
    same jump (entries record their index) must land on entry n for every n."""
    out = (ctypes.c_int * 64)()
    hip.sfsn_debug_vmcnt_table.restype = ctypes.c_int
    hip.sfsn_debug_vmcnt_table.argtypes = [ctypes.c_void_p]
    assert hip.sfsn_debug_vmcnt_table(out) == 0
    assert list(out) == list(range(64))


@pytest.mark.gpu
@pytest.mark.parametrize("R,T", [(21, 30), (64, 40), (130, 22)])
def test_split_scan_vs_oracle(hip, R, T):
    """sfsn_gsn_layer_scan_split against the ORACLE's gsn_layer directly (round-5 review: the split kernel met the oracle only through
    its equality with the streamed kernel): separate gate weights, H = 320 (baseline_xl's full-band layer, baseline_xl.toml:61,64;
    efficient_spiking_neuron.py:137-139), ragged / whole / many row blocks, non-zero initial state, the causal parity rule."""
    I, H = 40, 320
    rng = np.random.default_rng(7000 + R)
    sd, alpha, beta, bnp = make_layer(rng, I, H, False, True)
    x = rng.standard_normal((T, R, I)).astype(np.float32)
    h0 = (rng.random((R, H)) > 0.5).astype(np.float32)
    c0 = rng.standard_normal((R, H)).astype(np.float32)
    o = Oracle("f32")
    zin = o.linear(x, sd["weight_ih"])
    ref_spk, ref_mem, ref_h, ref_c = o.gsn_layer(x, sd["weight_ih"], sd["weight_hh"], sd["bias_ih"], bn=bnp, shared=False, h0=h0, c0=c0)
    spk, mem, s8, hT, cT = run_scan(hip, zin, sd["weight_hh"], sd["bias_ih"], alpha, beta, False, h0, c0, split=True)
    t_valid, st = parity.check_chain(spk, ref_spk, np.abs(ref_mem) < parity.TAU, np.full(R, T), f"split scan R={R}", mem, ref_mem)
    assert st["spike_agreement"] > 0.999, st
    np.testing.assert_array_equal(s8[:, :, :H], spk.astype(np.int8))
    assert not s8[:, :, H:].any()
    ok = t_valid == T
    assert ok.sum() >= R // 2, "too few rows survive to the end for the final-state check to mean anything"
    np.testing.assert_array_equal(hT[ok], ref_h[ok])
    np.testing.assert_array_equal(hT, spk[-1])
    np.testing.assert_allclose(cT[ok], ref_c[ok], atol=parity.MEM_ATOL, rtol=parity.MEM_RTOL)


@pytest.mark.gpu
def test_split_scan_refuses_what_one_compute_unit_serves(hip):
    from spiking_fullsubnet_amd import _lib
    from spiking_fullsubnet_amd._lib import ScanSegment
    seg = (ScanSegment * 1)()
    one = ctypes.c_void_p(256)
    assert hip.sfsn_gsn_layer_scan_split(seg, 1, 5, 224, 0, one, 1 << 20, None) == _lib.SFSN_EUNSUPPORTED   # fits one CU
    assert hip.sfsn_gsn_layer_scan_split(seg, 1, 5, 320, 1, one, 1 << 20, None) == _lib.SFSN_EUNSUPPORTED   # shared gates
    assert hip.sfsn_gsn_layer_scan_split(seg, 2, 5, 320, 0, one, 1 << 20, None) == _lib.SFSN_EUNSUPPORTED   # one segment per launch
    assert hip.sfsn_gsn_layer_scan_split(seg, 1, 5, 320, 0, one, 1 << 20, None) == _lib.SFSN_EINVAL         # empty descriptor
    assert hip.sfsn_gsn_layer_scan_split(seg, 1, 5, 320, 0, None, 0, None) == _lib.SFSN_EINVAL


def test_checkpoint_directory_loaded_on_the_device_passes_parity(tmp_path):
    """SURVEY 8f rank 3 on the GPU: an accelerate-style checkpoint directory (pytorch_model.bin of the generator next to a
    discriminator file, audiozen/trainer.py:225,238-242) with the trained model_zoo baseline_m weights, loaded into a module
    that is ALREADY on the device through checkpoint.load_checkpoint(prepack=True) -- the kernel-side weight images are built
    at load time -- then the golden fixture's parity check, spike for spike."""
    import spiking_fullsubnet_amd as pkg
    from safetensors.torch import save_file
    from spiking_fullsubnet_amd import checkpoint
    gold = load("frozen_m_zoo.npz")
    zoo = {k[3:]: torch.from_numpy(gold[k]) for k in gold if k.startswith("sd/")}
    d_bin, d_st = tmp_path / "best", tmp_path / "epoch_0001"
    d_bin.mkdir(), d_st.mkdir()
    torch.save(zoo, d_bin / "pytorch_model.bin")
    torch.save({"junk": torch.zeros(1)}, d_bin / "pytorch_model_1.bin")
    save_file({("module." + k): v.contiguous() for k, v in zoo.items()}, str(d_st / "model.safetensors"))
    spec = omodel.spec_from_frozen_kwargs(rw.FROZEN_M)
    for d in (d_bin, d_st):
        model = pkg.Separator(**rw.FROZEN_M).eval().to(DEV)
        e0 = model._engine if hasattr(model, "_engine") else None
        missing, unexpected = checkpoint.load_checkpoint(model, str(d), prepack=True)
        assert missing == [] and unexpected == []
        eng = model._engine
        assert eng is not None and eng is not e0 and eng.fb.cells[0].w_hh_q.device.type == "cuda"  # packed at load time
        out = hip_result(model, gold["stft"])
        assert model._engine is eng  # the forward reused the pre-packed weights
        stats = parity.check_model(out, gold, spec, tag=f"checkpoint {d.name}:")
        parity.report(f"checkpoint-bridge:{d.name}:frozen_m_zoo", stats)
        for st in stats:
            assert st["diverged"] == 0 and st["spike_agreement"] == 1.0, st


def test_scratch_cache_is_keyed_on_capacity_and_bounded():
    """evaluation over clips of many lengths must not pin one scratch set per length.  One buffer set per (stack,
    rows, stream), grown to the longest clip; a byte budget evicts least-recently-used sets; results do not depend on what the
    cache held before (bit identity with a fresh engine)."""
    kw, seed = rw.LIVE_M, 5
    sd = rw.live_state_dict(kw, seed)
    model = build_module("live", kw, sd)
    eng = model.engine()
    waves = {T: torch.from_numpy(rw.synth_wave(2, T, T)).to(DEV) for T in (40, 90, 33, 64, 90, 17)}
    outs = {}
    for T, w in waves.items():
        outs[T] = eng.forward_stft(model._stft(w))
    torch.cuda.synchronize()
    stacks = [k for k in eng._ws if k[0] in ("fb", "sb")]
    assert len(stacks) == 2, list(eng._ws)  # one full-band and one sub-band set, not one per length
    assert all(eng._ws[k]["T"] == 90 for k in stacks)
    fresh = build_module("live", kw, sd).engine()
    for T in (33, 17):
        ref = fresh.forward_stft(model._stft(waves[T]))
        assert torch.equal(torch.view_as_real(ref["enh_stft"]), torch.view_as_real(outs[T]["enh_stft"]))
        for x, y in zip(ref["fb_all"] + sum(ref["sb_all"], []), outs[T]["fb_all"] + sum(outs[T]["sb_all"], [])):
            assert torch.equal(x, y)
    # other batch sizes add sets; a tiny budget keeps only the most recent ones
    eng.ws_budget_bytes = 1
    for B in (1, 3, 4):
        eng.forward_stft(model._stft(torch.from_numpy(rw.synth_wave(B, 20, B)).to(DEV)))
    torch.cuda.synchronize()
    assert len(eng._ws) <= 3, list(eng._ws)


def test_stft_and_istft_attributes_keep_the_reference_signatures():
    """`model.stft(y)` is `partial(audio_feature.stft, ...)` in the reference (MODEL:404-405): (mag, phase, real, imag), or by
    output_type; `model.istft(feature, length=, input_type=)` accepts the three feature forms (audio_feature.py:236-347)."""
    model = build_module("live", rw.LIVE_TINY, rw.live_state_dict(rw.LIVE_TINY, 11))
    y = torch.from_numpy(rw.synth_wave(2, 30, 1)).to(DEV)
    mag, phase, real, imag = model.stft(y)
    c = torch.stft(y, 512, 128, 512, window=torch.hann_window(512, device=DEV), return_complex=True, pad_mode="constant")
    assert mag.shape == c.shape and torch.allclose(real, c.real, atol=2e-5) and torch.allclose(imag, c.imag, atol=2e-5)
    assert torch.allclose(mag, c.abs(), atol=2e-5) and torch.allclose(torch.polar(mag, phase), c, atol=3e-5)
    assert torch.is_complex(model.stft(y, output_type="complex")) and len(model.stft(y, output_type="real_imag")) == 2
    assert len(model.stft(y, output_type="mag_phase")) == 2
    assert model.stft(y.reshape(1, 2, -1))[0].shape == (1, 2, 257, c.shape[-1])  # multi-channel input, audio_feature.py:265-284
    ref = torch.istft(c, 512, 128, 512, window=torch.hann_window(512, device=DEV), length=y.shape[-1])
    for feat, kind in ((c, "complex"), ((c.real, c.imag), "real_imag"), ((c.abs(), c.angle()), "mag_phase")):
        assert torch.allclose(model.istft(feat, length=y.shape[-1], input_type=kind), ref, atol=2e-5)
    with pytest.raises(ValueError):
        model.istft(c.real, input_type="complex")
    with pytest.raises(ValueError):
        model.stft(y.reshape(1, 1, 2, -1))


def test_sixteen_bit_weight_mode_report():
    """BASELINE configs[2] asks for a 16-bit mode; SURVEY 0: 16-bit weights cannot meet 1e-4 against the fp32 oracle, so the parity
    gate stays the exact mode and THIS mode reports, per layer, its spike agreement and, at the output, its error -- beside the
    floor any fp32 evaluation has (fp32 oracle vs fp64 oracle on the same input).  Trained zoo weights (baseline_m), 400 frames.
    What it shows (numbers in the parity report): a 2^-16 weight perturbation flips the first spikes within a few frames and the
    chains then decorrelate like any perturbed spiking recurrence -- ~90 % per-layer spike agreement and ~0.4 relative L2 at the
    output after 400 frames, against 0.995+ / 0.11 for fp32-vs-fp64 and 0.996+ / 0.03 for the exact mode.  That is why the exact
    mode is the parity gate and the headline.  Asserted here: the mode runs every kernel path (same layout, zero least-
    significant digit plane), its output is finite and correlated with the reference, and the exact mode sits on the fp32 floor."""
    T = 400
    gold = load("frozen_m_zoo.npz")
    kw = rw.FROZEN_M
    sd = {k[3:]: v for k, v in gold.items() if k.startswith("sd/")}
    spec = omodel.spec_from_frozen_kwargs(kw)
    wave = torch.from_numpy(rw.synth_wave(2, T, 17, modulated=True))
    stft = torch.stft(wave, 512, 128, 512, window=torch.hann_window(512), return_complex=True, pad_mode="constant").numpy()
    o32 = omodel.forward_from_stft(spec, sd, stft, "f32")
    o64 = omodel.forward_from_stft(spec, sd, stft, "f64")
    model = build_module("frozen", kw, sd)
    exact = hip_result(model, stft, want_membrane=False)
    model.weight_bits = 16
    fast = hip_result(model, stft, want_membrane=False)
    assert model.engine().weight_bits == 16
    model.weight_bits = 24

    def layers(res):
        return [np.asarray(a) > 0.5 for a in res["fb_all"][1:-1]] + [np.asarray(a) > 0.5 for l in res["sb_all"] for a in l[1:-1]]

    def rel(a, b):
        return float(np.linalg.norm((np.asarray(a) - np.asarray(b)).ravel()) / np.linalg.norm(np.asarray(b).ravel()))

    agree16 = [float((a == b).mean()) for a, b in zip(layers(fast), layers(o32))]
    agree24 = [float((a == b).mean()) for a, b in zip(layers(exact), layers(o32))]
    floor = [float((a == b).mean()) for a, b in zip(layers(o32), layers(o64))]
    def first_flip(a, b):
        d = (a != b).reshape(a.shape[0], -1).any(1)
        return int(np.argmax(d)) if d.any() else T

    rep = dict(frames=T, clips=2, weights="model_zoo baseline_m (trained)",
               first_differing_frame_vs_fp32_oracle=dict(weight_bits_16=[first_flip(a, b) for a, b in zip(layers(fast), layers(o32))],
                                                         exact_mode=[first_flip(a, b) for a, b in zip(layers(exact), layers(o32))],
                                                         fp32_oracle_vs_fp64_oracle=[first_flip(a, b) for a, b in zip(layers(o32), layers(o64))]),
               spike_agreement_vs_fp32_oracle=dict(weight_bits_16=agree16, exact_mode=agree24, fp32_oracle_vs_fp64_oracle=floor),
               enh_stft_rel_l2_vs_fp32_oracle=dict(weight_bits_16=rel(fast["enh_stft"], o32["enh_stft"]), exact_mode=rel(exact["enh_stft"], o32["enh_stft"]),
                                                   fp32_oracle_vs_fp64_oracle=rel(o32["enh_stft"], o64["enh_stft"])))
    parity.report("sixteen-bit-weight-mode:frozen_m_zoo", [], extra=rep)
    assert np.isfinite(fast["enh_stft"]).all() and min(agree16) > 0.8 and rep["enh_stft_rel_l2_vs_fp32_oracle"]["weight_bits_16"] < 0.7, rep
    assert min(agree24) >= min(floor) - 2e-3, rep
    assert rep["enh_stft_rel_l2_vs_fp32_oracle"]["exact_mode"] <= 1.15 * rep["enh_stft_rel_l2_vs_fp32_oracle"]["fp32_oracle_vs_fp64_oracle"] + 1e-4, rep


def test_scan_teacher_forced_single_steps(hip):
    """Each step started from the ORACLE's state (T=1 launches): membranes within 1e-5 + 2e-6*|c| per step and spikes
    equal wherever the oracle membrane is outside the +-TAU band -- no error can accumulate along the chain."""
    rng = np.random.default_rng(7)
    I, H, R, T = 38, 224, 16, 12
    sd, alpha, beta, bnp = make_layer(rng, I, H, True, True)
    x = rng.standard_normal((T, R, I)).astype(np.float32)
    o = Oracle("f32")
    zin = o.linear(x, sd["weight_ih"])
    ref_spk, ref_mem, _, _ = o.gsn_layer(x, sd["weight_ih"], sd["weight_hh"], sd["bias_ih"], bn=bnp)
    h, c = np.zeros((R, H), np.float32), np.zeros((R, H), np.float32)
    for t in range(T):
        spk, mem, _, hT, cT = run_scan(hip, zin[t:t + 1], sd["weight_hh"], sd["bias_ih"], alpha, beta, True, h, c)
        err = np.abs(mem[0].astype(np.float64) - ref_mem[t])
        assert (err <= 1e-5 + 2e-6 * np.abs(ref_mem[t])).all(), (t, err.max())
        far = np.abs(ref_mem[t]) >= parity.TAU
        np.testing.assert_array_equal(spk[0][far], ref_spk[t][far])
        h, c = ref_spk[t], ref_mem[t]


def test_scan_multi_segment_and_chunked_state_carry(hip):
    """Three segments of different R in one launch == three single launches; two half-length launches carrying
    (h, c) == one full launch, bit for bit (state in/out contract of StackedGSU, efficient_spiking_neuron.py:50-62)."""
    from spiking_fullsubnet_amd._lib import ScanSegment, check
    from spiking_fullsubnet_amd.engine import pack_w3
    rng = np.random.default_rng(3)
    H, T = 160, 22
    Rs = [40, 9, 17]
    layers = [make_layer(rng, 8, H, True, True) for _ in Rs]
    zins = [rng.standard_normal((T, R, H)).astype(np.float32) for R in Rs]
    singles = [run_scan(hip, z, l[0]["weight_hh"], l[0]["bias_ih"], l[1], l[2], True) for z, l in zip(zins, layers)]
    keep, seg = [], (ScanSegment * 3)()
    outs = []
    for i, (R, z, l) in enumerate(zip(Rs, zins, layers)):
        pk, dq = pack_w3(l[0]["weight_hh"])
        ts = [_t((z + l[0]["bias_ih"][:H]).astype(np.float32)), _t(pk), _t(dq), _t(l[0]["bias_ih"]), _t(l[1]), _t(l[2]), torch.zeros((R, H), device=DEV), torch.zeros((R, H), device=DEV),
              torch.empty((T, R, H), device=DEV), torch.zeros((T, R, (H + 63) // 64 * 64), dtype=torch.int8, device=DEV)]
        keep.append(ts)
        s = seg[i]
        s.zin, s.w_hh, s.w_dq, s.bias, s.bn_alpha, s.bn_beta, s.h_state, s.c_state, s.spikes_f32, s.spikes_i8 = [_p(x) for x in ts]
        s.membrane, s.R = None, R
        outs.append(ts[8])
    check(hip.sfsn_gsn_layer_scan(seg, 3, T, H, 1, 8, None), "scan3")
    torch.cuda.synchronize()
    for o3, s1 in zip(outs, singles):
        np.testing.assert_array_equal(o3.cpu().numpy(), s1[0])
    z, l = zins[0], layers[0]
    a = run_scan(hip, z[:9], l[0]["weight_hh"], l[0]["bias_ih"], l[1], l[2], True)
    b = run_scan(hip, z[9:], l[0]["weight_hh"], l[0]["bias_ih"], l[1], l[2], True, a[3], a[4])
    np.testing.assert_array_equal(np.concatenate([a[0], b[0]]), singles[0][0])
    np.testing.assert_array_equal(b[4], singles[0][4])


@pytest.mark.parametrize("M,K,N", [(50, 32, 24), (333, 224, 224), (100, 160, 64), (64, 320, 320), (47, 224, 40), (200, 224, 192),
                                   (31, 48, 6), (90, 240, 64)])
def test_spike_proj_matches_exact_product(hip, M, K, N):
    """sfsn_spike_proj == sum of the selected fp32 weights, to one rounding of the digit-quantised weights."""
    from spiking_fullsubnet_amd._lib import check
    from spiking_fullsubnet_amd.engine import pack_w3, unpack_w3
    rng = np.random.default_rng(M + K + N)
    w = rng.uniform(-0.1, 0.1, (N, K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    s = (rng.random((M, K)) > 0.6)
    KP = (K + 63) // 64 * 64
    s8 = np.zeros((M, KP), np.int8)
    s8[:, :K] = s
    pk, dq = pack_w3(w)
    y = torch.full((M, N), float("nan"), device=DEV)
    ts = [_t(s8), _t(pk), _t(dq), _t(bias)]
    check(hip.sfsn_spike_proj(_p(ts[0]), _p(ts[1]), _p(ts[2]), _p(ts[3]), _p(y), M, K, N, N, None), "spike_proj")
    torch.cuda.synchronize()
    wq = unpack_w3(pk, dq, N, K).astype(np.float64)
    exact_q = (s.astype(np.float64) @ wq.T).astype(np.float32) + bias  # what the kernel computes, one rounding each
    np.testing.assert_array_equal(y.cpu().numpy(), exact_q)
    exact = s.astype(np.float64) @ w.astype(np.float64).T + bias
    np.testing.assert_allclose(y.cpu().numpy(), exact, atol=2e-6, rtol=1e-6)


def test_spike_proj_accepts_a_4_byte_aligned_output(hip):
    """A projection whose row pitch is not a multiple of 4 floats (P = 2*ctr*df*S with ctr = 1, df = 3) written at an odd
    chunk offset: the output pointer is only 4-byte aligned.  The fast (vector-store) kernels need 16 bytes; the entry point
    must fall back to scalar stores instead of refusing (the chunked / streaming schedules hit this)."""
    from spiking_fullsubnet_amd._lib import check
    from spiking_fullsubnet_amd.engine import pack_w3
    rng = np.random.default_rng(4)
    M, K, N = 150, 96, 6
    KP = (K + 63) // 64 * 64
    s8 = np.zeros((M, KP), np.int8)
    s8[:, :K] = rng.random((M, K)) > 0.6
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    pk, dq = pack_w3(w)
    ts, tp, td, tb = _t(s8), _t(pk), _t(dq), _t(b)
    ya = torch.zeros((M * N + 8,), device=DEV)
    yb = torch.zeros((M * N + 8,), device=DEV)
    check(hip.sfsn_spike_proj(_p(ts), _p(tp), _p(td), _p(tb), _p(ya), M, K, N, N, None), "aligned")
    check(hip.sfsn_spike_proj(_p(ts), _p(tp), _p(td), _p(tb), ctypes.c_void_p(yb.data_ptr() + 4), M, K, N, N, None), "offset by one float")
    torch.cuda.synchronize()
    assert torch.equal(ya[:M * N], yb[1:M * N + 1]) and float(yb[0]) == 0.0 and float(yb[M * N + 1]) == 0.0
    assert hip.sfsn_spike_proj(_p(ts), _p(tp), _p(td), _p(tb), ctypes.c_void_p(yb.data_ptr() + 2), M, K, N, N, None) < 0  # not even 4-byte aligned


@pytest.mark.parametrize("M,K,N", [(100, 38, 224), (77, 94, 160), (64, 158, 224), (130, 64, 320), (33, 12, 32), (40, 64, 240)])
def test_input_proj_f32(hip, M, K, N):
    from spiking_fullsubnet_amd._lib import check
    rng = np.random.default_rng(M * K)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = rng.uniform(-0.1, 0.1, (N, K)).astype(np.float32)
    z = torch.full((M, N), float("nan"), device=DEV)
    ts = [_t(x), _t(w)]
    bias = rng.standard_normal(N).astype(np.float32)
    ts.append(_t(bias))
    check(hip.sfsn_input_proj_f32(_p(ts[0]), _p(ts[1]), _p(ts[2]), _p(z), M, K, N, N, None), "input_proj")
    torch.cuda.synchronize()
    exact = x.astype(np.float64) @ w.astype(np.float64).T + bias
    np.testing.assert_allclose(z.cpu().numpy(), exact, atol=3e-6, rtol=1e-5)  # k-ordered fp32 fma chain, K <= 160


def test_products_of_several_groups_in_one_launch_equal_the_single_launches(hip):
    """sfsn_spike_proj_multi / sfsn_input_proj_f32_multi (round 5): the independent products of a stage -- the sub-band groups'
    projections, their layer-0 input products -- in ONE launch, a block range per job and every job on its own tiling: bit for bit
    what one call per job writes (baseline_m's shapes at a chunk's row counts, a ragged last tile, a single-tile job); jobs the
    single entries would not run on their fast kernels are refused (the engine then issues them one by one)."""
    from spiking_fullsubnet_amd._lib import InProjJob, ProjJob, SFSN_EUNSUPPORTED, check
    from spiking_fullsubnet_amd.engine import pack_w3
    rng = np.random.default_rng(77)
    K = 224
    KP = (K + 63) // 64 * 64
    keep = []
    shapes = [(512 * 5 + 13, 40), (192 * 7, 192), (128 * 9 + 1, 128), (64, 64)]
    jobs = (ProjJob * len(shapes))()
    ya, yb = [], []
    for a, (M, N) in zip(jobs, shapes):
        s8 = np.zeros((M, KP), np.int8)
        s8[:, :K] = rng.random((M, K)) > 0.7
        w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        pk, dq = pack_w3(w)
        ts = [_t(s8), _t(pk), _t(dq), _t(rng.standard_normal(N).astype(np.float32))]
        y1, y2 = torch.full((M, N), float("nan"), device=DEV), torch.full((M, N), float("nan"), device=DEV)
        check(hip.sfsn_spike_proj(_p(ts[0]), _p(ts[1]), _p(ts[2]), _p(ts[3]), _p(y1), M, K, N, N, None), "single")
        a.s, a.w_packed, a.w_dq, a.bias, a.y, a.M, a.K, a.N, a.ldy = [t.data_ptr() for t in ts] + [y2.data_ptr(), M, K, N, N]
        keep.append(ts)
        ya.append(y1)
        yb.append(y2)
    check(hip.sfsn_spike_proj_multi(jobs, len(shapes), None), "sfsn_spike_proj_multi")
    torch.cuda.synchronize()
    for y1, y2 in zip(ya, yb):
        assert torch.equal(y1, y2)
    jobs[1].M = 40  # fewer than 64 rows: not the fast kernel's shape
    assert hip.sfsn_spike_proj_multi(jobs, len(shapes), None) == SFSN_EUNSUPPORTED
    jobs[1].M, jobs[1].K = shapes[1][0], 100  # another k-step count than the other jobs
    assert hip.sfsn_spike_proj_multi(jobs, len(shapes), None) == SFSN_EUNSUPPORTED
    # ---- the real-valued input products (groups 1, 2 of baseline_m: I = 94 / 158 -> H = 224; and a narrow one)
    ishapes = [(192 * 6 + 5, 94, 224), (128 * 8, 158, 224), (300, 38, 160), (64 * 3, 64, 64)]
    ij = (InProjJob * len(ishapes))()
    za, zb = [], []
    for a, (M, Ki, N) in zip(ij, ishapes):
        ts = [_t(rng.standard_normal((M, Ki)).astype(np.float32)), _t(rng.uniform(-0.1, 0.1, (N, Ki)).astype(np.float32)),
              _t(rng.standard_normal(N).astype(np.float32))]
        z1, z2 = torch.full((M, N), float("nan"), device=DEV), torch.full((M, N), float("nan"), device=DEV)
        check(hip.sfsn_input_proj_f32(_p(ts[0]), _p(ts[1]), _p(ts[2]), _p(z1), M, Ki, N, N, None), "single")
        a.x, a.w, a.bias, a.z, a.M, a.K, a.N, a.ldz = [t.data_ptr() for t in ts] + [z2.data_ptr(), M, Ki, N, N]
        keep.append(ts)
        za.append(z1)
        zb.append(z2)
    check(hip.sfsn_input_proj_f32_multi(ij, len(ishapes), None), "sfsn_input_proj_f32_multi")
    torch.cuda.synchronize()
    for z1, z2 in zip(za, zb):
        assert torch.equal(z1, z2)
    ij[0].K = 93  # odd K: the fp32-MFMA fallback's shape
    assert hip.sfsn_input_proj_f32_multi(ij, len(ishapes), None) == SFSN_EUNSUPPORTED


MODEL_CASES = [
    ("live_tiny.npz", "live", rw.LIVE_TINY, 11), ("live_tiny_2spk.npz", "live", rw.LIVE_TINY_2SPK, 12),
    ("live_tiny_unshared.npz", "live", rw.LIVE_TINY_UNSHARED, 13), ("live_m.npz", "live", rw.LIVE_M, 21),
    ("frozen_tiny.npz", "frozen", rw.FROZEN_TINY, 31), ("frozen_s_zoo.npz", "frozen", rw.FROZEN_S, None),
    ("frozen_m_zoo.npz", "frozen", rw.FROZEN_M, None), ("frozen_l.npz", "frozen", rw.FROZEN_L, 33),
    ("frozen_tiny_gauss.npz", "frozen", rw.FROZEN_TINY_GAUSS, 37), ("frozen_tiny_cum.npz", "frozen", rw.FROZEN_TINY_CUM, 35), ("frozen_m_cum.npz", "frozen", rw.FROZEN_M_CUM, 36),
    ("frozen_xl.npz", "frozen", rw.FROZEN_XL, 34),
    # round 3: BASELINE configs[0] as written (trained baseline_s, ONE 4 s clip = 501 frames; weights: frozen_s_zoo.npz) and the
    # bench's sizes on two clips x 200 frames of amplitude-modulated noise (SURVEY 8d's second input distribution)
    ("frozen_s_zoo_4s.npz", "frozen", rw.FROZEN_S, "frozen_s_zoo.npz"), ("live_m_am.npz", "live", rw.LIVE_M, 21),
]


def build_module(front, kw, sd):
    import spiking_fullsubnet_amd as pkg
    m = (pkg.SpikingFullSubNet if front == "live" else pkg.Separator)(**kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.eval().to(DEV)


def hip_result(model, stft, want_membrane=True):
    res = model.forward_stft(_t(stft), want_membrane=want_membrane)
    torch.cuda.synchronize()
    out = dict(enh_stft=res["enh_stft"].cpu().numpy(), enh_mag=res["enh_mag"].cpu().numpy(),
               fb_all=[a.cpu().numpy() for a in res["fb_all"]], sb_all=[[a.cpu().numpy() for a in l] for l in res["sb_all"]])
    if want_membrane:
        out["mem"] = {("fb", l): m.cpu().numpy() for l, m in enumerate(res["fb_mem"])}
        for g, mems in enumerate(res["sb_mem"]):
            out["mem"].update({(f"sb{g}", l): m.cpu().numpy() for l, m in enumerate(mems)})
    return out


def case_setup(fname, front, kw, seed):
    gold = load(fname)
    if front == "live":
        spec, sd = omodel.spec_from_live_kwargs(kw), rw.live_state_dict(kw, seed)
    else:
        spec = omodel.spec_from_frozen_kwargs(kw)
        if isinstance(seed, str):  # the weights live in another fixture (the trained zoo checkpoints are stored once)
            sd = {k[3:]: v for k, v in load(seed).items() if k.startswith("sd/")}
        else:
            sd = {k[3:]: v for k, v in gold.items() if k.startswith("sd/")} if seed is None else rw.frozen_state_dict(kw, seed)
    return gold, spec, sd


@pytest.mark.parametrize("fname,front,kw,seed", MODEL_CASES, ids=[c[0][:-4] for c in MODEL_CASES])
def test_module_vs_reference_golden(fname, front, kw, seed):
    """Drop-in module on the golden STFT vs the reference's recorded outputs: every layer input, spike tensor,
    projection, the enhanced spectrum (<= 1e-4 rel, parity.REL) and, from the waveform, enh_y / enh_mag."""
    gold, spec, sd = case_setup(fname, front, kw, seed)
    model = build_module(front, kw, sd)
    out = hip_result(model, gold["stft"])
    stats = parity.check_model(out, gold, spec, tag=fname + ":")
    parity.report("golden:" + fname, stats)
    # the short fixtures (T <= 126 frames): NOT A SINGLE spike of any layer may differ from the reference's recording -- so none of
    # the output checks below can ever be skipped because "a row diverged".  The two long ones (round 3: T = 501 / 200) run under
    # the causal rule alone, which check_model has asserted: a chain may leave the recording only at a membrane inside the
    # don't-care band, and every output is compared up to that frame (on the trained baseline_s weights one of 14 sub-band chains
    # does, at frame 250 of 501)
    long_fixture = gold["stft"].shape[-1] > 126
    for st in stats:
        if long_fixture:
            assert st.get("own_unexplained", 0) == 0 and st["spike_agreement"] > 0.99 and st["diverged"] <= 2, st
        else:
            assert st["diverged"] == 0 and st["spike_agreement"] == 1.0, st
    if long_fixture and any(st["diverged"] for st in stats):
        return
    outs = model(_t(gold["wave"]))
    torch.cuda.synchronize()
    assert outs[0].shape == gold["enh_y"].shape
    B, S, F, T = out["enh_mag"].shape
    if "enh_mag" in gold:
        np.testing.assert_allclose(out["enh_mag"].reshape(B * S, F, T), gold["enh_mag"], rtol=parity.REL, atol=parity.ATOL)
        np.testing.assert_allclose(outs[1].cpu().numpy(), gold["enh_mag"], rtol=2e-4, atol=1e-4)  # includes the device STFT
    np.testing.assert_allclose(outs[0].cpu().numpy(), gold["enh_y"], rtol=1e-3, atol=2e-5)          # device STFT + iSTFT
    if "synops" in gold:
        assert omodel.compute_synops(out["fb_all"], out["sb_all"], spec["shared"]) == pytest.approx(float(gold["synops"]), rel=1e-6)


@pytest.mark.parametrize("front,kw,seed,B,T", [("live", rw.LIVE_M, 5, 3, 60), ("frozen", rw.FROZEN_S, 6, 2, 50),
                                                ("live", rw.LIVE_TINY_2SPK, 8, 5, 33), ("frozen", rw.FROZEN_L, 9, 2, 24)])
def test_module_vs_oracle_seeded(front, kw, seed, B, T):
    """Same seeded synthetic input through the HIP path and the CPU oracle (sizes the oracle finishes in seconds)."""
    sd = rw.live_state_dict(kw, seed) if front == "live" else rw.frozen_state_dict(kw, seed)
    spec = omodel.spec_from_live_kwargs(kw) if front == "live" else omodel.spec_from_frozen_kwargs(kw)
    wave = torch.from_numpy(rw.synth_wave(B, T, seed, modulated=True))
    stft = torch.stft(wave, 512, 128, 512, window=torch.hann_window(512), return_complex=True, pad_mode="constant").numpy()
    ora = omodel.forward_from_stft(spec, sd, stft, "f32", want_membrane=True)
    out = hip_result(build_module(front, kw, sd), stft)
    stats = parity.check_model(out, parity.gold_from_oracle(ora), spec, tag="oracle:")
    parity.report(f"oracle-seeded:{front}:sbH{spec['sb_hidden']}:B{B}xT{T}", stats)
    for st in stats:  # (a first flip outside the don't-care band already failed inside check_model)
        assert st["spike_agreement"] > 0.999, st
    if front == "frozen":  # Laplace means themselves
        assert np.isfinite(out["enh_mag"]).all()


def test_wsj0_separation_config_vs_oracle_and_streaming():
    """The reference's other recipe for this model (recipes/wsj0-mix/spiking_fullsubnet/default.toml: 256-point frames at 8 kHz,
    two speakers, 32-bin full-band input): HIP path against the CPU oracle on a seeded input, and the one-launch streaming
    session against the offline forward, bit for bit."""
    kw, seed, B, T = rw.LIVE_WSJ0, 17, 3, 48
    sd = rw.live_state_dict(kw, seed)
    spec = omodel.spec_from_live_kwargs(kw)
    wave = torch.from_numpy(rw.synth_wave(B, T, seed, hop=kw["hop_length"], modulated=True))
    stft_c = torch.stft(wave, kw["n_fft"], kw["hop_length"], kw["win_length"], window=torch.hann_window(kw["win_length"]), return_complex=True,
                        pad_mode="constant")
    ora = omodel.forward_from_stft(spec, sd, stft_c.numpy(), "f32", want_membrane=True)
    model = build_module("live", kw, sd)
    out = hip_result(model, stft_c.numpy())
    stats = parity.check_model(out, parity.gold_from_oracle(ora), spec, tag="oracle:")
    parity.report(f"oracle-seeded:live-wsj0:B{B}xT{T}", stats)
    for st in stats:
        assert st["spike_agreement"] > 0.999, st
    stft = stft_c.to(DEV)[..., :T].contiguous()
    off = model.engine().forward_stft(stft, want_layers=False)
    sess = model.streaming(batch=B, hop=1)
    assert sess._hop is not None
    outs = [sess.step(stft[..., t:t + 1].contiguous()) for t in range(T)]
    sess.check_errors()
    assert torch.equal(torch.view_as_real(torch.cat([e for e, _ in outs], -1)), torch.view_as_real(off["enh_stft"]))
    assert torch.equal(torch.cat([m for _, m in outs], -1), off["enh_mag"])
    with pytest.raises(NotImplementedError):
        model.streaming(batch=B, waveform=True)  # waveform streaming: 512-point frames only


@pytest.mark.parametrize("fname,kw", [("frozen_m_zoo.npz", rw.FROZEN_M), ("frozen_s_zoo.npz", rw.FROZEN_S)])
def test_thirty_second_clip_sits_on_the_fp32_noise_floor(fname, kw):
    """The recipes validate on 30 s clips = 3751 frames in one pass (dataloader.py:73-99).  Over thousands of steps
    the strict causal rule of the short tests cannot hold for ANY pair of fp32 evaluations: slowly integrating neurons
    accumulate rounding noise (measured with the trained baseline_s weights: |fp32 oracle - fp64 oracle| membranes 4e-7 at
    t=0, 4.5e-4 at t=1000, first spike flip at t=2235; HIP vs fp32 oracle 5e-7, 5.7e-5, first flip at t=2320), and the
    frozen front-end's utterance-level Laplace mean then carries one late full-band flip to every sub-band frame.  The
    statement that can be made, with the TRAINED zoo weights (the seeded random models overflow over such lengths, in the
    reference too): measured against the fp64 oracle, the HIP path is no further away than the fp32 oracle is -- per layer
    in spike agreement and at the output in relative L2."""
    T = 3751
    gold = load(fname)
    sd = {k[3:]: v for k, v in gold.items() if k.startswith("sd/")}
    spec = omodel.spec_from_frozen_kwargs(kw)
    wave = torch.from_numpy(rw.synth_wave(1, T, 17, modulated=True))
    stft = torch.stft(wave, 512, 128, 512, window=torch.hann_window(512), return_complex=True, pad_mode="constant").numpy()
    assert stft.shape[-1] == T
    o32 = omodel.forward_from_stft(spec, sd, stft, "f32")
    o64 = omodel.forward_from_stft(spec, sd, stft, "f64")
    out = hip_result(build_module("frozen", kw, sd), stft, want_membrane=False)
    assert np.isfinite(out["enh_stft"]).all()

    def layers(res):
        return [np.asarray(a) > 0.5 for a in res["fb_all"][1:-1]] + [np.asarray(a) > 0.5 for l in res["sb_all"] for a in l[1:-1]]

    def first_flip(a, b):
        d = (a != b).reshape(a.shape[0], -1).any(1)
        return int(np.argmax(d)) if d.any() else T

    for mine, f32, f64 in zip(layers(out), layers(o32), layers(o64)):
        assert (mine == f64).mean() >= (f32 == f64).mean() - 0.02
    # the full-band chain (nothing upstream but the input): the first disagreement with fp64 comes no earlier than fp32's own / 2
    assert first_flip(layers(out)[0], layers(o64)[0]) >= first_flip(layers(o32)[0], layers(o64)[0]) // 2
    ref64 = np.asarray(o64["enh_stft"])
    floor = np.linalg.norm((np.asarray(o32["enh_stft"]) - ref64).ravel()) / np.linalg.norm(ref64.ravel())
    mine = np.linalg.norm((out["enh_stft"] - ref64).ravel()) / np.linalg.norm(ref64.ravel())
    assert mine <= 1.15 * floor + 1e-4, (mine, floor)


@pytest.mark.parametrize("B,T", [(1, 1), (2, 2), (1, 33), (5, 31)])
def test_tiny_and_ragged_clip_shapes_vs_oracle(B, T):
    """Edge shapes of the offline forward: a single frame, fewer frames than a deep-filter order, one frame past a 32-frame
    tile, an odd batch -- live baseline_m sizes against the oracle (short clips: no divergence expected at all)."""
    kw, seed = rw.LIVE_M, 14
    sd = rw.live_state_dict(kw, seed)
    spec = omodel.spec_from_live_kwargs(kw)
    rng = np.random.default_rng(B * 100 + T)
    stft = (0.3 * (rng.standard_normal((B, 257, T)) + 1j * rng.standard_normal((B, 257, T)))).astype(np.complex64)
    ora = omodel.forward_from_stft(spec, sd, stft, "f32", want_membrane=True)
    out = hip_result(build_module("live", kw, sd), stft)
    stats = parity.check_model(out, parity.gold_from_oracle(ora), spec, tag=f"B{B}T{T}:")
    assert all(st["spike_agreement"] > 0.999 for st in stats), stats
    ref = np.asarray(ora["enh_stft"])
    if all(st["diverged"] == 0 for st in stats):
        np.testing.assert_allclose(out["enh_stft"], ref, rtol=parity.REL, atol=parity.ATOL + parity.REL * np.abs(ref).max())


@pytest.mark.parametrize("fname,kw", [("frozen_m_zoo.npz", rw.FROZEN_M), ("frozen_s_zoo.npz", rw.FROZEN_S)])
def test_config2_full_band_path_b32_t500(fname, kw):
    """BASELINE.json configs[1]: B=32, T=500, the full-band path (R=32, I=64, H=320 / 240, two layers, projection 64), fp32 --
    trained zoo weights, every full-band chain against the oracle under the strict causal rule."""
    B, T = 32, 500
    gold = load(fname)
    sd = {k[3:]: v for k, v in gold.items() if k.startswith("sd/")}
    spec = omodel.spec_from_frozen_kwargs(kw)
    wave = torch.from_numpy(rw.synth_wave(B, T, 23, modulated=True))
    stft = torch.stft(wave, 512, 128, 512, window=torch.hann_window(512), return_complex=True, pad_mode="constant").numpy()
    ora = omodel.forward_from_stft(spec, sd, stft, "f32", want_membrane=True)
    out = hip_result(build_module("frozen", kw, sd), stft)
    g = parity.gold_from_oracle(ora)
    t_valid = np.full(B, T)
    parity.check_continuous(out["fb_all"][0], g["fb/x"], t_valid, "cfg2:fb/x")
    for l in range(2):
        shape = tuple(int(v) for v in g[f"fb/spikes_shape/{l}"])
        ref, near = parity.unpack(g[f"fb/spikes_packed/{l}"], shape), parity.unpack(g[f"fb/near{parity.TAU:g}/{l}"], shape)
        # membranes too for baseline_m; baseline_s has runaway neurons (|c| grows geometrically until it overflows, in the
        # reference as well) whose membranes integrate rounding noise faster than any fixed relative tolerance: spikes only
        mems = (out["mem"][("fb", l)], g[f"fb/membrane/{l}"]) if "_m_" in fname else (None, None)
        t_valid, st = parity.check_chain(out["fb_all"][1 + l], ref, near, t_valid, f"cfg2:fb/L{l}", *mems)
        assert st["spike_agreement"] > 0.99 and st["diverged"] <= 4, st  # a diverged row is one whose first flip the rule allowed
    parity.check_continuous(out["fb_all"][-1], g["fb/proj"], t_valid, "cfg2:fb/proj")
    assert (t_valid == T).mean() > 0.8  # nearly every clip's full-band chains run the 500 frames without a single spike flip


def test_full_size_properties():
    """BASELINE.json config 3 (live M, B=64, T=1000): properties that do not need the oracle at full size.
    (i) run-to-run bit stability; (ii) batch independence: clips 5..12 computed alone == inside the batch, bit for bit;
    (iii) sub-sampled oracle check: 2 clips x first 120 frames against the CPU oracle (the model is causal in T)."""
    kw, seed, B, T = rw.LIVE_M, 21, 64, 1000
    sd = rw.live_state_dict(kw, seed)
    model = build_module("live", kw, sd)
    wave = torch.from_numpy(rw.synth_wave(B, T, 0)).to(DEV)
    stft = model._stft(wave)
    assert stft.shape == (B, 257, T)
    r1 = model.forward_stft(stft)
    r2 = model.forward_stft(stft)
    torch.cuda.synchronize()
    assert torch.equal(torch.view_as_real(r1["enh_stft"]), torch.view_as_real(r2["enh_stft"]))
    for a, b in zip(r1["sb_all"][0], r2["sb_all"][0]):
        assert torch.equal(a, b)
    sub = model.forward_stft(stft[5:13].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(torch.view_as_real(sub["enh_stft"]), torch.view_as_real(r1["enh_stft"][5:13]))
    assert torch.equal(sub["fb_all"][2], r1["fb_all"][2][:, 5:13])
    # oracle on 8 clips spread over the batch (first, middle, last) x ALL 1000 frames: clips are independent, so the long run
    # restricted to those clips must match the oracle run of those clips -- for THREE launch geometries: (i) the default schedule
    # of a forward alone (full-band stack launch + both sub-band layers side by side in ONE launch at 8 rows per workgroup:
    # FUSEDX3 / scan3 roles for layer 1, FUSED3 roles for layer 2 -- round 4's pair launch); (ii) `pair_scan = False`: the
    # per-layer sub-band launches of round 3 (the IO-wave scan at 4 rows per workgroup + sfsn_spike_proj), which every model with
    # more sub-band workgroups than n_cu - 40, H > 224 or unshared gates still takes; (iii) the geometry the bench's timed region
    # uses (full-band stack at 8, sub-band scans at 16 rows per workgroup with both fused-input variants).  The causal rule
    # handles late divergence: a chain may leave the reference only where the reference membrane is inside the don't-care band
    # (own_unexplained == 0 is asserted per layer; the per-layer first-flip frames go to the parity report).
    Tc, clips = T, [0, 1, 30, 31, 32, 33, 62, 63]
    spec = omodel.spec_from_live_kwargs(kw)
    ora = omodel.forward_from_stft(spec, sd, stft[clips, :, :Tc].cpu().numpy(), "f32", want_membrane=True)
    gold_sub = parity.gold_from_oracle(ora)
    eng = model.engine()
    pair0 = eng.pair_scan
    for label, rpw, pair in (("default", (0, 0), pair0), ("per-layer sub-band scans at 4 rows", (0, 0), False), ("timed-region geometry", (8, 16), pair0)):
        eng.rows_per_wg, eng.pair_scan = rpw, pair
        eng.stack_rows_fb_auto = 8 if rpw[0] == 8 else 4  # (bench.py's set_geometry: the full-band stack at 8 rows per workgroup)
        n0 = dict(eng.launches)
        rr = r1 if (rpw == (0, 0) and pair == pair0) else model.forward_stft(stft)
        torch.cuda.synchronize()
        if rpw != (0, 0):
            assert eng.launches.get("fused", 0) > n0.get("fused", 0) and eng.launches.get("fused_x", 0) > n0.get("fused_x", 0)
        elif not pair:
            # only the full-band stack went through a stack launch (one per chunk); the sub-band layers were per-layer launches
            assert eng.launches.get("stack", 0) - n0.get("stack", 0) == rr["n_chunks"], (eng.launches, n0, rr["n_chunks"])
            assert torch.equal(torch.view_as_real(rr["enh_stft"]), torch.view_as_real(r1["enh_stft"]))  # (and the pair launch agrees bit for bit)
        ci = torch.tensor(clips, device=DEV)
        out = dict(enh_stft=rr["enh_stft"][ci][:, :, :, :Tc].cpu().numpy(), fb_all=[a[:Tc, ci].cpu().numpy() for a in rr["fb_all"]], sb_all=[])
        for g, lst in enumerate(rr["sb_all"]):
            N = spec_units(spec, g)
            rows = torch.cat([torch.arange(c * N, (c + 1) * N, device=DEV) for c in clips])
            out["sb_all"].append([a[:Tc, rows].cpu().numpy() for a in lst])
        stats = parity.check_model(out, gold_sub, spec, tag=f"full-size ({label}):")
        parity.report(f"full-size:B64xT1000:{label}:clips{clips}:T{Tc}", stats, extra=dict(scan_kernels={k: v for k, v in eng.launches.items()}))
        for st in stats:
            assert st.get("own_unexplained", 0) == 0, st  # every first flip sits inside the don't-care band of the reference membrane
            assert st["valid_frac"] > 0.5, st             # ... and most chain-frames are compared strictly (a full-band flip voids the clip's sub-band rows from there on)
        # 8 clips x 14 rows x 4 layers x 1000 frames: a handful of chains leave the reference at a near-threshold membrane
        assert sum(st["diverged"] for st in stats) <= 0.1 * sum(st["rows"] for st in stats), stats
    eng.rows_per_wg, eng.stack_rows_fb_auto, eng.pair_scan = (0, 0), 4, pair0
    rates = [float(a.mean()) for a in r1["fb_all"][1:3]]
    assert all(0.02 < r < 0.98 for r in rates), rates  # the synthetic model is alive, not saturated


@pytest.mark.parametrize("front,kw,seed", [("live", rw.LIVE_M, 5), ("frozen", rw.FROZEN_S, 6), ("live", rw.LIVE_TINY_UNSHARED, 7)])
def test_pipelined_schedule_is_bit_identical(front, kw, seed):
    """The time-pipelined multi-stream schedule (chunks carried through h_state / c_state) == the sequential schedule, bit for bit."""
    sd = rw.live_state_dict(kw, seed) if front == "live" else rw.frozen_state_dict(kw, seed)
    model = build_module(front, kw, sd)
    wave = torch.from_numpy(rw.synth_wave(3, 300, seed)).to(DEV)
    stft = torch.stft(wave, 512, 128, 512, window=torch.hann_window(512, device=DEV), return_complex=True, pad_mode="constant")
    eng = model.engine()
    eng.pipeline_chunk = 96  # 300 frames -> chunks of 96, 96, 96, 12
    a = eng.forward_stft(stft, pipeline=False)
    b = eng.forward_stft(stft, pipeline=True)
    torch.cuda.synchronize()
    assert b["pipelined"] and b["n_chunks"] == 4 and not a["pipelined"]
    assert torch.equal(torch.view_as_real(a["enh_stft"]), torch.view_as_real(b["enh_stft"]))
    assert torch.equal(a["enh_mag"], b["enh_mag"])
    for x, y in zip(a["fb_all"] + sum(a["sb_all"], []), b["fb_all"] + sum(b["sb_all"], [])):
        assert torch.equal(x, y)


def test_forwards_in_flight_on_separate_streams_are_independent():
    """Batch-level pipelining (bench.py --inflight): forwards issued back to back on different HIP streams, with different
    launch geometries (rows per scan workgroup), give the results of the same forwards run one at a time -- per-stream scratch
    buffers do not alias and the 16-, 8- and 4-row scan variants agree bit for bit."""
    kw, seed = rw.LIVE_M, 9
    model = build_module("live", kw, rw.live_state_dict(kw, seed))
    eng = model.engine()
    stfts = []
    for i in range(3):
        wave = torch.from_numpy(rw.synth_wave(4, 150, 100 + i)).to(DEV)
        stfts.append(torch.stft(wave, 512, 128, 512, window=torch.hann_window(512, device=DEV), return_complex=True, pad_mode="constant"))
    eng.rows_per_wg = (0, 0)
    ref = [eng.forward_stft(s) for s in stfts]
    torch.cuda.synchronize()
    lanes = [torch.cuda.Stream(device=DEV) for _ in range(3)]
    outs = []
    for rpw, (s_, x) in zip([(4, 16), (16, 8), (8, 4)], zip(lanes, stfts)):
        s_.wait_stream(torch.cuda.current_stream())
        eng.rows_per_wg = rpw
        with torch.cuda.stream(s_):
            outs.append(eng.forward_stft(x))
    torch.cuda.synchronize()
    eng.rows_per_wg = (0, 0)
    for a, b in zip(ref, outs):
        assert torch.equal(torch.view_as_real(a["enh_stft"]), torch.view_as_real(b["enh_stft"]))
        for x, y in zip(a["fb_all"] + sum(a["sb_all"], []), b["fb_all"] + sum(b["sb_all"], [])):
            assert torch.equal(x, y)


@pytest.mark.parametrize("one_launch", ["auto", False])
@pytest.mark.parametrize("kw,seed,B,hop,graph", [(rw.LIVE_TINY, 11, 2, 1, True), (rw.LIVE_M, 5, 1, 1, True), (rw.LIVE_M, 5, 3, 4, True),
                                                  (rw.LIVE_TINY_2SPK, 12, 2, 3, False), (rw.LIVE_TINY_UNSHARED, 7, 1, 1, True),
                                                  (rw.LIVE_TINY_UNSHARED, 7, 3, 3, True), (dict(rw.LIVE_M, shared_weights=False), 8, 2, 1, True),
                                                  (rw.LIVE_M, 5, 37, 1, True),
                                                  (dict(rw.LIVE_TINY, use_pre_layer_norm_fb=False, use_pre_layer_norm_sb=False, bn=False), 13, 2, 1, True),
                                                  (dict(rw.LIVE_TINY, df_orders=[1, 1, 1]), 14, 3, 2, True)])
def test_streaming_session_equals_offline_forward(kw, seed, B, hop, graph, one_launch):
    """BASELINE configs[4] (streaming, state carried, hop frames per call): the frame-by-frame session reproduces the offline
    forward on the same clip bit for bit (the model is causal after the STFT), and a reset starts a new utterance -- both as ONE
    launch per hop (sfsn_stream_hop, the default wherever the library covers the model) and as the offline kernels replayed
    from a HIP graph.  (Offline forward vs oracle / golden vectors: the tests above.)"""
    model = build_module("live", kw, rw.live_state_dict(kw, seed))
    T = 24 * hop if hop > 1 else 40
    wave = torch.from_numpy(rw.synth_wave(B, T, seed)).to(DEV)
    stft = torch.stft(wave, kw["n_fft"], kw["hop_length"], kw["win_length"], window=torch.hann_window(kw["win_length"], device=DEV),
                      return_complex=True, pad_mode="constant")[..., :T].contiguous()
    assert stft.shape[-1] == T
    off = model.engine().forward_stft(stft, want_layers=False)
    sess = model.streaming(batch=B, hop=hop, graph=graph, rows_per_wg=None if seed != 12 else (0, 0),  # default / unfused geometry
                           one_launch=one_launch)
    if one_launch == "auto":
        assert sess._hop is not None  # every recipe's sizes take the one-launch path (separate gate weights since round 4)
    else:
        assert sess._hop is None
    for rep in range(2):
        outs, mags = [], []
        for t0 in range(0, T, hop):
            x = stft[..., t0:t0 + hop]
            e, m = sess.step(x.contiguous() if (t0 // hop) % 2 == 0 else x.clone(memory_format=torch.contiguous_format))
            outs.append(e)
            mags.append(m)
        assert sess.frames_done == T
        sess.check_errors()
        e, m = torch.cat(outs, -1), torch.cat(mags, -1)
        assert torch.equal(torch.view_as_real(e), torch.view_as_real(off["enh_stft"])), rep
        assert torch.equal(m, off["enh_mag"])
        sess.reset()


@pytest.mark.parametrize("kw,seed,B", [(rw.LIVE_TINY, 11, 2), (rw.LIVE_M, 5, 1), (rw.LIVE_TINY_2SPK, 12, 3), (rw.LIVE_M, 5, 17), (rw.LIVE_M, 5, 35),
                                       (rw.LIVE_TINY_UNSHARED, 7, 2),
                                       (dict(rw.LIVE_TINY, df_orders=[1, 1, 1], use_pre_layer_norm_sb=False), 15, 1),
                                       (dict(rw.LIVE_M, df_orders=[1, 1, 1]), 16, 2)])  # recipes/intel_ndns/.../baseline_m_no_df.toml
def test_waveform_streaming_equals_offline_forward(kw, seed, B):
    """Samples in, samples out, 128 at a time (8 ms): the session with waveform=True -- STFT of the new frame, the whole model
    and the inverse STFT with its overlap-add state in ONE launch per hop -- reproduces the offline forward's waveform bit for
    bit, three hops late (the look-ahead of torch.stft(center=True) + the overlap-add); a reset starts a new utterance."""
    model = build_module("live", kw, rw.live_state_dict(kw, seed))
    n_hops = 44
    wave = torch.from_numpy(rw.synth_wave(B, n_hops + 1, seed)).to(DEV)  # [B, 128 * n_hops]
    y = model(wave)[0]
    y = y.reshape(B, -1, y.shape[-1])  # [B, S, L]
    sess = model.streaming(batch=B, waveform=True)
    with pytest.raises(RuntimeError):
        sess.step(torch.zeros((B, 257, 1), dtype=torch.complex64, device=DEV))
    for rep in range(2):
        outs = []
        for c in range(n_hops):
            o = sess.step_wave(wave[:, 128 * c:128 * (c + 1)].contiguous())
            if c < 3:
                assert not bool(o.any())
            else:
                outs.append(o)
        sess.check_errors()
        got = torch.cat(outs, -1)  # the samples that entered with calls 0 .. n_hops - 4
        assert got.shape[-1] == 128 * (n_hops - 3)
        ref = y[..., :got.shape[-1]]
        assert torch.equal(got, ref), (rep, float((got - ref).abs().max()))
        sess.reset()


@pytest.mark.parametrize("kw,seed,B", [(rw.LIVE_M, 5, 1), (rw.LIVE_TINY_2SPK, 12, 3), (rw.LIVE_M, 5, 35)])
def test_waveform_streaming_from_and_to_host_memory(kw, seed, B):
    """host_io=True: the launch reads the new samples from pinned host memory and writes the enhanced samples and a completion
    word per (clip, speaker) back into pinned host memory; the caller spins on the words (no copy launch, no stream
    synchronisation).  Same samples as the device-side session and the offline forward, bit for bit."""
    model = build_module("live", kw, rw.live_state_dict(kw, seed))
    n_hops = 30
    wave = torch.from_numpy(rw.synth_wave(B, n_hops + 1, seed))  # CPU [B, 128 * n_hops]
    y = model(wave.to(DEV))[0]
    y = y.reshape(B, -1, y.shape[-1]).cpu()
    sess = model.streaming(batch=B, waveform=True, host_io=True)
    outs = []
    for c in range(n_hops):
        o = sess.step_wave_host(wave[:, 128 * c:128 * (c + 1)])
        assert o.device.type == "cpu"
        if c >= 3:
            outs.append(o.clone())
    sess.check_errors()
    got = torch.cat(outs, -1)
    assert torch.equal(got, y[..., :got.shape[-1]])
    with pytest.raises(ValueError):
        model.streaming(batch=B, host_io=True)


def test_streaming_hops_beside_a_saturated_chip_are_right_or_loud():
    """A one-launch hop assumes its ~20 workgroups get compute units promptly.  Here they do not: B = 64 x T = 1000
    forwards are in flight on six other streams (every CU held by scan workgroups that stay for the whole launch) while a
    B = 1 session streams 200 hops.  The contract: every hop is either bit-identical to the offline forward or the session
    raises (a bounded hand-off wait expired -> error word -> check_errors()); it never returns wrong samples silently, the
    sticky word is cleared by the report, and after reset() on a quiet chip the session is exact again."""
    kw, seed = rw.LIVE_M, 5
    model = build_module("live", kw, rw.live_state_dict(kw, seed))
    T = 200
    wave = torch.from_numpy(rw.synth_wave(1, T + 1, seed)).to(DEV)
    stft = torch.stft(wave, 512, 128, 512, window=torch.hann_window(512, device=DEV), return_complex=True, pad_mode="constant")[..., :T].contiguous()
    off = model.engine().forward_stft(stft, want_layers=False)
    load_model = build_module("live", kw, rw.live_state_dict(kw, seed + 1))
    big = torch.from_numpy(rw.synth_wave(64, 1001, 77)).to(DEV)
    big = torch.stft(big, 512, 128, 512, window=torch.hann_window(512, device=DEV), return_complex=True, pad_mode="constant")[..., :1000].contiguous()
    leng = load_model.engine()
    leng.forward_stft(big, want_layers=False)  # (scratch allocated, weights packed)
    sess = model.streaming(batch=1)
    torch.cuda.synchronize()
    lanes = [torch.cuda.Stream(device=DEV) for _ in range(6)]
    for rep in range(3):  # ~3 x 6 forwards of ~3 ms: the hops below overlap them
        for s_ in lanes:
            with torch.cuda.stream(s_):
                leng.forward_stft(big, want_layers=False)
    outs, raised = [], False
    t_begin = time.perf_counter()
    try:
        for t in range(T):
            e, _ = sess.step(stft[..., t:t + 1].contiguous())
            outs.append(e)
        sess.check_errors()
    except RuntimeError as err:
        raised = True
        assert "hand-off wait expired" in str(err)
    busy_ms = (time.perf_counter() - t_begin) * 1e3
    torch.cuda.synchronize()
    if not raised:
        e = torch.cat(outs, -1)
        assert torch.equal(torch.view_as_real(e), torch.view_as_real(off["enh_stft"]))
    parity.report("streaming-beside-saturated-chip", [], extra=dict(hops=T, raised=raised, wall_ms_for_the_hops=round(busy_ms, 2),
                                                                     load="6 streams x 3 forwards of B=64 x T=1000 in flight"))
    # quiet chip, new utterance: exact again, and no stale error word
    sess.reset()
    outs = [sess.step(stft[..., t:t + 1].contiguous())[0] for t in range(T)]
    sess.check_errors()
    assert torch.equal(torch.view_as_real(torch.cat(outs, -1)), torch.view_as_real(off["enh_stft"]))


@pytest.mark.parametrize("kw,seed,B", [(rw.LIVE_M, 5, 1), (rw.LIVE_TINY_2SPK, 12, 3), (rw.LIVE_TINY_UNSHARED, 7, 2)])
def test_waveform_streaming_resident_launch(kw, seed, B):
    """resident=True: ONE launch serves hop after hop, rung through a doorbell word in pinned host memory
    (sfsn_stream_hop_resident).  Same samples as the offline forward, bit for bit -- across a doorbell left silent until the
    kernel's watchdog ended it (the session starts another), and again after reset()."""
    import time
    model = build_module("live", kw, rw.live_state_dict(kw, seed))
    n_hops = 30
    wave = torch.from_numpy(rw.synth_wave(B, n_hops + 1, seed))
    y = model(wave.to(DEV))[0]
    y = y.reshape(B, -1, y.shape[-1]).cpu()
    sess = model.streaming(batch=B, waveform=True, host_io=True, resident=True, idle_ms=200)
    for rep in range(2):
        outs = []
        for c in range(n_hops):
            if rep == 0 and c == 17:
                time.sleep(0.6)  # well past idle_ms: the resident kernel has left by now
                assert sess._hop["host"]["bell_np"][1] == 1
            o = sess.step_wave_host(wave[:, 128 * c:128 * (c + 1)])
            if c >= 3:
                outs.append(o.clone())
        got = torch.cat(outs, -1)
        assert torch.equal(got, y[..., :got.shape[-1]]), rep
        sess.reset()  # ends the resident launch, checks the error word
        assert sess._res is None
    sess.close()
    with pytest.raises(ValueError):
        model.streaming(batch=B, waveform=True, resident=True)


TINY_CUM = dict(rw.FROZEN_TINY_CUM, sb_df_orders=[3, 2, 1])  # (the fixture's orders [2, 1, 3] give the last group 384 projections: one launch covers 256)


XL_CUM = dict(rw.FROZEN_XL, norm_type="cumulative_laplace_norm")  # recipes/.../spiking_fullsubnet_freeze_phase/baseline_xl.toml as written


@pytest.mark.parametrize("kw,seed,B,hop", [(TINY_CUM, 35, 3, 1), (rw.FROZEN_M_CUM, 36, 2, 1), (TINY_CUM, 35, 2, 3), (XL_CUM, 34, 2, 1),
                                           (dict(TINY_CUM, shared_weights=False), 38, 3, 2)])
def test_frozen_front_end_with_cumulative_norm_streams(kw, seed, B, hop):
    """cumulative_laplace_norm makes the frozen (model_zoo-architecture) front-end causal: the one-launch streaming session --
    on spectra and on waveforms -- reproduces the offline forward bit for bit (running sums carried per row)."""
    model = build_module("frozen", kw, rw.frozen_state_dict(kw, seed))
    T = 36 * hop
    wave = torch.from_numpy(rw.synth_wave(B, T, seed)).to(DEV)
    stft = model._stft(wave)[..., :T].contiguous()
    off = model.engine().forward_stft(stft, want_layers=False)
    sess = model.streaming(batch=B, hop=hop)
    assert sess._hop is not None
    for rep in range(2):
        outs = [sess.step(stft[..., t0:t0 + hop].contiguous()) for t0 in range(0, T, hop)]
        sess.check_errors()
        assert torch.equal(torch.view_as_real(torch.cat([e for e, _ in outs], -1)), torch.view_as_real(off["enh_stft"])), rep
        assert torch.equal(torch.cat([m for _, m in outs], -1), off["enh_mag"])
        sess.reset()
    if hop == 1:
        y = model(wave)[0].reshape(B, 1, -1)
        ws = model.streaming(batch=B, waveform=True)
        outs = [ws.step_wave(wave[:, 128 * c:128 * (c + 1)].contiguous()) for c in range((T - 1))]
        ws.check_errors()
        got = torch.cat(outs[3:], -1)
        assert torch.equal(got, y[..., :got.shape[-1]])
    with pytest.raises(NotImplementedError):  # 384 projections in the last group: no one-launch hop, and no per-kernel form of this norm
        build_module("frozen", rw.FROZEN_TINY_CUM, rw.frozen_state_dict(rw.FROZEN_TINY_CUM, 35)).streaming(batch=1)


@pytest.mark.parametrize("kw,seed,B", [(TINY_CUM, 35, 3), (rw.FROZEN_M_CUM, 36, 2)])
def test_resident_launch_carries_the_cumulative_norm_sums(kw, seed, B):
    """The resident form of the hop with cumulative_laplace_norm (round-3 advisor finding): the running sum of a row is written by
    one workgroup of a hop and read by the others in the next hop with NO launch boundary in between -- it travels write-through
    with agent-scope loads, like the last spikes.  Enhanced samples equal the offline forward's, bit for bit, over 60 hops, after
    a watchdog exit and after reset()."""
    import time
    model = build_module("frozen", kw, rw.frozen_state_dict(kw, seed))
    n_hops = 60
    wave = torch.from_numpy(rw.synth_wave(B, n_hops + 1, seed))
    y = model(wave.to(DEV))[0]
    y = y.reshape(B, -1, y.shape[-1]).cpu()
    sess = model.streaming(batch=B, waveform=True, host_io=True, resident=True, idle_ms=200)
    for rep in range(2):
        outs = []
        for c in range(n_hops):
            if rep == 0 and c == 23:
                time.sleep(0.6)  # the resident kernel has left: the next hop starts another one, the sums continue from device memory
            o = sess.step_wave_host(wave[:, 128 * c:128 * (c + 1)])
            if c >= 3:
                outs.append(o.clone())
        got = torch.cat(outs, -1)
        assert torch.equal(got, y[..., :got.shape[-1]]), rep
        sess.reset()
    sess.close()


def test_stream_hop_argument_checks_and_fallback():
    """sfsn_stream_hop through the C ABI: malformed descriptors are refused, what the launch does not cover reports
    SFSN_EUNSUPPORTED (the session then replays the offline kernels), one_launch=True insists."""
    import ctypes
    from spiking_fullsubnet_amd import _lib
    L = _lib.lib()
    model = build_module("live", rw.LIVE_TINY, rw.live_state_dict(rw.LIVE_TINY, 11))
    sess = model.streaming(batch=1, hop=1)
    desc = sess._hop["desc"]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    good = (desc.scratch, desc.scratch_bytes, desc.hop, desc.inp_ri)
    desc.scratch_bytes = 8
    assert L.sfsn_stream_hop(ctypes.byref(desc), st) == _lib.SFSN_EINVAL
    desc.scratch_bytes = good[1]
    desc.inp_ri = None
    assert L.sfsn_stream_hop(ctypes.byref(desc), st) == _lib.SFSN_EINVAL
    desc.inp_ri = good[3]
    desc.hop = 40
    assert L.sfsn_stream_hop(ctypes.byref(desc), st) == _lib.SFSN_EUNSUPPORTED
    assert L.sfsn_hop_scratch_bytes(ctypes.byref(desc)) == 0
    desc.hop = good[2]
    assert L.sfsn_hop_scratch_bytes(ctypes.byref(desc)) == good[1]
    unshared = build_module("live", rw.LIVE_TINY_UNSHARED, rw.live_state_dict(rw.LIVE_TINY_UNSHARED, 7))
    assert unshared.streaming(batch=1, hop=1, one_launch=True)._hop is not None  # (separate gate weights: covered since round 4)
    deep = build_module("live", dict(rw.LIVE_TINY, sb_num_layers=4), rw.live_state_dict(dict(rw.LIVE_TINY, sb_num_layers=4), 7))
    with pytest.raises(NotImplementedError):  # more layers than the launch has stages for: one_launch=True insists, "auto" falls back
        deep.streaming(batch=1, hop=1, one_launch=True)
    assert deep.streaming(batch=1, hop=1)._hop is None
    mid = build_module("live", rw.LIVE_M, rw.live_state_dict(rw.LIVE_M, 5))
    # more workgroups than compute units in one launch: the batch is cut into equal parts, one launch each
    assert len(mid.streaming(batch=16, hop=1)._hop["parts"]) == 1 and len(mid.streaming(batch=64, hop=1)._hop["parts"]) == 2
    big = mid.streaming(batch=64, hop=1, one_launch=False)  # the per-kernel sequence still serves any batch
    big.step(torch.zeros((64, 257, 1), dtype=torch.complex64, device=DEV))
    torch.cuda.synchronize()


def test_streaming_rejects_the_non_causal_front_end_and_bad_frames():
    model = build_module("frozen", rw.FROZEN_TINY, rw.frozen_state_dict(rw.FROZEN_TINY, 31))
    with pytest.raises(NotImplementedError):
        model.streaming()
    live = build_module("live", rw.LIVE_TINY, rw.live_state_dict(rw.LIVE_TINY, 11))
    sess = live.streaming(batch=1, hop=2, graph=False)
    with pytest.raises(RuntimeError):
        sess.step(torch.zeros((1, rw.LIVE_TINY["n_fft"] // 2 + 1, 1), dtype=torch.complex64, device=DEV))
    with pytest.raises(RuntimeError):
        sess.step(torch.zeros((1, rw.LIVE_TINY["n_fft"] // 2 + 1, 2), dtype=torch.complex64))


@pytest.mark.parametrize("front,kw,seed", [("live", rw.LIVE_M, 5), ("frozen", rw.FROZEN_S, 6), ("live", rw.LIVE_TINY_UNSHARED, 7)])
def test_spike_counts_replace_the_fp32_spike_tensors(front, kw, seed):
    """layer_outputs="counts": the device-side counts of the int8 spikes are exactly the sums of the fp32 spike tensors of the
    default mode, the other outputs are bit-identical, and the SynOPs / NeuronOPs drop-ins give the same numbers either way."""
    from spiking_fullsubnet_amd import SpikeSummary, metric
    sd = rw.live_state_dict(kw, seed) if front == "live" else rw.frozen_state_dict(kw, seed)
    model = build_module(front, kw, sd)
    wave = torch.from_numpy(rw.synth_wave(3, 77, seed)).to(DEV)
    full = model(wave)
    model.layer_outputs = "counts"
    lean = model(wave)
    model.layer_outputs = "none"
    bare = model(wave)
    model.layer_outputs = "tensors"
    torch.cuda.synchronize()
    assert torch.equal(full[0], lean[0]) and torch.equal(full[1], lean[1]) and torch.equal(full[0], bare[0])
    fb_f, sb_f, fb_c, sb_c = full[-2], full[-1], lean[-2], lean[-1]
    n = 0
    for a, b, c in zip([fb_f] + list(sb_f), [fb_c] + list(sb_c), [bare[-2]] + list(bare[-1])):
        assert torch.equal(a[0], b[0]) and torch.equal(a[-1], b[-1])
        for x, y, z in zip(a[1:-1], b[1:-1], c[1:-1]):
            assert isinstance(y, SpikeSummary) and z is None and tuple(y.shape) == tuple(x.shape)
            assert int(y.count.item()) == int((x > 0).sum().item())
            n += 1
    assert n == 2 * (1 + len(sb_f))
    shared = kw.get("shared_weights", True)
    assert metric.compute_synops(fb_c, sb_c, shared) == pytest.approx(metric.compute_synops(fb_f, sb_f, shared), rel=1e-6)
    assert metric.compute_neuronops(fb_c, sb_c) == metric.compute_neuronops(fb_f, sb_f)
    oracle_syn = omodel.compute_synops([t.cpu().numpy() for t in fb_f], [[t.cpu().numpy() for t in l] for l in sb_f], shared)
    assert metric.compute_synops(fb_c, sb_c, shared) == pytest.approx(oracle_syn, rel=1e-6)


@pytest.mark.parametrize("front,kw,seed,geoms", [
    ("live", rw.LIVE_M, 5, [((0, 0), True), ((0, 0), False), ((8, 16), True), ((16, 16), True), ((4, 8), True)]),
    ("frozen", rw.FROZEN_S, 6, [((0, 0), True), ((8, 16), True)]),
    ("live", rw.LIVE_TINY_UNSHARED, 7, [((0, 0), True), ((16, 16), True)]),
    ("frozen", XL_CUM, 34, [((0, 0), True)]),
])
def test_spikes_are_counted_inside_the_scans(front, kw, seed, geoms):
    """SURVEY 8f-1 as worded: layer_outputs="counts" launches NOTHING extra -- every scan kernel family counts the spikes it flushes
    (sfsn_scan_segment.spike_count: the IO-wave roles' storer waves, round 2's flush, the streamed-weight kernel) and the counters
    are zeroed with the states by the forward's first feature launch.  For every launch geometry a model can take (pair launch /
    stack launches, per-layer IO-wave scans, the 16-row fused-input kernels of the timed region, chunked sequences with carried
    state, separate gate weights, the streamed-weight kernel of baseline_xl) the in-scan counts equal the sums of the fp32 spike
    tensors AND round 3's counting launch over the int8 copies, and the other outputs stay bit-identical."""
    from spiking_fullsubnet_amd import SpikeSummary
    sd = rw.live_state_dict(kw, seed) if front == "live" else rw.frozen_state_dict(kw, seed)
    model = build_module(front, kw, sd)
    B, T = 3, 300  # (>= 96 frames per chunk: the default schedule cuts the sequence in three and carries the state)
    wave = torch.from_numpy(rw.synth_wave(B, T, seed)).to(DEV)
    stft = model._stft(wave)
    eng = model.engine()
    keep = (eng.rows_per_wg, eng.pair_scan, eng.stack_rows_fb_auto, eng.count_in_scan)
    try:
        for rpw, pair in geoms:
            eng.rows_per_wg, eng.pair_scan = rpw, pair
            eng.stack_rows_fb_auto = rpw[0] if rpw[0] in (4, 8, 16) else 4
            full = eng.forward_stft(stft, want_layers=True)
            eng.count_in_scan = True
            n0 = dict(eng.launches)
            lean = eng.forward_stft(stft, want_layers=False, want_counts=True)
            eng.count_in_scan = False
            old = eng.forward_stft(stft, want_layers=False, want_counts=True)
            torch.cuda.synchronize()
            eng.check_stack_errors()
            assert torch.equal(torch.view_as_real(full["enh_stft"]), torch.view_as_real(lean["enh_stft"])), (rpw, pair)
            n = 0
            for a, b, c in zip([full["fb_all"]] + full["sb_all"], [lean["fb_all"]] + lean["sb_all"], [old["fb_all"]] + old["sb_all"]):
                assert torch.equal(a[0], b[0]) and torch.equal(a[-1], b[-1])
                for x, y, z in zip(a[1:-1], b[1:-1], c[1:-1]):
                    assert isinstance(y, SpikeSummary) and tuple(y.shape) == tuple(x.shape)
                    want = int((x > 0).sum().item())
                    assert int(y.count.item()) == want == int(z.count.item()), (rpw, pair, n, int(y.count.item()), want, int(z.count.item()))
                    assert want > 0
                    n += 1
            assert n == eng.spec.fb_layers + eng.spec.n_groups * eng.spec.sb_layers
    finally:
        eng.rows_per_wg, eng.pair_scan, eng.stack_rows_fb_auto, eng.count_in_scan = keep


def test_spike_count_rejects_bad_arguments(hip):
    from spiking_fullsubnet_amd._lib import CountTensor, SFSN_EINVAL
    s = torch.ones((4, 64), dtype=torch.int8, device=DEV)
    cnt = torch.zeros((1,), dtype=torch.int64, device=DEV)
    arr = (CountTensor * 1)()
    arr[0].spikes_i8, arr[0].n_bytes, arr[0].count = s.data_ptr(), s.numel(), cnt.data_ptr()
    assert hip.sfsn_spike_count(arr, 1, None) == 0
    torch.cuda.synchronize()
    assert int(cnt.item()) == 256
    arr[0].n_bytes = 250
    assert hip.sfsn_spike_count(arr, 1, None) == SFSN_EINVAL
    arr[0].n_bytes, arr[0].count = 256, None
    assert hip.sfsn_spike_count(arr, 1, None) == SFSN_EINVAL
    assert hip.sfsn_spike_count(arr, 0, None) == SFSN_EINVAL and hip.sfsn_spike_count(arr, 17, None) == SFSN_EINVAL


@pytest.mark.parametrize("B,L", [(2, 128 * 39), (1, 1000), (3, 128 * 16), (2, 128 * 17 + 5), (1, 300), (64, 128 * 99)])
def test_stft_istft_kernels_vs_oracle_and_torch(B, L):
    """sfsn_stft / sfsn_istft vs the oracle's float64 restatement of audio_feature.py:236-347 and vs torch.stft / torch.istft on
    the same device; ragged lengths (L not a multiple of the hop, T not a multiple of the 16-frame tile, clips shorter than a
    window), and the analysis-synthesis round trip."""
    from spiking_fullsubnet_amd import spectral
    rng = np.random.default_rng(B * 1000 + L)
    wave = (0.05 * rng.standard_normal((B, L))).astype(np.float32)
    y = _t(wave)
    X = spectral.stft(y, 512, 128)
    torch.cuda.synchronize()
    ref = omodel.stft(wave)
    assert tuple(X.shape) == ref.shape == (B, 257, 1 + L // 128)
    scale = np.abs(ref).max()
    np.testing.assert_allclose(X.cpu().numpy(), ref, atol=3e-6 * scale, rtol=0)
    win = torch.hann_window(512, device=DEV)
    Xt = torch.stft(y, 512, 128, 512, window=win, return_complex=True, pad_mode="constant")
    np.testing.assert_allclose(X.cpu().numpy(), Xt.cpu().numpy(), atol=3e-6 * scale, rtol=0)
    # inverse on an arbitrary (not STFT-consistent) spectrum, incl. non-zero imaginary DC / Nyquist parts
    T = X.shape[-1]
    Z = (rng.standard_normal((B, 257, T)) + 1j * rng.standard_normal((B, 257, T))).astype(np.complex64)
    for length in ((T - 1) * 128, max(1, (T - 1) * 128 - 37)):
        if length < 1:
            continue
        yi = spectral.istft(_t(Z), 512, 128, length=length)
        torch.cuda.synchronize()
        refi = omodel.istft(Z, length=length)
        assert tuple(yi.shape) == refi.shape == (B, length)
        np.testing.assert_allclose(yi.cpu().numpy(), refi, atol=3e-6 * np.abs(refi).max(), rtol=0)
        if length > 256:  # torch.istft refuses clips whose kept range touches a zero of the window envelope
            yt = torch.istft(_t(Z), 512, 128, 512, window=win, length=length)
            np.testing.assert_allclose(yi.cpu().numpy(), yt.cpu().numpy(), atol=3e-6 * np.abs(refi).max(), rtol=0)
    if L % 128 == 0 and L >= 512:
        back = spectral.istft(X, 512, 128, length=L).cpu().numpy()
        np.testing.assert_allclose(back, wave, atol=2e-6, rtol=0)


def test_stft_kernels_on_the_golden_edges_and_errors():
    from spiking_fullsubnet_amd import spectral
    for fname in ("live_m.npz", "frozen_m_zoo.npz", "live_tiny_2spk.npz"):
        gold = load(fname)
        X = spectral.stft(_t(gold["wave"]), 512, 128).cpu().numpy()
        np.testing.assert_allclose(X, gold["stft"], atol=3e-6 * np.abs(gold["stft"]).max(), rtol=0)
        enh = gold["enh_stft"]
        y = spectral.istft(_t(enh.reshape(-1, *enh.shape[-2:])), 512, 128, length=gold["wave"].shape[-1]).cpu().numpy()
        ref = gold["enh_y"].reshape(y.shape)
        np.testing.assert_allclose(y, ref, atol=3e-6 * max(1e-3, np.abs(ref).max()), rtol=0)
    with pytest.raises(NotImplementedError):
        spectral.stft(torch.zeros((1, 4000), device=DEV), 256, 64)
    with pytest.raises(RuntimeError):
        spectral.stft(torch.zeros((1, 4000)), 512, 128)
    with pytest.raises(ValueError):
        spectral.istft(torch.zeros((1, 129, 10), dtype=torch.complex64, device=DEV), 512, 128)


@pytest.mark.parametrize("front,kw,seed", [("live", rw.LIVE_M, 5), ("frozen", rw.FROZEN_S, 6), ("frozen", rw.FROZEN_L, 9),
                                           ("live", rw.LIVE_TINY_2SPK, 8)])
def test_fused_input_scan_is_bit_identical(front, kw, seed):
    """sfsn_gsn_layer_scan_fused (layers >= 1 compute x.W_ih^T + b inside the scan from the previous layer's int8 spikes) and
    sfsn_gsn_layer_scan_fused_x (layer 0 of groups with narrow feature rows, from the fp32 features with the bf16 split) ==
    sfsn_spike_proj / sfsn_input_proj_f32 + sfsn_gsn_layer_scan, bit for bit, for every tensor the module returns: hidden sizes 224 / 240 / 160 /
    256 (four and three 64-wide k steps, 14 / 15 / 10 / 16 output tiles), ragged row counts, state carried over chunks."""
    sd = rw.live_state_dict(kw, seed) if front == "live" else rw.frozen_state_dict(kw, seed)
    model = build_module(front, kw, sd)
    wave = torch.from_numpy(rw.synth_wave(4, 70, seed)).to(DEV)  # even batch: group 0's rows are a multiple of 16 (layer-0 fusion)
    stft = model._stft(wave)
    eng = model.engine()
    eng.stack_scan = False  # this test is about the per-layer entry points (tests/test_stack_scan.py covers the stack launch)
    eng.rows_per_wg = (16, 16)
    outs = []
    for fuse, chunk in ((False, 0), (True, 0), (True, 32)):
        eng.fuse_input, eng.seq_chunk = fuse, chunk
        outs.append(eng.forward_stft(stft))
        torch.cuda.synchronize()
    lean = []
    for fuse in (False, True):  # the int8-only kernel variants (no fp32 spike tensors): counts mode
        eng.fuse_input, eng.seq_chunk = fuse, 0
        lean.append(eng.forward_stft(stft, want_layers=False, want_counts=True))
        torch.cuda.synchronize()
    eng.fuse_input, eng.seq_chunk, eng.rows_per_wg = True, 0, (0, 0)
    if kw is not rw.LIVE_TINY_2SPK:  # both fused entry points really ran (the tiny model's hidden size is below their range)
        assert eng.launches.get("fused", 0) > 0 and eng.launches.get("fused_x", 0) > 0, eng.launches
    a = outs[0]
    for b in outs[1:]:
        assert torch.equal(torch.view_as_real(a["enh_stft"]), torch.view_as_real(b["enh_stft"]))
        for x, y in zip(a["fb_all"] + sum(a["sb_all"], []), b["fb_all"] + sum(b["sb_all"], [])):
            assert torch.equal(x, y)
    for b in lean:
        assert torch.equal(torch.view_as_real(a["enh_stft"]), torch.view_as_real(b["enh_stft"]))
        for x, y in zip(a["fb_all"] + sum(a["sb_all"], []), b["fb_all"] + sum(b["sb_all"], [])):
            if torch.is_tensor(y):
                assert torch.equal(x, y)
            else:
                assert int(y.count.item()) == int((x > 0).sum().item())


def test_full_size_fused_forwards_in_flight_equal_the_plain_forward():
    """BASELINE configs[2] sizes (B=64, T=1000, live baseline_m) as bench.py's timed region runs them -- several forwards in
    flight on separate streams, sub-band scans at 16 rows per workgroup with both fused-input scan variants -- against the
    plain single-stream forward (no fusion, 4 rows per workgroup): every returned tensor bit for bit, for every lane."""
    kw, seed = rw.LIVE_M, 21
    model = build_module("live", kw, rw.live_state_dict(kw, seed))
    eng = model.engine()
    eng.stack_scan = False  # this test is about the per-layer entry points (tests/test_stack_scan.py covers the stack launch)
    stft = model._stft(torch.from_numpy(rw.synth_wave(64, 1000, 3)).to(DEV))
    eng.fuse_input, eng.rows_per_wg = False, (0, 0)
    ref = eng.forward_stft(stft)
    torch.cuda.synchronize()
    eng.fuse_input, eng.rows_per_wg = True, (4, 16)
    lanes = [torch.cuda.Stream(device=DEV) for _ in range(6)]
    outs = []
    for s_ in lanes:
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            outs.append(eng.forward_stft(stft))
    torch.cuda.synchronize()
    eng.rows_per_wg = (0, 0)
    assert eng.launches.get("fused", 0) >= 6 and eng.launches.get("fused_x", 0) >= 6
    for b in outs:
        assert torch.equal(torch.view_as_real(ref["enh_stft"]), torch.view_as_real(b["enh_stft"]))
        assert torch.equal(ref["enh_mag"], b["enh_mag"])
        for x, y in zip(ref["fb_all"] + sum(ref["sb_all"], []), b["fb_all"] + sum(b["sb_all"], [])):
            assert torch.equal(x, y)


@pytest.mark.parametrize("T", [1, 2, 3, 5])
def test_fused_scans_on_very_short_sequences(T):
    """Fewer frames than the fused scans' input ring is deep (3 slots): prologue clamping, first-steps drains and the final
    flush, against the two-call form, bit for bit."""
    kw, seed = rw.LIVE_M, 31
    model = build_module("live", kw, rw.live_state_dict(kw, seed))
    eng = model.engine()
    eng.stack_scan = False  # this test is about the per-layer entry points (tests/test_stack_scan.py covers the stack launch)
    rng = np.random.default_rng(T)
    stft = _t((0.3 * (rng.standard_normal((2, 257, T)) + 1j * rng.standard_normal((2, 257, T)))).astype(np.complex64))
    eng.rows_per_wg = (16, 16)
    eng.fuse_input = False
    a = eng.forward_stft(stft)
    eng.fuse_input = True
    n0 = dict(eng.launches)
    b = eng.forward_stft(stft)
    torch.cuda.synchronize()
    eng.rows_per_wg = (0, 0)
    assert eng.launches.get("fused", 0) > n0.get("fused", 0) and eng.launches.get("fused_x", 0) > n0.get("fused_x", 0)
    assert torch.equal(torch.view_as_real(a["enh_stft"]), torch.view_as_real(b["enh_stft"]))
    for x, y in zip(a["fb_all"] + sum(a["sb_all"], []), b["fb_all"] + sum(b["sb_all"], [])):
        assert torch.equal(x, y)


def spec_units(spec, g):
    return (spec["cutoffs"][g + 1] - spec["cutoffs"][g]) // spec["ctr"][g]


def test_errors_match_reference_behaviour():
    """ValueError for an indivisible band (modeling:283-287), AssertionError for a non-2D input (:426), loud failure on CPU;
    train() is served by the differentiable path, not by the inference kernels."""
    import spiking_fullsubnet_amd as pkg
    bad = dict(rw.LIVE_TINY, freq_cutoffs=[0, 30, 128, 256])
    m = pkg.SpikingFullSubNet(**bad).eval().to(DEV)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 2048, device=DEV))
    m = pkg.SpikingFullSubNet(**rw.LIVE_TINY).eval().to(DEV)
    with pytest.raises(AssertionError):
        m(torch.zeros(1, 1, 2048, device=DEV))
    with pytest.raises(RuntimeError):
        pkg.SpikingFullSubNet(**rw.LIVE_TINY).eval()(torch.zeros(1, 2048))
    # training mode is the differentiable path (tests/test_training.py); the inference-only entry points refuse it
    mt = pkg.SpikingFullSubNet(**rw.LIVE_TINY).to(DEV).train()
    assert mt(torch.zeros(2, 2048, device=DEV))[0].requires_grad
    with pytest.raises(ValueError, match="more than 1 value per channel"):  # one clip = one full-band row: nn.BatchNorm1d's own refusal
        mt(torch.zeros(1, 2048, device=DEV))
    with pytest.raises(RuntimeError):
        mt.streaming(batch=1)


def test_sixteen_bit_weights_take_the_two_plane_scan_with_the_same_results():
    """weight_bits = 16 leaves digit plane 0 of every packed matrix zero; sfsn_gsn_layer_scan_w16 skips its matrix instructions in the
    scans it covers (8 instead of 12 per tile and step).  Same sums: the forward equals the three-plane kernels' on the same weights."""
    kw, seed, B, T = rw.LIVE_M, 5, 4, 180
    model = build_module("live", kw, rw.live_state_dict(kw, seed))
    model.weight_bits = 16
    wave = torch.from_numpy(rw.synth_wave(B, T + 1, seed)).to(DEV)
    stft = torch.stft(wave, 512, 128, 512, window=torch.hann_window(512, device=DEV), return_complex=True, pad_mode="constant")[..., :T].contiguous()
    eng = model.engine()
    assert eng.weight_bits == 16
    ref = None
    # per-layer launches (sfsn_gsn_layer_scan_w16) and -- round 6 -- the pair launch of the sub-band layers (sfsn_gsn_stack_scan_x_w16:
    # two-plane scan3 / FUSEDX3 / FUSED3 roles), as one whole-sequence launch and in the overlapped three-chunk schedule
    for stack, chunks in ((False, 0), ("auto", 0), ("auto", 3)):
        eng.stack_scan, eng.overlap_chunks = stack, chunks
        res = {}
        for fast in (False, True):
            eng.w16_fast = fast
            eng.launches = {}
            res[fast] = eng.forward_stft(stft)
            torch.cuda.synchronize()
            assert (eng.launches.get("stack_w16", 0) > 0) == (fast and stack == "auto"), eng.launches
        a, b = res[False], res[True]
        ref = ref or a
        for c in (b, ref):
            assert torch.equal(torch.view_as_real(a["enh_stft"]), torch.view_as_real(c["enh_stft"]))
            for x, y in zip(a["fb_all"] + sum(a["sb_all"], []), c["fb_all"] + sum(c["sb_all"], [])):
                assert torch.equal(x, y)
    eng.check_stack_errors()
    assert float(b["sb_all"][0][2].mean()) > 0.01  # (the cells do spike)


def test_fused_scan_entry_points_reject_what_they_do_not_cover(hip):
    """sfsn_gsn_layer_scan_fused / _fused_x return SFSN_EUNSUPPORTED (the caller then uses the two-call form) or SFSN_EINVAL."""
    from spiking_fullsubnet_amd._lib import FusedInput, FusedX, ScanSegment, SFSN_EINVAL, SFSN_EUNSUPPORTED
    from spiking_fullsubnet_amd.engine import fold_batchnorm, pack_w3
    rng = np.random.default_rng(0)

    def segment(H, R, T):
        w = rng.uniform(-0.1, 0.1, (H, H)).astype(np.float32)
        pk, dq = pack_w3(w)
        keep = [_t(pk), _t(dq), _t(rng.standard_normal(2 * H).astype(np.float32)), _t(np.ones(H, np.float32)), _t(np.zeros(H, np.float32)),
                torch.zeros((R, H), device=DEV), torch.zeros((R, H), device=DEV),
                torch.zeros((T, R, (H + 63) // 64 * 64), dtype=torch.int8, device=DEV),
                torch.zeros((T, R, (H + 63) // 64 * 64), dtype=torch.int8, device=DEV), _t(rng.standard_normal((T, R, 38)).astype(np.float32)),
                _t(rng.uniform(-0.1, 0.1, (H, 38)).astype(np.float32))]
        sg = (ScanSegment * 1)()
        sg[0].w_hh, sg[0].w_dq, sg[0].bias, sg[0].bn_alpha, sg[0].bn_beta = (_p(k) for k in keep[:5])
        sg[0].h_state, sg[0].c_state, sg[0].spikes_i8, sg[0].R = _p(keep[5]), _p(keep[6]), _p(keep[7]), R
        fi = (FusedInput * 1)()
        fi[0].spikes_in, fi[0].w_ih, fi[0].w_ih_dq = keep[8].data_ptr(), keep[0].data_ptr(), keep[1].data_ptr()
        fx = (FusedX * 1)()
        fx[0].x, fx[0].w_ih, fx[0].I = keep[9].data_ptr(), keep[10].data_ptr(), 38
        return sg, fi, fx, keep

    sg, fi, fx, keep = segment(224, 32, 4)
    assert hip.sfsn_gsn_layer_scan_fused(sg, fi, 1, 4, 224, None) == 0 and hip.sfsn_gsn_layer_scan_fused_x(sg, fx, 1, 4, 224, None) == 0
    torch.cuda.synchronize()
    for H in (64, 128, 320):  # outside 128 < H <= 256
        s2, f2, x2, k2 = segment(H, 32, 2)
        assert hip.sfsn_gsn_layer_scan_fused(s2, f2, 1, 2, H, None) == SFSN_EUNSUPPORTED
        assert hip.sfsn_gsn_layer_scan_fused_x(s2, x2, 1, 2, H, None) == SFSN_EUNSUPPORTED
    s3, f3, x3, k3 = segment(224, 24, 2)  # rows not a multiple of 16: only the real-valued variant minds
    assert hip.sfsn_gsn_layer_scan_fused(s3, f3, 1, 2, 224, None) == 0
    assert hip.sfsn_gsn_layer_scan_fused_x(s3, x3, 1, 2, 224, None) == SFSN_EUNSUPPORTED
    fx[0].I = 37
    assert hip.sfsn_gsn_layer_scan_fused_x(sg, fx, 1, 4, 224, None) == SFSN_EUNSUPPORTED
    fx[0].I, fx[0].x = 38, None
    assert hip.sfsn_gsn_layer_scan_fused_x(sg, fx, 1, 4, 224, None) == SFSN_EINVAL
    fi[0].spikes_in = None
    assert hip.sfsn_gsn_layer_scan_fused(sg, fi, 1, 4, 224, None) == SFSN_EINVAL
    assert hip.sfsn_gsn_layer_scan_fused(sg, fi, 0, 4, 224, None) == SFSN_EINVAL
    torch.cuda.synchronize()


def _run_fused(hip, s_in, sd, alpha, beta, h0, c0, want_f32=True, segs_split=None):
    """sfsn_gsn_layer_scan_fused on int8 input spikes s_in [T, R, HP] (a layer >= 1): fp32 spikes (or None), int8 spikes, h, c, count."""
    from spiking_fullsubnet_amd._lib import FusedInput, ScanSegment, check
    from spiking_fullsubnet_amd.engine import pack_w3
    T, R, HP = s_in.shape
    H = sd["weight_hh"].shape[1]
    pk, dq = pack_w3(sd["weight_hh"])
    pki, dqi = pack_w3(sd["weight_ih"])
    cuts = [0, R] if segs_split is None else [0, segs_split, R]
    ns = len(cuts) - 1
    keep = [_t(pk), _t(dq), _t(pki), _t(dqi), _t(sd["bias_ih"]), _t(alpha), _t(beta)]
    seg, fin, outs = (ScanSegment * ns)(), (FusedInput * ns)(), []
    for i in range(ns):
        r0, r1 = cuts[i], cuts[i + 1]
        t = dict(sin=_t(np.ascontiguousarray(s_in[:, r0:r1])), h=_t(h0[r0:r1]), c=_t(c0[r0:r1]),
                 spk=torch.empty((T, r1 - r0, H), device=DEV) if want_f32 else None,
                 s8=torch.zeros((T, r1 - r0, HP), dtype=torch.int8, device=DEV), cnt=torch.zeros((1,), dtype=torch.int64, device=DEV))
        s = seg[i]
        s.zin, s.w_hh, s.w_dq, s.bias, s.bn_alpha, s.bn_beta = None, _p(keep[0]), _p(keep[1]), _p(keep[4]), _p(keep[5]), _p(keep[6])
        s.h_state, s.c_state, s.spikes_f32, s.spikes_i8, s.membrane, s.R = _p(t["h"]), _p(t["c"]), _p(t["spk"]), _p(t["s8"]), None, r1 - r0
        s.spike_count = None if want_f32 else _p(t["cnt"])
        fin[i].spikes_in, fin[i].w_ih, fin[i].w_ih_dq = t["sin"].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr()
        outs.append(t)
    check(hip.sfsn_gsn_layer_scan_fused(seg, fin, ns, T, H, None), "sfsn_gsn_layer_scan_fused")
    torch.cuda.synchronize()
    cat = lambda k: None if outs[0][k] is None else torch.cat([o[k] for o in outs], dim=1 if k in ("spk", "s8") else 0).cpu().numpy()
    return cat("spk"), cat("s8"), cat("h"), cat("c"), sum(int(o["cnt"][0]) for o in outs)


@pytest.mark.parametrize("H,R,T,split", [(224, 37, 33, None), (224, 16, 1, None), (224, 5, 2, None), (160, 21, 19, None), (192, 40, 17, 16),
                                         (144, 9, 12, None), (208, 33, 11, 8), (176, 18, 9, None), (224, 64, 90, 32)])
def test_fused_scan_with_io_waves_equals_round_2_body_and_the_two_calls(hip, H, R, T, split, monkeypatch):
    """Round 6: sfsn_gsn_layer_scan_fused runs scan3j_role (16 rows per workgroup, IO-specialised waves) for H <= 224.  Bit for bit
    equal to round 2's body (SFSN_FUSED_V2=1, read per call) and to sfsn_spike_proj + sfsn_gsn_layer_scan -- fp32 / int8 spikes,
    final h and c -- for every k-step form (H mod 64 in (0, 32]: the 32-wide tail; full steps), ragged row blocks, one / two / odd
    frame counts, two segments in one launch, non-zero initial state; without fp32 spikes the launch's count equals their sum."""
    from test_stack_scan import _spike_proj
    rng = np.random.default_rng(H * 7 + R)
    sd, alpha, beta, bnp = make_layer(rng, H, H, True, True)
    HP = (H + 63) // 64 * 64
    s_in = np.zeros((T, R, HP), np.int8)
    s_in[:, :, :H] = rng.random((T, R, H)) < 0.25
    h0 = (rng.random((R, H)) > 0.5).astype(np.float32)
    c0 = rng.standard_normal((R, H)).astype(np.float32)
    new = _run_fused(hip, s_in, sd, alpha, beta, h0, c0, segs_split=split)
    monkeypatch.setenv("SFSN_FUSED_V2", "1")
    old = _run_fused(hip, s_in, sd, alpha, beta, h0, c0, segs_split=split)
    monkeypatch.delenv("SFSN_FUSED_V2")
    for a, b, nm in zip(new[:4], old[:4], ("fp32 spikes", "int8 spikes", "h", "c")):
        np.testing.assert_array_equal(a, b, err_msg=nm)
    assert new[1].any() and not new[1][:, :, H:].any()
    zin = _spike_proj(hip, s_in, sd["weight_ih"], H)
    spk, _, s8, hT, cT = run_scan(hip, zin, sd["weight_hh"], sd["bias_ih"], alpha, beta, True, h0, c0, want_mem=False)
    np.testing.assert_array_equal(new[0], spk)
    np.testing.assert_array_equal(new[1], s8)
    np.testing.assert_array_equal(new[2], hT)
    np.testing.assert_array_equal(new[3], cT)
    lean = _run_fused(hip, s_in, sd, alpha, beta, h0, c0, want_f32=False, segs_split=split)
    assert lean[0] is None and lean[4] == int(new[0].sum())
    np.testing.assert_array_equal(lean[1], new[1])
    np.testing.assert_array_equal(lean[3], new[3])


@pytest.mark.parametrize("R,T", [(37, 33), (16, 1), (64, 90)])
def test_fused_scan_with_the_io_waves_computing_two_tiles_input_terms_is_bit_identical(hip, R, T, monkeypatch):
    """scan3j_role's OFF form (H = 224, 14 tiles: the loader / storer waves compute the input terms of tiles 12 / 13 and hand them over
    through LDS at the step barrier) -- the default without fp32 spikes, forced here WITH them too (SFSN_S3J_OFF=2) -- against the plain
    form (SFSN_S3J_OFF=0): fp32 / int8 spikes, h, c and the spike count."""
    H = 224
    rng = np.random.default_rng(R * 31 + T)
    sd, alpha, beta, bnp = make_layer(rng, H, H, True, True)
    s_in = np.zeros((T, R, 256), np.int8)
    s_in[:, :, :H] = rng.random((T, R, H)) < 0.25
    h0 = (rng.random((R, H)) > 0.5).astype(np.float32)
    c0 = rng.standard_normal((R, H)).astype(np.float32)
    res = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("SFSN_S3J_OFF", mode)
        res[mode] = (_run_fused(hip, s_in, sd, alpha, beta, h0, c0), _run_fused(hip, s_in, sd, alpha, beta, h0, c0, want_f32=False))
    monkeypatch.delenv("SFSN_S3J_OFF")
    for k in range(2):
        for a, b, nm in zip(res["0"][k], res["2"][k], ("fp32 spikes", "int8 spikes", "h", "c", "count")):
            if a is None:
                assert b is None
            else:
                np.testing.assert_array_equal(a, b, err_msg=f"{nm} ({'with' if k == 0 else 'without'} fp32 spikes)")
    assert res["2"][0][1].any()


def _run_fused_x(hip, x, sd, alpha, beta, h0, c0, want_f32=True, segs_split=None):
    """sfsn_gsn_layer_scan_fused_x on fp32 feature rows x [T, R, I] (a layer 0): fp32 spikes (or None), int8 spikes, h, c, count."""
    from spiking_fullsubnet_amd._lib import FusedX, ScanSegment, check
    from spiking_fullsubnet_amd.engine import pack_w3
    T, R, I = x.shape
    H = sd["weight_hh"].shape[1]
    HP = (H + 63) // 64 * 64
    pk, dq = pack_w3(sd["weight_hh"])
    cuts = [0, R] if segs_split is None else [0, segs_split, R]
    ns = len(cuts) - 1
    keep = [_t(pk), _t(dq), _t(sd["weight_ih"].astype(np.float32)), _t(sd["bias_ih"]), _t(alpha), _t(beta)]
    seg, fin, outs = (ScanSegment * ns)(), (FusedX * ns)(), []
    for i in range(ns):
        r0, r1 = cuts[i], cuts[i + 1]
        t = dict(x=_t(np.ascontiguousarray(x[:, r0:r1])), h=_t(h0[r0:r1]), c=_t(c0[r0:r1]),
                 spk=torch.empty((T, r1 - r0, H), device=DEV) if want_f32 else None,
                 s8=torch.zeros((T, r1 - r0, HP), dtype=torch.int8, device=DEV), cnt=torch.zeros((1,), dtype=torch.int64, device=DEV))
        s = seg[i]
        s.zin, s.w_hh, s.w_dq, s.bias, s.bn_alpha, s.bn_beta = None, _p(keep[0]), _p(keep[1]), _p(keep[3]), _p(keep[4]), _p(keep[5])
        s.h_state, s.c_state, s.spikes_f32, s.spikes_i8, s.membrane, s.R = _p(t["h"]), _p(t["c"]), _p(t["spk"]), _p(t["s8"]), None, r1 - r0
        s.spike_count = None if want_f32 else _p(t["cnt"])
        fin[i].x, fin[i].w_ih, fin[i].I = t["x"].data_ptr(), keep[2].data_ptr(), I
        outs.append(t)
    check(hip.sfsn_gsn_layer_scan_fused_x(seg, fin, ns, T, H, None), "sfsn_gsn_layer_scan_fused_x")
    torch.cuda.synchronize()
    cat = lambda k: None if outs[0][k] is None else torch.cat([o[k] for o in outs], dim=1 if k in ("spk", "s8") else 0).cpu().numpy()
    return cat("spk"), cat("s8"), cat("h"), cat("c"), sum(int(o["cnt"][0]) for o in outs)


@pytest.mark.parametrize("I,H,R,T,split", [(38, 224, 32, 33, None), (38, 224, 16, 1, None), (64, 224, 48, 2, 16), (20, 160, 32, 19, None),
                                           (32, 192, 64, 17, 32), (34, 144, 16, 12, None), (62, 208, 32, 11, None), (6, 176, 16, 9, None),
                                           (38, 224, 512, 60, None)])
def test_fused_x_scan_with_io_waves_equals_round_2_body_and_the_two_calls(hip, I, H, R, T, split, monkeypatch):
    """Round 6: sfsn_gsn_layer_scan_fused_x runs scan3y_role (16 rows per workgroup, IO-specialised waves, the bf16 three-way split of
    the real-valued input product inside) for H <= 224.  Bit for bit equal to round 2's body (SFSN_FUSED_V2=1) and to
    sfsn_input_proj_f32 + sfsn_gsn_layer_scan: one and two 32-wide k-chunks, every k-step form of the recurrent product, one / two /
    odd frame counts, two segments, non-zero initial state, the spike count without fp32 spikes."""
    from test_stack_scan import _input_proj
    rng = np.random.default_rng(H * 5 + I)
    sd, alpha, beta, bnp = make_layer(rng, I, H, True, True)
    x = rng.standard_normal((T, R, I)).astype(np.float32)
    h0 = (rng.random((R, H)) > 0.5).astype(np.float32)
    c0 = rng.standard_normal((R, H)).astype(np.float32)
    new = _run_fused_x(hip, x, sd, alpha, beta, h0, c0, segs_split=split)
    monkeypatch.setenv("SFSN_FUSED_V2", "1")
    old = _run_fused_x(hip, x, sd, alpha, beta, h0, c0, segs_split=split)
    monkeypatch.delenv("SFSN_FUSED_V2")
    for a, b, nm in zip(new[:4], old[:4], ("fp32 spikes", "int8 spikes", "h", "c")):
        np.testing.assert_array_equal(a, b, err_msg=nm)
    assert new[1].any() and not new[1][:, :, H:].any()
    if T * R >= 64:  # (below that sfsn_input_proj_f32 takes its fp32-MFMA form: another summation order)
        zin = _input_proj(hip, x, sd["weight_ih"]).reshape(T, R, H)
        spk, _, s8, hT, cT = run_scan(hip, zin, sd["weight_hh"], sd["bias_ih"], alpha, beta, True, h0, c0, want_mem=False)
        np.testing.assert_array_equal(new[0], spk)
        np.testing.assert_array_equal(new[2], hT)
        np.testing.assert_array_equal(new[3], cT)
    lean = _run_fused_x(hip, x, sd, alpha, beta, h0, c0, want_f32=False, segs_split=split)
    assert lean[0] is None and lean[4] == int(new[0].sum())
    np.testing.assert_array_equal(lean[1], new[1])
    np.testing.assert_array_equal(lean[3], new[3])


def test_feature_launch_zeroes_the_scan_states_and_nothing_else(hip, monkeypatch):
    """sfsn_features_z: extra workgroups of the feature launch write the zero initial state of the forward's scans (MODEL:100-106)
    -- exactly the bytes asked for, the features themselves unchanged -- and the engine's forward gives the same bits with the
    states zeroed that way as with a fill launch of its own (SFSN_ZERO_FOLD=0)."""
    from spiking_fullsubnet_amd import _lib
    from spiking_fullsubnet_amd._lib import FeatureGroup
    B, F, T, I = 3, 33, 40, 32
    rng = np.random.default_rng(5)
    ri = _t(rng.standard_normal((B, F, T, 2)).astype(np.float32))
    xa, xb = torch.empty((T, B, I), device=DEV), torch.empty((T, B, I), device=DEV)
    for n16 in (0, 1, 100, 2048 * 3 + 5):
        buf = torch.full((4 * n16 + 64,), 7.0, device=DEV)
        for x, z in ((xa, None), (xb, buf)):
            g = (FeatureGroup * 1)()
            g[0].x, g[0].lo, g[0].n_units, g[0].ctr, g[0].nbr, g[0].ctr_fb, g[0].nbr_fb, g[0].norm, g[0].ln_eps = x.data_ptr(), 0, 1, I, 0, 0, 0, 0, 1e-5
            rc = hip.sfsn_features_z(ri.data_ptr(), None, B, F, T, 0, 0.5, g, 1, 0, T, None if z is None else ctypes.c_void_p(z.data_ptr() + 64),
                                     0 if z is None else 16 * n16, None)
            assert rc == 0, rc
        torch.cuda.synchronize()
        assert torch.equal(xa, xb)
        assert bool((buf[:16] == 7).all()) and bool((buf[16:16 + 4 * n16] == 0).all()) and bool((buf[16 + 4 * n16:] == 7).all()), n16
    assert hip.sfsn_features_z(ri.data_ptr(), None, B, F, T, 0, 0.5, g, 1, 0, T, ctypes.c_void_p(buf.data_ptr() + 4), 16, None) == _lib.SFSN_EINVAL
    assert hip.sfsn_features_z(ri.data_ptr(), None, B, F, T, 0, 0.5, g, 1, 0, T, ctypes.c_void_p(buf.data_ptr()), 24, None) == _lib.SFSN_EINVAL
    import spiking_fullsubnet_amd as pkg
    kw = rw.LIVE_TINY
    model = pkg.SpikingFullSubNet(**kw)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in rw.live_state_dict(kw, 3).items()}, strict=True)
    model = model.eval().to(DEV)
    stft = model._stft(_t(rw.synth_wave(3, 300, 9)))
    a = model.engine().forward_stft(stft)
    monkeypatch.setenv("SFSN_ZERO_FOLD", "0")
    b = model.engine().forward_stft(stft)
    assert torch.equal(torch.view_as_real(a["enh_stft"]), torch.view_as_real(b["enh_stft"]))
    assert all(torch.equal(u, v) for u, v in zip(a["fb_all"] + sum(a["sb_all"], []), b["fb_all"] + sum(b["sb_all"], [])))
    model.engine().check_stack_errors()


def _featproj_case(hip, rng, B, F, T, FB, groups, Hs, norm, t0, nt, with_x=True, zero=0, ref=True):
    """(x, z) per group from the two calls and from the fused launch; groups: (lo, n_units, ctr, nbr, ctr_fb, nbr_fb); Hs: H or None."""
    from spiking_fullsubnet_amd import _lib
    from spiking_fullsubnet_amd._lib import FeatProjJob, FeatureGroup, check
    ri = _t(rng.standard_normal((B, F, T, 2)).astype(np.float32) * 3.0)
    fbp = _t(rng.standard_normal((T, B, max(FB, 1))).astype(np.float32)) if FB else None
    n = len(groups)
    fg, jobs, keep, xa, xb, za, zb = (FeatureGroup * n)(), (FeatProjJob * n)(), [], [], [], [], []
    for i, ((lo, nu, ctr, nbr, cfb, nfb), H) in enumerate(zip(groups, Hs)):
        I = ctr + 2 * nbr + (cfb + 2 * nfb if cfb else 0)
        x1, x2 = torch.full((T, B * nu, I), float("nan"), device=DEV), torch.full((T, B * nu, I), float("nan"), device=DEV)
        g = fg[i]
        g.lo, g.n_units, g.ctr, g.nbr, g.ctr_fb, g.nbr_fb, g.norm, g.ln_eps = lo, nu, ctr, nbr, cfb, nfb, norm, 1e-5
        ts = []
        if norm == _lib.NORM_LAYERNORM:
            ts = [_t(rng.uniform(0.5, 1.5, I).astype(np.float32)), _t(rng.standard_normal(I).astype(np.float32) * 0.1)]
            g.ln_w, g.ln_b = ts[0].data_ptr(), ts[1].data_ptr()
        elif norm == _lib.NORM_LAPLACE:
            ts = [_t(rng.uniform(0.5, 2.0, B).astype(np.float32))]
            g.mu = ts[0].data_ptr()
        elif norm == _lib.NORM_GAUSSIAN:
            ts = [_t(rng.uniform(0.5, 2.0, B).astype(np.float32)), _t(rng.uniform(0.5, 2.0, B).astype(np.float32))]
            g.mu, g.ln_w = ts[0].data_ptr(), ts[1].data_ptr()
        keep.append(ts)
        jobs[i].feat = g
        fg[i].x = x1.data_ptr()
        jobs[i].feat.x = x2.data_ptr() if (with_x or H is None) else None
        xa.append(x1)
        xb.append(x2)
        if H is None:
            za.append(None)
            zb.append(None)
            continue
        w, bias = _t(rng.uniform(-0.1, 0.1, (H, I)).astype(np.float32)), _t(rng.standard_normal(H).astype(np.float32))
        z1, z2 = torch.full((nt, B * nu, H), float("nan"), device=DEV), torch.full((nt, B * nu, H), float("nan"), device=DEV)
        keep.append((w, bias))
        jobs[i].w, jobs[i].bias, jobs[i].z, jobs[i].H, jobs[i].ldz = w.data_ptr(), bias.data_ptr(), z2.data_ptr(), H, H
        za.append((z1, w, bias, I))
        zb.append(z2)
    if ref:
        check(hip.sfsn_features(_p(ri), None if fbp is None else _p(fbp), B, F, T, FB, 0.5, fg, n, t0, nt, None), "sfsn_features")
    for i, za_i in enumerate(za):
        if za_i is not None and ref:
            z1, w, bias, I = za_i
            R = xa[i].shape[1]
            check(hip.sfsn_input_proj_f32(ctypes.c_void_p(xa[i].data_ptr() + t0 * R * I * 4), _p(w), _p(bias), _p(z1), nt * R, I, z1.shape[2],
                                          z1.shape[2], None), "sfsn_input_proj_f32")
    zbuf = torch.full((4 * zero + 8,), 7.0, device=DEV)
    rc = hip.sfsn_features_proj(_p(ri), None if fbp is None else _p(fbp), B, F, T, FB, 0.5, jobs, n, t0, nt,
                                ctypes.c_void_p(zbuf.data_ptr() + 16) if zero else None, 16 * zero, None)
    torch.cuda.synchronize()
    if zero:
        assert bool((zbuf[:4] == 7).all()) and bool((zbuf[4:4 + 4 * zero] == 0).all()) and bool((zbuf[4 + 4 * zero:] == 7).all())
    return rc, xa, xb, [None if a is None else a[0] for a in za], zb


def _same(a, b):  # bit-identical including the NaN canaries of rows outside [t0, t0 + nt)
    return torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("norm", ["layernorm", "laplace", "gaussian", "none"])
def test_features_and_input_products_in_one_launch_equal_the_two_calls(hip, norm):
    """sfsn_features_proj (round 5, ABI 17): the feature rows and the layer-0 input terms of a chunk from ONE launch are bit for
    bit what sfsn_features + sfsn_input_proj_f32 write -- baseline_m's three sub-band groups (group 0 rows only, as beside the
    FUSEDX3 role), its full-band group (H = 320), a ragged frame range in the middle of the sequence, every normalisation, the
    zero side job; with feat.x = NULL the rows are simply not written."""
    from spiking_fullsubnet_amd import _lib
    nm = dict(layernorm=_lib.NORM_LAYERNORM, laplace=_lib.NORM_LAPLACE, gaussian=_lib.NORM_GAUSSIAN, none=_lib.NORM_NONE)[norm]
    rng = np.random.default_rng(123)
    B, F, T, FB = 5, 257, 150, 64
    sb = [(0, 8, 4, 15, 4, 0), (32, 3, 32, 15, 32, 0), (128, 2, 64, 15, 64, 0)]
    for t0, nt, Hs, zero in ((0, T, (None, 224, 224), 0), (37, 77, (None, 224, 224), 100), (8, 64, (32, 48, 224), 5000)):
        rc, xa, xb, za, zb = _featproj_case(hip, rng, B, F, T, FB, sb, Hs, nm, t0, nt, zero=zero)
        assert rc == 0, rc
        for g in range(3):
            assert _same(xa[g], xb[g]), (norm, t0, g)
            if za[g] is not None:
                assert _same(za[g], zb[g]), (norm, t0, g)
    # the full-band group (no tiled full-band input), H = 320
    rc, xa, xb, za, zb = _featproj_case(hip, rng, B, F, T, 0, [(0, 1, 64, 0, 0, 0)], (320,), nm, 16, 100, zero=64)
    assert rc == 0 and _same(xa[0], xb[0]) and _same(za[0], zb[0])
    # rows not written when nobody reads them
    rc, xa, xb, za, zb = _featproj_case(hip, rng, B, F, T, FB, sb, (None, 224, 224), nm, 0, T, with_x=False)
    assert rc == 0 and _same(xa[0], xb[0]) and bool(torch.isnan(xb[1]).all()) and bool(torch.isnan(xb[2]).all())
    assert _same(za[1], zb[1]) and _same(za[2], zb[2])


@pytest.mark.gpu
def test_features_proj_refuses_what_the_two_calls_would_run_differently(hip):
    """Shapes sfsn_input_proj_f32 runs on its fp32-MFMA kernels (another rounding), fewer than 64 rows, odd widths: refused
    (SFSN_EUNSUPPORTED), the engine then issues the two calls; bad arguments are SFSN_EINVAL."""
    from spiking_fullsubnet_amd import _lib
    rng = np.random.default_rng(9)
    B, F, T = 2, 257, 40
    ok = [(32, 3, 32, 15, 32, 0)]
    case = lambda *a, **k: _featproj_case(hip, rng, B, F, T, *a, ref=False, **k)[0]
    assert case(64, ok, (224,), 0, 0, T) == 0
    assert case(64, ok, (224,), 0, 0, 8) == _lib.SFSN_EUNSUPPORTED                         # 48 rows
    assert case(64, [(32, 3, 32, 15, 31, 0)], (224,), 0, 0, T) == _lib.SFSN_EUNSUPPORTED   # odd I
    assert case(64, [(128, 2, 64, 15, 64, 0)], (400,), 0, 0, T) == _lib.SFSN_EUNSUPPORTED  # H > 384
    assert case(64, [(0, 1, 160, 0, 0, 0)], (224,), 0, 0, T) == _lib.SFSN_EUNSUPPORTED     # 160 bins
    assert case(64, ok, (222,), 0, 0, T) == _lib.SFSN_EUNSUPPORTED                         # H % 4
    assert case(48, ok, (224,), 0, 0, T) == _lib.SFSN_EUNSUPPORTED                         # 512 % FB
    assert case(64, ok, (224,), 0, 30, 20) == _lib.SFSN_EINVAL                             # frames past the end


@pytest.mark.gpu
@pytest.mark.parametrize("front,kw,seed", [("live", rw.LIVE_M, 21), ("frozen", rw.FROZEN_M, 32), ("live", rw.LIVE_TINY, 11)])
def test_forward_with_the_fused_feature_product_launch_is_bit_identical(front, kw, seed, monkeypatch):
    """The engine's forward with features + layer-0 input products in one launch (the default) against the two launches
    (fuse_featproj = False): the same bits in every output, in the strict schedule and at 16 rows per workgroup; with
    layer_outputs = "counts" the skipped feature rows come back as shape-only (meta) entries and the counts agree."""
    sd = rw.live_state_dict(kw, seed) if front == "live" else rw.frozen_state_dict(kw, seed)
    model = build_module(front, kw, sd)
    eng = model.engine()
    stft = model._stft(_t(rw.synth_wave(4, 128 * 420, 5)))
    for rows in (None, (16, 16)):
        if rows is not None:
            eng.rows_per_wg, eng.stack_scan = rows, False
        eng.fuse_featproj = True
        a = eng.forward_stft(stft)
        n_fused = eng.launches.get("featproj", 0)
        eng.fuse_featproj = False
        b = eng.forward_stft(stft)
        assert n_fused > 0 and eng.launches.get("featproj", 0) == n_fused
        assert torch.equal(torch.view_as_real(a["enh_stft"]), torch.view_as_real(b["enh_stft"])) and torch.equal(a["enh_mag"], b["enh_mag"])
        assert all(torch.equal(u, v) for u, v in zip(a["fb_all"] + sum(a["sb_all"], []), b["fb_all"] + sum(b["sb_all"], [])))
        eng.fuse_featproj = True
        c = eng.forward_stft(stft, want_layers=False, want_counts=True)
        assert torch.equal(torch.view_as_real(a["enh_stft"]), torch.view_as_real(c["enh_stft"]))
        assert c["fb_all"][0].device.type == "meta" and c["fb_all"][0].shape == a["fb_all"][0].shape
        for la, lc in zip([a["fb_all"]] + a["sb_all"], [c["fb_all"]] + c["sb_all"]):
            for u, v in zip(la[1:-1], lc[1:-1]):
                assert int((u > 0).sum()) == int(v.count)
    eng.check_stack_errors()


def _projdf_case(hip, rng, B, F, T, S, H, groups, t0, nt, write_proj=True):
    """groups: [(n_units, fc, df)].  Runs sfsn_spike_proj (per group) + sfsn_deepfilter and sfsn_proj_deepfilter on the same random
    spikes / weights / spectrum; returns both result sets (torch tensors; rows / frames outside [t0, t0 + nt) carry NaN canaries)."""
    from spiking_fullsubnet_amd._lib import DfGroup, ProjDfGroup, check
    from spiking_fullsubnet_amd.engine import pack_w3
    HP = (H + 63) // 64 * 64
    stft = _t(rng.standard_normal((B, F, T, 2)).astype(np.float32))
    keep, per = [], []
    for (N, fc, df) in groups:
        P = 2 * fc * df * S
        s8 = np.zeros((T, B * N, HP), np.int8)
        s8[:, :, :H] = rng.random((T, B * N, H)) < 0.3
        w = (rng.standard_normal((P, H)) * 0.2).astype(np.float32)
        pk, dq = pack_w3(w)
        per.append(dict(N=N, fc=fc, df=df, P=P, s8=_t(s8), pk=_t(pk), dq=_t(dq), bias=_t(rng.standard_normal(P).astype(np.float32))))
    out = {}
    for how in ("two", "one"):
        enh = torch.full((B, S, F, T, 2), float("nan"), device=DEV)
        mag = torch.full((B, S, F, T), float("nan"), device=DEV)
        projs = [torch.full((T, B * g["N"], g["P"]), float("nan"), device=DEV) for g in per]
        if how == "two":
            dfg = (DfGroup * len(per))()
            for a, g, y in zip(dfg, per, projs):
                R = B * g["N"]
                check(hip.sfsn_spike_proj(ctypes.c_void_p(g["s8"].data_ptr() + t0 * R * HP), _p(g["pk"]), _p(g["dq"]), _p(g["bias"]),
                                          ctypes.c_void_p(y.data_ptr() + t0 * R * g["P"] * 4), nt * R, H, g["P"], g["P"], None), "spike_proj")
                a.proj, a.n_units, a.fc, a.df = y.data_ptr(), g["N"], g["fc"], g["df"]
            check(hip.sfsn_deepfilter(_p(stft), B, F, T, S, dfg, len(per), _p(enh), _p(mag), t0, nt, None), "deepfilter")
        else:
            arr = (ProjDfGroup * len(per))()
            for a, g, y in zip(arr, per, projs):
                a.spikes_i8, a.w_packed, a.w_dq, a.bias = g["s8"].data_ptr(), g["pk"].data_ptr(), g["dq"].data_ptr(), g["bias"].data_ptr()
                a.proj = y.data_ptr() if write_proj else None
                a.n_units, a.fc, a.df = g["N"], g["fc"], g["df"]
            rc = hip.sfsn_proj_deepfilter(_p(stft), B, F, T, S, H, arr, len(per), _p(enh), _p(mag), t0, nt, None)
            if rc != 0:
                return rc, None
        torch.cuda.synchronize()
        out[how] = (enh, mag, projs)
    return 0, out


def _same_nan(a, b):
    return torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize("B,F,T,S,H,groups,t0,nt", [
    (3, 257, 50, 1, 224, [(8, 4, 5), (3, 32, 3), (2, 64, 1)], 0, 50),      # baseline_m's groups, a ragged last tile
    (2, 257, 77, 1, 224, [(8, 4, 5), (3, 32, 3), (2, 64, 1)], 19, 41),     # a chunk in the middle: history from before t0
    (2, 129, 40, 2, 64, [(4, 8, 2), (2, 16, 1), (1, 32, 1)], 0, 40),       # two speakers (wsj0-mix shape class), 129 bins
    (1, 257, 16, 1, 32, [(8, 4, 2), (3, 32, 3), (2, 64, 1)], 0, 16),       # tiny hidden size, exactly one tile
    (2, 257, 33, 1, 256, [(16, 2, 3), (7, 32, 1)], 5, 28),                 # many units in a group (several unit passes), bins left over
    (5, 257, 9, 1, 160, [(8, 4, 5), (3, 32, 3), (2, 64, 1)], 0, 9),        # fewer frames than one tile
])
def test_projection_and_deep_filter_in_one_launch_equal_the_two_calls(hip, B, F, T, S, H, groups, t0, nt):
    """sfsn_proj_deepfilter (round 6) == sfsn_spike_proj + sfsn_deepfilter, bit for bit: coefficient rows, enhanced spectrum incl. the
    pass-through bins, magnitude; frames outside [t0, t0 + nt) untouched; with proj = NULL the rows are not written and the spectrum is
    the same."""
    rng = np.random.default_rng(B * 1000 + T)
    rc, out = _projdf_case(hip, rng, B, F, T, S, H, groups, t0, nt)
    assert rc == 0
    (e2, m2, p2), (e1, m1, p1) = out["two"], out["one"]
    assert _same_nan(e1, e2) and _same_nan(m1, m2)
    for a, b in zip(p1, p2):
        assert _same_nan(a, b)
    assert not torch.isnan(e1[:, :, :, t0:t0 + nt]).any() and (t0 == 0 or torch.isnan(e1[:, :, :, :t0]).all())
    rng = np.random.default_rng(B * 1000 + T)
    rc, out = _projdf_case(hip, rng, B, F, T, S, H, groups, t0, nt, write_proj=False)
    assert rc == 0 and _same_nan(out["one"][0], e2) and _same_nan(out["one"][1], m2)
    assert all(torch.isnan(y).all() for y in out["one"][2])


def test_proj_deepfilter_argument_checks(hip):
    from spiking_fullsubnet_amd import _lib
    from spiking_fullsubnet_amd._lib import ProjDfGroup
    arr = (ProjDfGroup * 1)()
    one = ctypes.c_void_p(256)
    assert hip.sfsn_proj_deepfilter(one, 1, 257, 8, 1, 224, arr, 1, one, one, 0, 8, None) == _lib.SFSN_EINVAL       # empty descriptor
    a = arr[0]
    a.spikes_i8, a.w_packed, a.w_dq, a.n_units, a.fc, a.df = 256, 256, 256, 2, 3, 1                                    # P = 6: not a multiple of 4
    assert hip.sfsn_proj_deepfilter(one, 1, 257, 8, 1, 224, arr, 1, one, one, 0, 8, None) == _lib.SFSN_EUNSUPPORTED
    a.fc = 4
    assert hip.sfsn_proj_deepfilter(one, 1, 257, 8, 1, 320, arr, 1, one, one, 0, 8, None) == _lib.SFSN_EUNSUPPORTED  # H > 256
    assert hip.sfsn_proj_deepfilter(one, 1, 257, 8, 1, 224, arr, 1, one, one, 4, 8, None) == _lib.SFSN_EINVAL        # frames past T
    assert hip.sfsn_proj_deepfilter(one, 1, 5, 8, 1, 224, arr, 1, one, one, 0, 8, None) == _lib.SFSN_EINVAL          # more bins than F


@pytest.mark.parametrize("front,kw,seed,covered", [("live", rw.LIVE_TINY, 11, True), ("live", rw.LIVE_TINY_2SPK, 12, True), ("live", rw.LIVE_M, 21, True),
                                                   ("frozen", rw.FROZEN_S, 32, True), ("live", rw.LIVE_TINY_UNSHARED, 13, True),
                                                   ("frozen", rw.FROZEN_TINY, 31, False)])  # (P = 384 coefficients per row: the two launches)
def test_forward_with_the_fused_projection_filter_launch_is_bit_identical(front, kw, seed, covered):
    """A whole forward with the sub-band epilogue in one launch per chunk (Engine.fuse_projdf, the default) against the two launches:
    every tensor of the module API, in the overlapped three-chunk schedule and as one whole-sequence chunk; layer_outputs "counts":
    the same spectrum, the coefficient rows not written (a meta tensor of the same shape)."""
    sd = rw.live_state_dict(kw, seed) if front == "live" else rw.frozen_state_dict(kw, seed)
    model = build_module(front, kw, sd)
    eng = model.engine()
    stft = model._stft(_t(rw.synth_wave(3, 128 * 330, 5)))
    for chunks in (3, 0):
        eng.overlap_chunks = chunks
        eng.fuse_projdf = True
        n0 = eng.launches.get("projdf", 0)
        a = eng.forward_stft(stft)
        assert (eng.launches.get("projdf", 0) > n0) == covered
        eng.fuse_projdf = False
        n1 = eng.launches.get("projdf", 0)
        b = eng.forward_stft(stft)
        assert eng.launches.get("projdf", 0) == n1
        assert torch.equal(torch.view_as_real(a["enh_stft"]), torch.view_as_real(b["enh_stft"])) and torch.equal(a["enh_mag"], b["enh_mag"])
        assert all(torch.equal(u, v) for u, v in zip(a["fb_all"] + sum(a["sb_all"], []), b["fb_all"] + sum(b["sb_all"], [])))
        eng.fuse_projdf = True
        for skip in (False, True):
            eng.lean_skips_proj = skip
            c = eng.forward_stft(stft, want_layers=False, want_counts=True)
            assert torch.equal(torch.view_as_real(a["enh_stft"]), torch.view_as_real(c["enh_stft"])) and torch.equal(a["enh_mag"], c["enh_mag"])
            for la, lc in zip(a["sb_all"], c["sb_all"]):
                assert lc[-1].shape == la[-1].shape and (lc[-1].device.type == "meta") == (covered and skip)
                if not (covered and skip):
                    assert torch.equal(la[-1], lc[-1])
        eng.lean_skips_proj = False
    eng.check_stack_errors()


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher re-executes itself under torch.distributed.run (one rank per GPU; with
    SFSN_BENCH_BACKEND=gloo the two ranks share this box's GPU -- a plumbing check of the N > 1 path): one JSON line, n_gpus 2,
    the whole-job value of both ranks' clips."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SFSN_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
                          "--no-phase-a", "--inflight", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["world_size"] == 2 and d["config"]["backend"] == "gloo"
    assert d["scaling"] == "weak" and d["value"] > 0 and d["steps"] == 4


def test_bench_rccl_path_with_a_single_rank():
    """SFSN_BENCH_FORCE_DIST=1: bench.py initialises RCCL and runs the per-step all_gather_into_tensor of the enhanced magnitudes,
    the barrier and the max-over-ranks reduction with ONE rank -- the N > 1 code path on the real backend, on a one-GPU box.
    stdout carries the JSON line and nothing else (RCCL's banner goes to stderr)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SFSN_BENCH_FORCE_DIST="1", MASTER_PORT="29533")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-phase-a",
                          "--inflight", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["config"]["backend"] == "nccl" and d["config"]["world_size"] == 1 and d["n_gpus"] == 1 and d["value"] > 0
