"""Pins the CPU oracle (oracle/) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

import parity
import refweights as rw
from oracle import Oracle
from oracle import model as omodel

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(G, name), allow_pickle=False))


@pytest.fixture(scope="module")
def cells():
    return load("gsn_cells.npz")


def _bn(d, p):
    k = p + "batchnorm."
    return (d[k + "weight"], d[k + "bias"], d[k + "running_mean"], d[k + "running_var"]) if k + "weight" in d else None


@pytest.mark.parametrize("ci", range(5))
def test_gsn_layer_matches_reference_cells(cells, ci):
    """Layer scan (efficient_spiking_neuron.py:75-81,132-153): spikes, per-step membranes, final state."""
    name = str(cells["cases"][ci])
    I, H, L, R, T, shared, bn = (int(v) for v in cells["dims"][ci])
    o = Oracle("f32")
    x = cells[f"{name}/x"]
    sd = {k[len(name) + 4:]: v for k, v in cells.items() if k.startswith(name + "/sd/")}
    t_valid = np.full(R, T)
    for l in range(L):
        p = f"layers.{l}.cell."
        spk, mem, hT, cT = o.gsn_layer(x, sd[p + "weight_ih"], sd[p + "weight_hh"], sd[p + "bias_ih"], bn=_bn(sd, p),
                                       shared=bool(shared), h0=cells[f"{name}/h0/{l}"], c0=cells[f"{name}/c0/{l}"])
        ref, mem_ref = cells[f"{name}/spikes/{l}"], cells[f"{name}/membrane/{l}"]
        t_valid, st = parity.check_chain(spk, ref, np.abs(mem_ref) < parity.TAU, t_valid, f"{name}/L{l}", mem, mem_ref)
        rows_ok = t_valid == T
        np.testing.assert_array_equal(hT[rows_ok], cells[f"{name}/hT/{l}"][rows_ok])
        np.testing.assert_allclose(cT[rows_ok], cells[f"{name}/cT/{l}"][rows_ok], atol=parity.MEM_ATOL, rtol=parity.MEM_RTOL)
        assert st["spike_agreement"] > 0.999, st
        x = spk  # free-running: the oracle's own spikes feed the next layer


def test_gsn_layer_chunked_state_carry(cells):
    """StackedGSU passes states in and out (efficient_spiking_neuron.py:50-62): two half scans == one scan, bit for bit."""
    name = "tiny_shared_bn"
    o = Oracle("f32")
    sd = {k[len(name) + 4:]: v for k, v in cells.items() if k.startswith(name + "/sd/")}
    p = "layers.0.cell."
    x = cells[f"{name}/x"]
    full = o.gsn_layer(x, sd[p + "weight_ih"], sd[p + "weight_hh"], sd[p + "bias_ih"], bn=_bn(sd, p))
    a = o.gsn_layer(x[:17], sd[p + "weight_ih"], sd[p + "weight_hh"], sd[p + "bias_ih"], bn=_bn(sd, p))
    b = o.gsn_layer(x[17:], sd[p + "weight_ih"], sd[p + "weight_hh"], sd[p + "bias_ih"], bn=_bn(sd, p), h0=a[2], c0=a[3])
    np.testing.assert_array_equal(np.concatenate([a[0], b[0]]), full[0])
    np.testing.assert_array_equal(b[3], full[3])


CASES = [
    ("live_tiny.npz", "live", rw.LIVE_TINY, 11),
    ("live_tiny_2spk.npz", "live", rw.LIVE_TINY_2SPK, 12),
    ("live_tiny_unshared.npz", "live", rw.LIVE_TINY_UNSHARED, 13),
    ("live_m.npz", "live", rw.LIVE_M, 21),
    ("frozen_tiny.npz", "frozen", rw.FROZEN_TINY, 31),
    ("frozen_s_zoo.npz", "frozen", rw.FROZEN_S, None), ("frozen_m_zoo.npz", "frozen", rw.FROZEN_M, None),
    ("frozen_l.npz", "frozen", rw.FROZEN_L, 33), ("frozen_xl.npz", "frozen", rw.FROZEN_XL, 34),
    ("frozen_tiny_gauss.npz", "frozen", rw.FROZEN_TINY_GAUSS, 37), ("frozen_tiny_cum.npz", "frozen", rw.FROZEN_TINY_CUM, 35), ("frozen_m_cum.npz", "frozen", rw.FROZEN_M_CUM, 36),
    # round 3: BASELINE configs[0] as written (trained baseline_s, ONE 4 s clip = 501 frames; weights: frozen_s_zoo.npz) and the
    # bench's sizes on two clips x 200 frames of amplitude-modulated noise (SURVEY 8d's second input distribution)
    ("frozen_s_zoo_4s.npz", "frozen", rw.FROZEN_S, "frozen_s_zoo.npz"), ("live_m_am.npz", "live", rw.LIVE_M, 21),
]


def case_inputs(fname, front, kw, seed):
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


@pytest.mark.parametrize("fname,front,kw,seed", CASES, ids=[c[0][:-4] for c in CASES])
def test_whole_model_matches_reference(fname, front, kw, seed):
    """Whole path (modeling_spiking_fullsubnet.py:415-474 / model_low_freq.py:561-618) from the complex STFT:
    layer inputs, spikes (causal rule), projections, enhanced spectrum, enhanced magnitude."""
    gold, spec, sd = case_inputs(fname, front, kw, seed)
    res = omodel.forward_from_stft(spec, sd, gold["stft"], "f32", want_membrane=True)
    res["mem"] = {("fb", l): m for l, m in enumerate(res["fb_mem"])}
    for g, mems in enumerate(res["sb_mem"]):
        res["mem"].update({(f"sb{g}", l): m for l, m in enumerate(mems)})
    stats = parity.check_model(res, gold, spec, tag=fname + ":")
    for st in stats:
        assert st["spike_agreement"] > 0.995, st
    if "enh_mag" in gold and all(st["diverged"] == 0 for st in stats):
        B, S, F, T = res["enh_mag"].shape
        np.testing.assert_allclose(res["enh_mag"].reshape(B * S, F, T), gold["enh_mag"], rtol=parity.REL, atol=parity.ATOL)
    if "synops" in gold and all(st["diverged"] == 0 for st in stats):
        assert omodel.compute_synops(res["fb_all"], res["sb_all"], spec["shared"]) == pytest.approx(float(gold["synops"]), rel=1e-6)
        assert omodel.compute_neuronops(res["fb_all"], res["sb_all"]) == float(gold["neuronops"])


def test_gather_rejects_indivisible_band():
    """_freq_unfold raises ValueError when (hi-lo) % ctr != 0 (modeling_spiking_fullsubnet.py:283-287)."""
    o = Oracle("f32")
    with pytest.raises(ValueError):
        o.gather_group(np.zeros((1, 256, 4), np.float32), np.zeros((4, 1, 64), np.float32), 0, 30, 4, 15)


def test_reflect_index_map():
    """Reflect (no edge repeat) at both spectrum ends: f<0 -> -f, f>255 -> 510-f (F.pad reflect, modeling:294,299)."""
    o = Oracle("f32")
    mag = np.arange(256, dtype=np.float32)[None, :, None].repeat(2, 2)
    fb = np.zeros((2, 1, 64), np.float32)
    x0 = o.gather_group(mag, fb, 0, 32, 4, 15)      # unit 0, feature j -> bin |j-15|
    assert x0[0, 0, :34].tolist() == [abs(j - 15) for j in range(34)]
    x2 = o.gather_group(mag, fb, 128, 256, 64, 15)  # last unit reaches past bin 255
    want = [f if f <= 255 else 510 - f for f in range(192 - 15, 256 + 15)]
    assert x2[0, 1, :94].tolist() == want


def test_fp32_vs_fp64_noise_floor_is_reported():
    """The oracle's own fp32-vs-fp64 disagreement (the parity noise floor): small but not zero-guaranteed."""
    gold, spec, sd = case_inputs(*CASES[0])
    r32 = omodel.forward_from_stft(spec, sd, gold["stft"], "f32")
    r64 = omodel.forward_from_stft(spec, sd, gold["stft"], "f64")
    agree = [float(((a > .5) == (b > .5)).mean()) for a, b in zip(r32["fb_all"][1:-1], r64["fb_all"][1:-1])]
    assert min(agree) > 0.99


@pytest.mark.parametrize("fname", ["live_tiny.npz", "live_m.npz", "frozen_s_zoo.npz", "frozen_m_zoo.npz"])
def test_stft_and_istft_restatements_match_the_reference(golden_dir, fname):
    """The edges of the path (audio_feature.py:236-347): the recorded waveform -> the STFT the reference computed from it, and
    the recorded enhanced STFT -> the waveform the reference returned."""
    gold = np.load(os.path.join(golden_dir, fname))
    X = omodel.stft(gold["wave"])
    assert X.shape == gold["stft"].shape
    np.testing.assert_allclose(X, gold["stft"], atol=2e-6 * np.abs(gold["stft"]).max(), rtol=0)
    enh = gold["enh_stft"]
    y = omodel.istft(enh.reshape(-1, *enh.shape[-2:]), length=gold["wave"].shape[-1])
    ref = gold["enh_y"].reshape(y.shape)
    np.testing.assert_allclose(y, ref, atol=2e-6 * max(1e-3, np.abs(ref).max()), rtol=0)
