"""Deterministic synthetic weights for the reference's two model front-ends, as numpy state dicts.

Shared by tests/golden/make_golden.py (which loads them into the *reference* modules to produce the
golden outputs) and by the tests / bench (which load them into the oracle and the HIP modules), so the
large weight sets never need to be stored in fixtures: a (kwargs, seed) pair reproduces them bit for
bit anywhere (numpy's PCG64 streams are platform independent).

Distributions follow SURVEY 8c: reference-style U(-1/sqrt(H), 1/sqrt(H)) for cell weights and biases
(efficient_spiking_neuron.py:127-130), nn.Linear-style U(-1/sqrt(fan_in), ..) for projections, and
*randomised* BatchNorm statistics / affine terms and LayerNorm affine terms so that those code paths
are non-trivial.
"""
from __future__ import annotations

import numpy as np


def _cell(rng, prefix, I, H, shared, bn, sd):
    G = 1 if shared else 2
    s = 1.0 / np.sqrt(H)
    sd[prefix + "weight_ih"] = rng.uniform(-s, s, (G * H, I)).astype(np.float32)
    sd[prefix + "weight_hh"] = rng.uniform(-s, s, (G * H, H)).astype(np.float32)
    sd[prefix + "bias_ih"] = rng.uniform(-s, s, (2 * H,)).astype(np.float32)
    if bn:
        sd[prefix + "batchnorm.weight"] = rng.normal(1.0, 0.2, (H,)).astype(np.float32)
        sd[prefix + "batchnorm.bias"] = rng.normal(0.0, 0.3, (H,)).astype(np.float32)
        sd[prefix + "batchnorm.running_mean"] = rng.normal(0.0, 0.5, (H,)).astype(np.float32)
        sd[prefix + "batchnorm.running_var"] = rng.uniform(0.3, 1.5, (H,)).astype(np.float32)
        sd[prefix + "batchnorm.num_batches_tracked"] = np.asarray(0, dtype=np.int64)


def _sequence_model(rng, prefix, I, H, L, P, shared, bn, ln, proj_name, sd):
    if ln:
        sd[prefix + "pre_layer_norm.weight"] = rng.normal(1.0, 0.1, (I,)).astype(np.float32)
        sd[prefix + "pre_layer_norm.bias"] = rng.normal(0.0, 0.1, (I,)).astype(np.float32)
    for l in range(L):
        _cell(rng, f"{prefix}sequence_model.layers.{l}.cell.", I if l == 0 else H, H, shared, bn, sd)
    s = 1.0 / np.sqrt(H)
    sd[f"{prefix}{proj_name}.weight"] = rng.uniform(-s, s, (P, H)).astype(np.float32)
    sd[f"{prefix}{proj_name}.bias"] = rng.uniform(-s, s, (P,)).astype(np.float32)


def live_state_dict(kw: dict, seed: int) -> dict:
    """State dict for ``SpikingFullSubNet(**kw)`` (key names: SURVEY 8b, probed against the reference)."""
    rng = np.random.default_rng(seed)
    sd: dict = {}
    shared, bn = kw.get("shared_weights", False), kw.get("bn", False)
    S = kw.get("num_spks", 1)
    _sequence_model(rng, "fb_model.", kw["fb_input_size"], kw["fb_hidden_size"], kw["fb_num_layers"], kw["fb_proj_size"],
                    shared, bn, kw.get("use_pre_layer_norm_fb", True), "proj", sd)
    for g, (c, n, d) in enumerate(zip(kw["center_freq_sizes"], kw["neighbor_freq_sizes"], kw["df_orders"])):
        _sequence_model(rng, f"sb_model.sb_models.{g}.", (c + 2 * n) + c, kw["sb_hidden_size"], kw["sb_num_layers"],
                        2 * c * d * S, shared, bn, kw.get("use_pre_layer_norm_sb", True), "proj", sd)
    return sd


def frozen_state_dict(kw: dict, seed: int) -> dict:
    """State dict for the frozen ``Separator(**kw)`` (no pre_layer_norm, ``fc_output_layer``)."""
    rng = np.random.default_rng(seed)
    sd: dict = {}
    shared, bn = kw.get("shared_weights", False), kw.get("bn", False)
    _sequence_model(rng, "fb_model.", kw["fb_freqs"], kw["fb_hidden_size"], 2, kw["fb_freqs"], shared, bn, False,
                    "fc_output_layer", sd)
    for g, (c, n, cf, nf, d) in enumerate(zip(kw["sb_num_center_freqs"], kw["sb_num_neighbor_freqs"],
                                              kw["fb_num_center_freqs"], kw["fb_num_neighbor_freqs"], kw["sb_df_orders"])):
        _sequence_model(rng, f"sb_model.sb_models.{g}.", (c + 2 * n) + (cf + 2 * nf), kw["sb_hidden_size"], 2, 2 * c * d,
                        shared, bn, False, "fc_output_layer", sd)
    return sd


# ---- the named configurations used by fixtures, tests and bench -------------------------------------
LIVE_M = dict(  # recipes/intel_ndns/spiking_fullsubnet/baseline_m.toml:33-56
    n_fft=512, hop_length=128, win_length=512, fdrc=0.5, fb_input_size=64, fb_hidden_size=320, fb_num_layers=2,
    fb_proj_size=64, fb_output_activate_function=False, sb_hidden_size=224, sb_num_layers=2,
    freq_cutoffs=[0, 32, 128, 256], df_orders=[5, 3, 1], center_freq_sizes=[4, 32, 64], neighbor_freq_sizes=[15, 15, 15],
    use_pre_layer_norm_fb=True, use_pre_layer_norm_sb=True, bn=True, shared_weights=True, sequence_model="GSN", num_spks=1,
)

LIVE_WSJ0 = dict(  # recipes/wsj0-mix/spiking_fullsubnet/default.toml [model.args]: 8 kHz, 256-point frames, two speakers
    LIVE_M, n_fft=256, hop_length=64, win_length=256, fb_input_size=32, fb_proj_size=32, freq_cutoffs=[0, 16, 64, 128],
    center_freq_sizes=[2, 16, 32], neighbor_freq_sizes=[7, 7, 7], num_spks=2)

LIVE_TINY = dict(LIVE_M, fb_hidden_size=48, sb_hidden_size=32, df_orders=[2, 3, 1])
LIVE_TINY_2SPK = dict(LIVE_TINY, num_spks=2, df_orders=[2, 1, 1])
LIVE_TINY_UNSHARED = dict(LIVE_TINY, shared_weights=False, bn=False, use_pre_layer_norm_sb=False)

FROZEN_S = dict(  # model_zoo/intel_ndns/spike_fsb/baseline_s/baseline_s.toml [model_g.args]
    sr=16000, fdrc=0.5, n_fft=512, fb_freqs=64, hop_length=128, win_length=512, num_freqs=256, sequence_model="GSU",
    fb_hidden_size=240, fb_output_activate_function=False, freq_cutoffs=[32, 128], sb_df_orders=[3, 1, 1],
    sb_num_center_freqs=[4, 32, 64], sb_num_neighbor_freqs=[15, 15, 15], fb_num_center_freqs=[4, 32, 64],
    fb_num_neighbor_freqs=[0, 0, 0], sb_hidden_size=160, sb_output_activate_function=False,
    norm_type="offline_laplace_norm", shared_weights=True, bn=True,
)

FROZEN_M = dict(FROZEN_S, fb_hidden_size=320, sb_hidden_size=224, sb_df_orders=[5, 3, 1])  # .../baseline_m/baseline_m.toml [model_g.args]
FROZEN_M_CUM = dict(FROZEN_M, norm_type="cumulative_laplace_norm")

FROZEN_L = dict(FROZEN_S, fb_hidden_size=320, sb_hidden_size=256, freq_cutoffs=[32, 128, 192], sb_df_orders=[5, 3, 1, 1],
                sb_num_center_freqs=[2, 4, 32, 64], sb_num_neighbor_freqs=[15, 15, 15, 15], fb_num_center_freqs=[2, 4, 32, 64],
                fb_num_neighbor_freqs=[0, 0, 0, 0])  # recipes/.../spiking_fullsubnet_freeze_phase/baseline_l.toml (offline norm)

FROZEN_XL = dict(FROZEN_M, shared_weights=False)  # .../spiking_fullsubnet_freeze_phase/baseline_xl.toml: separate gate weights

FROZEN_TINY = dict(FROZEN_S, fb_hidden_size=48, sb_hidden_size=32, sb_df_orders=[2, 1, 3])
FROZEN_TINY_GAUSS = dict(FROZEN_TINY, norm_type="offline_gaussian_norm")  # model_low_freq.py:205-218 (no recipe uses it)
FROZEN_TINY_CUM = dict(FROZEN_TINY, norm_type="cumulative_laplace_norm")  # recipes/.../baseline_m_cumulative_laplace_norm.toml's norm


def synth_wave(B: int, T: int, seed: int = 0, hop: int = 128, modulated: bool = False) -> np.ndarray:
    """0.05*randn waveform whose centred STFT has exactly T frames (SURVEY 8d synthetic inputs)."""
    rng = np.random.default_rng(seed)
    n = (T - 1) * hop
    x = (0.05 * rng.standard_normal((B, n))).astype(np.float32)
    if modulated:
        t = np.arange(n, dtype=np.float64) / 16000.0
        x = (x * (0.5 * (1 + np.sin(2 * np.pi * 3.0 * t)))[None, :]).astype(np.float32)
    return x
