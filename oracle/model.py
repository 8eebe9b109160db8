"""Whole-model composition of the CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, on numpy arrays and the C primitives of ``sfsn_oracle.c``, the two model front-ends of the
reference from the complex STFT to the enhanced spectrum:

* live   ``SpikingFullSubNet.forward``  audiozen/models/spiking_fullsubnet/modeling_spiking_fullsubnet.py:415-474
* frozen ``Separator.forward``          recipes/intel_ndns/spiking_fullsubnet_freeze_phase/model_low_freq.py:561-618

The STFT/iSTFT either side (audio_feature.py:236-347, plain ``torch.stft``/``torch.istft``) is the edge
of the path and is not restated: callers hand in the complex spectrum ``[B, 257, T]``.
"""
from __future__ import annotations

import numpy as np

from . import Oracle


def spec_from_live_kwargs(kw: dict) -> dict:
    """Normalise ``SpikingFullSubNet(**kw)`` (modeling_spiking_fullsubnet.py:350-373) to the oracle's spec."""
    if kw.get("sequence_model", "GSN") != "GSN":
        raise NotImplementedError("oracle restates the GSN sequence model only")
    n_groups = len(kw["center_freq_sizes"])
    return dict(
        front="live", n_fft=kw["n_fft"], fdrc=kw["fdrc"], fb_in=kw["fb_input_size"], fb_hidden=kw["fb_hidden_size"],
        fb_layers=kw["fb_num_layers"], fb_proj=kw["fb_proj_size"], sb_hidden=kw["sb_hidden_size"],
        sb_layers=kw["sb_num_layers"], cutoffs=list(kw["freq_cutoffs"]), ctr=list(kw["center_freq_sizes"]),
        nbr=list(kw["neighbor_freq_sizes"]), ctr_fb=list(kw["center_freq_sizes"]), nbr_fb=[0] * n_groups,
        df=list(kw["df_orders"]), num_spks=kw.get("num_spks", 1), shared=kw.get("shared_weights", False),
        bn=kw.get("bn", False), ln_fb=kw.get("use_pre_layer_norm_fb", True), ln_sb=kw.get("use_pre_layer_norm_sb", True),
        laplace=False, proj_name="proj",
    )


def spec_from_frozen_kwargs(kw: dict) -> dict:
    """Normalise ``Separator(**kw)`` (model_low_freq.py:486-509); interior cut points -> full cutoffs (:446-456)."""
    if kw["sequence_model"] != "GSU":
        raise NotImplementedError(f"Not implemented {kw['sequence_model']}")
    if kw["norm_type"] not in ("offline_laplace_norm", "cumulative_laplace_norm", "offline_gaussian_norm"):
        raise NotImplementedError("oracle restates offline_laplace_norm (the zoo checkpoints' setting), offline_gaussian_norm and cumulative_laplace_norm")
    return dict(
        front="frozen", n_fft=kw["n_fft"], fdrc=kw["fdrc"], fb_in=kw["fb_freqs"], fb_hidden=kw["fb_hidden_size"],
        fb_layers=2, fb_proj=kw["fb_freqs"], sb_hidden=kw["sb_hidden_size"], sb_layers=2,
        cutoffs=[0] + list(kw["freq_cutoffs"]) + [kw["num_freqs"]], ctr=list(kw["sb_num_center_freqs"]),
        nbr=list(kw["sb_num_neighbor_freqs"]), ctr_fb=list(kw["fb_num_center_freqs"]),
        nbr_fb=list(kw["fb_num_neighbor_freqs"]), df=list(kw["sb_df_orders"]), num_spks=1,
        shared=kw.get("shared_weights", False), bn=kw.get("bn", False), ln_fb=False, ln_sb=False,
        laplace=kw["norm_type"] == "offline_laplace_norm", gaussian=kw["norm_type"] == "offline_gaussian_norm",
        cum_laplace=kw["norm_type"] == "cumulative_laplace_norm",
        proj_name="fc_output_layer",
    )


def _sequence_model(o: Oracle, x, sd: dict, prefix: str, spec: dict, n_layers: int, use_ln: bool, want_membrane: bool):
    """SequenceModel.forward (modeling:81-125 / model_low_freq:100-139) on a time-major input x [T,R,I].

    Returns (proj [T,R,P], all_layer_outputs [x_norm, S1.., proj], membranes [per layer]).
    """
    if use_ln:
        x = o.layer_norm(x, sd[prefix + "pre_layer_norm.weight"], sd[prefix + "pre_layer_norm.bias"])
    outs = [x]
    mems = []
    cur = x
    for l in range(n_layers):
        p = f"{prefix}sequence_model.layers.{l}.cell."
        bn = None
        if spec["bn"]:
            bn = (sd[p + "batchnorm.weight"], sd[p + "batchnorm.bias"], sd[p + "batchnorm.running_mean"],
                  sd[p + "batchnorm.running_var"])
        cur, mem, _, _ = o.gsn_layer(cur, sd[p + "weight_ih"], sd[p + "weight_hh"], sd[p + "bias_ih"], bn=bn,
                                     shared=spec["shared"], want_membrane=want_membrane)
        outs.append(cur)
        mems.append(mem)
    pn = prefix + spec["proj_name"]
    proj = o.linear(cur, sd[pn + ".weight"], sd[pn + ".bias"])
    outs.append(proj)
    return proj, outs, mems


def forward_from_stft(spec: dict, sd: dict, stft, precision: str = "f32", want_membrane: bool = False) -> dict:
    """The hot path: complex noisy STFT [B, n_fft/2+1, T] -> dict(enh_stft [B,S,F,T], enh_mag, fb_all, sb_all, ...).

    ``sd`` is the reference state dict as numpy arrays (reference key names).
    """
    o = Oracle(precision)
    sd = {k: np.asarray(v) for k, v in sd.items()}
    stft = np.ascontiguousarray(stft, dtype=o.cdtype)
    B, F, T = stft.shape
    S = spec["num_spks"]
    mag = o.front_mag(stft, spec["fdrc"])  # [B, F-1, T]
    nf = F - 1
    res = {}
    # ---- full-band model (modeling:438-443 / model_low_freq:577-582)
    x_fb = o.gather_fullband(mag, spec["fb_in"])
    if spec["laplace"]:
        x_fb, res["mu_fb"] = o.laplace_norm(x_fb, B)
    elif spec.get("gaussian"):
        x_fb, res["mu_fb"] = o.gaussian_norm(x_fb, B)
    elif spec.get("cum_laplace"):
        x_fb = o.cum_laplace_norm(x_fb)
    fb_proj, fb_all, fb_mem = _sequence_model(o, x_fb, sd, "fb_model.", spec, spec["fb_layers"], spec["ln_fb"], want_membrane)
    # ---- sub-band models (modeling:216-263 / model_low_freq:433-482)
    cut = spec["cutoffs"]
    if cut[0] == 0 and cut[-1] == nf and len(cut) == 2:
        raise NotImplementedError("single-group models hit a latent reflect-pad quirk of the reference; not restated")
    sb_all, sb_mem, sb_proj = [], [], []
    for g in range(len(spec["ctr"])):
        x = o.gather_group(mag, fb_proj, cut[g], cut[g + 1], spec["ctr"][g], spec["nbr"][g], spec["ctr_fb"][g], spec["nbr_fb"][g])
        if spec["laplace"]:
            x, _ = o.laplace_norm(x, B)
        elif spec.get("gaussian"):
            x, _ = o.gaussian_norm(x, B)
        elif spec.get("cum_laplace"):
            x = o.cum_laplace_norm(x)
        proj, outs, mems = _sequence_model(o, x, sd, f"sb_model.sb_models.{g}.", spec, spec["sb_layers"], spec["ln_sb"], want_membrane)
        sb_all.append(outs)
        sb_mem.append(mems)
        sb_proj.append(proj)
    # ---- deep filtering + reconstruction (modeling:450-472 / model_low_freq:588-607)
    enh = np.zeros((B, S, F, T), o.cdtype)
    lo = 0
    for g, proj in enumerate(sb_proj):
        N = (cut[g + 1] - cut[g]) // spec["ctr"][g]
        o.deepfilter_group(stft, proj, enh, lo, N, spec["ctr"][g], spec["df"][g], S)
        lo += N * spec["ctr"][g]
    enh_mag = o.finish_spectrum(stft, enh, lo)
    res.update(enh_stft=enh, enh_mag=enh_mag, fb_all=fb_all, sb_all=sb_all, fb_mem=fb_mem, sb_mem=sb_mem, mag=mag)
    return res


def compute_synops(fb_all, sb_all, shared_weights=True) -> float:
    """audiozen/metric.py:303-327 on numpy lists (entries 1..-2 are the spike tensors)."""
    syn = 0.0
    lists = [fb_all] + list(sb_all)
    for outs in lists:
        for i in range(1, len(outs) - 1):
            syn += float(np.float32((outs[i] > 0).astype(np.float32).mean())) * outs[i].shape[-1] * (
                outs[i + 1].shape[-1] + outs[i].shape[-1])
    return syn if shared_weights else 2 * syn


def compute_neuronops(fb_all, sb_all) -> float:
    """audiozen/metric.py:330-340."""
    return float(sum(o.shape[-1] for o in fb_all) + sum(o.shape[-1] for outs in sb_all for o in outs))


# ---- the two edges of the path (audiozen/acoustics/audio_feature.py:236-347) ---------------------------------------
def hann_window(n: int) -> np.ndarray:
    """torch.hann_window(n) (periodic), float32 as the reference builds it (audio_feature.py:269,337)."""
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n)).astype(np.float32)


def stft(wave, n_fft: int = 512, hop: int = 128) -> np.ndarray:
    """audio_feature.py:269-279 = torch.stft(y, n_fft, hop, n_fft, hann, center=True, pad_mode="constant", return_complex=True):
    float [B, L] -> complex64 [B, n_fft/2+1, 1 + L // hop].  Frames in float64, rounded once."""
    wave = np.asarray(wave, np.float64)
    B, L = wave.shape
    T = 1 + L // hop
    pad = np.zeros((B, L + n_fft), np.float64)
    pad[:, n_fft // 2:n_fft // 2 + L] = wave
    win = hann_window(n_fft).astype(np.float64)
    idx = np.arange(T)[:, None] * hop + np.arange(n_fft)[None, :]
    frames = pad[:, idx] * win  # [B, T, n_fft]
    return np.fft.rfft(frames, axis=-1).transpose(0, 2, 1).astype(np.complex64)


def istft(spec, n_fft: int = 512, hop: int = 128, length=None) -> np.ndarray:
    """audio_feature.py:337-345 = torch.istft(X, n_fft, hop, n_fft, hann, length=length): inverse real transform per frame
    (imaginary parts of the DC / Nyquist bins ignored), synthesis window, overlap-add, division by the overlap-added
    squared window, n_fft/2 samples trimmed at the front, `length` samples kept."""
    spec = np.asarray(spec, np.complex128)
    B, F, T = spec.shape
    win = hann_window(n_fft).astype(np.float64)
    frames = np.fft.irfft(spec.transpose(0, 2, 1), n=n_fft, axis=-1) * win  # [B, T, n_fft]
    total = (T - 1) * hop + n_fft
    y = np.zeros((B, total), np.float64)
    env = np.zeros(total, np.float64)
    for t in range(T):
        y[:, t * hop:t * hop + n_fft] += frames[:, t]
        env[t * hop:t * hop + n_fft] += win * win
    if length is None:
        length = (T - 1) * hop
    sl = slice(n_fft // 2, n_fft // 2 + length)
    return (y[:, sl] / env[sl]).astype(np.float32)
