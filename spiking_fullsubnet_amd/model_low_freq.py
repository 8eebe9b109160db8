"""Drop-in for the frozen competition model ``model_low_freq.Separator``
(recipes/intel_ndns/spiking_fullsubnet_freeze_phase/model_low_freq.py:485-618) -- the architecture of
``baseline_s`` and of every ``model_zoo`` checkpoint.

Same constructor keywords (:486-509), same state-dict names (``fc_output_layer`` instead of ``proj``, no
``pre_layer_norm``; the zoo checkpoints load with ``strict=True``), same ``forward`` return tuple (:618).
A frozen-recipe TOML switches over with ``[model_g] path = "spiking_fullsubnet_amd.model_low_freq.Separator"``.

Input normalisation: the utterance-level ``offline_laplace_norm`` (:147-169, the setting of all zoo checkpoints) or
``offline_gaussian_norm`` (:205-218: (x - mean) / (std + eps) per clip, no recipe uses it), or the causal
``cumulative_laplace_norm`` of ``baseline_m_cumulative_laplace_norm.toml`` -- in the reference that one raises on the 5-D sub-band
tensor (:172-202 unpacks four dimensions); it is built here in the form of ``model_low_freq_count_time.py:182-204`` (every row by its
own running mean), which also makes this front-end streamable (``streaming()``).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .engine import PathSpec
from .modeling_spiking_fullsubnet import StackedGSU, _EngineMixin


class SequenceModel(nn.Module):
    """Container for model_low_freq.py:42-98 (sequence_model, fc_output_layer)."""

    def __init__(self, input_size, output_size, hidden_size, num_layers, bidirectional, sequence_model="GSU",
                 output_activate_function="Tanh", num_groups=4, mogrify_steps=5, dropout=0.0, shared_weights=False, bn=False):
        super().__init__()
        if sequence_model != "GSU":
            raise NotImplementedError(f"Not implemented {sequence_model}")
        if bidirectional:
            raise NotImplementedError("bidirectional GSU is not used by the reference (model_low_freq.py:524,332 pass False)")
        self.sequence_model = StackedGSU(input_size, hidden_size, num_layers, shared_weights, bn)
        if not int(output_size):
            raise NotImplementedError("output_size = 0 is not used by any reference config")
        self.fc_output_layer = nn.Linear(hidden_size, output_size)
        if output_activate_function:
            raise NotImplementedError("output activations are not fused yet; every zoo config uses `false`")
        self.output_activate_function, self.output_size = output_activate_function, output_size
        self.sequence_model_name, self.hidden_size, self.num_layers = sequence_model, hidden_size, num_layers


class SubBandSequenceWrapper(SequenceModel):
    def __init__(self, df_order, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.df_order = df_order


class SubbandModel(nn.Module):
    """Container for model_low_freq.py:293-348."""

    def __init__(self, freq_cutoffs, sb_num_center_freqs, sb_num_neighbor_freqs, fb_num_center_freqs, fb_num_neighbor_freqs,
                 sb_df_orders, sequence_model, hidden_size, activate_function=False, norm_type="offline_laplace_norm",
                 shared_weights=False, bn=False):
        super().__init__()
        self.sb_models = nn.ModuleList([
            SubBandSequenceWrapper(df_order=d, input_size=(c + n * 2) + (cf + nf * 2), output_size=c * 2 * d, hidden_size=hidden_size,
                                   num_layers=2, sequence_model=sequence_model, bidirectional=False,
                                   output_activate_function=activate_function, shared_weights=shared_weights, bn=bn)
            for c, n, cf, nf, d in zip(sb_num_center_freqs, sb_num_neighbor_freqs, fb_num_center_freqs, fb_num_neighbor_freqs, sb_df_orders)])
        self.freq_cutoffs = freq_cutoffs
        self.sb_num_center_freqs, self.sb_num_neighbor_freqs = sb_num_center_freqs, sb_num_neighbor_freqs
        self.fb_num_center_freqs, self.fb_num_neighbor_freqs = fb_num_center_freqs, fb_num_neighbor_freqs


class Separator(_EngineMixin, nn.Module):
    def __init__(self, sr, n_fft, hop_length, win_length, fdrc, num_freqs, fb_freqs, freq_cutoffs, sb_num_center_freqs,
                 sb_num_neighbor_freqs, fb_num_center_freqs, fb_num_neighbor_freqs, fb_hidden_size, sb_hidden_size, sb_df_orders,
                 sequence_model, fb_output_activate_function, sb_output_activate_function, norm_type, shared_weights=False, bn=False):
        super().__init__()
        if norm_type not in ("offline_laplace_norm", "cumulative_laplace_norm", "offline_gaussian_norm"):
            # (the reference: model_low_freq.py:216-231 norm_wrapper raises NotImplementedError for anything else, with this message)
            raise NotImplementedError("You must set up a type of Norm. e.g. offline_laplace_norm, cumulative_laplace_norm, forgetting_norm, etc.")
        self.n_fft, self.hop_length, self.win_length, self.fdrc = n_fft, hop_length, win_length, fdrc
        self.freq_cutoffs, self.sb_df_orders = freq_cutoffs, sb_df_orders
        self.num_repeats, self.fb_freqs = num_freqs // fb_freqs, fb_freqs
        self.fb_model = SequenceModel(input_size=fb_freqs, output_size=fb_freqs, hidden_size=fb_hidden_size, num_layers=2,
                                      bidirectional=False, sequence_model=sequence_model,
                                      output_activate_function=fb_output_activate_function, shared_weights=shared_weights, bn=bn)
        self.sb_model = SubbandModel(freq_cutoffs=freq_cutoffs, sb_num_center_freqs=sb_num_center_freqs,
                                     sb_num_neighbor_freqs=sb_num_neighbor_freqs, fb_num_center_freqs=fb_num_center_freqs,
                                     fb_num_neighbor_freqs=fb_num_neighbor_freqs, sb_df_orders=sb_df_orders,
                                     hidden_size=sb_hidden_size, sequence_model=sequence_model,
                                     activate_function=sb_output_activate_function, shared_weights=shared_weights, bn=bn,
                                     norm_type=norm_type)
        if num_freqs != n_fft // 2 or num_freqs % fb_freqs != 0:
            raise NotImplementedError("num_freqs must be n_fft/2 and a multiple of fb_freqs (as in every zoo config)")
        self._path_spec = PathSpec(
            front="frozen", n_fft=n_fft, fdrc=fdrc, fb_in=fb_freqs, fb_hidden=fb_hidden_size, fb_layers=2, fb_proj=fb_freqs,
            sb_hidden=sb_hidden_size, sb_layers=2, cutoffs=[0] + list(freq_cutoffs) + [num_freqs], ctr=list(sb_num_center_freqs),
            nbr=list(sb_num_neighbor_freqs), ctr_fb=list(fb_num_center_freqs), nbr_fb=list(fb_num_neighbor_freqs),
            df=list(sb_df_orders), num_spks=1, shared=shared_weights, bn=bn, ln_fb=False, ln_sb=False,
            laplace=norm_type in ("offline_laplace_norm", "offline_gaussian_norm"), gaussian=norm_type == "offline_gaussian_norm",
            cum_laplace=norm_type == "cumulative_laplace_norm", proj_name="fc_output_layer")

    def _spec(self) -> PathSpec:
        return self._path_spec

    @torch.no_grad()
    def forward_stft(self, complex_stft, want_layers=True, want_membrane=False, want_counts=False):
        self._check_mode(complex_stft)
        return self.engine().forward_stft(complex_stft, want_layers=want_layers, want_membrane=want_membrane, want_counts=want_counts)

    def _kernel_path(self) -> bool:
        return True  # (every constructor option that is accepted is served by the kernels)

    def forward(self, noisy_y):
        """model_low_freq.py:561-618.  In training mode, or when gradients can flow into the input, the differentiable path of
        training.py (``forward_frozen``: ATen front / back end, HIP training kernels for the cell loop); else the kernels."""
        if self._wants_autograd(noisy_y):
            from . import training
            return training.forward_frozen(self, noisy_y)
        with torch.no_grad():
            return self._forward_inference(noisy_y)

    def _forward_inference(self, noisy_y):
        ndim = noisy_y.dim()
        assert ndim in (2, 3), "Input must be 2D (B, T) or 3D tensor (B, 1, T)"
        if ndim == 3:
            assert noisy_y.size(1) == 1, "Input must be 2D (B, T) or 3D tensor (B, 1, T)"
            noisy_y = noisy_y.squeeze(1)
        self._check_mode(noisy_y)
        res = self.engine().forward_stft(self._stft(noisy_y), **self._layer_kwargs())
        enhanced_y = self._istft(res["enh_stft"][:, 0], length=noisy_y.size(-1))
        return enhanced_y, res["enh_mag"][:, 0], res["fb_all"], res["sb_all"]
