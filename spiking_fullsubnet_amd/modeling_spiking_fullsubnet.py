"""Drop-in for ``audiozen.models.spiking_fullsubnet.modeling_spiking_fullsubnet.SpikingFullSubNet``.

Same constructor keywords (modeling_spiking_fullsubnet.py:350-373), same parameter / buffer names (so
``accelerator.load_state`` / ``load_state_dict(strict=True)`` of a reference checkpoint works),
same ``forward(input[B, samples])`` return tuple (:415-474).  A recipe switches over by changing only

    [model]
    path = "spiking_fullsubnet_amd.modeling_spiking_fullsubnet.SpikingFullSubNet"

The sub-modules below are parameter containers that mirror the reference's module tree; in ``eval()`` mode all arithmetic
between ``stft`` and ``istft`` runs in the gfx950 inference kernels of ``libsfsn_hip.so`` via ``Engine`` (no autograd graph).
In ``train()`` mode -- or when gradients can flow into the input -- ``forward()`` takes the differentiable path of ``training.py``:
the cell loop with per-step batch-statistics BatchNorm and the triangle surrogate's backward on the HIP training kernels
(``csrc/sfsn_train.hip``), everything time-parallel as ATen operations.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
from torch.nn import Parameter

from .engine import Engine, PathSpec


# ---- parameter containers mirroring efficient_spiking_neuron.py:43-130 ---------------------------------
class GSUCell(nn.Module):
    """Parameters of one gated spiking cell (efficient_spiking_neuron.py:104-130), reference initialisation."""

    def __init__(self, input_size, hidden_size, shared_weights=False, bn=False):
        super().__init__()
        self.input_size, self.hidden_size, self.shared_weights, self.use_bn = input_size, hidden_size, shared_weights, bn
        rows = hidden_size if shared_weights else 2 * hidden_size
        self.weight_ih = Parameter(torch.empty(rows, input_size))
        self.weight_hh = Parameter(torch.empty(rows, hidden_size))
        self.bias_ih = Parameter(torch.zeros(2 * hidden_size))
        stdv = 1.0 / math.sqrt(hidden_size) if hidden_size > 0 else 0
        for weight in self.parameters():
            torch.nn.init.uniform_(weight, -stdv, stdv)
        if bn:
            self.batchnorm = nn.BatchNorm1d(hidden_size)


class GSULayer(nn.Module):
    def __init__(self, *cell_args):
        super().__init__()
        self.cell = GSUCell(*cell_args)


class StackedGSU(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers, shared_weights, bn):
        super().__init__()
        self.layers = nn.ModuleList([GSULayer(input_size if l == 0 else hidden_size, hidden_size, shared_weights, bn)
                                     for l in range(num_layers)])


class SequenceModel(nn.Module):
    """Container for modeling_spiking_fullsubnet.py:12-79 (pre_layer_norm, sequence_model, proj)."""

    def __init__(self, input_size, hidden_size, num_layers, sequence_model="GSN", proj_size=0, shared_weights=False,
                 output_activate_function=None, bn=False, use_pre_layer_norm=True):
        super().__init__()
        if use_pre_layer_norm:
            self.pre_layer_norm = nn.LayerNorm(input_size)
        if sequence_model == "GSN":
            self.sequence_model = StackedGSU(input_size, hidden_size, num_layers, shared_weights, bn)
        elif sequence_model == "LSTM":  # the reference's nn.LSTM ablation (:38-45): served by the ATen path (training.py), not by the kernels
            self.sequence_model = nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers, batch_first=True,
                                          bidirectional=False)
        else:
            raise NotImplementedError(f"Sequence model {sequence_model} not implemented.")
        self.proj = nn.Linear(hidden_size, proj_size) if proj_size > 0 else nn.Identity()
        self.output_activate_function = {"tanh": nn.Tanh, "sigmoid": nn.Sigmoid, "relu": nn.ReLU}.get(output_activate_function, nn.Identity)()
        # what the inference kernels cover: GSN cells, a Linear projection, Identity output (every reference config); the rest
        # runs on the ATen path in eval mode too (SURVEY 8b's torch fallback)
        self.kernel_path = sequence_model == "GSN" and proj_size > 0 and isinstance(self.output_activate_function, nn.Identity)
        self.hidden_size, self.num_layers = hidden_size, num_layers
        self.use_pre_layer_norm, self.sequence_model_name = use_pre_layer_norm, sequence_model


class SubBandSequenceModel(SequenceModel):
    def __init__(self, df_order, num_spks, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.df_order, self.num_spks = df_order, num_spks


class SubbandModel(nn.Module):
    """Container for modeling_spiking_fullsubnet.py:172-214."""

    def __init__(self, freq_cutoffs, center_freq_sizes, neighbor_freq_sizes, df_orders, num_spks, **kwargs):
        super().__init__()
        assert len(freq_cutoffs) - 1 == len(center_freq_sizes), "Number of subbands must be equal to len(cutoffs)."
        self.sb_models = nn.ModuleList([
            SubBandSequenceModel(input_size=(c + n * 2) + c, proj_size=2 * c * d * num_spks, df_order=d, num_spks=num_spks, **kwargs)
            for c, n, d in zip(center_freq_sizes, neighbor_freq_sizes, df_orders)])
        self.freq_cutoffs, self.center_freq_sizes = freq_cutoffs, center_freq_sizes
        self.neighbor_freq_sizes, self.df_orders = neighbor_freq_sizes, df_orders


class _EngineMixin:
    """Lazy (re)packing of the module's weights for the kernels, keyed on parameter versions and device."""

    _engine = None
    _engine_key = None
    # 24 = the exact fp32-parity mode; 16 = the 16-bit-weight fast mode (recurrent / spike-input / projection weights rounded to 16
    # significant bits of their row grid): set `module.weight_bits = 16` before the first forward (or any time: the weights are re-packed)
    weight_bits = 24
    # What forward() returns in place of the reference's fp32 spike tensors (entries 1..L of every all_layer_outputs list):
    #   "tensors" (default, the reference's API), "counts" (SpikeSummary: exact spike counts + shape -- all that
    #   metric.compute_synops / compute_neuronops read), "none" (None entries).
    layer_outputs = "tensors"

    # The two edges of the path (audio_feature.py:236-347).  "device": the package's own STFT / inverse-STFT kernels (n_fft = 512,
    # win_length = n_fft, 1..4 hops per window: every reference config); "torch": torch.stft / torch.istft (rocFFT).  With
    # "device" a configuration the kernels do not cover falls back to torch -- that is the documented edge, not a CPU path.
    spectral_backend = "device"
    # eval-mode forward() with gradients to the PARAMETERS (grad mode on): off by default -- parameters require grad by default, and an
    # inference call that merely forgot torch.no_grad() should not leave the kernels; an input that requires grad always does
    autograd_in_eval = False

    def _device_fft(self, t: torch.Tensor) -> bool:
        n_fft, hop = self.n_fft, self.hop_length
        return (self.spectral_backend == "device" and t.device.type == "cuda" and n_fft == 512 and self.win_length == n_fft
                and n_fft % hop == 0 and n_fft // hop <= 4)

    def _stft(self, y: torch.Tensor) -> torch.Tensor:
        if self._device_fft(y) and y.dtype == torch.float32:
            from . import spectral
            return spectral.stft(y, self.n_fft, self.hop_length)
        window = torch.hann_window(self.n_fft, device=y.device)
        return torch.stft(y, self.n_fft, self.hop_length, self.win_length, window=window, return_complex=True, pad_mode="constant")

    def _istft(self, spec: torch.Tensor, length=None) -> torch.Tensor:
        if self._device_fft(spec) and spec.dtype == torch.complex64 and (length is None or 0 < length <= (spec.shape[-1] - 1) * self.hop_length
                                                                         + self.n_fft // 2):
            from . import spectral
            return spectral.istft(spec, self.n_fft, self.hop_length, length=length)
        window = torch.hann_window(self.n_fft, device=spec.device)
        return torch.istft(spec, self.n_fft, self.hop_length, self.win_length, window=window, length=length)

    # ---- the public attributes the reference modules carry: `self.stft = partial(audio_feature.stft, n_fft, hop, win)` and
    #      `self.istft = partial(audio_feature.istft, ...)` (MODEL:404-405, FROZEN:547-558) -- same call signatures and return
    #      values (audio_feature.py:236-347), computed by the device kernels; forward() uses the complex fast path directly
    def stft(self, y, output_type=None, **kwargs):
        """(mag, phase, real, imag) [B, F, T] -- or, by output_type, (mag, phase) / (real, imag) / the complex tensor."""
        if kwargs:
            raise NotImplementedError(f"extra torch.stft arguments are not supported here: {sorted(kwargs)}")
        if y.ndim not in (2, 3):
            raise ValueError(f"Only support single-/multi-channel signals. Received {y.ndim=}.")  # audio_feature.py:259-260
        shape = y.shape
        c = self._stft(y.reshape(-1, shape[-1]) if y.ndim == 3 else y)
        if y.ndim == 3:
            c = c.reshape(shape[0], -1, *c.shape[-2:])
        if output_type == "complex":
            return c
        if output_type == "real_imag":
            return c.real, c.imag
        mag, phase = torch.abs(c), torch.angle(c)
        if output_type == "mag_phase":
            return mag, phase
        return mag, phase, c.real, c.imag

    def istft(self, feature, length=None, input_type="complex"):
        """[B, F, T] spectrogram(s) -> [B, length] (audio_feature.py:297-347: "complex" tensor, ("real", "imag") or (mag, phase))."""
        if input_type == "real_imag":
            if not (isinstance(feature, (tuple, list)) and len(feature) == 2):
                raise ValueError(f"Only support tuple or list. Received {type(feature)}.")
            c = torch.complex(real=feature[0], imag=feature[1])
        elif input_type == "complex":
            if not (torch.is_tensor(feature) and torch.is_complex(feature)):
                raise ValueError(f"Only support complex-valued tensor. Received {type(feature)}")
            c = feature
        elif input_type == "mag_phase":
            if not (isinstance(feature, (tuple, list)) and len(feature) == 2):
                raise ValueError(f"Only support tuple or list. Received {type(feature)}.")
            c = torch.polar(feature[0], feature[1])
        else:
            raise ValueError(f"Only support 'real_imag', 'complex', and 'mag_phase'. Received {input_type=}")
        return self._istft(c, length)

    def _layer_kwargs(self) -> dict:
        if self.layer_outputs not in ("tensors", "counts", "none"):
            raise ValueError(f"layer_outputs must be 'tensors', 'counts' or 'none', got {self.layer_outputs!r}")
        return dict(want_layers=self.layer_outputs == "tensors", want_counts=self.layer_outputs == "counts")

    def _spec(self) -> PathSpec:  # pragma: no cover - provided by subclasses
        raise NotImplementedError

    def engine(self) -> Engine:
        tensors = list(self.state_dict(keep_vars=True).items())
        dev = tensors[0][1].device
        key = (str(dev), self.weight_bits) + tuple((k, t.data_ptr(), t._version) for k, t in tensors)
        if self._engine is None or self._engine_key != key:
            if dev.type != "cuda":
                raise RuntimeError("spiking_fullsubnet_amd has no CPU path: move the module to a HIP device (`.to('cuda')`) first")
            sd = {k: t.detach().cpu().numpy() for k, t in tensors}
            self._engine = Engine(self._spec(), sd, dev, weight_bits=self.weight_bits)
            self._engine_key = key
        return self._engine

    def streaming(self, batch: int = 1, hop: int = 1, graph: bool = True, rows_per_wg=None, one_launch="auto", waveform: bool = False,
                  host_io: bool = False, resident: bool = False, idle_ms: int = 1000):
        """Frame-by-frame session (``streaming.StreamingSession``): state and deep-filter history stay on the device between
        calls; one launch per hop (``sfsn_stream_hop``) where the library covers the model, else the offline kernels replayed
        from a HIP graph.  ``waveform=True``: samples in, samples out (``step_wave``).  Live front-end only."""
        from .streaming import StreamingSession
        self._check_mode()
        return StreamingSession(self.engine(), batch=batch, hop=hop, graph=graph, rows_per_wg=rows_per_wg, owner=self, one_launch=one_launch,
                                waveform=waveform, host_io=host_io, resident=resident, idle_ms=idle_ms)

    def _check_mode(self, x=None):
        """The inference kernels have no autograd graph and no training-mode BatchNorm: the entry points that use them
        (forward_stft, streaming sessions) refuse a module in training mode or an input that requires grad.  forward() itself
        routes such calls to the differentiable path (training.py)."""
        if x is not None and x.requires_grad:
            raise RuntimeError("the input requires grad: the inference kernels have no backward pass -- call the module itself "
                               "(forward() takes the differentiable path of training.py), or detach the input")
        if self.training:
            raise RuntimeError("the module is in training mode: the inference kernels evaluate BatchNorm with the running statistics -- "
                               "call the module itself (forward() takes the training path of training.py), or .eval()")

    def _wants_autograd(self, x) -> bool:
        """forward() takes the differentiable ATen + training-step-kernel path when the module is in training mode, when gradients
        can flow (grad mode on and the input or a parameter requires grad ... the recipes' validation runs under no_grad and stays on
        the inference kernels), or when the constructor options are ones the inference kernels do not cover."""
        if self.training:
            return True
        if torch.is_grad_enabled() and (x.requires_grad or self.autograd_in_eval):
            return True  # (parameters always "require grad": an eval-mode call outside no_grad stays on the kernels unless asked)
        return not self._kernel_path()


class SpikingFullSubNet(_EngineMixin, nn.Module):
    def __init__(self, n_fft, hop_length, win_length, fdrc, fb_input_size, fb_hidden_size, fb_num_layers, fb_proj_size,
                 fb_output_activate_function, sb_hidden_size, sb_num_layers, freq_cutoffs, df_orders, center_freq_sizes,
                 neighbor_freq_sizes, use_pre_layer_norm_fb=True, use_pre_layer_norm_sb=True, bn=False, shared_weights=False,
                 sequence_model="GSN", num_spks=1):
        super().__init__()
        self.fb_model = SequenceModel(input_size=fb_input_size, hidden_size=fb_hidden_size, num_layers=fb_num_layers,
                                      shared_weights=shared_weights, sequence_model=sequence_model, proj_size=fb_proj_size,
                                      output_activate_function=fb_output_activate_function, bn=bn,
                                      use_pre_layer_norm=use_pre_layer_norm_fb)
        self.sb_model = SubbandModel(freq_cutoffs=freq_cutoffs, center_freq_sizes=center_freq_sizes,
                                     neighbor_freq_sizes=neighbor_freq_sizes, df_orders=df_orders, num_spks=num_spks,
                                     hidden_size=sb_hidden_size, num_layers=sb_num_layers, shared_weights=shared_weights,
                                     sequence_model=sequence_model, bn=bn, use_pre_layer_norm=use_pre_layer_norm_sb)
        self.subband_model = None
        self.fb_input_size, self.n_fft, self.hop_length, self.win_length = fb_input_size, n_fft, hop_length, win_length
        self.fdrc, self.df_orders, self.num_spks = fdrc, df_orders, num_spks
        self._path_spec = PathSpec(
            front="live", n_fft=n_fft, fdrc=fdrc, fb_in=fb_input_size, fb_hidden=fb_hidden_size, fb_layers=fb_num_layers,
            fb_proj=fb_proj_size, sb_hidden=sb_hidden_size, sb_layers=sb_num_layers, cutoffs=list(freq_cutoffs),
            ctr=list(center_freq_sizes), nbr=list(neighbor_freq_sizes), ctr_fb=list(center_freq_sizes),
            nbr_fb=[0] * len(center_freq_sizes), df=list(df_orders), num_spks=num_spks, shared=shared_weights, bn=bn,
            ln_fb=use_pre_layer_norm_fb, ln_sb=use_pre_layer_norm_sb, laplace=False, proj_name="proj")
        if fb_proj_size != fb_input_size or (n_fft // 2) % fb_input_size != 0:
            # the reference tiles the full-band output (n_fft//2+1)//fb_input_size times to cover the spectrum (:442-443)
            raise NotImplementedError("fb_proj_size must equal fb_input_size and divide n_fft/2 (as in every reference config)")

    def _spec(self) -> PathSpec:
        return self._path_spec

    @torch.no_grad()
    def forward_stft(self, noisy_cmp, want_layers=True, want_membrane=False, want_counts=False):
        """The hot path alone: complex64 [B, 257, T] -> Engine.forward_stft result dict."""
        self._check_mode(noisy_cmp)
        return self.engine().forward_stft(noisy_cmp, want_layers=want_layers, want_membrane=want_membrane, want_counts=want_counts)

    def _kernel_path(self) -> bool:
        return self.fb_model.kernel_path and all(s.kernel_path for s in self.sb_model.sb_models)

    def forward(self, input):
        assert input.ndim == 2, f"Input tensor must be 2D, but got {input.ndim}D."
        if self._wants_autograd(input):
            from . import training
            return training.forward_live(self, input)
        with torch.no_grad():
            return self._forward_inference(input)

    def _forward_inference(self, input):
        batch_size, sequence_length = input.shape
        res = self.engine().forward_stft(self._stft(input), **self._layer_kwargs())
        enh_stft = res["enh_stft"]  # [B, S, F, T]
        if self.num_spks > 1:
            enh_y = self._istft(enh_stft.reshape(batch_size * self.num_spks, *enh_stft.shape[2:]), length=sequence_length)
            return enh_y.reshape(batch_size, self.num_spks, -1), res["fb_all"], res["sb_all"]
        enh_y = self._istft(enh_stft[:, 0], length=sequence_length)
        return enh_y, res["enh_mag"][:, 0], res["fb_all"], res["sb_all"]
