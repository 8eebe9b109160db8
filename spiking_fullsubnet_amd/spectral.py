"""STFT / inverse STFT of the models on the device kernels (``sfsn_stft`` / ``sfsn_istft``).

Same contract as ``audiozen.acoustics.audio_feature.stft(..., output_type="complex")`` / ``istft(..., input_type="complex")``
(audio_feature.py:236-347): Hann window of ``n_fft``, centred frames, zero padding, ``[B, F, T]`` complex64.  The kernels are
built for the 512-point analysis every reference config uses; any other ``n_fft`` raises ``NotImplementedError`` (the
modules then keep ``torch.stft`` / ``torch.istft``, see ``SpikingFullSubNet.stft``).
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from ._lib import check

_windows = {}


def hann(n_fft: int, device) -> torch.Tensor:
    key = (n_fft, str(device))
    if key not in _windows:
        _windows[key] = torch.hann_window(n_fft, device=device)  # audio_feature.py:269,337
    return _windows[key]


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def stft(y: torch.Tensor, n_fft: int, hop_length: int, win_length: Optional[int] = None, window: Optional[torch.Tensor] = None) -> torch.Tensor:
    """float32 [B, L] on the device -> complex64 [B, n_fft/2+1, 1 + L // hop]."""
    if y.dim() != 2 or y.dtype != torch.float32 or y.device.type != "cuda":
        raise RuntimeError(f"expected a float32 [B, L] tensor on a HIP device, got {y.dtype} {tuple(y.shape)} on {y.device}")
    if win_length not in (None, n_fft):
        raise NotImplementedError("win_length != n_fft")
    B, L = y.shape
    T = 1 + L // hop_length
    w = hann(n_fft, y.device) if window is None else window.to(device=y.device, dtype=torch.float32).contiguous()
    out = torch.empty((B, n_fft // 2 + 1, T), dtype=torch.complex64, device=y.device)
    y = y.contiguous()
    with torch.cuda.device(y.device):  # the C ABI launches on the calling thread's current device
        check(_lib.lib().sfsn_stft(y.data_ptr(), B, L, n_fft, hop_length, w.data_ptr(), torch.view_as_real(out).data_ptr(), T, _stream(y.device)),
              "sfsn_stft")
    return out


def istft(spec: torch.Tensor, n_fft: int, hop_length: int, win_length: Optional[int] = None, window: Optional[torch.Tensor] = None,
          length: Optional[int] = None) -> torch.Tensor:
    """complex64 [B, n_fft/2+1, T] on the device -> float32 [B, length] (default length: (T - 1) * hop, as torch.istft)."""
    if spec.dim() != 3 or spec.dtype != torch.complex64 or spec.device.type != "cuda":
        raise RuntimeError(f"expected a complex64 [B, F, T] tensor on a HIP device, got {spec.dtype} {tuple(spec.shape)} on {spec.device}")
    if win_length not in (None, n_fft):
        raise NotImplementedError("win_length != n_fft")
    B, F, T = spec.shape
    if F != n_fft // 2 + 1:
        raise ValueError(f"expected {n_fft // 2 + 1} frequency bins, got {F}")
    if length is None:
        length = (T - 1) * hop_length
    w = hann(n_fft, spec.device) if window is None else window.to(device=spec.device, dtype=torch.float32).contiguous()
    spec = spec.contiguous()
    out = torch.empty((B, length), dtype=torch.float32, device=spec.device)
    with torch.cuda.device(spec.device):
        check(_lib.lib().sfsn_istft(torch.view_as_real(spec).data_ptr(), B, T, n_fft, hop_length, w.data_ptr(), out.data_ptr(), length,
                                    _stream(spec.device)), "sfsn_istft")
    return out
