"""CPU oracle for the Spiking-FullSubNet hot path -- TEST INFRASTRUCTURE ONLY.

ctypes bindings over ``oracle/libsfsn_oracle.so`` (built from ``oracle/sfsn_oracle.c`` by
``oracle/Makefile``; see that file's header for the parity status and the reference citations) plus
``oracle.model``: the whole-model composition (live ``SpikingFullSubNet`` and frozen ``Separator``
front-ends) on numpy arrays.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package -- as the checker or as the timed CPU baseline, never as a compute path of the product.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsfsn_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (``make -C oracle``). Returns the .so path."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
        os.path.join(_HERE, "sfsn_oracle.c")
    ):
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


_P = ctypes.c_void_p
_I = ctypes.c_int
_D = ctypes.c_double


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_P)


class Oracle:
    """One precision of the oracle: ``Oracle("f32")`` (the oracle proper) or ``Oracle("f64")`` (noise floor)."""

    def __init__(self, precision: str = "f32"):
        assert precision in ("f32", "f64")
        self.dtype = np.float32 if precision == "f32" else np.float64
        self.cdtype = np.complex64 if precision == "f32" else np.complex128
        self._pfx = f"sfsn_oracle_{precision}_"
        self._lib = lib()

    def _fn(self, name, argtypes, restype=None):
        f = getattr(self._lib, self._pfx + name)
        f.argtypes = argtypes
        f.restype = restype
        return f

    def arr(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    # -- primitives ---------------------------------------------------------------------------
    def gsn_layer(self, x, w_ih, w_hh, bias, bn=None, shared=True, h0=None, c0=None, eps=1e-5, want_membrane=True):
        """x [T,R,I] -> (spikes [T,R,H], membrane [T,R,H] | None, hT [R,H], cT [R,H]).

        bn = (weight, bias, running_mean, running_var) or None.
        """
        x = self.arr(x)
        T, R, I = x.shape
        H = bias.shape[0] // 2
        w_ih, w_hh, bias = self.arr(w_ih), self.arr(w_hh), self.arr(bias)
        G = 1 if shared else 2
        assert w_ih.shape == (G * H, I) and w_hh.shape == (G * H, H), (w_ih.shape, w_hh.shape, G, H, I)
        h = np.zeros((R, H), self.dtype) if h0 is None else self.arr(h0).copy()
        c = np.zeros((R, H), self.dtype) if c0 is None else self.arr(c0).copy()
        spikes = np.empty((T, R, H), self.dtype)
        mem = np.empty((T, R, H), self.dtype) if want_membrane else None
        bnp = [self.arr(b) for b in bn] if bn is not None else [None] * 4
        f = self._fn("gsn_layer", [_P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _D, _P, _P, _P, _P])
        f(_ptr(x), T, R, I, H, int(bool(shared)), _ptr(w_ih), _ptr(w_hh), _ptr(bias), int(bn is not None),
          _ptr(bnp[0]), _ptr(bnp[1]), _ptr(bnp[2]), _ptr(bnp[3]), float(eps), _ptr(h), _ptr(c), _ptr(spikes), _ptr(mem))
        return spikes, mem, h, c

    def linear(self, x, w, b=None):
        x = self.arr(x)
        w = self.arr(w)
        lead = x.shape[:-1]
        K = x.shape[-1]
        N = w.shape[0]
        M = int(np.prod(lead))
        y = np.empty((M, N), self.dtype)
        b = None if b is None else self.arr(b)
        self._fn("linear", [_P, _I, _I, _I, _P, _P, _P])(_ptr(x), M, K, N, _ptr(w), _ptr(b), _ptr(y))
        return y.reshape(*lead, N)

    def layer_norm(self, x, g, b, eps=1e-5):
        x = self.arr(x).copy()
        I = x.shape[-1]
        self._fn("layer_norm", [_P, _I, _I, _P, _P, _D])(_ptr(x), x.size // I, I, _ptr(self.arr(g)), _ptr(self.arr(b)), eps)
        return x

    def front_mag(self, stft, fdrc=0.5):
        """complex [B,F,T] -> compressed magnitude [B,F-1,T]."""
        stft = np.ascontiguousarray(stft, dtype=self.cdtype)
        B, F, T = stft.shape
        mag = np.empty((B, F - 1, T), self.dtype)
        self._fn("front_mag", [_P, _I, _I, _I, _D, _P])(_ptr(stft), B, F, T, float(fdrc), _ptr(mag))
        return mag

    def gather_fullband(self, mag, FB):
        mag = self.arr(mag)
        B, nf, T = mag.shape
        x = np.empty((T, B, FB), self.dtype)
        self._fn("gather_fullband", [_P, _I, _I, _I, _I, _P])(_ptr(mag), B, nf, T, FB, _ptr(x))
        return x

    def gather_group(self, mag, fb_tbf, lo, hi, ctr, nbr, ctr_fb=None, nbr_fb=0):
        mag = self.arr(mag)
        fb_tbf = self.arr(fb_tbf)
        B, nf, T = mag.shape
        FB = fb_tbf.shape[-1]
        ctr_fb = ctr if ctr_fb is None else ctr_fb
        N = (hi - lo) // ctr
        I = ctr + 2 * nbr + ctr_fb + 2 * nbr_fb
        x = np.empty((T, B * N, I), self.dtype)
        rc = self._fn("gather_group", [_P, _P] + [_I] * 10 + [_P], _I)(
            _ptr(mag), _ptr(fb_tbf), B, nf, T, FB, lo, hi, ctr, nbr, ctr_fb, nbr_fb, _ptr(x))
        if rc != 0:
            raise ValueError(f"Number of frequency bins must be divisible by the center frequency. {ctr=}, {hi=}, {lo=}")
        return x

    def laplace_norm(self, x, B):
        """x [T, B*N, I] -> normalised copy, mu [B]."""
        x = self.arr(x).copy()
        T, BN, I = x.shape
        mu = np.empty((B,), self.dtype)
        self._fn("laplace_norm", [_P, _I, _I, _I, _I, _P])(_ptr(x), T, B, BN // B, I, _ptr(mu))
        return x, mu

    def gaussian_norm(self, x, B):
        """x [T, B*N, I] -> (x - mean) / (std + eps) per clip (offline_gaussian_norm), mu [B]."""
        x = self.arr(x).copy()
        T, BN, I = x.shape
        mu = np.empty((B,), self.dtype)
        self._fn("gaussian_norm", [_P, _I, _I, _I, _I, _P, _P])(_ptr(x), T, B, BN // B, I, _ptr(mu), None)
        return x, mu

    def cum_laplace_norm(self, x):
        """x [T, R, I] -> copy with every row divided by the running mean of what it has seen so far."""
        x = self.arr(x).copy()
        T, R, I = x.shape
        self._fn("cum_laplace_norm", [_P, _I, _I, _I])(_ptr(x), T, R, I)
        return x

    def deepfilter_group(self, stft, proj, enh, lo, N, fc, df, S):
        """Writes bins lo..lo+N*fc-1 of enh (complex [B,S,F,T]) in place."""
        stft = np.ascontiguousarray(stft, dtype=self.cdtype)
        proj = self.arr(proj)
        B, F, T = stft.shape
        assert enh.dtype == self.cdtype and enh.shape == (B, S, F, T) and enh.flags.c_contiguous
        assert proj.shape == (T, B * N, 2 * fc * df * S), (proj.shape, (T, B * N, 2 * fc * df * S))
        self._fn("deepfilter_group", [_P, _P] + [_I] * 8 + [_P])(_ptr(stft), _ptr(proj), B, F, T, lo, N, fc, df, S, _ptr(enh))

    def finish_spectrum(self, stft, enh, f0):
        stft = np.ascontiguousarray(stft, dtype=self.cdtype)
        B, S, F, T = enh.shape
        mag = np.empty((B, S, F, T), self.dtype)
        self._fn("finish_spectrum", [_P] + [_I] * 5 + [_P, _P])(_ptr(stft), B, S, F, T, f0, _ptr(enh), _ptr(mag))
        return mag
