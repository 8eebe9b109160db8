"""ctypes binding of ``libsfsn_hip.so`` (the C ABI declared in ``include/sfsn.h``).

There is no fallback: if the library has not been built the import of any compute entry point raises,
and every launch on a box without a gfx950 device raises ``RuntimeError`` (``SFSN_EHIP``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libsfsn_hip.so")

SFSN_OK, SFSN_EINVAL, SFSN_EUNSUPPORTED, SFSN_EHIP, SFSN_EDIVISIBLE = 0, -1, -2, -3, -4
NORM_NONE, NORM_LAYERNORM, NORM_LAPLACE, NORM_CUMLAPLACE, NORM_GAUSSIAN = 0, 1, 2, 3, 4
MAX_SEGMENTS, MAX_GROUPS, MAX_HIDDEN = 8, 8, 320
ABI_VERSION = 19  # = SFSN_ABI_VERSION of include/sfsn.h; bumped with every struct / signature change

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float


class ScanSegment(ctypes.Structure):
    _fields_ = [("zin", _P), ("w_hh", _P), ("w_dq", _P), ("bias", _P), ("bn_alpha", _P), ("bn_beta", _P),
                ("h_state", _P), ("c_state", _P), ("spikes_f32", _P), ("spikes_i8", _P), ("membrane", _P), ("R", _I),
                ("spike_count", _P)]


class FusedInput(ctypes.Structure):
    _fields_ = [("spikes_in", _P), ("w_ih", _P), ("w_ih_dq", _P)]


class FusedX(ctypes.Structure):
    _fields_ = [("x", _P), ("w_ih", _P), ("I", _I)]


class FeatureGroup(ctypes.Structure):
    _fields_ = [("x", _P), ("ln_w", _P), ("ln_b", _P), ("mu", _P), ("lo", _I), ("n_units", _I), ("ctr", _I), ("nbr", _I),
                ("ctr_fb", _I), ("nbr_fb", _I), ("norm", _I), ("ln_eps", _F)]


class DfGroup(ctypes.Structure):
    _fields_ = [("proj", _P), ("n_units", _I), ("fc", _I), ("df", _I)]


class ProjDfGroup(ctypes.Structure):  # sfsn_projdf_group
    _fields_ = [("spikes_i8", _P), ("w_packed", _P), ("w_dq", _P), ("bias", _P), ("proj", _P), ("n_units", _I), ("fc", _I), ("df", _I)]


class ProjJob(ctypes.Structure):
    _fields_ = [("s", _P), ("w_packed", _P), ("w_dq", _P), ("bias", _P), ("y", _P), ("M", _I), ("K", _I), ("N", _I), ("ldy", _I)]


class InProjJob(ctypes.Structure):
    _fields_ = [("x", _P), ("w", _P), ("bias", _P), ("z", _P), ("M", _I), ("K", _I), ("N", _I), ("ldz", _I)]


class FeatProjJob(ctypes.Structure):  # sfsn_featproj_job
    _fields_ = [("feat", FeatureGroup), ("w", _P), ("bias", _P), ("z", _P), ("H", _I), ("ldz", _I)]


class CountTensor(ctypes.Structure):
    _fields_ = [("spikes_i8", _P), ("n_bytes", ctypes.c_ulonglong), ("count", _P)]


class TrainSeqFwd(ctypes.Structure):  # SfsnTrainSeqFwd: one layer call of a multi-call training launch
    _fields_ = [("z", _P), ("w_hh", _P), ("bias", _P), ("bn_w", _P), ("bn_b", _P), ("running_mean", _P), ("running_var", _P),
                ("momentum", _F), ("eps", _F), ("R", _I), ("spikes", _P), ("u", _P), ("xhat", _P), ("f", _P), ("g", _P), ("invstd", _P),
                ("scratch", _P), ("h0", _P), ("c0", _P), ("T", _I)]


class TrainSeqBwd(ctypes.Structure):  # SfsnTrainSeqBwd
    _fields_ = [("w_hh", _P), ("dh_up", _P), ("u", _P), ("xhat", _P), ("f", _P), ("g", _P), ("invstd", _P), ("bn_w", _P), ("R", _I),
                ("d_gates", _P), ("d_z", _P), ("d_bn_w", _P), ("d_bn_b", _P), ("scratch", _P), ("dc_in", _P), ("dc_out", _P), ("has_prev", _I), ("T", _I)]


TRAIN_MAX_CALLS = 8
MAX_COUNT_TENSORS = 16
HOP_MAX_LAYERS = 3
HOP_MAX_GROUPS = 4


class HopLayer(ctypes.Structure):
    _fields_ = [("w_ih_frag", _P), ("w_ih", _P), ("w_ih_dq", _P), ("w_hh", _P), ("w_hh_dq", _P), ("bias", _P), ("bn_alpha", _P),
                ("bn_beta", _P), ("h", _P * 2), ("c", _P), ("spikes", _P)]


class HopSeq(ctypes.Structure):
    _fields_ = [("layer", HopLayer * HOP_MAX_LAYERS), ("n_layers", _I), ("H", _I), ("P", _I), ("feat", FeatureGroup), ("w_p", _P),
                ("w_p_dq", _P), ("b_p", _P), ("df", _I), ("fc", _I), ("cum", _P * 2)]


class HopDesc(ctypes.Structure):
    _fields_ = [("fb", HopSeq), ("sb", HopSeq * HOP_MAX_GROUPS), ("n_groups", _I), ("B", _I), ("F", _I), ("S", _I), ("hop", _I),
                ("D", _I), ("fdrc", _F), ("inp_ri", _P), ("hist_ri", _P), ("enh_ri", _P), ("enh_mag", _P),
                ("scratch", _P), ("scratch_bytes", ctypes.c_size_t), ("launch_index", ctypes.c_uint), ("wave_in", _P), ("wave_state", _P),
                ("ola_state", _P), ("wave_out", _P), ("window", _P), ("spec_g", _P), ("enh_g", _P), ("frame_index", _I), ("done", _P), ("frames_before", _I), ("unshared", _I)]


def _sources():
    """The files the library is made of, in the order the Makefile hashes them (SRCS)."""
    return [os.path.join(_HERE, "..", "include", "sfsn.h")] + [
        os.path.join(CSRC, f) for f in ("sfsn_scan_dev.h", "sfsn_scan3_dev.h", "sfsn_scan3i_dev.h", "sfsn_scan3x_dev.h", "sfsn_scan3w_dev.h", "sfsn_scan3j_dev.h", "sfsn_scan3g_dev.h", "sfsn_feat_dev.h", "sfsn_fft_dev.h", "sfsn_kernels.hip", "sfsn_stack.hip", "sfsn_hop.hip", "sfsn_fft.hip", "sfsn_train.hip",
                                  "sfsn_featproj.hip", "sfsn_projdf.hip", "sfsn_pack.cpp")]


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the sources; equals ``sfsn_source_hash()`` of a library built from them."""
    import hashlib
    h = hashlib.sha256()
    for s in _sources():
        with open(s, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (``make -C csrc``); hipcc cross-compiles without a GPU."""
    srcs = _sources()
    stale = not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", CSRC, "-s", "-j4"] + (["-B"] if force else []), check=True)
    return LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    """The loaded library; raises ImportError with the build recipe when it is missing (no fallback)."""
    global _lib, LIB_PATH
    if _lib is not None:
        return _lib
    if os.environ.get("SFSN_LIB_PATH"):  # an experiment build beside the product's (scripts/build_exp_lib.sh); same ABI / source-hash checks
        LIB_PATH = os.environ["SFSN_LIB_PATH"]
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is not built. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C spiking_fullsubnet_amd/csrc`). There is no CPU or eager fallback for this path.")
    L = ctypes.CDLL(LIB_PATH)
    L.sfsn_abi_version.restype = _I
    if L.sfsn_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {L.sfsn_abi_version()} != {ABI_VERSION}; rebuild (make -C {CSRC})")
    # a stale build (sources edited after the last make) is refused, not called with mismatched struct layouts; file times
    # do not survive a copy of the tree, so the check is a hash of the sources compiled into the library
    if all(os.path.exists(s) for s in _sources()):
        L.sfsn_source_hash.restype = ctypes.c_char_p
        built, want = L.sfsn_source_hash().decode(), source_hash()
        if built != want:
            raise ImportError(f"{LIB_PATH} was built from other sources (hash {built}, tree {want}); rebuild: make -C {CSRC}")
    L.sfsn_strerror.restype = ctypes.c_char_p
    L.sfsn_strerror.argtypes = [_I]
    L.sfsn_device_count.restype = _I
    L.sfsn_w3_packed_bytes.restype = ctypes.c_size_t
    L.sfsn_w3_packed_bytes.argtypes = [_I, _I]
    L.sfsn_w3_padded_rows.restype = _I
    L.sfsn_w3_padded_rows.argtypes = [_I]
    L.sfsn_w3_pack.restype = _I
    L.sfsn_w3_pack.argtypes = [_P, _I, _I, _P, _P]
    L.sfsn_w3_pack_bits.restype = _I
    L.sfsn_w3_pack_bits.argtypes = [_P, _I, _I, _I, _P, _P]
    L.sfsn_w3_unpack.restype = _I
    L.sfsn_w3_unpack.argtypes = [_P, _P, _I, _I, _P]
    L.sfsn_gsn_layer_scan.restype = _I
    L.sfsn_gsn_layer_scan.argtypes = [ctypes.POINTER(ScanSegment), _I, _I, _I, _I, _I, _P]
    L.sfsn_gsn_layer_scan_w16.restype = _I
    L.sfsn_gsn_layer_scan_w16.argtypes = [ctypes.POINTER(ScanSegment), _I, _I, _I, _I, _I, _P]
    L.sfsn_gsn_train_step_fwd.restype = _I
    L.sfsn_gsn_train_step_fwd.argtypes = [_P] * 9 + [_F, _F, _I, _I, _I] + [_P] * 7 + [ctypes.c_uint, _P]
    L.sfsn_train_scratch_bytes.restype = ctypes.c_size_t
    L.sfsn_train_scratch_bytes.argtypes = [_I]
    L.sfsn_train_seq_scratch_bytes.restype = ctypes.c_size_t
    L.sfsn_train_seq_scratch_bytes.argtypes = [_I, _I]
    L.sfsn_gsn_train_multi_check.restype = _I
    L.sfsn_gsn_train_multi_check.argtypes = [ctypes.POINTER(_I), _I, _I, _I]
    L.sfsn_gsn_train_seq_fwd_multi.restype = _I
    L.sfsn_gsn_train_seq_fwd_multi.argtypes = [ctypes.POINTER(TrainSeqFwd), _I, _I, _I, _I, _P]
    L.sfsn_gsn_train_seq_bwd_multi.restype = _I
    L.sfsn_gsn_train_seq_bwd_multi.argtypes = [ctypes.POINTER(TrainSeqBwd), _I, _I, _I, _I, _P]
    L.sfsn_gsn_train_check.restype = _I
    L.sfsn_gsn_train_check.argtypes = [_I, _I, _I]
    L.sfsn_gsn_train_step_check.restype = _I
    L.sfsn_gsn_train_step_check.argtypes = [_I, _I, _I]
    L.sfsn_gsn_train_step_bwd.restype = _I
    L.sfsn_gsn_train_step_bwd.argtypes = [_P] * 12 + [_I, _I, _I] + [_P] * 6 + [ctypes.c_uint, _P]
    L.sfsn_gsn_train_seq_fwd.restype = _I  # z, w_hh, bias, bn_w, bn_b, running_mean, running_var | momentum, eps | T, R, H, shared | zero, 6 outputs, scratch, stream
    L.sfsn_gsn_train_seq_fwd.argtypes = [_P] * 7 + [_F, _F, _I, _I, _I, _I] + [_P] * 9
    L.sfsn_gsn_train_seq_bwd.restype = _I  # w_hh, dh_up, u, xhat, f, g, invstd, bn_w | T, R, H, shared | zero, d_gates, d_z, dc_work, d_bn_w, d_bn_b, scratch, stream
    L.sfsn_gsn_train_seq_bwd.argtypes = [_P] * 8 + [_I, _I, _I, _I] + [_P] * 8
    L.sfsn_gsn_layer_scan_fused.restype = _I
    L.sfsn_gsn_layer_scan_fused.argtypes = [ctypes.POINTER(ScanSegment), ctypes.POINTER(FusedInput), _I, _I, _I, _P]
    L.sfsn_gsn_layer_scan_fused_x.restype = _I
    L.sfsn_gsn_layer_scan_fused_x.argtypes = [ctypes.POINTER(ScanSegment), ctypes.POINTER(FusedX), _I, _I, _I, _P]
    L.sfsn_stack_scratch_bytes.restype = ctypes.c_size_t
    L.sfsn_stack_scratch_bytes.argtypes = [_I, _I, _I]
    L.sfsn_gsn_stack_scan.restype = _I
    L.sfsn_gsn_stack_scan.argtypes = [ctypes.POINTER(ScanSegment), ctypes.POINTER(FusedInput), _I, _I, _I, _I, ctypes.POINTER(_I), _I, _P,
                                      ctypes.c_size_t, _P]
    L.sfsn_gsn_stack_scan_x.restype = _I
    L.sfsn_gsn_stack_scan_x.argtypes = [ctypes.POINTER(ScanSegment), ctypes.POINTER(FusedInput), ctypes.POINTER(FusedX), _I, _I, _I, _I,
                                        ctypes.POINTER(_I), _I, _P, ctypes.c_size_t, _P]
    L.sfsn_gsn_stack_scan_x_w16.restype = _I
    L.sfsn_gsn_stack_scan_x_w16.argtypes = L.sfsn_gsn_stack_scan_x.argtypes
    L.sfsn_input_proj_f32.restype = _I
    L.sfsn_input_proj_f32.argtypes = [_P, _P, _P, _P, _I, _I, _I, _I, _P]
    L.sfsn_spike_proj.restype = _I
    L.sfsn_spike_proj.argtypes = [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]
    L.sfsn_spike_proj_multi.restype = _I
    L.sfsn_spike_proj_multi.argtypes = [ctypes.POINTER(ProjJob), _I, _P]
    L.sfsn_input_proj_f32_multi.restype = _I
    L.sfsn_input_proj_f32_multi.argtypes = [ctypes.POINTER(InProjJob), _I, _P]
    L.sfsn_features.restype = _I
    L.sfsn_features.argtypes = [_P, _P, _I, _I, _I, _I, _F, ctypes.POINTER(FeatureGroup), _I, _I, _I, _P]
    L.sfsn_scan_split_scratch_bytes.restype = ctypes.c_size_t
    L.sfsn_scan_split_scratch_bytes.argtypes = [_I, _I]
    L.sfsn_gsn_layer_scan_split.restype = _I
    L.sfsn_gsn_layer_scan_split.argtypes = [ctypes.POINTER(ScanSegment), _I, _I, _I, _I, _P, ctypes.c_size_t, _P]
    L.sfsn_features_proj.restype = _I
    L.sfsn_features_proj.argtypes = [_P, _P, _I, _I, _I, _I, _F, ctypes.POINTER(FeatProjJob), _I, _I, _I, _P, ctypes.c_size_t, _P]
    L.sfsn_features_z.restype = _I
    L.sfsn_features_z.argtypes = [_P, _P, _I, _I, _I, _I, _F, ctypes.POINTER(FeatureGroup), _I, _I, _I, _P, ctypes.c_size_t, _P]
    L.sfsn_laplace_means.restype = _I
    L.sfsn_laplace_means.argtypes = [_P, _P, _I, _I, _I, _I, _F, ctypes.POINTER(FeatureGroup), _I, _P, _P, _P]
    L.sfsn_gaussian_stats.restype = _I
    L.sfsn_gaussian_stats.argtypes = [_P, _P, _I, _I, _I, _I, _F, ctypes.POINTER(FeatureGroup), _I, _P, _P, _P, _P]
    L.sfsn_deepfilter.restype = _I
    L.sfsn_deepfilter.argtypes = [_P, _I, _I, _I, _I, ctypes.POINTER(DfGroup), _I, _P, _P, _I, _I, _P]
    L.sfsn_proj_deepfilter.restype = _I
    L.sfsn_proj_deepfilter.argtypes = [_P, _I, _I, _I, _I, _I, ctypes.POINTER(ProjDfGroup), _I, _P, _P, _I, _I, _P]
    L.sfsn_hist_shift.restype = _I
    L.sfsn_hist_shift.argtypes = [_P, _P, _I, _I, _I, _P]
    L.sfsn_cum_laplace_norm.restype = _I
    L.sfsn_cum_laplace_norm.argtypes = [_P, _I, _I, _I, _P, _I, _P, _P]
    L.sfsn_hop_scratch_bytes.restype = ctypes.c_size_t
    L.sfsn_hop_scratch_bytes.argtypes = [ctypes.POINTER(HopDesc)]
    L.sfsn_hop_stages.restype = _I
    L.sfsn_hop_stages.argtypes = [ctypes.POINTER(HopDesc), ctypes.POINTER(_I), _I]
    L.sfsn_stream_hop.restype = _I
    L.sfsn_stream_hop.argtypes = [ctypes.POINTER(HopDesc), _P]
    L.sfsn_stream_hop_resident.restype = _I
    L.sfsn_stream_hop_resident.argtypes = [ctypes.POINTER(HopDesc), _P, ctypes.c_uint, _P]
    L.sfsn_spike_count.restype = _I
    L.sfsn_spike_count.argtypes = [ctypes.POINTER(CountTensor), _I, _P]
    L.sfsn_stft.restype = _I
    L.sfsn_stft.argtypes = [_P, _I, _I, _I, _I, _P, _P, _I, _P]
    L.sfsn_istft.restype = _I
    L.sfsn_istft.argtypes = [_P, _I, _I, _I, _I, _P, _P, _I, _P]
    if L.sfsn_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {L.sfsn_abi_version()} != {ABI_VERSION}; rebuild (make -C {CSRC})")
    _lib = L
    return L


EXPORTS = ("sfsn_abi_version", "sfsn_source_hash", "sfsn_strerror", "sfsn_device_count", "sfsn_w3_packed_bytes", "sfsn_w3_padded_rows",
           "sfsn_w3_pack", "sfsn_w3_pack_bits", "sfsn_w3_unpack", "sfsn_gsn_layer_scan", "sfsn_gsn_layer_scan_fused", "sfsn_gsn_layer_scan_fused_x", "sfsn_stack_scratch_bytes", "sfsn_gsn_stack_scan",
           "sfsn_input_proj_f32", "sfsn_spike_proj", "sfsn_features",
           "sfsn_laplace_means", "sfsn_cum_laplace_norm", "sfsn_deepfilter", "sfsn_hist_shift", "sfsn_hop_scratch_bytes", "sfsn_stream_hop", "sfsn_stream_hop_resident", "sfsn_hop_stages", "sfsn_spike_count", "sfsn_stft", "sfsn_istft", "sfsn_gsn_train_step_fwd", "sfsn_gsn_train_step_bwd", "sfsn_train_scratch_bytes",
           "sfsn_gsn_train_seq_fwd", "sfsn_gsn_train_seq_bwd", "sfsn_gsn_layer_scan_w16", "sfsn_gsn_train_check", "sfsn_gsn_stack_scan_x", "sfsn_train_seq_scratch_bytes", "sfsn_gsn_train_multi_check",
           "sfsn_gsn_train_seq_fwd_multi", "sfsn_gsn_train_seq_bwd_multi", "sfsn_features_z", "sfsn_gaussian_stats", "sfsn_gsn_train_step_check",
           "sfsn_spike_proj_multi", "sfsn_input_proj_f32_multi", "sfsn_features_proj",
           "sfsn_scan_split_scratch_bytes", "sfsn_gsn_layer_scan_split", "sfsn_proj_deepfilter", "sfsn_gsn_stack_scan_x_w16")


def check(rc: int, what: str = "") -> None:
    """Map a C-ABI status to the exception the reference's own code would raise for the same condition."""
    if rc == SFSN_OK:
        return
    msg = f"{what}: {lib().sfsn_strerror(rc).decode()}" if what else lib().sfsn_strerror(rc).decode()
    if rc == SFSN_EDIVISIBLE:
        raise ValueError(msg)  # modeling_spiking_fullsubnet.py:283-287
    if rc == SFSN_EUNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == SFSN_EINVAL:
        raise ValueError(msg)
    raise RuntimeError(msg)
