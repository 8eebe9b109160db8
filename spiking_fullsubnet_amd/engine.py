"""Host-side driver of the hot path: packs a model's weights for the gfx950 kernels and enqueues the
launch sequence of one forward (complex STFT in HBM -> enhanced STFT / magnitude + the per-layer
tensors the reference's modules return) on torch's current HIP stream, through the C ABI only.

PyTorch is used here for device memory and streams; every arithmetic step of the path runs in
``libsfsn_hip.so``.  There is no CPU / eager fallback: CPU tensors raise.

Reference call stack replaced: ``SpikingFullSubNet.forward`` modeling_spiking_fullsubnet.py:434-472
(frozen twin ``Separator.forward`` model_low_freq.py:574-607) and everything below it.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import CountTensor, DfGroup, FeatProjJob, FeatureGroup, FusedInput, FusedX, InProjJob, ProjDfGroup, ProjJob, ScanSegment, check


@dataclass
class PathSpec:
    """Geometry of one model, normalised over the live and frozen front-ends."""
    front: str                    # "live" (LayerNorm inputs, `proj`) | "frozen" (offline Laplace norm, `fc_output_layer`)
    n_fft: int
    fdrc: float
    fb_in: int
    fb_hidden: int
    fb_layers: int
    fb_proj: int
    sb_hidden: int
    sb_layers: int
    cutoffs: List[int]
    ctr: List[int]
    nbr: List[int]
    ctr_fb: List[int]
    nbr_fb: List[int]
    df: List[int]
    num_spks: int = 1
    shared: bool = False
    bn: bool = False
    ln_fb: bool = True
    ln_sb: bool = True
    laplace: bool = False         # frozen front-end with an utterance-level (offline, non-causal) norm: offline_laplace_norm, or ...
    gaussian: bool = False        # ... offline_gaussian_norm (set together with `laplace`: same schedule, other statistics)
    cum_laplace: bool = False     # frozen front-end with cumulative_laplace_norm (causal: every row by its own running mean)
    proj_name: str = "proj"

    @property
    def n_groups(self) -> int:
        return len(self.ctr)

    @property
    def num_freqs(self) -> int:  # bins the network processes (Nyquist excluded)
        return self.n_fft // 2

    def units(self, g: int) -> int:
        return (self.cutoffs[g + 1] - self.cutoffs[g]) // self.ctr[g]

    def sb_input_size(self, g: int) -> int:
        return (self.ctr[g] + 2 * self.nbr[g]) + (self.ctr_fb[g] + 2 * self.nbr_fb[g])

    def sb_proj_size(self, g: int) -> int:
        return 2 * self.ctr[g] * self.df[g] * self.num_spks


def _dev(a: np.ndarray, device) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def pack_w3(w: np.ndarray, bits: int = 24):
    """fp32 [N, K] -> (int8 digits in MFMA fragment order, dq [pad16(N)]) as numpy arrays (host packing).  bits = 16: the
    16-bit-weight mode (weights rounded to 16 significant bits of the row grid, least-significant digit plane zero)."""
    L = _lib.lib()
    w = np.ascontiguousarray(w, dtype=np.float32)
    n, k = w.shape
    packed = np.empty(L.sfsn_w3_packed_bytes(n, k), np.int8)
    dq = np.empty(L.sfsn_w3_padded_rows(n), np.float32)
    check(L.sfsn_w3_pack_bits(w.ctypes.data, n, k, bits, packed.ctypes.data, dq.ctypes.data), "sfsn_w3_pack_bits")
    return packed, dq


def unpack_w3(packed: np.ndarray, dq: np.ndarray, n: int, k: int) -> np.ndarray:
    L = _lib.lib()
    w = np.empty((n, k), np.float32)
    check(L.sfsn_w3_unpack(packed.ctypes.data, dq.ctypes.data, n, k, w.ctypes.data), "sfsn_w3_unpack")
    return w


def fold_batchnorm(weight, bias, mean, var, eps=1e-5):
    """Eval-mode BatchNorm1d as ATen's CPU kernel evaluates it (probed bit-exact, see oracle/sfsn_oracle.c):
    alpha = gamma / sqrt(var + eps), beta = fma(-mean, alpha, bias), y = fma(x, alpha, beta)."""
    f32 = np.float32
    invstd = (f32(1) / np.sqrt(var.astype(f32) + f32(eps))).astype(f32)
    alpha = (invstd * weight.astype(f32)).astype(f32)
    beta = (bias.astype(np.float64) - mean.astype(np.float64) * alpha.astype(np.float64)).astype(f32)
    return alpha, beta


@dataclass
class _Cell:
    H: int
    G: int
    w_ih_f32: Optional[torch.Tensor] = None                 # layer 0: [G*H, I] fp32
    w_ih_q: List[tuple] = field(default_factory=list)        # layer >= 1: per gate (packed, dq) of [H, H]
    w_hh_q: Optional[torch.Tensor] = None                    # packed [G*H, H]
    w_hh_dq: Optional[torch.Tensor] = None
    bias: Optional[torch.Tensor] = None
    alpha: Optional[torch.Tensor] = None
    beta: Optional[torch.Tensor] = None


@dataclass
class _Seq:
    I: int
    H: int
    P: int
    cells: List[_Cell]
    proj_q: torch.Tensor
    proj_dq: torch.Tensor
    proj_b: torch.Tensor
    ln_w: Optional[torch.Tensor] = None
    ln_b: Optional[torch.Tensor] = None


def _pack_seq(sd: Dict[str, np.ndarray], prefix: str, spec: PathSpec, I: int, H: int, L: int, P: int, use_ln: bool, device, bits: int = 24) -> _Seq:
    G = 1 if spec.shared else 2
    cells = []
    for l in range(L):
        p = f"{prefix}sequence_model.layers.{l}.cell."
        w_ih, w_hh, b = sd[p + "weight_ih"], sd[p + "weight_hh"], sd[p + "bias_ih"]
        Il = I if l == 0 else H
        assert w_ih.shape == (G * H, Il) and w_hh.shape == (G * H, H) and b.shape == (2 * H,), (p, w_ih.shape, w_hh.shape)
        cell = _Cell(H=H, G=G)
        if l == 0:
            cell.w_ih_f32 = _dev(w_ih.astype(np.float32), device)
        else:
            for g in range(G):
                pk, dq = pack_w3(w_ih[g * H:(g + 1) * H], bits)
                cell.w_ih_q.append((_dev(pk, device), _dev(dq, device)))
        pk, dq = pack_w3(w_hh, bits)
        cell.w_hh_q, cell.w_hh_dq = _dev(pk, device), _dev(dq, device)
        cell.bias = _dev(b.astype(np.float32), device)
        if spec.bn:
            a, be = fold_batchnorm(sd[p + "batchnorm.weight"], sd[p + "batchnorm.bias"], sd[p + "batchnorm.running_mean"],
                                   sd[p + "batchnorm.running_var"])
        else:
            a, be = np.ones(H, np.float32), np.zeros(H, np.float32)
        cell.alpha, cell.beta = _dev(a, device), _dev(be, device)
        cells.append(cell)
    pn = prefix + spec.proj_name
    pw, pb = sd[pn + ".weight"], sd[pn + ".bias"]
    assert pw.shape == (P, H), (pn, pw.shape, (P, H))
    pk, dq = pack_w3(pw, bits)
    seq = _Seq(I=I, H=H, P=P, cells=cells, proj_q=_dev(pk, device), proj_dq=_dev(dq, device), proj_b=_dev(pb.astype(np.float32), device))
    if use_ln:
        seq.ln_w = _dev(sd[prefix + "pre_layer_norm.weight"].astype(np.float32), device)
        seq.ln_b = _dev(sd[prefix + "pre_layer_norm.bias"].astype(np.float32), device)
    return seq


class SpikeSummary:
    """Stand-in for one fp32 spike tensor of ``all_layer_outputs`` when only its statistics are wanted
    (``layer_outputs="counts"``): the exact number of spikes, counted on the device from the int8 spikes the scan writes,
    plus the shape the tensor would have had.  ``metric.compute_synops`` / ``compute_neuronops`` accept it in place of
    the tensor (audiozen/metric.py:303-340 read only ``gt(x, 0).float().mean()`` and ``size(-1)``)."""

    def __init__(self, count: torch.Tensor, shape):
        self.count, self.shape = count, torch.Size(shape)

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def numel(self) -> int:
        return self.shape.numel()

    def rate(self) -> torch.Tensor:
        """fp32 firing rate = what ``torch.gt(x, 0).float().mean()`` estimates (here exact count / numel, rounded once)."""
        return (self.count.to(torch.float64) / self.numel()).to(torch.float32)

    def __repr__(self):
        return f"SpikeSummary(shape={tuple(self.shape)})"


SOAKED_HW_QUEUES = (4, 24)  # GPU_MAX_HW_QUEUES values the two-stream schedule has been soaked under (unset = HIP's default of 4)
_hw_queues_warned = False


def _check_hw_queues() -> int:
    """The overlapped schedule of a forward alone runs two launches of resident workgroups with in-launch waits on two HIP streams
    (the sub-band pair launch and the full-band stack), and bench.py's timed region a dozen forwards on as many streams.  How the
    runtime maps streams onto hardware queues is `GPU_MAX_HW_QUEUES` (read by HIP once, at its initialisation).  Three such launches
    on three streams stopped being dispatched within 3-120 iterations at 4 and 8 queues and never at 2 (the training path's layer
    calls, scripts/dbg_train_hang.py, profiles/EXPERIMENTS.md; root cause unknown -- training was moved to one grid).  The inference
    schedules were soaked at the default (4: the -m gpu suite, scripts/soak_r04.py, 3000 forwards) and at 24 (bench.py): every
    in-launch wait is bounded and reported, so another value can cost a loud failure, never wrong results -- warn once."""
    global _hw_queues_warned
    raw = os.environ.get("GPU_MAX_HW_QUEUES", "")
    try:
        n = int(raw) if raw else 4
    except ValueError:
        n = 4
    if n not in SOAKED_HW_QUEUES and not _hw_queues_warned:
        _hw_queues_warned = True
        import warnings
        warnings.warn(f"GPU_MAX_HW_QUEUES={raw}: the two-stream schedule of spiking_fullsubnet_amd was soaked at {SOAKED_HW_QUEUES} "
                      "hardware queues only (see README.md, limits); in-launch waits are bounded and raise, results are never silently wrong",
                      RuntimeWarning, stacklevel=3)
    return n


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class Engine:
    """Packed weights + launch sequence for one model on one device."""

    def __init__(self, spec: PathSpec, state_dict: Dict[str, np.ndarray], device, weight_bits: int = 24):
        """weight_bits = 24: the exact fp32-parity mode.  16: the 16-bit-weight fast mode -- the recurrent, spike-input and
        projection weights rounded to 16 significant bits of their row grid (the real-valued layer-0 input product stays
        exact); reported against the fp32 oracle by tests/test_hip_parity.py::test_sixteen_bit_weight_mode_report."""
        if weight_bits not in (24, 16):
            raise ValueError("weight_bits must be 24 (exact) or 16")
        self.weight_bits = weight_bits
        self.w16_fast = os.environ.get("SFSN_W16_FAST", "1") != "0"  # 16-bit mode: skip the zero digit plane in the scans that can
        self.spec = spec
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("spiking_fullsubnet_amd runs on a HIP device only (no CPU path); move the module to 'cuda'")
        self.lib = _lib.lib()
        for H in (spec.fb_hidden, spec.sb_hidden):
            if H % 16 != 0 or H > _lib.MAX_HIDDEN:
                raise NotImplementedError(f"hidden size {H}: the gfx950 scan holds W_hh register-resident for H % 16 == 0, H <= {_lib.MAX_HIDDEN}")
        if spec.n_groups > _lib.MAX_SEGMENTS:
            raise NotImplementedError(f"more than {_lib.MAX_SEGMENTS} sub-band groups")
        if len(spec.cutoffs) == 2:
            raise NotImplementedError("single-group models hit a latent reflect-pad quirk of the reference (SubbandModel._freq_unfold)")
        sd = {k: np.asarray(v) for k, v in state_dict.items()}
        self.fb = _pack_seq(sd, "fb_model.", spec, spec.fb_in, spec.fb_hidden, spec.fb_layers, spec.fb_proj, spec.ln_fb, self.device, weight_bits)
        self.sb = [_pack_seq(sd, f"sb_model.sb_models.{g}.", spec, spec.sb_input_size(g), spec.sb_hidden, spec.sb_layers,
                             spec.sb_proj_size(g), spec.ln_sb, self.device, weight_bits) for g in range(spec.n_groups)]
        self._ws: Dict[tuple, dict] = {}
        self.ws_budget_bytes = int(float(os.environ.get("SFSN_WS_BUDGET_GB", "96")) * 2**30)  # scratch cache budget (of 288 GB HBM)
        self.timers: Optional[dict] = None  # set to {} to record HIP events around each launch group (bench.py)
        self.seq_chunk = 0  # frames per chunk of the single-stream schedule (0 = whole sequence per launch)
        self.rows_per_wg = (0, 0)  # (full-band, sub-band) rows per scan workgroup; 0 = let the library spread over all CUs
        self.fuse_input = True  # input term inside the scan where the geometry allows it (see _fusable / _fusable_x)
        self.launches: Dict[str, int] = {}  # launches per C-ABI scan entry point (tests assert which path ran)
        self.timer_tags = None  # optional set of tags to time (each timed group costs ~10 us of launch gap)
        self._stream_objs: Dict[int, torch.cuda.Stream] = {}
        self._side: List[torch.cuda.Stream] = []
        self.pipeline_chunk = 128  # frames per chunk of the time-pipelined schedule
        self.pipeline_default = False  # opt-in until the stages own disjoint CU sets
        self.pipeline_two_phase = bool(int(os.environ.get('SFSN_TWO_PHASE', '1')))
        # layer-pipelined stack scan (sfsn_gsn_stack_scan): all layers of a stack in one launch, layer l+1 trailing layer l
        # by a few frames.  Shared gate weights only; membrane outputs (a test tap of the per-layer kernel) use the per-layer path.
        _ss = os.environ.get('SFSN_STACK_SCAN', 'auto')
        self.stack_scan = "auto" if _ss == "auto" else bool(int(_ss))
        self.stack_rows_fb_auto = int(os.environ.get("SFSN_FB_STACK_ROWS", "4"))  # rows per workgroup of the full-band stack under "auto"
        self.split_scan = os.environ.get("SFSN_SPLIT_SCAN", "1") != "0"  # large separate gate weights: sfsn_gsn_layer_scan_split (see _stage_scan)
        self.merge_products = os.environ.get("SFSN_MERGE_PRODUCTS", "1") != "0"  # the independent products of a stage in one launch
        # features + layer-0 input products of a chunk in ONE launch (sfsn_features_proj): the rows never make the round trip through
        # HBM between the two, and are not written at all when nobody reads them (layer_outputs "counts" / "none").  OFF by default:
        # bit-identical and 0.34 GB less traffic per forward, but slower -- the row arithmetic is bound by VALU issue (~110 wave
        # instructions per row) and the product's W pieces (120 registers per wave) leave two waves per SIMD where the feature kernel
        # has three: 316 us per sub-band chunk against 67 + 72 (profiles/EXPERIMENTS.md).  SFSN_FEATPROJ=1 switches it on.
        self.fuse_featproj = os.environ.get("SFSN_FEATPROJ", "0") == "1"
        # round 6: the sub-band projection, the output re-index, the deep filter and |.| in ONE launch (sfsn_proj_deepfilter): the
        # coefficient rows are written once and never re-read (0.3 GB per forward at B = 64, T = 1000).  SFSN_PROJDF=0: the two launches.
        self.fuse_projdf = os.environ.get("SFSN_PROJDF", "1") != "0"
        # opt-in for the lean modes (want_layers False): the coefficient rows (`all_layer_outputs[-1]`) are not written either -- the live
        # recipe discards the lists (trainer.py:31,52) and metric.compute_neuronops reads size(-1) only; the entry keeps its shape as a
        # meta tensor, like the feature rows a fused feature launch skips.  Off by default: `layer_outputs = "counts"` promises the
        # projection tensor (tests).  bench.py's no-layer-outputs leg switches it on and says so.
        self.lean_skips_proj = os.environ.get("SFSN_LEAN_SKIP_PROJ", "0") == "1"
        self.count_in_scan = os.environ.get("SFSN_COUNT_IN_SCAN", "1") != "0"  # layer_outputs="counts": counted by the scans themselves
        self.pair_scan = os.environ.get("SFSN_PAIR_SCAN", "1") != "0"  # H <= 224 stacks as one launch of FUSED3 roles (see _stack_choice)
        self.stack_rows_per_wg = {"fb": 4, "sb": 8}  # rows per workgroup of every layer of a stack: sum of workgroups <= CUs
        # frames a consumer role that had to wait lets its producers run ahead before it resumes (the hand-offs' hysteresis).  Round 2: 16
        # (every wave of a workgroup stood in the poll); since the IO-wave roles one lane polls, and a launch ends lag + ring frames after
        # its first layer does: 16 / 8 / 4 / 2 / 1 -> strict forward 2.74-2.76 / 2.65-2.70 / 2.62 / 2.63 / 2.63 ms (scripts/exp_lag_r05.sh)
        self.stack_lag = int(os.environ.get("SFSN_STACK_LAG", "4"))
        # full-band / sub-band overlap of ONE forward: the sequence is cut into this many chunks, the full-band model runs them
        # on one stream, the sub-band models follow one chunk behind on a second stream (they need the full-band output of the
        # SAME frames only, MODEL:441-447; states are carried by the ABI's h_state / c_state).  The full-band chain (few
        # workgroups, long) then hides behind the sub-band work of the previous chunk.  0 / 1 = off.
        # Measured (B=64, T=1000, scripts/exp_overlap.py): 3.93 ms without, 3.67 / 3.60 / 3.62 / 3.87 / 4.38 ms with 2 / 3 / 4 / 5 / 8
        # chunks -- every chunk costs ~0.15 ms of launch boundaries, scan prologues and hand-off lag.  Off when several forwards
        # are in flight anyway (bench.py's timed region sets 0).
        self.overlap_chunks = int(os.environ.get("SFSN_OVERLAP_CHUNKS", "3"))
        self.overlap_fracs = None  # optional explicit chunk lengths (fractions of T) of the overlapped schedule
        if os.environ.get("SFSN_OVERLAP_FRACS"):
            self.overlap_fracs = [float(v) for v in os.environ["SFSN_OVERLAP_FRACS"].split(",")]
        # frames of the first chunk: -1 = 0.24 T (measured, scripts/exp_overlap.py: 3.55 -> 3.40 ms at B=64, the same 3-4 % at B=4..32 and
        # T=500; 64..128 frames and 0.32 T gain nothing), 0 = equal chunks
        self.overlap_first = int(os.environ.get("SFSN_OVERLAP_FIRST", "-1"))
        # True: forward_stft() returns only after the error words of its stack launches have been read (a launch whose bounded
        # hand-off wait expired raises HERE instead of at the next forward); False: non-blocking, see check_stack_errors()
        self.strict_errors = os.environ.get("SFSN_STRICT_ERRORS", "0") == "1"
        self._ov_streams = None
        # overlapped schedule, EXPERIMENT (off): the full-band model's features and layer-0 input products of ALL chunks go out at once
        # on a stream of their own (nothing gates them), so its stack launches follow one another without the time-parallel kernels
        # between them.  Bit-identical, but 2.97 instead of 2.80 ms per forward at B = 64, T = 1000: an input term produced a
        # millisecond before it is read has left the 256 MB Infinity Cache (the sub-band kernels move > 1 GB in between), and the
        # full-band scan -- two ring slots deep at H = 320 -- runs 1.9 instead of 1.6 us per frame from HBM (profiles/EXPERIMENTS.md)
        self.overlap_prep_ahead = os.environ.get("SFSN_PREP_AHEAD", "0") != "0"
        # overlapped schedule, round 6 (profiles/r06_strict_timeline.txt: the full-band chain gates, ~100 us of small kernels between its
        # stack launches, and they run beside the sub-band stream's feature / input-product launches that the same event releases):
        #   prep_next   -- the full-band model's features + layer-0 input product of chunk c + 1 go out BEFORE chunk c's projection (which
        #                  is what releases the sub-band stream): they run alone (27 instead of 82 us) and the next stack launch follows
        #   post_stream -- the sub-band epilogue (sfsn_proj_deepfilter) of chunk c on a third stream: the sub-band stream goes on with
        #                  chunk c + 1's features as soon as the pair launch is done
        # Both bit-identical and both OFF: measured (scripts/exp_ovsched_r06.sh, three interleaved rounds, strict forward API / lean)
        # default 2.41-2.45 / 2.36-2.38 ms; prep_next 2.42-2.46 / 2.34-2.36 (the small kernels leave the chain, the earlier stack launch
        # meets the sub-band stream's feature launches instead); post_stream 2.54 / 2.46-2.48 (the epilogue's 0.84 GB beside the next
        # pair launch costs that launch more than the 100-136 us it takes off the stream); both 2.60-2.64 / 2.53-2.56.
        self.overlap_prep_next = os.environ.get("SFSN_OV_PREP_NEXT", "0") != "0"
        self.overlap_post_stream = os.environ.get("SFSN_OV_POST_STREAM", "0") != "0"
        self.stack_wide = True
        self._stack_scratch: List[torch.Tensor] = []
        self._stack_err_pending: List[tuple] = []  # (event, pinned copy of a launch's error word): polled at the next forward
        self._defer_err = None  # overlapped schedule: [(what, scratch)] of the running forward's stack launches (one copy per forward)
        self._err_stream = None
        self.hw_queues = _check_hw_queues()
        self._last_forward: Dict[int, torch.cuda.Event] = {}  # per calling stream: recorded behind its most recent forward

    # ---------------------------------------------------------------------------------------------
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    class _Timed:
        """HIP events on the launch stream around a group of launches; no-op unless ``engine.timers`` is a dict."""

        def __init__(self, eng, tag, stream):
            self.eng, self.tag, self.stream = eng, tag, stream

        def __enter__(self):
            self.on = self.eng.timers is not None and (self.eng.timer_tags is None or self.tag in self.eng.timer_tags)
            if self.on:
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e1 = torch.cuda.Event(enable_timing=True)
                self.e0.record(self.eng._tstream(self.stream))
            return self

        def __exit__(self, *exc):
            if self.on:
                self.e1.record(self.eng._tstream(self.stream))
                self.eng.timers.setdefault(self.tag, []).append((self.e0, self.e1))
            return False

    def timed(self, tag, st=None):
        return Engine._Timed(self, tag, st)

    def _tstream(self, st):
        """torch stream object for a raw stream handle handed to the C ABI (None = current stream)."""
        if st is None:
            return torch.cuda.current_stream(self.device)
        return self._stream_objs.get(st.value, torch.cuda.current_stream(self.device))

    def timer_summary(self) -> dict:
        """{tag: {mean_ms, n}} per launch group (one group = the launches of one call site in one forward)."""
        if not self.timers:
            return {}
        torch.cuda.synchronize(self.device)
        out = {}
        for tag, evs in self.timers.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[tag] = dict(mean_ms=float(np.mean(ms)), min_ms=float(np.min(ms)), n=len(ms))
        return out

    @staticmethod
    def _nbytes(obj) -> int:
        if torch.is_tensor(obj):
            return obj.numel() * obj.element_size()
        if isinstance(obj, dict):
            return sum(Engine._nbytes(v) for v in obj.values())
        if isinstance(obj, (list, tuple)):
            return sum(Engine._nbytes(v) for v in obj)
        return 0

    def _workspace(self, key, make):
        """Scratch cache: least-recently-used entries are dropped once the total exceeds `ws_budget_bytes` (the entry being
        asked for is never dropped).  Entries are keyed on (tag, rows, stream) and sized by CAPACITY (see _alloc_stack): a
        stream that sees clips of many lengths keeps one buffer set, grown to the longest."""
        ws = self._ws.get(key)
        if ws is None:
            ws = self._ws[key] = make()
            total = sum(self._nbytes(v) for v in self._ws.values())
            for k in list(self._ws.keys()):
                if total <= self.ws_budget_bytes or k == key:
                    continue
                total -= self._nbytes(self._ws.pop(k))
        else:
            self._ws[key] = self._ws.pop(key)  # most recently used last (dicts keep insertion order)
        return ws

    # ---- per-stage launch helpers: every one works on the frame range [t0, t0+nt) and on an explicit stream, so
    #      that the same code serves the sequential path (one chunk, current stream) and the time-pipelined path ----
    def _stage_input(self, seqs, l, srcs, zins, t0, nt, st, tag):
        """zin (chunk-local, [nt, R, G*H]) = layer input . W_ih^T + bias_ih.  srcs: x (l = 0, full [T,R,I] fp32) or the
        previous layer's int8 spikes (l >= 1, full [T,R,HP])."""
        L = self.lib
        with self.timed(("inproj:" if l == 0 else "spikeproj_in:") + tag, st):
            jobs = []  # (x or s, w, dq, bias, z, M, K, N, ld) per (sequence model, gate)
            for seq, src, z in zip(seqs, srcs, zins):
                cell, H, G, R = seq.cells[l], seq.H, seq.cells[l].G, src.shape[1]
                M = nt * R
                for g in range(G):
                    zp = z.data_ptr() + g * H * 4
                    bp = cell.bias.data_ptr() + g * H * 4
                    if l == 0:
                        jobs.append((src.data_ptr() + t0 * R * seq.I * 4, cell.w_ih_f32.data_ptr() + g * H * seq.I * 4, None, bp, zp, M, seq.I, H, G * H))
                    else:
                        pk, dq = cell.w_ih_q[g]
                        jobs.append((src.data_ptr() + t0 * R * src.shape[2], pk.data_ptr(), dq.data_ptr(), bp, zp, M, H, H, G * H))
            if self.merge_products and 1 < len(jobs) <= _lib.MAX_SEGMENTS:
                # the products of a stage in ONE launch (a block range per job, every job its own tiling: the same results)
                if l == 0:
                    arr = (InProjJob * len(jobs))()
                    for a, (x_, w_, _, b_, z_, M, K, N, ld) in zip(arr, jobs):
                        a.x, a.w, a.bias, a.z, a.M, a.K, a.N, a.ldz = x_, w_, b_, z_, M, K, N, ld
                    rc = L.sfsn_input_proj_f32_multi(arr, len(jobs), st)
                else:
                    arr = (ProjJob * len(jobs))()
                    for a, (s_, w_, dq_, b_, z_, M, K, N, ld) in zip(arr, jobs):
                        a.s, a.w_packed, a.w_dq, a.bias, a.y, a.M, a.K, a.N, a.ldy = s_, w_, dq_, b_, z_, M, K, N, ld
                    rc = L.sfsn_spike_proj_multi(arr, len(jobs), st)
                if rc == 0:
                    self.launches["merged_products"] = self.launches.get("merged_products", 0) + 1
                    return
                if rc != _lib.SFSN_EUNSUPPORTED:
                    check(rc, "sfsn_input_proj_f32_multi" if l == 0 else "sfsn_spike_proj_multi")
            P = ctypes.c_void_p
            for a_, w_, dq_, b_, z_, M, K, N, ld in jobs:
                if l == 0:
                    check(L.sfsn_input_proj_f32(P(a_), P(w_), P(b_), P(z_), M, K, N, ld, st), "sfsn_input_proj_f32")
                else:
                    check(L.sfsn_spike_proj(P(a_), P(w_), P(dq_), P(b_), P(z_), M, K, N, ld, st), "sfsn_spike_proj")

    def _stage_featproj(self, fg, seqs, idx, zins, stft_ri, fb_proj, dims, t0, nt, zero, need_x, st, tag) -> bool:
        """Features of every group of `fg` AND the layer-0 input term of the groups `idx` (zins: their chunk-local buffers) in one
        launch.  False (nothing launched) when the library would not run the pair bit-identically (sfsn_features_proj's
        SFSN_EUNSUPPORTED, separate gate weights): the caller issues the two calls then.  need_x False: the rows of the groups in
        `idx` are not written."""
        B, F, T, FB, fdrc = dims
        n = len(seqs)
        if not self.fuse_featproj or not idx or n > _lib.MAX_GROUPS or any(seqs[i].cells[0].G != 1 for i in idx):
            return False
        jobs = (FeatProjJob * n)()
        for i in range(n):
            jobs[i].feat = fg[i]
            if i in idx:
                seq, z = seqs[i], zins[idx.index(i)]
                cell = seq.cells[0]
                jobs[i].w, jobs[i].bias, jobs[i].z = cell.w_ih_f32.data_ptr(), cell.bias.data_ptr(), z.data_ptr()
                jobs[i].H, jobs[i].ldz = seq.H, seq.H
                if not need_x:
                    jobs[i].feat.x = None
        with self.timed("featproj:" + tag, st):
            rc = self.lib.sfsn_features_proj(_ptr(stft_ri), None if fb_proj is None else _ptr(fb_proj), B, F, T, FB, fdrc, jobs, n, t0, nt,
                                             None if zero is None else _ptr(zero), 0 if zero is None else zero.numel() * 4, st)
        if rc == _lib.SFSN_EUNSUPPORTED:
            return False
        check(rc, "sfsn_features_proj")
        self.launches["featproj"] = self.launches.get("featproj", 0) + 1
        return True

    def _stage_scan(self, seqs, l, zins, states, spks, s8s, mems, t0, nt, st, tag, rpw, cnts=None):
        L, spec = self.lib, self.spec
        H = seqs[0].H
        HP = (H + 63) // 64 * 64
        segs = (ScanSegment * len(seqs))()
        for i, seq in enumerate(seqs):
            cell, sg, R = seq.cells[l], segs[i], s8s[i].shape[1]
            sg.zin, sg.w_hh, sg.w_dq, sg.bias = _ptr(zins[i]), _ptr(cell.w_hh_q), _ptr(cell.w_hh_dq), _ptr(cell.bias)
            sg.bn_alpha, sg.bn_beta, sg.h_state, sg.c_state = _ptr(cell.alpha), _ptr(cell.beta), _ptr(states[i][0]), _ptr(states[i][1])
            sg.spikes_f32 = None if spks[i] is None else ctypes.c_void_p(spks[i].data_ptr() + t0 * R * H * 4)
            sg.membrane = None if mems[i] is None else ctypes.c_void_p(mems[i].data_ptr() + t0 * R * H * 4)
            sg.spikes_i8 = ctypes.c_void_p(s8s[i].data_ptr() + t0 * R * HP)
            sg.R = R
            sg.spike_count = None if cnts is None else _ptr(cnts[i])
        with self.timed("scan:" + tag, st):
            rc = _lib.SFSN_EUNSUPPORTED
            if (self.split_scan and not spec.shared and H > 256 and len(seqs) == 1 and nt >= 16
                    and not torch.cuda.is_current_stream_capturing()):
                # separate gate weights too large for one compute unit (baseline_xl's full-band model): the tiles of a row block split
                # over workgroups with resident weights and a spike exchange per step, instead of streaming all of W_hh every step
                # (10 us per step): same results; its scratch (tagged exchange words, error word first) is zeroed per call.  Not for a
                # few frames (a streaming hop: the workgroups' weights are loaded per call) and not under graph capture (the scratch
                # buffer is this call's)
                R0 = s8s[0].shape[1]
                stream = self._tstream(st)
                with torch.cuda.stream(stream):
                    # allocated AND zeroed under the launch's stream: the block then comes from that stream's pool, so its previous use
                    # is ordered before the fill (round-5 advisor finding: a block freed on the caller's stream could still be in use
                    # there when the stage stream zeroed it -- a corrupted epoch tag ends in a bounded-spin error)
                    scr = torch.zeros((L.sfsn_scan_split_scratch_bytes(R0, H) // 4,), dtype=torch.int32, device=self.device)
                rc = L.sfsn_gsn_layer_scan_split(segs, 1, nt, H, 0, _ptr(scr), scr.numel() * 4, st)
                if rc == 0:
                    with torch.cuda.stream(stream):
                        pin = torch.empty((1,), dtype=torch.int32, pin_memory=True)
                        pin.copy_(scr[:1], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(stream)
                    self._stack_err_pending.append((ev, pin, f"split scan {tag} H={H}", scr))
                    self.launches["split_scan"] = self.launches.get("split_scan", 0) + 1
                    return
                if rc != _lib.SFSN_EUNSUPPORTED:
                    check(rc, "sfsn_gsn_layer_scan_split")
            if self.weight_bits == 16 and self.w16_fast:  # the two-plane scan where the library has it (same results)
                rc = L.sfsn_gsn_layer_scan_w16(segs, len(seqs), nt, H, int(spec.shared), rpw, st)
                if rc not in (0, _lib.SFSN_EUNSUPPORTED):
                    check(rc, "sfsn_gsn_layer_scan_w16")
            if rc != 0:
                check(L.sfsn_gsn_layer_scan(segs, len(seqs), nt, H, int(spec.shared), rpw, st), "sfsn_gsn_layer_scan")

    def _fusable(self, seqs, rpw, want_membrane) -> bool:
        """Layers >= 1 can take their input term inside the scan (sfsn_gsn_layer_scan_fused): shared gates, 128 < H <= 256,
        16 rows per workgroup -- the launch geometry of a full chip, where the scan is HBM-bound and the saved round trip of
        the fp32 input term is pure gain; with 4 rows per workgroup (one forward alone) the doubled MFMA work would cost more."""
        return bool(self.fuse_input and self.spec.shared and 128 < seqs[0].H <= 256 and rpw == 16 and not want_membrane)

    def _fusable_x(self, seq, x, rpw, want_membrane) -> bool:
        """Layer 0 of a group can take its real-valued input product inside the scan (sfsn_gsn_layer_scan_fused_x) when, on
        top of _fusable's conditions, the feature rows are narrow (even I <= 64: W_ih pieces fit in registers) and the row
        count is a multiple of 16 (a step's rows are copied as one flat block)."""
        return bool(self.fuse_input and self.spec.shared and 128 < seq.H <= 256 and rpw == 16 and not want_membrane
                    and seq.I % 2 == 0 and seq.I <= 64 and x.shape[1] % 16 == 0)

    def _stage_scan_fused_x(self, seqs, xs_, states, spks, s8s, t0, nt, st, tag, cnts=None):
        """Layer 0 of the given sequence models, input product inside the scan."""
        L = self.lib
        H = seqs[0].H
        HP = (H + 63) // 64 * 64
        segs = (ScanSegment * len(seqs))()
        fin = (FusedX * len(seqs))()
        for i, (seq, x) in enumerate(zip(seqs, xs_)):
            cell, sg, R = seq.cells[0], segs[i], x.shape[1]
            sg.zin, sg.w_hh, sg.w_dq, sg.bias = None, _ptr(cell.w_hh_q), _ptr(cell.w_hh_dq), _ptr(cell.bias)
            sg.bn_alpha, sg.bn_beta, sg.h_state, sg.c_state = _ptr(cell.alpha), _ptr(cell.beta), _ptr(states[i][0]), _ptr(states[i][1])
            sg.spikes_f32 = None if spks[i] is None else ctypes.c_void_p(spks[i].data_ptr() + t0 * R * H * 4)
            sg.membrane = None
            sg.spikes_i8 = ctypes.c_void_p(s8s[i].data_ptr() + t0 * R * HP)
            sg.R = R
            sg.spike_count = None if cnts is None else _ptr(cnts[i])
            fin[i].x = x.data_ptr() + t0 * R * seq.I * 4
            fin[i].w_ih, fin[i].I = cell.w_ih_f32.data_ptr(), seq.I
        self.launches["fused_x"] = self.launches.get("fused_x", 0) + 1
        with self.timed("scanx:" + tag, st):
            check(L.sfsn_gsn_layer_scan_fused_x(segs, fin, len(seqs), nt, H, st), "sfsn_gsn_layer_scan_fused_x")

    def _stage_scan_fused(self, seqs, l, states, spks, s8s, t0, nt, st, tag, cnts=None):
        L = self.lib
        H = seqs[0].H
        HP = (H + 63) // 64 * 64
        segs = (ScanSegment * len(seqs))()
        fin = (FusedInput * len(seqs))()
        for i, seq in enumerate(seqs):
            cell, sg, R = seq.cells[l], segs[i], s8s[l][i].shape[1]
            pk, dq = cell.w_ih_q[0]
            sg.zin, sg.w_hh, sg.w_dq, sg.bias = None, _ptr(cell.w_hh_q), _ptr(cell.w_hh_dq), _ptr(cell.bias)
            sg.bn_alpha, sg.bn_beta, sg.h_state, sg.c_state = _ptr(cell.alpha), _ptr(cell.beta), _ptr(states[i][0]), _ptr(states[i][1])
            sg.spikes_f32 = None if spks[i] is None else ctypes.c_void_p(spks[i].data_ptr() + t0 * R * H * 4)
            sg.membrane = None
            sg.spikes_i8 = ctypes.c_void_p(s8s[l][i].data_ptr() + t0 * R * HP)
            sg.R = R
            sg.spike_count = None if cnts is None else _ptr(cnts[i])
            fin[i].spikes_in = s8s[l - 1][i].data_ptr() + t0 * R * HP
            fin[i].w_ih, fin[i].w_ih_dq = pk.data_ptr(), dq.data_ptr()
        self.launches["fused"] = self.launches.get("fused", 0) + 1
        with self.timed("scanf:" + tag, st):
            check(L.sfsn_gsn_layer_scan_fused(segs, fin, len(seqs), nt, H, st), "sfsn_gsn_layer_scan_fused")

    def _stack_choice(self, seqs, Rs, want_membrane):
        """(use the stack launch?, wide flavour?, rows per workgroup) for one stack of sequence models.

        sfsn_gsn_stack_scan runs all layers of a stack in one launch, layer l+1 trailing layer l.  It wins where one layer's
        workgroups leave most of the chip idle; where a single layer already fills the chip at 4 rows per workgroup the layers
        side by side get half the CUs each and the input terms' traffic through L2 / HBM doubles up, and full-chip launches
        in a row are faster.  Measured on MI355X, T=1000, baseline_m sizes (scripts/exp_stack.py), sub-band stack
        (per-layer scans + input products / narrow stack / wide stack, ms): B=4 1.9 / 1.7 / 1.2, B=16 1.9 / 1.7 / 1.2,
        B=32 2.0 / 1.7 / 1.95, B=64 1.66 / 1.83 / 2.33; full-band stack (H=320, B rows) 2.26 -> 1.25 at every B.
        `stack_scan`: "auto" (the rule below), True (always, `stack_rows_per_wg` / `stack_wide` as set), False (never)."""
        nl, H = len(seqs[0].cells), seqs[0].H
        ok = bool(self.stack_scan and self.spec.shared and not want_membrane and nl >= 2
                  and len(seqs) * (1 + (nl - 1) * 2) <= 24)
        if not ok:
            return False, False, 0
        tag_rpw = self.stack_rows_per_wg
        if self.stack_scan != "auto":
            return True, self.stack_wide, None
        rows = sum(Rs)
        n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
        if H > 256:  # the full-band model: few rows, PROJ + gated scan roles
            rp = self.stack_rows_fb_auto
            return (rows + rp - 1) // rp * nl + (rows + 15) // 16 <= n_cu, False, rp
        # round 4: H <= 224 stacks without input-term buffers run their layers >= 1 as FUSED3 roles (the input product inside the
        # 8-row IO-wave scan): all layers side by side in one launch, no fp32 input term for the layers >= 1.  Every workgroup of
        # the launch has to be resident beside the full-band stack of the same forward (<= 40 workgroups), and the forward has
        # to be alone on the chip -- `rows_per_wg` (0 or 8 for the sub-band models) says so: bench.py's timed region, with a
        # dozen forwards in flight, sets 16 and keeps its per-layer launches.  Measured at B = 64, T = 1000 (scripts/exp_pair.py):
        # 1.03 ms against 1.43 (scan3 at 4 rows + sfsn_spike_proj), 1.72 (8-wave FUSED roles), 2.4 (PROJ roles).
        wgs8 = nl * sum((R + 7) // 8 for R in Rs)
        # (the launch assumes the forward has the chip to itself: its 2 x 104 workgroups wait for each other inside the launch.  A second
        #  forward of this engine still in flight on another stream (its end-of-forward event has not fired) takes the per-layer
        #  launches instead: round-4 advisor finding.  Work of OTHER processes or libraries cannot be seen from here; the waits
        #  are bounded and reported.)
        me = torch.cuda.current_stream(self.device).cuda_stream
        alone = all(ev.query() for k, ev in self._last_forward.items() if k != me)
        if self.pair_scan and alone and H <= 224 and self.rows_per_wg[1] in (0, 8) and wgs8 <= n_cu - 40:
            return True, False, 8
        if rows <= n_cu:       # every layer's workgroups at 8 rows + the PROJ workgroups fit several times over
            return True, True, 8
        if rows <= 2 * n_cu:   # 8 rows per workgroup: both layers side by side still fit
            return True, False, 8
        return False, False, 0

    def _stack_x_groups(self, seqs, xs_, nt, wide, rpw_stack, want_membrane=False):
        """Groups whose LAYER 0 takes its real-valued input product inside the stack launch (FUSEDX3 role, sfsn_gsn_stack_scan_x):
        the layout H <= 224 stacks without input-term buffers get (8 rows per workgroup), even I <= 64, whole 8-row blocks, and at
        least 64 row-frames (below that sfsn_input_proj_f32 itself takes its fp32-MFMA form: the schedules would differ in the last bit)."""
        if not (self.fuse_input and self.pair_scan and self.spec.shared and not wide and rpw_stack == 8 and not want_membrane
                and seqs[0].H <= 224 and len(seqs[0].cells) >= 2 and not os.environ.get("SFSN_STACK_FUSED8")
                and not os.environ.get("SFSN_SCAN_V2")):  # (the library refuses the FUSEDX3 role under either switch)
            return []
        return [i for i, (seq, x) in enumerate(zip(seqs, xs_))
                if seq.I % 2 == 0 and seq.I <= 64 and x.shape[1] % 8 == 0 and nt * x.shape[1] >= 64]

    def _stage_stack(self, seqs, d, t0, nt, st, tag, wide, rpw_stack, lag=None, scratch=None, xs_=None, xg=()):
        """Every layer of the given sequence models in one launch (layer 0's input term is already in d["zin"][0]; for the groups
        in `xg` the launch forms it itself from the feature rows xs_[i])."""
        L = self.lib
        H, nl, ns = seqs[0].H, len(seqs[0].cells), len(seqs)
        HP = (H + 63) // 64 * 64
        segs = (ScanSegment * (nl * ns))()
        fin = (FusedInput * (nl * ns))()
        fx = (FusedX * ns)() if xg else None
        for i in xg:
            x, cell = xs_[i], seqs[i].cells[0]
            fx[i].x, fx[i].w_ih, fx[i].I = x.data_ptr() + t0 * x.shape[1] * seqs[i].I * 4, cell.w_ih_f32.data_ptr(), seqs[i].I
        rows = 0
        for l in range(nl):
            for i, seq in enumerate(seqs):
                cell, sg, R = seq.cells[l], segs[l * ns + i], d["s8"][l][i].shape[1]
                rows += R if l == 0 else 0
                # layers >= 1: an input-term buffer selects the wide flavour for H <= 256 (16-wave scans fed by PROJ workgroups
                # of the same launch); without it the 8-wave fused-input roles run
                sg.zin = _ptr(d["zin"][l][i]) if ((l == 0 and i not in xg) or (l > 0 and (H > 256 or wide))) else None
                sg.w_hh, sg.w_dq, sg.bias = _ptr(cell.w_hh_q), _ptr(cell.w_hh_dq), _ptr(cell.bias)
                sg.bn_alpha, sg.bn_beta = _ptr(cell.alpha), _ptr(cell.beta)
                sg.h_state, sg.c_state = _ptr(d["states"][l][i][0]), _ptr(d["states"][l][i][1])
                spk = d["spk"][l][i]
                sg.spikes_f32 = None if spk is None else ctypes.c_void_p(spk.data_ptr() + t0 * R * H * 4)
                sg.membrane = None
                sg.spikes_i8 = ctypes.c_void_p(d["s8"][l][i].data_ptr() + t0 * R * HP)
                sg.R = R
                sg.spike_count = _ptr(d["cnt"][l][i]) if d.get("cnt") is not None else None
                if l > 0:
                    pk, dq = cell.w_ih_q[0]
                    fin[l * ns + i].spikes_in = d["s8"][l - 1][i].data_ptr() + t0 * R * HP
                    fin[l * ns + i].w_ih, fin[l * ns + i].w_ih_dq = pk.data_ptr(), dq.data_ptr()
        nbytes = L.sfsn_stack_scratch_bytes(nl, ns, rows)
        own_scratch = scratch is None
        if scratch is None:  # (a streaming session owns its own: its captured graph must not outlive a cache entry)
            def make():
                # zero-filled on the stream the kernel is launched on: on the overlapped schedule that is a side stream which
                # only waits for the fork event, and a fill enqueued on the caller's stream after that event would not be
                # ordered before the kernel that polls these counters (round-2 advisor finding)
                with torch.cuda.stream(self._tstream(st)):
                    return dict(t=torch.zeros((nbytes // 4,), dtype=torch.int32, device=self.device))
            scratch = self._workspace(("stack_scratch", tag, nl, ns, rows, torch.cuda.current_stream(self.device).cuda_stream,
                                       self._tstream(st).cuda_stream), make)["t"]
        assert scratch.numel() * 4 >= nbytes
        if not any(scratch is t for t in self._stack_scratch):
            self._stack_scratch.append(scratch)
        rp = rpw_stack if rpw_stack else self.stack_rows_per_wg[tag]
        rpw = (ctypes.c_int * nl)(*([rp] * nl))
        self.launches["stack"] = self.launches.get("stack", 0) + 1
        with self.timed("stack:" + tag, st):
            rc16 = _lib.SFSN_EUNSUPPORTED
            if self.weight_bits == 16 and self.w16_fast:  # the two-plane roles where the library has them (the pair layout): same results
                rc16 = L.sfsn_gsn_stack_scan_x_w16(segs, fin, fx, nl, ns, nt, H, rpw, self.stack_lag if lag is None else lag, _ptr(scratch), nbytes, st)
                if rc16 not in (0, _lib.SFSN_EUNSUPPORTED):
                    check(rc16, "sfsn_gsn_stack_scan_x_w16")
                if rc16 == 0:
                    self.launches["stack_w16"] = self.launches.get("stack_w16", 0) + 1
            if rc16 == 0:
                pass
            elif xg:
                check(L.sfsn_gsn_stack_scan_x(segs, fin, fx, nl, ns, nt, H, rpw, self.stack_lag if lag is None else lag, _ptr(scratch), nbytes, st),
                      "sfsn_gsn_stack_scan_x")
            else:
                check(L.sfsn_gsn_stack_scan(segs, fin, nl, ns, nt, H, rpw, self.stack_lag if lag is None else lag, _ptr(scratch), nbytes, st),
                      "sfsn_gsn_stack_scan")
        # the launch's error word (a bounded hand-off wait expired) travels to pinned host memory behind the launch; it is looked
        # at without blocking at the next forward (and by check_stack_errors): a failed launch cannot go unnoticed for long
        if not torch.cuda.is_current_stream_capturing():
            stream = self._tstream(st)
            what = f"{tag} rows={rows} frames={nt} wide={wide} rows_per_wg={rp} lag={self.stack_lag if lag is None else lag}"
            if self._defer_err is not None and own_scratch:
                # the overlapped schedule of a forward alone: the copy (a 4-byte blit + its completion) would sit between this launch
                # and the projection that follows it on the same stream, ~10 us on the forward's chain per launch -- the forward
                # collects its launches' words once, off the chain, when its streams have joined (_forward_stft)
                self._defer_err.append((what, scratch))
                return
            with torch.cuda.stream(stream):
                pin = torch.empty((1,), dtype=torch.int32, pin_memory=True)
                pin.copy_(scratch[:1], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(stream)
            self._stack_err_pending.append((ev, pin, what, scratch))

    def _poll_stack_errors(self, block: bool = False) -> None:
        keep = []
        for ev, pin, what, scratch in self._stack_err_pending:
            if block:
                ev.synchronize()
            if ev.query():
                if int(pin[0]) != 0:
                    self._stack_err_pending = []
                    self._clear_stack_scratch(scratch)
                    raise RuntimeError("sfsn_gsn_stack_scan: a layer-to-layer hand-off wait expired in an earlier launch "
                                       f"(that forward's results are invalid): {what}")
            else:
                keep.append((ev, pin, what, scratch))
        self._stack_err_pending = keep

    def _clear_stack_scratch(self, scratch) -> None:
        """After a failed launch: the error word is sticky by design (nobody clears it on the device) and the progress counters
        of a launch that gave up are not all zeroed by its last workgroup -- reset both, or every later launch on this scratch
        buffer would report the old failure (round-2 advisor finding)."""
        torch.cuda.synchronize(self.device)
        for sc in (scratch if isinstance(scratch, (list, tuple)) else [scratch]):
            sc.zero_()
        torch.cuda.synchronize(self.device)

    def check_stack_errors(self) -> None:
        """Raise if a hand-off wait of a stack launch expired (synchronises; tests and bench call it after a forward)."""
        torch.cuda.synchronize(self.device)
        self._poll_stack_errors(block=True)
        for t in self._stack_scratch:
            if int(t[0].item()) != 0:
                self._clear_stack_scratch(t)
                raise RuntimeError("sfsn_gsn_stack_scan: a layer-to-layer hand-off wait expired (results invalid)")

    def _stage_proj(self, seqs, s8s, projs, t0, nt, st, tag):
        L = self.lib
        with self.timed("proj:" + tag, st):
            if self.merge_products and 1 < len(seqs) <= _lib.MAX_SEGMENTS:  # the groups' projections in ONE launch
                arr = (ProjJob * len(seqs))()
                for a, seq, s8, y in zip(arr, seqs, s8s, projs):
                    R = s8.shape[1]
                    a.s, a.w_packed, a.w_dq, a.bias = s8.data_ptr() + t0 * R * s8.shape[2], seq.proj_q.data_ptr(), seq.proj_dq.data_ptr(), seq.proj_b.data_ptr()
                    a.y, a.M, a.K, a.N, a.ldy = y.data_ptr() + t0 * R * seq.P * 4, nt * R, seq.H, seq.P, seq.P
                rc = L.sfsn_spike_proj_multi(arr, len(seqs), st)
                if rc == 0:
                    self.launches["merged_products"] = self.launches.get("merged_products", 0) + 1
                    return
                if rc != _lib.SFSN_EUNSUPPORTED:
                    check(rc, "sfsn_spike_proj_multi(proj)")
            for seq, s8, y in zip(seqs, s8s, projs):
                R = s8.shape[1]
                check(L.sfsn_spike_proj(ctypes.c_void_p(s8.data_ptr() + t0 * R * s8.shape[2]), _ptr(seq.proj_q), _ptr(seq.proj_dq),
                                        _ptr(seq.proj_b), ctypes.c_void_p(y.data_ptr() + t0 * R * seq.P * 4), nt * R, seq.H, seq.P, seq.P,
                                        st), "sfsn_spike_proj(proj)")

    def _stage_projdf(self, seqs, s8s, projs, ri, enh_ri, enh_mag, dims, t0, nt, st, write_proj=True) -> bool:
        """Projection of every group's last-layer spikes + deep filter + pass-through + |.| in one launch (sfsn_proj_deepfilter).
        False (nothing launched): the library does not cover the shapes, the caller issues _stage_proj and sfsn_deepfilter."""
        if not self.fuse_projdf:
            return False
        B, F, T, S = dims
        spec = self.spec
        arr = (ProjDfGroup * len(seqs))()
        for g, (a, seq, s8, y) in enumerate(zip(arr, seqs, s8s, projs)):
            a.spikes_i8, a.w_packed, a.w_dq, a.bias = s8.data_ptr(), seq.proj_q.data_ptr(), seq.proj_dq.data_ptr(), seq.proj_b.data_ptr()
            a.proj = y.data_ptr() if write_proj else None
            a.n_units, a.fc, a.df = spec.units(g), spec.ctr[g], spec.df[g]
        with self.timed("projdf", st):
            rc = self.lib.sfsn_proj_deepfilter(_ptr(ri), B, F, T, S, seqs[0].H, arr, len(seqs), _ptr(enh_ri), _ptr(enh_mag), t0, nt, st)
        if rc == _lib.SFSN_EUNSUPPORTED:
            return False
        check(rc, "sfsn_proj_deepfilter")
        self.launches["projdf"] = self.launches.get("projdf", 0) + 1
        return True

    def _zero_states(self, Rs, H, nl, flat=None):
        """(h, c) state pairs of a stack as views of one flat buffer; ``flat`` given = a slice of the forward's state buffer, which the
        forward's first feature launch zeroes (sfsn_features_z) -- otherwise a fill of its own."""
        if flat is None:
            flat = torch.zeros((2 * nl * sum(Rs) * H,), dtype=torch.float32, device=self.device)
        out, pos = [], 0
        for _ in range(nl):
            row = []
            for R in Rs:
                h = flat[pos:pos + R * H].view(R, H)
                c = flat[pos + R * H:pos + 2 * R * H].view(R, H)
                pos += 2 * R * H
                row.append((h, c))
            out.append(row)
        return out

    def _alloc_stack(self, seqs, Rs, T, nt_max, want_layers, want_membrane, tag, state_flat=None, counts=None):
        """Per-forward tensors of a stack of sequence models sharing (H, L): API outputs are fresh, scratch is cached."""
        dev, H, G, nl = self.device, seqs[0].H, seqs[0].cells[0].G, len(seqs[0].cells)
        HP = (H + 63) // 64 * 64
        key = (tag, tuple(Rs), torch.cuda.current_stream(dev).cuda_stream)
        old = self._ws.get(key)
        if old is not None and (old["T"] < T or old["nt"] < nt_max):  # grow: one buffer set per (tag, rows, stream), sized for the
            T_cap, nt_cap = max(T, old["T"]), max(nt_max, old["nt"])  # longest clip seen -- not one set per clip length
            del self._ws[key], old
        else:
            T_cap, nt_cap = T, nt_max
        ws = self._workspace(key, lambda: dict(
            T=T_cap, nt=nt_cap,
            zin=[[torch.empty((nt_cap, R, G * H), dtype=torch.float32, device=dev) for R in Rs] for _ in range(nl)],
            s8=[[torch.zeros((T_cap, R, HP), dtype=torch.int8, device=dev) for R in Rs] for _ in range(nl)]))  # (pad columns stay 0)
        f32 = dict(dtype=torch.float32, device=dev)
        return dict(
            zin=[[z[:nt_max] for z in zl] for zl in ws["zin"]], s8=[[b[:T] for b in bl] for bl in ws["s8"]],
            states=self._zero_states(Rs, H, nl, state_flat),  # zero init, modeling:100-106
            spk=[[torch.empty((T, R, H), **f32) if want_layers else None for R in Rs] for _ in range(nl)],
            mem=[[torch.empty((T, R, H), **f32) if want_membrane else None for R in Rs] for _ in range(nl)],
            proj=[torch.empty((T, R, seq.P), **f32) for seq, R in zip(seqs, Rs)],
            # in-scan spike counters (layer_outputs="counts"): one zeroed int64 per (layer, sequence model), or None
            cnt=None if counts is None else [[counts[l * len(Rs) + i] for i in range(len(Rs))] for l in range(nl)])

    def _feature_groups(self, which: str, xs, mu, sd=None):
        spec = self.spec
        if which == "fb":
            g = FeatureGroup()
            g.x, g.lo, g.n_units, g.ctr, g.nbr, g.ctr_fb, g.nbr_fb = _ptr(xs[0]), 0, 1, spec.fb_in, 0, 0, 0
            g.ln_eps = 1e-5
            if spec.laplace and spec.gaussian:
                g.norm, g.mu, g.ln_w = _lib.NORM_GAUSSIAN, _ptr(mu), _ptr(sd)  # (the clips' standard deviations travel in ln_w)
            elif spec.laplace:
                g.norm, g.mu = _lib.NORM_LAPLACE, _ptr(mu)
            elif spec.ln_fb:
                g.norm, g.ln_w, g.ln_b = _lib.NORM_LAYERNORM, _ptr(self.fb.ln_w), _ptr(self.fb.ln_b)
            else:
                g.norm = _lib.NORM_NONE
            arr = (FeatureGroup * 1)()
            arr[0] = g
            return arr
        arr = (FeatureGroup * spec.n_groups)()
        for i in range(spec.n_groups):
            g = arr[i]
            g.x = _ptr(xs[i]) if xs is not None else None
            g.lo, g.n_units, g.ctr, g.nbr = spec.cutoffs[i], spec.units(i), spec.ctr[i], spec.nbr[i]
            g.ctr_fb, g.nbr_fb, g.ln_eps = spec.ctr_fb[i], spec.nbr_fb[i], 1e-5
            if spec.laplace:
                g.norm = _lib.NORM_GAUSSIAN if spec.gaussian else _lib.NORM_LAPLACE
                g.mu = ctypes.c_void_p(mu.data_ptr() + i * mu.shape[1] * 4) if mu is not None else None
                if spec.gaussian:
                    g.ln_w = ctypes.c_void_p(sd.data_ptr() + i * sd.shape[1] * 4) if sd is not None else None
            elif spec.ln_sb:
                g.norm, g.ln_w, g.ln_b = _lib.NORM_LAYERNORM, _ptr(self.sb[i].ln_w), _ptr(self.sb[i].ln_b)
            else:
                g.norm = _lib.NORM_NONE
        return arr

    _hip = None

    def _masked_stream(self, cus: List[int]) -> torch.cuda.Stream:
        """A HIP stream restricted to the given compute units (hipExtStreamCreateWithCUMask), wrapped for torch.  The
        four recurrent scans of the pipelined schedule each get their own CUs and the time-parallel kernels the rest, so
        that a 2000-block GEMM cannot flood the chip and starve a 52-workgroup scan (or vice versa)."""
        if Engine._hip is None:
            Engine._hip = ctypes.CDLL("libamdhip64.so")
            Engine._hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
            Engine._hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
        n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
        words = (n_cu + 31) // 32
        mask = (ctypes.c_uint32 * words)()
        for c in cus:
            mask[c // 32] |= (1 << (c % 32))
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = Engine._hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), words, mask)
        if rc != 0 or not h.value:
            raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
        return torch.cuda.ExternalStream(h.value, device=self.device)

    def _pipeline_streams(self, n_fb: int, n_sb: int, fb_tiles: int, sb_tiles: int):
        """(scan streams per stage, G streams per stage): disjoint CU sets, created once per geometry."""
        key = (n_fb, n_sb, fb_tiles, sb_tiles)
        if getattr(self, "_pipe_key", None) != key:
            # CU ids are (32-bit mask word w, bit b) = w*32 + b; measured on MI355X (scripts/exp_cumask.py): a mask word
            # is one XCD, and masks are honoured when every word enables a contiguous bit range.  Pools are therefore bit
            # COLUMNS: the same range of CUs in every XCD -- each pool spans all eight L2s.
            n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
            words = n_cu // 32
            def columns(b0, b1, w0=0, w1=None):
                return [w * 32 + b for w in range(w0, words if w1 is None else w1) for b in range(b0, b1)]
            pools, col = [], 0
            for tiles in [fb_tiles] * n_fb + [sb_tiles] * n_sb:
                ncol = -(-tiles // words)  # bit columns needed: ceil(tiles / XCDs)
                pools.append(columns(col, col + ncol))
                col += ncol
            if col > 24:
                raise NotImplementedError("pipelined schedule: the scans would leave fewer than a quarter of the CUs to the other kernels")
            g_pool = columns(col, 32)
            if os.environ.get("SFSN_PLAIN_STREAMS"):  # diagnostic: no CU partitioning
                self._pipe_scan = [torch.cuda.Stream(device=self.device) for _ in pools]
                self._pipe_g = [torch.cuda.Stream(device=self.device) for _ in pools]
                self._pipe_key = key
                return self._pipe_scan, self._pipe_g
            self._pipe_scan = [self._masked_stream(p) for p in pools]
            self._pipe_g = [self._masked_stream(g_pool) for _ in pools]  # same CU pool, one queue per stage (a stage's
            # time-parallel kernels wait for its own scan; a shared queue would serialise all four stages)
            self._pipe_key = key
        return self._pipe_scan, self._pipe_g

    def _handle(self, stream: torch.cuda.Stream):
        self._stream_objs[stream.cuda_stream] = stream
        return ctypes.c_void_p(stream.cuda_stream)

    def _count_spikes(self, s8_lists, shapes, st):
        """One launch of sfsn_spike_count over every layer's int8 spike tensor -> list of SpikeSummary."""
        n = len(s8_lists)
        if n > _lib.MAX_COUNT_TENSORS:
            raise NotImplementedError(f"more than {_lib.MAX_COUNT_TENSORS} spike tensors to count")
        counts = torch.zeros((n,), dtype=torch.int64, device=self.device)
        arr = (CountTensor * n)()
        for i, t in enumerate(s8_lists):
            arr[i].spikes_i8, arr[i].n_bytes, arr[i].count = t.data_ptr(), t.numel(), counts.data_ptr() + 8 * i
        with self.timed("spike_count", st):
            check(self.lib.sfsn_spike_count(arr, n, st), "sfsn_spike_count")
        return [SpikeSummary(counts[i], shp) for i, shp in enumerate(shapes)]

    def forward_stft(self, stft: torch.Tensor, want_layers: bool = True, want_membrane: bool = False, pipeline: Optional[bool] = None,
                     want_counts: bool = False) -> dict:
        """See ``_forward_stft``; runs with this engine's device current (the C ABI launches on the calling thread's device)."""
        with torch.cuda.device(self.device):
            if self._stack_err_pending:
                self._poll_stack_errors()
            out = self._forward_stft(stft, want_layers, want_membrane, pipeline, want_counts)
            if not torch.cuda.is_current_stream_capturing():
                cur = torch.cuda.current_stream(self.device)
                ev = self._last_forward.get(cur.cuda_stream)
                if ev is None:
                    ev = self._last_forward[cur.cuda_stream] = torch.cuda.Event()
                ev.record(cur)  # (what _stack_choice asks: is another stream's forward still running?)
            if self.strict_errors and self._stack_err_pending:
                # results are only handed out once every stack launch of THIS forward is known to have completed its hand-offs
                # (one stream synchronisation per forward: for callers that keep several forwards in flight leave it off and
                # call check_stack_errors() at their own synchronisation points)
                self._poll_stack_errors(block=True)
            return out

    def _forward_stft(self, stft: torch.Tensor, want_layers: bool = True, want_membrane: bool = False, pipeline: Optional[bool] = None,
                      want_counts: bool = False) -> dict:
        """complex64 [B, n_fft/2+1, T] on the device -> dict(enh_stft [B,S,F,T] complex64, enh_mag [B,S,F,T],
        fb_all, sb_all (the reference's all_layer_outputs lists; spike entries are None when want_layers=False)).

        Schedule.  The four recurrent scans of a model (full-band layer 1/2, sub-band layer 1/2) are each a chain of T
        dependent steps that occupies only a fraction of the chip, and layer l+1 / the sub-band model need frame t of
        their producer only at frame t.  With ``pipeline`` (default: live front-end and T >= 2 chunks) the sequence is
        cut into chunks of ``pipeline_chunk`` frames and the stages run as a software pipeline on separate HIP streams
        chained by events -- stage s works on chunk c while stage s+1 works on chunk c-1 -- with the scan state carried in
        the ABI's h_state / c_state buffers.  Results are bit-identical to the sequential schedule (same kernels, same
        arithmetic; tested).  The frozen front-end's utterance-level Laplace mean needs the whole full-band output, so
        only its two layer pairs overlap.
        """
        spec, L = self.spec, self.lib
        self._defer_err = None  # (a forward that raised half way must not leave the next one's error words uncollected)
        want_layers = want_layers or want_membrane  # membranes are a test output of the fp32-spike kernel variant
        if stft.device != self.device or stft.dtype != torch.complex64 or stft.dim() != 3:
            raise RuntimeError(f"expected a complex64 [B, F, T] tensor on {self.device}, got {stft.dtype} {tuple(stft.shape)} on {stft.device}")
        B, F, T = stft.shape
        if F != spec.n_fft // 2 + 1:
            raise ValueError(f"expected {spec.n_fft // 2 + 1} frequency bins, got {F}")
        for g in range(spec.n_groups):  # the reference's ValueError, modeling_spiking_fullsubnet.py:283-287
            lo, hi, c = spec.cutoffs[g], spec.cutoffs[g + 1], spec.ctr[g]
            if (hi - lo) % c != 0:
                raise ValueError(f"Number of frequency bins must be divisible by the center frequency.GOT: ctr_freq={c}, "
                                 f"upper_cutoff_freq={hi}, lower_cutoff_freq={lo}")
        dev = self.device
        main = torch.cuda.current_stream(dev)
        ri = torch.view_as_real(stft.contiguous())  # [B, F, T, 2] float32 view, no copy
        f32 = dict(dtype=torch.float32, device=dev)
        chunk = self.pipeline_chunk
        if pipeline is None:
            pipeline = self.pipeline_default and chunk > 0 and T >= 2 * chunk
        if spec.cum_laplace:
            pipeline = False  # (the running-mean state and its scratch are per forward, not per stage: single-stream order only)
        # sequential schedule: optionally still cut the sequence into chunks (single stream): the chunk-sized input-term
        # buffer is then produced and consumed while it is still in the 256 MB Infinity Cache instead of making a round
        # trip through HBM (745 MB per sub-band layer at B=64, T=1000)
        overlap = bool(not pipeline and self.overlap_chunks > 1 and not spec.laplace and not spec.cum_laplace and not (0 < self.seq_chunk < T)
                       and T >= 96 * self.overlap_chunks)
        if overlap and self.stack_scan is True:
            # forced stack launches for both models: side by side on two streams they must not ask for more workgroups than the
            # chip has compute units -- in-launch hand-offs rely on producers being resident (the "auto" rule never gets here)
            def blocks(seqs, Rs, tag):
                use, wide, rp = self._stack_choice(seqs, Rs, want_membrane)
                rp = rp or self.stack_rows_per_wg[tag]
                nl = len(seqs[0].cells)
                scan = sum(-(-R // rp) for R in Rs)
                proj = sum(-(-R // 16) for R in Rs) if (wide or seqs[0].H > 256) else 0
                return (nl * (scan + 7) // 8 * 8 + (nl - 1) * (proj + 7) // 8 * 8) if use else 0
            n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
            b_fb, b_sb = blocks([self.fb], [B], "fb"), blocks(self.sb, [B * spec.units(g) for g in range(spec.n_groups)], "sb")
            if b_fb and b_sb and b_fb + b_sb > n_cu:
                overlap = False
        if overlap:
            # a short first chunk: the sub-band models can only start once the full-band model has finished a chunk, so the first
            # one is the lead-in of the whole forward; the rest is cut evenly (the full-band model is ~2x faster per frame and
            # stays ahead)
            first = self.overlap_first if self.overlap_first >= 0 else int(round(0.24 * T / 8.0)) * 8
            first = min(max(int(first), 0), T // 2)
            fracs = self.overlap_fracs
            if (not fracs and self.overlap_first < 0 and self.overlap_chunks == 3 and T >= 256
                    and B * sum(spec.units(g) for g in range(spec.n_groups)) < 600):
                # round 5 (scripts/exp_chunks_ab*_r05.sh, interleaved A/B on two boxes): below ~600 sub-band rows (B <= 32 of
                # baseline_m) a LONGER lead-in chunk wins -- (.36,.32,.32) T: 2.03-2.06 / 2.13 / 2.34 ms at B = 4 / 16 / 32 against
                # 2.08-2.10 / 2.19 / 2.39-2.42 with 0.24 T.  At B = 64 the balance sits on a knife edge (full-band chunk c+1 must fit
                # beside sub-band chunk c): (.32,.33,.35) measured 2.65-2.69 on one box and 2.80-2.82 on the next against 2.70-2.76
                # for 0.24 T on both, so 0.24 T stays there.
                fracs = (0.36, 0.32, 0.32)
            if fracs:  # explicit chunk lengths as fractions of T (experiments: scripts/exp_forward_r04.py)
                cuts = [int(round(f * T / 8.0)) * 8 for f in fracs]
                lens = [c for c in cuts if c > 0]
                lens = lens[:-1] + [T - sum(lens[:-1])] if sum(lens[:-1]) < T else [T]
                bounds, t0 = [], 0
                for n_ in lens:
                    bounds.append((t0, n_))
                    t0 += n_
            elif first:
                rest = -(-(T - first) // (self.overlap_chunks - 1))
                bounds = [(0, first)] + [(t0, min(rest, T - t0)) for t0 in range(first, T, rest)]
            else:
                nt = -(-T // self.overlap_chunks)
                bounds = [(t0, min(nt, T - t0)) for t0 in range(0, T, nt)]
            nt_max = max(n for _, n in bounds)
        else:
            nt_max = chunk if pipeline else (self.seq_chunk if 0 < self.seq_chunk < T else T)
            bounds = [(t0, min(nt_max, T - t0)) for t0 in range(0, T, nt_max)]
        S, ng = spec.num_spks, spec.n_groups
        nl_fb, nl_sb = spec.fb_layers, spec.sb_layers

        # ---------------- tensors
        x_fb = torch.empty((T, B, spec.fb_in), **f32)
        xs = [torch.empty((T, B * spec.units(g), spec.sb_input_size(g)), **f32) for g in range(ng)]
        prep_ahead = bool(overlap and self.overlap_prep_ahead)
        # the zero initial state of every scan of the forward: ONE buffer, zeroed by the forward's first feature launch (every scan
        # depends on that launch through its input) -- no fill launch on the forward's chain
        n_fb = 2 * nl_fb * B * self.fb.H
        n_sb = 2 * nl_sb * sum(x.shape[1] for x in xs) * self.sb[0].H
        fold = os.environ.get("SFSN_ZERO_FOLD", "1") != "0"  # (0: a fill launch of its own, for comparison)
        # layer_outputs="counts": the scans count the spikes they write (sfsn_scan_segment.spike_count) -- the int64 counters sit behind
        # the states in the same zeroed buffer (two floats each; the buffer's length stays a multiple of 16 bytes)
        in_scan = bool(want_counts and not want_layers and self.count_in_scan)
        n_cnt = (nl_fb + ng * nl_sb) if in_scan else 0
        n_cnt_f = (2 * n_cnt + 3) // 4 * 4
        state_flat = torch.empty((n_fb + n_sb + n_cnt_f,), **f32) if fold else torch.zeros((n_fb + n_sb + n_cnt_f,), **f32)
        zero_job = [state_flat] if fold else []
        cnt_all = state_flat[n_fb + n_sb:n_fb + n_sb + 2 * n_cnt].view(torch.int64) if in_scan else None
        fb = self._alloc_stack([self.fb], [B], T, T if prep_ahead else nt_max, want_layers, want_membrane, "fb", state_flat[:n_fb],
                               None if cnt_all is None else cnt_all[:nl_fb])
        sb = self._alloc_stack(self.sb, [x.shape[1] for x in xs], T, nt_max, want_layers, want_membrane, "sb", state_flat[n_fb:n_fb + n_sb],
                               None if cnt_all is None else cnt_all[nl_fb:])
        enh = torch.empty((B, S, F, T), dtype=torch.complex64, device=dev)
        enh_mag = torch.empty((B, S, F, T), **f32)
        enh_ri = torch.view_as_real(enh)
        dfg = (DfGroup * ng)()
        for g in range(ng):
            dfg[g].proj, dfg[g].n_units, dfg[g].fc, dfg[g].df = _ptr(sb["proj"][g]), spec.units(g), spec.ctr[g], spec.df[g]
        mu_fb = mu_sb = sd_fb = sd_sb = scratch = None
        if spec.laplace:
            mu_fb, mu_sb = torch.empty((1, B), **f32), torch.empty((ng, B), **f32)
            if spec.gaussian:
                sd_fb, sd_sb = torch.empty((1, B), **f32), torch.empty((ng, B), **f32)
            scratch = torch.empty(((5 if spec.gaussian else 1) * B * (F - 1 + spec.fb_proj) + 2,), **f32)
        fg_fb = self._feature_groups("fb", [x_fb], mu_fb, sd_fb)
        fg_sb = self._feature_groups("sb", xs, mu_sb, sd_sb)
        fb_proj = fb["proj"][0]

        # ---------------- streams: sequential = the current stream; pipelined = per stage one scan stream (own CUs) and one
        #                  stream for its time-parallel kernels (shared CU pool), chained by events
        n_stage = nl_fb + nl_sb
        post_stream = None
        if pipeline:
            rpw_fb, rpw_sb = 16, 16
            fb_tiles = (B + rpw_fb - 1) // rpw_fb
            sb_tiles = sum((x.shape[1] + rpw_sb - 1) // rpw_sb for x in xs)
            sstreams, gstreams = self._pipeline_streams(nl_fb, nl_sb, fb_tiles, sb_tiles)
            fork = torch.cuda.Event()
            fork.record(main)
            for s_ in sstreams + gstreams:
                s_.wait_event(fork)
        elif overlap:
            if self._ov_streams is None:
                self._ov_streams = {}
            if main.cuda_stream not in self._ov_streams:  # a pair per calling stream: forwards in flight stay independent
                nx = int(os.environ.get("SFSN_OV_XCD_SPLIT", "0"))
                if nx > 0:
                    # experiment (round 5): the full-band stream on the first `nx` XCDs only, the sub-band stream on the others (a mask
                    # word = one XCD, scripts/exp_cumask.py): the full-band stack's hand-offs then stay inside nx L2s
                    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
                    self._ov_streams[main.cuda_stream] = (self._masked_stream(list(range(0, 32 * nx))), self._masked_stream(list(range(32 * nx, n_cu))))
                else:
                    self._ov_streams[main.cuda_stream] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
            sa, sb_ = self._ov_streams[main.cuda_stream]
            if self.overlap_post_stream:
                if ("post", main.cuda_stream) not in self._ov_streams:
                    self._ov_streams[("post", main.cuda_stream)] = torch.cuda.Stream(device=dev)
                post_stream = self._ov_streams[("post", main.cuda_stream)]
            self._defer_err = []
            sstreams = gstreams = [sa] * nl_fb + [sb_] * nl_sb
            rpw_fb, rpw_sb = self.rows_per_wg
            fork = torch.cuda.Event()
            fork.record(main)
            sa.wait_event(fork)
            sb_.wait_event(fork)
            if post_stream is not None:
                post_stream.wait_event(fork)
            if prep_ahead:
                if ("aux", main.cuda_stream) not in self._ov_streams:
                    self._ov_streams[("aux", main.cuda_stream)] = torch.cuda.Stream(device=dev)
                aux_stream = self._ov_streams[("aux", main.cuda_stream)]
                aux_stream.wait_event(fork)
        else:
            sstreams = gstreams = [main] * n_stage
            rpw_fb, rpw_sb = self.rows_per_wg
        if rpw_sb == 0 and self.fuse_input and spec.shared and 128 < self.sb[0].H <= 256:
            # Many sub-band rows (baseline_l: 43 units per clip = 2752 rows at B = 64): left to itself the library spreads them at 8
            # rows per workgroup (344 workgroups: two rounds on 256 compute units, round 2's body for 16 tiles) behind a separate
            # layer-0 input product and a layer-2 spike product (5.6 GB of fp32 input terms written and read).  At 16 rows per workgroup
            # the launch is one round and the fused-input kernels apply (input products inside the scan, defined at 16 rows): said
            # explicitly here whenever 8 rows would not fit the chip in one round.
            n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
            rows_sb = sum(x.shape[1] for x in xs)
            if (rows_sb + 7) // 8 > n_cu:
                rpw_sb = 16
        staged = pipeline or overlap  # several streams chained by events
        hS = [self._handle(s_) for s_ in sstreams]
        hG = [self._handle(s_) for s_ in gstreams]

        def link(src, dst):
            """dst waits for everything enqueued on src so far (no-op on the sequential path)."""
            if staged and src is not dst:
                ev = torch.cuda.Event()
                ev.record(src)
                dst.wait_event(ev)

        # per-chunk completion events of the producers the next model / layer gates on
        def run_model(seqs, d, xs_, first, feat_fn, tag, rpw, post_fn, gate_events, fused_post=None):
            nl = len(seqs[0].cells)
            fused = self._fusable(seqs, rpw, want_membrane)
            done = []
            pick = lambda lst, idx: [lst[i] for i in idx]
            use_stack, wide, rpw_stack = self._stack_choice(seqs, [x.shape[1] for x in xs_], want_membrane)
            if not pipeline and use_stack:
                # all layers in one launch: features, layer 0's input term, the stack scan, the projection
                ahead = prep_ahead and gate_events is None and d["zin"][0][0].shape[0] >= T
                def prep(t0, nt, dz, st):
                    xg = self._stack_x_groups(seqs, xs_, nt, wide, rpw_stack or self.stack_rows_per_wg[tag], want_membrane)
                    zr = [i for i in range(len(seqs)) if i not in xg]
                    if not feat_fn(t0, nt, st, (zr, pick(dz["zin"][0], zr))) and zr:  # (True: one launch made the rows and the input terms)
                        self._stage_input(pick(seqs, zr), 0, pick(xs_, zr), pick(dz["zin"][0], zr), t0, nt, st, tag)
                    return xg
                if ahead:  # layer 0's input term has a buffer for the whole sequence: chunk c's rows are [t0, t0 + nt)
                    views = [dict(d, zin=[[z[t0:t0 + nt] if l_ == 0 else z[:nt] for z in zl] for l_, zl in enumerate(d["zin"])]) for t0, nt in bounds]
                    h_aux, ready = self._handle(aux_stream), []
                    for (t0, nt), dv in zip(bounds, views):
                        xg = prep(t0, nt, dv, h_aux)
                        ev = torch.cuda.Event()
                        ev.record(aux_stream)
                        ready.append((ev, xg))
                # prep_next (ungated model of the overlapped schedule = the full-band one; needs the chunk-local input term buffer free:
                # the stack launch of chunk c is behind us on the same stream when chunk c + 1's product is enqueued)
                nxt = bool(overlap and self.overlap_prep_next and gate_events is None and not ahead and sstreams[first] is gstreams[first])
                post_h = hG[first]
                on_post = bool(overlap and fused_post is not None and post_stream is not None and gate_events is not None)
                xg_next = prep(bounds[0][0], bounds[0][1], d, hG[first]) if nxt else None
                for c, (t0, nt) in enumerate(bounds):
                    if ahead:
                        dv, (ev, xg) = views[c], ready[c]
                        sstreams[first].wait_event(ev)
                    elif nxt:
                        dv, xg = d, xg_next
                    else:
                        dv = d
                        if staged and gate_events is not None:
                            gstreams[first].wait_event(gate_events[c])
                        xg = prep(t0, nt, dv, hG[first])
                    self._stage_stack(seqs, dv, t0, nt, hS[first], tag, wide, rpw_stack, xs_=xs_, xg=xg)
                    if nxt and c + 1 < len(bounds):
                        xg_next = prep(bounds[c + 1][0], bounds[c + 1][1], d, hG[first])
                    if on_post:
                        link(sstreams[first], post_stream)
                        post_h = self._handle(post_stream)
                    if fused_post is None or not fused_post(t0, nt, post_h):  # (the two-launch epilogue reads the same tensors)
                        self._stage_proj(seqs, d["s8"][nl - 1], d["proj"], t0, nt, post_h, tag)
                        if post_fn is not None:
                            post_fn(t0, nt, post_h)
                    if staged:
                        ev = torch.cuda.Event()
                        ev.record(post_stream if on_post else gstreams[first])
                        done.append(ev)
                return done
            # layer 0: groups whose real-valued input product can run inside the scan / the rest (input product first)
            fx = [i for i in range(len(seqs)) if self._fusable_x(seqs[i], xs_[i], rpw, want_membrane)]
            rest = [i for i in range(len(seqs)) if i not in fx]
            cn = d.get("cnt")
            for c, (t0, nt) in enumerate(bounds):
                for l in range(nl):
                    si = first + l
                    g, sc = gstreams[si], sstreams[si]
                    # ---- what the scan of this layer consumes
                    if l == 0:
                        if staged and gate_events is not None:
                            g.wait_event(gate_events[c])
                        if not feat_fn(t0, nt, hG[si], (rest, pick(d["zin"][0], rest))) and rest:
                            self._stage_input(pick(seqs, rest), 0, pick(xs_, rest), pick(d["zin"][0], rest), t0, nt, hG[si], tag)
                    else:
                        link(sstreams[si - 1], g)  # previous layer's scan of this chunk
                        if not fused:
                            self._stage_input(seqs, l, d["s8"][l - 1], d["zin"][l], t0, nt, hG[si], tag)
                    link(g, sc)
                    # ---- the scan(s)
                    if l == 0:
                        if fx:
                            self._stage_scan_fused_x(pick(seqs, fx), pick(xs_, fx), pick(d["states"][0], fx), pick(d["spk"][0], fx),
                                                     pick(d["s8"][0], fx), t0, nt, hS[si], tag, cnts=None if cn is None else pick(cn[0], fx))
                        if rest:
                            self._stage_scan(pick(seqs, rest), 0, pick(d["zin"][0], rest), pick(d["states"][0], rest), pick(d["spk"][0], rest),
                                             pick(d["s8"][0], rest), pick(d["mem"][0], rest), t0, nt, hS[si], tag, rpw,
                                             cnts=None if cn is None else pick(cn[0], rest))
                    elif fused:
                        self._stage_scan_fused(seqs, l, d["states"][l], d["spk"][l], d["s8"], t0, nt, hS[si], tag, cnts=None if cn is None else cn[l])
                    else:
                        self._stage_scan(seqs, l, d["zin"][l], d["states"][l], d["spk"][l], d["s8"][l], d["mem"][l], t0, nt, hS[si], tag, rpw,
                                         cnts=None if cn is None else cn[l])
                    if staged:
                        link(sc, g)  # the chunk-local zin buffer is reused by the next chunk's input product
                    # ---- after the last layer: projection and whatever follows the model
                    if l == nl - 1 and (fused_post is None or not fused_post(t0, nt, hG[si])):
                        self._stage_proj(seqs, d["s8"][l], d["proj"], t0, nt, hG[si], tag)
                        if post_fn is not None:
                            post_fn(t0, nt, hG[si])
                    if l == nl - 1:
                        if staged:
                            ev = torch.cuda.Event()
                            ev.record(g)
                            done.append(ev)
            return done

        cum = None
        if spec.cum_laplace:
            # cumulative_laplace_norm: the rows leave sfsn_features unnormalised and are divided by their running means (state per
            # row, so a sequence fed in pieces continues where it stopped)
            rows = [B] + [x.shape[1] for x in xs]
            cum = dict(state=[torch.zeros((R,), **f32) for R in rows], scratch=torch.empty((nt_max * max(rows),), **f32))

        def cum_norm(x, i, t0, nt, st):
            R, I = x.shape[1], x.shape[2]
            check(L.sfsn_cum_laplace_norm(ctypes.c_void_p(x.data_ptr() + t0 * R * I * 4), nt, R, I, _ptr(cum["state"][i]), t0,
                                          _ptr(cum["scratch"]), st), "sfsn_cum_laplace_norm")

        # feature rows a fused launch did not write (layer_outputs "counts" / "none": nobody reads them)
        x_skipped = dict(fb=set(), sb=set())
        dims_fb, dims_sb = (B, F, T, 0, spec.fdrc), (B, F, T, spec.fb_proj, spec.fdrc)

        def feat_fb(t0, nt, st, proj=None):
            z = zero_job.pop() if zero_job else None  # (the first full-band feature launch of the forward carries the state zeroing)
            if proj is not None and cum is None and self._stage_featproj(fg_fb, [self.fb], proj[0], proj[1], ri, None, dims_fb, t0, nt, z,
                                                                         want_layers, st, "fb"):
                if not want_layers:
                    x_skipped["fb"].update(proj[0])
                return True
            with self.timed("features:fb", st):
                check(L.sfsn_features_z(_ptr(ri), None, B, F, T, 0, spec.fdrc, fg_fb, 1, t0, nt, None if z is None else _ptr(z),
                                        0 if z is None else z.numel() * 4, st), "sfsn_features(fb)")
                if cum is not None:
                    cum_norm(x_fb, 0, t0, nt, st)
            return False

        def feat_sb(t0, nt, st, proj=None):
            if proj is not None and cum is None and self._stage_featproj(fg_sb, self.sb, proj[0], proj[1], ri, fb_proj, dims_sb, t0, nt, None,
                                                                         want_layers, st, "sb"):
                if not want_layers:
                    x_skipped["sb"].update(proj[0])
                return True
            with self.timed("features:sb", st):
                check(L.sfsn_features(_ptr(ri), _ptr(fb_proj), B, F, T, spec.fb_proj, spec.fdrc, fg_sb, ng, t0, nt, st), "sfsn_features(sb)")
                if cum is not None:
                    for g in range(ng):
                        cum_norm(xs[g], 1 + g, t0, nt, st)
            return False

        def post_sb(t0, nt, st):
            with self.timed("deepfilter", st):
                check(L.sfsn_deepfilter(_ptr(ri), B, F, T, S, dfg, ng, _ptr(enh_ri), _ptr(enh_mag), t0, nt, st), "sfsn_deepfilter")

        # round 6: projection + deep filter of a chunk in one launch; in the lean modes the coefficient rows are not written at all
        write_proj = bool(want_layers or want_membrane or not self.lean_skips_proj)
        proj_skipped = [False]

        def fused_post_sb(t0, nt, st):
            if not self._stage_projdf(self.sb, sb["s8"][nl_sb - 1], sb["proj"], ri, enh_ri, enh_mag, (B, F, T, S), t0, nt, st, write_proj):
                return False
            proj_skipped[0] = not write_proj
            return True

        if spec.laplace and spec.gaussian:
            check(L.sfsn_gaussian_stats(_ptr(ri), None, B, F, T, 0, spec.fdrc, fg_fb, 1, _ptr(mu_fb), _ptr(sd_fb), _ptr(scratch), hG[0]),
                  "sfsn_gaussian_stats(fb)")
        elif spec.laplace:
            check(L.sfsn_laplace_means(_ptr(ri), None, B, F, T, 0, spec.fdrc, fg_fb, 1, _ptr(mu_fb), _ptr(scratch), hG[0]), "sfsn_laplace_means(fb)")
        fb_done = run_model([self.fb], fb, [x_fb], 0, feat_fb, "fb", rpw_fb, None, None)
        if spec.laplace:
            # the utterance-level Laplace mean of the sub-band input needs the whole full-band output: no chunk overlap fb -> sb
            if pipeline:
                gstreams[nl_fb].wait_event(fb_done[-1])
            if spec.gaussian:
                check(L.sfsn_gaussian_stats(_ptr(ri), _ptr(fb_proj), B, F, T, spec.fb_proj, spec.fdrc, fg_sb, ng, _ptr(mu_sb), _ptr(sd_sb),
                                            _ptr(scratch), hG[nl_fb]), "sfsn_gaussian_stats(sb)")
            else:
                check(L.sfsn_laplace_means(_ptr(ri), _ptr(fb_proj), B, F, T, spec.fb_proj, spec.fdrc, fg_sb, ng, _ptr(mu_sb), _ptr(scratch),
                                           hG[nl_fb]), "sfsn_laplace_means(sb)")
            sb_done = run_model(self.sb, sb, xs, nl_fb, feat_sb, "sb", rpw_sb, post_sb, None, fused_post_sb)
        elif pipeline and self.pipeline_two_phase:
            # full-band model first (its two layers overlapped), then the sub-band models (their two layers overlapped)
            gstreams[nl_fb].wait_event(fb_done[-1])
            sb_done = run_model(self.sb, sb, xs, nl_fb, feat_sb, "sb", rpw_sb, post_sb, None, fused_post_sb)
        else:
            sb_done = run_model(self.sb, sb, xs, nl_fb, feat_sb, "sb", rpw_sb, post_sb, fb_done if staged else None, fused_post_sb)
        if staged:
            extra = ([aux_stream] if prep_ahead else []) + ([post_stream] if (overlap and post_stream is not None) else [])
            for s_ in {id(x): x for x in sstreams + gstreams + extra}.values():
                link(s_, main)
        if self._defer_err:
            # the error words of this forward's stack launches: one maximum, one copy to pinned memory, on a stream of its own behind
            # the joined forward (nothing of the forward waits for it)
            items, self._defer_err = self._defer_err, None
            if self._err_stream is None:
                self._err_stream = torch.cuda.Stream(device=dev)
            es = self._err_stream
            ev0 = torch.cuda.Event()
            ev0.record(main)
            es.wait_event(ev0)
            with torch.cuda.stream(es):
                uniq = list({id(sc): sc for _, sc in items}.values())
                word = torch.stack([sc[0] for sc in uniq]).max().reshape(1)
                pin = torch.empty((1,), dtype=torch.int32, pin_memory=True)
                pin.copy_(word, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(es)
            for sc in uniq:
                sc.record_stream(es)
            self._stack_err_pending.append((ev, pin, "; ".join(sorted({w for w, _ in items})), uniq))
        self._defer_err = None

        if want_counts and not want_layers:
            # SynOPs without the fp32 spike tensors: the scans have counted what they wrote (no launch, no pass over
            # the int8 copies); `count_in_scan = False`: round 3's counting launch over the int8 spikes, one launch for all layers
            tens = [fb["s8"][l][0] for l in range(nl_fb)] + [sb["s8"][l][g] for g in range(ng) for l in range(nl_sb)]
            shapes = [(T, B, self.fb.H)] * nl_fb + [(T, xs[g].shape[1], self.sb[g].H) for g in range(ng) for _ in range(nl_sb)]
            if in_scan:
                summ = [SpikeSummary(fb["cnt"][l][0], shapes[l]) for l in range(nl_fb)] + \
                       [SpikeSummary(sb["cnt"][l][g], shapes[nl_fb + g * nl_sb + l]) for g in range(ng) for l in range(nl_sb)]
            else:
                summ = self._count_spikes(tens, shapes, self._handle(main))
            for l in range(nl_fb):
                fb["spk"][l][0] = summ[l]
            for g in range(ng):
                for l in range(nl_sb):
                    sb["spk"][l][g] = summ[nl_fb + g * nl_sb + l]

        def outs(x, d, i, skipped, no_proj=False):
            if i in skipped:  # never written: the entry keeps its shape (compute_neuronops reads size(-1)) and nothing else
                x = torch.empty(x.shape, dtype=x.dtype, device="meta")
            pr = d["proj"][i]
            if no_proj:  # (lean modes through sfsn_proj_deepfilter: the coefficient rows stayed in LDS)
                pr = torch.empty(pr.shape, dtype=pr.dtype, device="meta")
            return [x] + [d["spk"][l][i] for l in range(len(d["spk"]))] + [pr]
        return dict(enh_stft=enh, enh_mag=enh_mag, fb_all=outs(x_fb, fb, 0, x_skipped["fb"]),
                    sb_all=[outs(xs[g], sb, g, x_skipped["sb"], proj_skipped[0]) for g in range(ng)],
                    fb_mem=[fb["mem"][l][0] for l in range(nl_fb)], sb_mem=[[sb["mem"][l][g] for l in range(nl_sb)] for g in range(ng)],
                    mu_fb=mu_fb, mu_sb=mu_sb, pipelined=bool(pipeline), overlapped=overlap, n_chunks=len(bounds))
