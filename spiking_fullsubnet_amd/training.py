"""Training-mode forward and backward of the live model (SURVEY 8b "autograd", 8f rank 4).

What the recipes' training step needs (recipes/intel_ndns/spiking_fullsubnet/trainer.py:24-48: ``self.model(noisy)`` in
``.train()`` mode, a loss on the enhanced waveform / magnitude, ``accelerator.backward(loss)``):

* the recurrent cell loop with ``nn.BatchNorm1d`` in TRAINING mode inside the cell -- every time step normalises with that step's
  batch statistics and updates the running statistics (efficient_spiking_neuron.py:123,149-150) -- and its backward pass through the
  triangle surrogate of the spike (:94-101): ``GSNLayerTrainFn`` / ``GSNLayersTrainFn``, ``torch.autograd.Function``s whose whole time
  loop is ONE HIP launch per direction (``sfsn_gsn_train_seq_fwd`` / ``_bwd``, csrc/sfsn_train.hip: the workgroups of a layer call --
  16 neurons x a block of rows each -- stay resident for all T steps, recurrent products on the fp32 matrix pipe, carried state in
  LDS, what a step needs from other workgroups exchanged through the L2; layer l of all sub-band groups in one grid,
  ``_multi``); the time-parallel products around them (input product, weight gradients, input gradient) are library GEMMs
  (``torch.mm``; the weight gradients as batched GEMMs over slices of the (frame, row) axis, ``_tn_gemm``).  Round 5: ``GSNStackTrainFn``
  runs the layers of a model's stacks in ONE grid per stage, layer l + 1 a chunk of frames behind layer l (chunked layer calls carry
  their state through ``SfsnTrainSeqFwd.h0 / c0`` and ``SfsnTrainSeqBwd.dc_in / dc_out``) -- ``gsn_stack`` / ``gsn_stacks`` take it
  wherever the library can hold the calls resident together;
* everything between ``stft`` and ``istft`` that is time-parallel (band selection, reflect-gathered sub-band features, LayerNorm,
  projections, deep filter) as differentiable ATen operations on index tensors built once per module -- the kernels of the
  inference engine have no backward.

The same ATen path serves the constructor options the inference kernels do not cover (``sequence_model="LSTM"`` = ``nn.LSTM``,
``proj_size=0``, output activations), in ``eval()`` mode too: SURVEY 8b's torch fallback.  HIP tensors only -- like the rest of the
package this has no CPU path (a CPU module raises).
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import threading
import warnings
from typing import List

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import check


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# debugging / cross-checks: True = round 3's launch per time step (sfsn_gsn_train_step_fwd / _bwd, scalar products) instead of one launch
# per layer and direction (scripts/exp_train_shapes.py compares the two over odd shapes)
STEP_LAUNCHES = os.environ.get("SFSN_TRAIN_STEP_LAUNCHES", "0") != "0"
# the same layer of all sub-band groups in ONE launch per direction (SFSN_TRAIN_GROUPS_TOGETHER=0: one group after the other)
GROUPS_TOGETHER = os.environ.get("SFSN_TRAIN_GROUPS_TOGETHER", "1") != "0"

# The layers of ONE stack in one grid, layer l + 1 a chunk of frames behind layer l (GSNStackTrainFn: the layer calls are cut into
# STACK_CHUNKS chunks; stage s launches {layer l, chunk s - l} together).  0 / 1 = one layer call after the other.
STACK_CHUNKS = int(os.environ.get("SFSN_TRAIN_STACK_CHUNKS", "20"))
# GSNStackTrainFn.backward: the weight gradients (and layer 0's dL/dx) in STACK_GRAD_BLOCKS blocks of chunks, optionally on a SIDE stream
# beside the following stages' launches (they are not on the chain).  Measured at B = 64 and OFF: one block behind the last stage
# 82.2 ms per step; 2 / 4 / 5 blocks on the side stream 83.8 / 81.9 / 82.3 (the launches beside the GEMMs take 31.7 instead of 28.5 ms:
# what is hidden is paid back), 4 blocks on the main stream 85.4, a GEMM per chunk 96.5 (87 on the side stream).
STACK_SIDE_STREAM = os.environ.get("SFSN_TRAIN_SIDE_STREAM", "0") != "0"
STACK_GRAD_BLOCKS = int(os.environ.get("SFSN_TRAIN_GRAD_BLOCKS", "1"))  # the sequence's weight-gradient GEMMs in this many blocks of chunks
_side_streams: dict = {}
_STACK_CALLS = 0  # (tests: how many stacks went through GSNStackTrainFn)
STACK_MIN_FRAMES = 16  # (a chunk shorter than this is all launch ramp; the tests lower it to run the reference's short fixtures in chunks)

_EXCHANGE_FAILED = ("sfsn_gsn_train_step: the row blocks of a step did not all arrive (a launch's workgroups were not co-resident); "
                    "the outputs of that layer call are invalid")
launch_log = None  # bench.py --training sets a list: (direction, T, [(R, H, G*H) per layer call], start event, end event) per launch
_debug_scratch = None  # scripts/dbg_train_hang.py sets a list: every layer call appends (what, R, H, T, scratch tensor)
_pending: list = []  # (event, pinned copy of a layer call's error word); forward thread and autograd thread both touch it: _lock
_lock = threading.Lock()
_final_check_queued = False


def check_pending() -> None:
    """Wait for every layer call issued so far and raise if one of them reported a failed row-block exchange (its outputs and
    gradients are invalid; the gradients are NaN).  Called by the package itself at the end of every backward pass (see
    _queue_final_check), so a training loop needs no call of its own."""
    _poll_pending(block=True)


def _poll_pending(block: bool = False, new_forward: bool = False) -> None:
    """Raise if an earlier layer call reported a failed row-block exchange (a forward nobody followed with a backward, or a
    backward: its gradients are NaN).  new_forward (the layer calls' forward()): a backward pass that queued its end-of-pass check and
    then ABORTED in another node (out of memory, a user exception) never ran the callback -- autograd drops the graph task's callbacks
    -- and would leave the "queued" flag set for the rest of the process, so that no later loss.backward() checked anything (round-5
    advisor finding).  A forward is outside any pass that could still run that callback: the flag is cleared here, the next backward
    queues afresh (a forward recomputed INSIDE a backward pass -- activation checkpointing -- merely queues a second, idempotent check)."""
    global _final_check_queued
    if _capturing():
        return  # (no event queries / host reads inside a capture; GraphedTrainStep checked before it began)
    with _lock:
        if new_forward:
            _final_check_queued = False
        items, _pending[:] = list(_pending), []
    keep, failed = [], False
    for ev, pin in items:
        if block:
            ev.synchronize()
        if ev.query():
            failed = failed or int(pin.max().item()) != 0
        else:
            keep.append((ev, pin))
    with _lock:
        if failed:
            _pending.clear()
        else:
            _pending[:0] = keep
    if failed:
        raise RuntimeError(_EXCHANGE_FAILED)


def _push_pending(ev, pin) -> None:
    with _lock:
        _pending.append((ev, pin))


_capture_errs = None  # GraphedTrainStep sets a list while it captures: the layer calls' error words (device tensors of the graph's pool)


def _capturing() -> bool:
    return torch.cuda.is_current_stream_capturing()


def _report(errs: torch.Tensor, dev, final: bool) -> None:
    """A layer call's error word(s) (int32 device tensor, any shape) on their way to the host without blocking it: to pinned memory
    behind the launches, looked at by the next layer call / at the end of the backward pass (final = called from a backward()).
    While a HIP graph is being captured (GraphedTrainStep) nothing host-side may happen: the words are handed to the capturing
    object, which reduces them inside the graph and reads the result after every replay."""
    if _capturing():
        if _capture_errs is None:
            raise RuntimeError("a training layer call inside a HIP graph capture that training.GraphedTrainStep does not own: its error "
                               "word would go unread -- capture the step with GraphedTrainStep")
        _capture_errs.append(errs.reshape(-1).max())
        return
    pin = torch.empty((errs.numel(),), dtype=torch.int32, pin_memory=True)
    pin.copy_(errs.reshape(-1), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    _push_pending(ev, pin)
    if final:
        _queue_final_check()


def _queue_final_check() -> None:
    """Called from a layer call's backward(): have the autograd engine run check_pending() ONCE when the backward pass that is
    executing has finished (the mechanism DistributedDataParallel uses for its final reductions).  `loss.backward()` then raises on a
    failed exchange itself -- before the recipe's `optimizer.step()` can apply NaN gradients to the weights (round-4 advisor finding:
    the reference's trainer never calls check_pending(), recipes/intel_ndns/spiking_fullsubnet/trainer.py:24-48).  One host
    synchronisation per training step, at a point where every launch of the step has been enqueued; the layer calls themselves
    still do not block (the sub-band groups share a grid, the autograd thread keeps enqueueing)."""
    global _final_check_queued
    with _lock:
        if _final_check_queued:
            return
        _final_check_queued = True

    def _final():
        global _final_check_queued
        with _lock:
            _final_check_queued = False
        check_pending()
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_final)
    except Exception:  # (not inside a backward pass of the engine -- e.g. a test calling backward() by hand): block here instead
        with _lock:
            _final_check_queued = False
        check_pending()


# The weight gradients are a^T . b over ALL (frame, row) pairs: a [M, N1], b [M, N2] with M = T R ~ 10^5 .. 10^6 and a result of a few
# hundred squared.  As ONE library GEMM that is (N1 / 32) x (N2 / 64) ~ 30 workgroups walking the whole of M (hipBLASLt picks no split
# over k here): 0.5 ms each, ~10 ms of a training step at B = 64.  Cut into WGRAD_SPLIT slices of M as a batched GEMM plus a sum of
# the slices' results, every compute unit has a workgroup.
WGRAD_SPLIT = int(os.environ.get("SFSN_TRAIN_WGRAD_SPLIT", "64"))


def _tn_gemm(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a^T . b for a [M, N1], b [M, N2] (contiguous views), M large."""
    M = a.shape[0]
    S = min(WGRAD_SPLIT, M // 2048)
    if S < 2:
        return torch.mm(a.t(), b)
    m = M // S
    out = torch.bmm(a[:S * m].view(S, m, a.shape[1]).transpose(1, 2), b[:S * m].view(S, m, b.shape[1])).sum(0)
    if S * m < M:
        out.addmm_(a[S * m:].t(), b[S * m:])
    return out


class _LinearTN(torch.autograd.Function):
    """nn.Linear over [T, R, H] with the weight gradient through _tn_gemm (ATen's is one GEMM of ~30 workgroups over all T R rows)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        dx = torch.mm(dy2, w).view(x.shape) if ctx.needs_input_grad[0] else None
        dw = _tn_gemm(dy2 if dy2.is_contiguous() else dy2.contiguous(), x.reshape(-1, x.shape[-1])) if ctx.needs_input_grad[1] else None
        db = dy2.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db


class _LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last axis of [T, R, I] with I = 38 .. 158 (the sub-band feature rows; MODEL:91-93): ATen's kernels give a
    block to every row (RowwiseMoments: 0.8 ms for 512,000 rows of 38 floats, its backward another 1.2) -- the same arithmetic out of a
    handful of element-wise / short-reduction operations over the whole tensor is 2 x faster in both directions (round 6,
    scripts/exp_train_glue_r06.py: 3.75 -> 1.73 ms per training step for the three groups; values within 3e-6)."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        var, mean = torch.var_mean(x, dim=-1, unbiased=False, keepdim=True)
        rstd = torch.rsqrt(var + eps)
        xhat = (x - mean) * rstd
        ctx.save_for_backward(xhat, rstd, w)
        return torch.addcmul(b, xhat, w)

    @staticmethod
    def backward(ctx, dy):
        xhat, rstd, w = ctx.saved_tensors
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            g = dy * w
            dx = (g - g.mean(-1, keepdim=True) - xhat * (g * xhat).mean(-1, keepdim=True)) * rstd
        n = dy.shape[-1]
        if ctx.needs_input_grad[1]:
            dw = (dy.reshape(-1, n) * xhat.reshape(-1, n)).sum(0)
        if ctx.needs_input_grad[2]:
            db = dy.reshape(-1, n).sum(0)
        return dx, dw, db, None


FAST_GLUE = os.environ.get("SFSN_TRAIN_FAST_GLUE", "1") != "0"  # round 6: _LayerNormFn, time-major feature rows, complex deep filter


def _pre_ln(ln, x):
    """seq.pre_layer_norm(x): an affine nn.LayerNorm over the last axis of a float32 HIP tensor through _LayerNormFn, else the module."""
    if (FAST_GLUE and isinstance(ln, torch.nn.LayerNorm) and ln.elementwise_affine and ln.bias is not None and len(ln.normalized_shape) == 1
            and x.is_cuda and x.dtype == torch.float32 and ln.weight.dtype == torch.float32):
        return _LayerNormFn.apply(x, ln.weight, ln.bias, ln.eps)
    return ln(x)


def _proj(lin, x):
    """seq.proj(x) (MODEL:49-52,118): nn.Linear -> _LinearTN on contiguous HIP tensors; anything else (Identity, ...) as it is."""
    if isinstance(lin, torch.nn.Linear) and x.is_cuda and x.is_contiguous() and x.dtype == torch.float32 and lin.weight.dtype == torch.float32:
        return _LinearTN.apply(x, lin.weight, lin.bias)
    return lin(x)


class _Logged:
    """HIP events around a layer-call launch when ``launch_log`` is a list (bench.py's per-step figures); nothing otherwise."""
    def __init__(self, kind, T, shapes):
        self.rec = (kind, T, shapes) if launch_log is not None else None

    def __enter__(self):
        if self.rec is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.rec is not None:
            self.e1.record()
            launch_log.append(self.rec + (self.e0, self.e1))
        return False


class GSNLayerTrainFn(torch.autograd.Function):
    """One GSN layer over all T steps: x [T, R, I] -> spikes [T, R, H] (zero initial state, modeling_spiking_fullsubnet.py:100-106).

    ``bn_w`` / ``bn_b`` None = no BatchNorm.  ``stats`` = (running_mean, running_var, num_batches_tracked) or None; with
    ``batch_stats`` (the module is in training mode) every step uses its own batch statistics and updates ``stats`` in place;
    without it (eval mode, gradients still wanted) the running statistics are folded into an affine map."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, bias, bn_w, bn_b, stats, shared, batch_stats, momentum, eps):
        if not x.is_cuda:
            raise RuntimeError("spiking_fullsubnet_amd has no CPU path: move the module and its input to a HIP device")
        L = _lib.lib()
        _poll_pending(new_forward=True)
        T, R, I = x.shape
        GH, H = w_hh.shape
        dev = x.device
        # both directions' geometry up front (a forward pass must not succeed -- and update the BatchNorm buffers -- where the backward
        # pass would be refused): the one-launch layer call needs every workgroup of BOTH directions resident (LDS and occupancy:
        # sfsn_gsn_train_multi_check with one call).  Where it cannot be, round 3's launch per step is taken IF those kernels can hold
        # a step's workgroups (sfsn_gsn_train_step_check: their own, smaller LDS needs -- the eval-mode-BatchNorm path always runs them);
        # a geometry neither form holds (e.g. H = 320, R ~ 1200: 300 workgroups against 256 slots of either backward kernel) is
        # refused here, before anything has run
        use_bn = bn_w is not None
        fold = use_bn and not batch_stats  # eval-mode BatchNorm: y = x * alpha + beta with the running statistics
        seq = not fold and not STEP_LAUNCHES
        if seq:
            rc = L.sfsn_gsn_train_multi_check((ctypes.c_int * 1)(R), 1, H, int(shared))
            if rc == _lib.SFSN_EUNSUPPORTED:
                seq = False
            else:
                check(rc, f"sfsn_gsn_train_multi_check(R={R}, H={H})")
        if not seq:
            check(L.sfsn_gsn_train_step_check(R, H, int(shared)), f"sfsn_gsn_train_step_check(R={R}, H={H})")
        if bn_w is not None and batch_stats:
            if R == 1:  # nn.BatchNorm1d in training mode (torch/nn/functional.py: _verify_batch_size)
                raise ValueError(f"Expected more than 1 value per channel when training, got input size torch.Size([{R}, {H}])")
            if stats is not None:
                for nm, t_ in (("running_mean", stats[0]), ("running_var", stats[1])):
                    # the step kernel updates these in place through raw float pointers
                    if t_.dtype != torch.float32 or not t_.is_contiguous() or t_.device != dev or t_.numel() != H:
                        raise TypeError(f"BatchNorm {nm} must be a contiguous float32 tensor of {H} elements on {dev}, got "
                                        f"{t_.dtype} {tuple(t_.shape)} on {t_.device}")
        x = x.contiguous().float()
        w_ih_c, w_hh_c, bias_c = w_ih.detach().contiguous().float(), w_hh.detach().contiguous().float(), bias.detach().contiguous().float()
        z = torch.mm(x.reshape(T * R, I), w_ih_c.t()).view(T, R, GH)  # x_t . W_ih^T for all t (bias is added inside the step)
        f32 = dict(dtype=torch.float32, device=dev)
        spikes, u = torch.empty((T, R, H), **f32), torch.empty((T, R, H), **f32)
        fg, gg = torch.empty((T, R, H), **f32), torch.empty((T, R, H), **f32)
        if fold:
            rm, rv = stats[0].float(), stats[1].float()
            alpha = bn_w.detach().float() / torch.sqrt(rv + eps)
            beta = bn_b.detach().float() - rm * alpha
        xhat = torch.empty((T, R, H), **f32) if (use_bn and not fold) else None
        invstd = torch.empty((T, H), **f32) if (use_bn and not fold) else None
        zero = torch.zeros((R, H), **f32)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        bw = bn_w.detach().contiguous().float() if (use_bn and not fold) else None
        bb = bn_b.detach().contiguous().float() if (use_bn and not fold) else None
        rmean = stats[0] if (use_bn and batch_stats and stats is not None) else None
        rvar = stats[1] if (use_bn and batch_stats and stats is not None) else None
        # raw pointers per step (no tensor views: the interpreter's share of a step is what bounds small batches)
        P = ctypes.c_void_p
        sRH, sRG = R * H * 4, R * GH * 4
        pz, psp, pu, pf, pg = z.data_ptr(), spikes.data_ptr(), u.data_ptr(), fg.data_ptr(), gg.data_ptr()
        pxh = xhat.data_ptr() if xhat is not None else 0
        pis = invstd.data_ptr() if invstd is not None else 0
        pzero, pw, pb = zero.data_ptr(), w_hh_c.data_ptr(), bias_c.data_ptr()
        a_bw, a_bb, a_rm, a_rv = _p(bw), _p(bb), _p(rmean), _p(rvar)
        # momentum=None = cumulative moving average (torch/nn/modules/batchnorm.py): the factor of step t is 1 / num_batches_tracked after
        # that step's increment; handed to the kernels as -(count before the call + 1) (one host read of the counter per layer call)
        n0 = int(stats[2].item()) if (momentum is None and use_bn and batch_stats and stats is not None and stats[2] is not None) else 0
        mom, ep, sh = float(-(n0 + 1) if momentum is None else momentum), float(eps), int(shared)
        fwd = L.sfsn_gsn_train_step_fwd
        # zeroed per call: packed-spike slots and publish counters of the one-launch layer call, partial-sum granules, error word (last 4 words)
        scr = torch.zeros(((L.sfsn_train_seq_scratch_bytes(R, H) if seq else L.sfsn_train_scratch_bytes(H)) // 4,), dtype=torch.int32, device=dev)
        p_scr = P(scr.data_ptr())
        if _debug_scratch is not None:
            _debug_scratch.append(("fwd", R, H, T, scr))
        if seq:  # ONE launch for the T steps of the layer (csrc/sfsn_train.hip: workgroups resident over the sequence)
            with torch.cuda.device(dev), _Logged("fwd", T, [(R, H, GH)]):
                check(L.sfsn_gsn_train_seq_fwd(P(pz), P(pw), P(pb), a_bw, a_bb, a_rm, a_rv, mom, ep, T, R, H, sh, P(pzero), P(psp), P(pu),
                                               P(pxh) if pxh else None, P(pf), P(pg), P(pis) if pis else None, p_scr, st), "sfsn_gsn_train_seq_fwd")
        with torch.cuda.device(dev):
            for t in (() if seq else range(T)):
                rc = fwd(P(pz + t * sRG), P(pw), P(pb), P(pzero if t == 0 else psp + (t - 1) * sRH), P(pzero if t == 0 else pu + (t - 1) * sRH),
                         a_bw, a_bb, a_rm, a_rv, (mom if mom >= 0 else 1.0 / (n0 + t + 1)), ep, R, H, sh, P(psp + t * sRH), P(pu + t * sRH), P(pxh + t * sRH) if pxh else None,
                         P(pf + t * sRH), P(pg + t * sRH), P(pis + t * H * 4) if pis else None, p_scr, t + 1, st)
                if rc:
                    check(rc, "sfsn_gsn_train_step_fwd")
                if fold:  # (the folded affine map and the threshold, on the pre-normalisation membrane the step left in u)
                    u[t].mul_(alpha).add_(beta)
                    spikes[t].copy_((u[t] >= 0).float())
        ctx.scr = scr  # (its error word is read in backward: no synchronisation there until the gradients are assembled)
        if not fold:
            if not any(ctx.needs_input_grad):
                # nothing will call backward() (BatchNorm recalibration, validation with the module left in train()): an exchange that
                # timed out leaves spikes / u / running statistics unwritten -- find out before handing them out
                if int(scr[-4:].max().item()) != 0:
                    raise RuntimeError(_EXCHANGE_FAILED)
            else:
                # backward() reads the word too, but a forward that is never followed by one must not go unnoticed either: the word
                # travels to pinned host memory behind the launches and the next layer call looks at it without blocking
                _report(scr[-4:], dev, final=False)
        if use_bn and batch_stats and stats is not None and stats[2] is not None:
            stats[2].add_(T)  # num_batches_tracked: one BatchNorm call per time step
        ctx.save_for_backward(x, w_ih_c, w_hh_c, spikes, u, fg, gg, xhat if xhat is not None else zero, invstd if invstd is not None else zero,
                              bw if bw is not None else zero, alpha if fold else zero)
        ctx.meta = (bool(shared), use_bn, fold, T, R, I, H, GH)
        ctx.seq = seq  # (backward takes the same form as forward did)
        return spikes

    @staticmethod
    def backward(ctx, dy):
        x, w_ih, w_hh, spikes, u, fg, gg, xhat, invstd, bw, alpha = ctx.saved_tensors
        shared, use_bn, fold, T, R, I, H, GH = ctx.meta
        L = _lib.lib()
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        dy = dy.contiguous().float()
        d_gates = torch.empty((T, R, 2 * H), **f32)
        d_z = torch.empty((T, R, H), **f32) if shared else None
        dc_buf = [torch.empty((R, H), **f32), torch.empty((R, H), **f32)]
        d_bn_w, d_bn_b = torch.zeros((H,), **f32), torch.zeros((H,), **f32)
        zero = torch.zeros((R, H), **f32)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        bn_kernel = use_bn and not fold
        P = ctypes.c_void_p
        sRH, s2H, sRG = R * H * 4, R * 2 * H * 4, R * GH * 4
        pdy, pu, pf, pg = dy.data_ptr(), u.data_ptr(), fg.data_ptr(), gg.data_ptr()
        pxh, pis = (xhat.data_ptr(), invstd.data_ptr()) if bn_kernel else (0, 0)
        pdg = d_gates.data_ptr()
        pdz = d_z.data_ptr() if shared else pdg      # the gradient of the (shared or per-gate) products: [T][R][G*H]
        pzero, pw = zero.data_ptr(), w_hh.data_ptr()
        a_bw = _p(bw) if bn_kernel else None
        a_dw, a_db = (_p(d_bn_w), _p(d_bn_b)) if bn_kernel else (None, None)
        pdc = [dc_buf[0].data_ptr(), dc_buf[1].data_ptr()]
        sh = int(shared)
        bwd = L.sfsn_gsn_train_step_bwd
        seq = ctx.seq
        scr = torch.zeros(((L.sfsn_train_seq_scratch_bytes(R, H) if seq else L.sfsn_train_scratch_bytes(H)) // 4,), dtype=torch.int32, device=dev)
        p_scr = P(scr.data_ptr())
        if _debug_scratch is not None:
            _debug_scratch.append(("bwd", R, H, T, scr))
        dh_rec = dc = None
        if seq:
            with torch.cuda.device(dev), _Logged("bwd", T, [(R, H, GH)]):
                check(L.sfsn_gsn_train_seq_bwd(P(pw), P(pdy), P(pu), P(pxh) if bn_kernel else None, P(pf), P(pg), P(pis) if bn_kernel else None, a_bw,
                                               T, R, H, sh, None, P(pdg), P(pdz) if shared else None, None, a_dw, a_db, p_scr, st),
                      "sfsn_gsn_train_seq_bwd")
        with torch.cuda.device(dev):
            for t in (() if seq else range(T - 1, -1, -1)):
                last = t == T - 1
                p_cp = P(pzero if t == 0 else pu + (t - 1) * sRH)
                p_dzn = None if last else P(pdz + (t + 1) * sRG)   # dL/dh_t through step t+1: formed inside the step from its d_z
                if fold:
                    # eval-mode BatchNorm folded into an affine map: du -> dc' is a per-neuron scale, applied by pre-scaling the
                    # incoming gradient of u (the step kernel then runs without normalisation, its dc_next carrying everything)
                    tri = torch.clamp(1.0 - u[t].abs(), min=0.0)
                    dh = dy[t] if last else dy[t] + torch.mm((d_z if shared else d_gates)[t + 1], w_hh)
                    du = dh * tri
                    if dc is not None:
                        du = du + dc
                    dcy = (du * alpha).contiguous()
                    rc = bwd(None, None, None, None, _p(dcy), P(pu + t * sRH), None, P(pf + t * sRH), P(pg + t * sRH), p_cp, None, None, R, H, sh,
                             P(pdg + t * s2H), P(pdz + t * sRG) if shared else None, P(pdc[t & 1]), None, None, p_scr, T - t, st)
                    dc = dc_buf[t & 1]
                else:
                    rc = bwd(p_dzn, P(pw), P(pdy + t * sRH), None, None if last else P(pdc[(t + 1) & 1]), P(pu + t * sRH),
                             P(pxh + t * sRH) if bn_kernel else None, P(pf + t * sRH), P(pg + t * sRH), p_cp,
                             P(pis + t * H * 4) if bn_kernel else None, a_bw, R, H, sh, P(pdg + t * s2H),
                             P(pdz + t * sRG) if shared else None, P(pdc[t & 1]), a_dw, a_db, p_scr, T - t, st)
                if rc:
                    check(rc, "sfsn_gsn_train_step_bwd")
        # a failed row-block exchange (this call's or its forward's) must not yield gradients that look like numbers.  This function
        # does not block the host (the autograd thread keeps enqueueing the layers below while this one runs): every returned gradient
        # is poisoned with NaN ON THE DEVICE when either error word is set, the words travel to pinned memory, and the autograd
        # engine runs check_pending() when THIS backward pass has finished (_queue_final_check): `loss.backward()` raises, so the
        # recipe's optimizer.step() never sees the NaN gradients.
        bad = (scr[-4:].max() + ctx.scr[-4:].max()) > 0
        poison = torch.where(bad, torch.full((), float("nan"), **f32), torch.zeros((), **f32))
        _report(torch.maximum(scr[-4:], ctx.scr[-4:]), dev, final=True)
        dz = (d_z if shared else d_gates).reshape(T * R, GH)
        dx = torch.mm(dz, w_ih).view(T, R, I).add_(poison)
        dw_ih = _tn_gemm(dz, x.reshape(T * R, I)).add_(poison)
        # dL/dW_hh = sum_t dz_t^T h_{t-1}: h_{-1} = 0, so steps 1 .. T-1 against spikes 0 .. T-2 (views: no shifted copy of the spikes)
        dw_hh = (_tn_gemm(dz[R:], spikes[:-1].reshape((T - 1) * R, H)) if T > 1 else torch.zeros((GH, H), **f32)).add_(poison)
        dbias = d_gates.reshape(T * R, 2 * H).sum(0).add_(poison)
        if bn_kernel:
            d_bn_w.add_(poison)
            d_bn_b.add_(poison)
        if not use_bn:
            d_bn_w = d_bn_b = None
        elif fold:
            d_bn_w = d_bn_b = None  # (eval-mode gradients of gamma / beta are not produced: parameters are frozen in eval use)
        return dx, dw_ih, dw_hh, dbias, d_bn_w, d_bn_b, None, None, None, None, None


class GSNLayersTrainFn(torch.autograd.Function):
    """The same layer of n independent sequence models in ONE launch per direction (sfsn_gsn_train_seq_fwd_multi / _bwd_multi): the
    sub-band groups of a model do not depend on one another, and a training-mode layer call -- T dependent steps of a few hundred
    small workgroups -- leaves most of the chip idle; side by side in one grid the three groups of baseline_m take the time of the
    largest.  ``forward(ctx, meta, *flat)``: flat = n x (x [T, R_i, I_i], w_ih, w_hh, bias, bn_w, bn_b); meta = dict(shared, stats
    [n x (running_mean, running_var, num_batches_tracked) or None], momentum [n], eps [n]).  Training-mode BatchNorm (per-step batch
    statistics) or no BatchNorm; returns the n spike tensors [T, R_i, H]."""

    @staticmethod
    def forward(ctx, meta, *flat):
        L = _lib.lib()
        _poll_pending(new_forward=True)
        n = len(flat) // 6
        shared = bool(meta["shared"])
        xs = [flat[6 * i].contiguous().float() for i in range(n)]
        if not all(x.is_cuda for x in xs):
            raise RuntimeError("spiking_fullsubnet_amd has no CPU path: move the module and its input to a HIP device")
        dev = xs[0].device
        T = xs[0].shape[0]
        GH, H = flat[2].shape
        use_bn = flat[4] is not None
        f32 = dict(dtype=torch.float32, device=dev)
        calls = (_lib.TrainSeqFwd * n)()
        keep, saved, geo = [], [], []
        P = ctypes.c_void_p
        for i in range(n):
            x, w_ih, w_hh, bias, bn_w, bn_b = xs[i], *flat[6 * i + 1:6 * i + 6]
            Ti, R, I = x.shape
            assert Ti == T and tuple(w_hh.shape) == (GH, H) and (bn_w is not None) == use_bn
            stats = meta["stats"][i]
            if use_bn:
                if R == 1:  # nn.BatchNorm1d in training mode (torch/nn/functional.py: _verify_batch_size)
                    raise ValueError(f"Expected more than 1 value per channel when training, got input size torch.Size([{R}, {H}])")
                if stats is not None:
                    for nm, t_ in (("running_mean", stats[0]), ("running_var", stats[1])):
                        if t_.dtype != torch.float32 or not t_.is_contiguous() or t_.device != dev or t_.numel() != H:
                            raise TypeError(f"BatchNorm {nm} must be a contiguous float32 tensor of {H} elements on {dev}, got "
                                            f"{t_.dtype} {tuple(t_.shape)} on {t_.device}")
            w_ih_c, w_hh_c, bias_c = w_ih.detach().contiguous().float(), w_hh.detach().contiguous().float(), bias.detach().contiguous().float()
            z = torch.mm(x.reshape(T * R, I), w_ih_c.t()).view(T, R, GH)
            spikes, u = torch.empty((T, R, H), **f32), torch.empty((T, R, H), **f32)
            fg, gg = torch.empty((T, R, H), **f32), torch.empty((T, R, H), **f32)
            xhat = torch.empty((T, R, H), **f32) if use_bn else None
            invstd = torch.empty((T, H), **f32) if use_bn else None
            bw = bn_w.detach().contiguous().float() if use_bn else None
            bb = bn_b.detach().contiguous().float() if use_bn else None
            scr = torch.zeros((L.sfsn_train_seq_scratch_bytes(R, H) // 4,), dtype=torch.int32, device=dev)
            c = calls[i]
            c.z, c.w_hh, c.bias, c.bn_w, c.bn_b = z.data_ptr(), w_hh_c.data_ptr(), bias_c.data_ptr(), _dp(bw), _dp(bb)
            c.running_mean = _dp(stats[0]) if (use_bn and stats is not None) else None
            c.running_var = _dp(stats[1]) if (use_bn and stats is not None) else None
            if meta["momentum"][i] is None:  # cumulative moving average: see GSNLayerTrainFn.forward
                n0 = int(stats[2].item()) if (use_bn and stats is not None and stats[2] is not None) else 0
                c.momentum = float(-(n0 + 1))
            else:
                c.momentum = float(meta["momentum"][i])
            c.eps, c.R = float(meta["eps"][i]), R
            c.spikes, c.u, c.xhat, c.f, c.g, c.invstd, c.scratch = spikes.data_ptr(), u.data_ptr(), _dp(xhat), fg.data_ptr(), gg.data_ptr(), _dp(invstd), scr.data_ptr()
            keep.append((z, scr))
            zero = torch.zeros((1,), **f32)
            saved += [x, w_ih_c, w_hh_c, spikes, u, fg, gg, xhat if use_bn else zero, invstd if use_bn else zero, bw if use_bn else zero]
            geo.append((R, I))
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev), _Logged("fwd", T, [(R, H, GH) for R, _ in geo]):
            check(L.sfsn_gsn_train_seq_fwd_multi(calls, n, T, H, int(shared), st), "sfsn_gsn_train_seq_fwd_multi")
        errs = torch.stack([scr[-4:] for _, scr in keep]).max()
        if not any(ctx.needs_input_grad):
            if int(errs.item()) != 0:
                raise RuntimeError(_EXCHANGE_FAILED)
        else:
            _report(errs, dev, final=False)
        if use_bn:
            for i in range(n):
                stats = meta["stats"][i]
                if stats is not None and stats[2] is not None:
                    stats[2].add_(T)  # num_batches_tracked: one BatchNorm call per time step
        ctx.save_for_backward(*saved)
        ctx.fwd_err = errs
        ctx.meta = (shared, use_bn, n, T, H, GH, geo)
        outs = tuple(saved[10 * i + 3] for i in range(n))
        return outs

    @staticmethod
    def backward(ctx, *dys):
        shared, use_bn, n, T, H, GH, geo = ctx.meta
        saved = ctx.saved_tensors
        L = _lib.lib()
        dev = saved[0].device
        f32 = dict(dtype=torch.float32, device=dev)
        calls = (_lib.TrainSeqBwd * n)()
        work = []
        for i in range(n):
            x, w_ih, w_hh, spikes, u, fg, gg, xhat, invstd, bw = saved[10 * i:10 * i + 10]
            R, I = geo[i]
            dy = dys[i]
            dy = torch.zeros((T, R, H), **f32) if dy is None else dy.contiguous().float()
            d_gates = torch.empty((T, R, 2 * H), **f32)
            d_z = torch.empty((T, R, H), **f32) if shared else None
            d_bn_w, d_bn_b = (torch.zeros((H,), **f32), torch.zeros((H,), **f32)) if use_bn else (None, None)
            scr = torch.zeros((L.sfsn_train_seq_scratch_bytes(R, H) // 4,), dtype=torch.int32, device=dev)
            c = calls[i]
            c.w_hh, c.dh_up, c.u, c.f, c.g, c.R = w_hh.data_ptr(), dy.data_ptr(), u.data_ptr(), fg.data_ptr(), gg.data_ptr(), R
            c.xhat, c.invstd, c.bn_w = (xhat.data_ptr(), invstd.data_ptr(), bw.data_ptr()) if use_bn else (None, None, None)
            c.d_gates, c.d_z, c.d_bn_w, c.d_bn_b, c.scratch = d_gates.data_ptr(), _dp(d_z), _dp(d_bn_w), _dp(d_bn_b), scr.data_ptr()
            work.append((dy, d_gates, d_z, d_bn_w, d_bn_b, scr))
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev), _Logged("bwd", T, [(R, H, GH) for R, _ in geo]):
            check(L.sfsn_gsn_train_seq_bwd_multi(calls, n, T, H, int(shared), st), "sfsn_gsn_train_seq_bwd_multi")
        # no host synchronisation here (see GSNLayerTrainFn.backward): NaN-poisoned gradients on a failed exchange, the error word to
        # pinned memory, check_pending() when the backward pass has finished
        errs = torch.maximum(torch.stack([w[5][-4:] for w in work]).max(), ctx.fwd_err)
        poison = torch.where(errs > 0, torch.full((), float("nan"), **f32), torch.zeros((), **f32))
        _report(errs, dev, final=True)
        grads = []
        for i in range(n):
            x, w_ih, w_hh, spikes, u, fg, gg, xhat, invstd, bw = saved[10 * i:10 * i + 10]
            R, I = geo[i]
            dy, d_gates, d_z, d_bn_w, d_bn_b, scr = work[i]
            dz = (d_z if shared else d_gates).reshape(T * R, GH)
            dx = torch.mm(dz, w_ih).view(T, R, I).add_(poison)
            dw_ih = _tn_gemm(dz, x.reshape(T * R, I)).add_(poison)
            dw_hh = (_tn_gemm(dz[R:], spikes[:-1].reshape((T - 1) * R, H)) if T > 1 else torch.zeros((GH, H), **f32)).add_(poison)
            dbias = d_gates.reshape(T * R, 2 * H).sum(0).add_(poison)
            if use_bn:
                d_bn_w.add_(poison)
                d_bn_b.add_(poison)
            grads += [dx, dw_ih, dw_hh, dbias, d_bn_w, d_bn_b]
        return (None, *grads)


class GSNStackTrainFn(torch.autograd.Function):
    """All L layers of n independent cell stacks (one stack, or the sub-band groups of a model), pipelined over chunks of frames
    (round 5): a layer call is T dependent steps on <= ~160 workgroups, and layer l + 1 needs layer l's spikes of frame t only at frame
    t (NEURON:50-62 runs layer l over all T first only because it is written layer by layer).  The sequence is cut into K chunks; stage
    s = 0 .. K + L - 2 is ONE launch (sfsn_gsn_train_seq_fwd_multi) of the calls {(stack i, layer l, chunk s - l)}, layer l >= 1's input
    product of that chunk (a GEMM over the chunk) just in front of it; a call continues from the state its previous chunk left
    (SfsnTrainSeqFwd.h0 / c0 = rows of the spike / membrane tensors), the BatchNorm running statistics continue in launch order.
    Backward the same in reverse (last chunk and last layer first; SfsnTrainSeqBwd.dc_in / dc_out carry dL/dc across the cut, d_z of
    the later chunk is read from the tensor).  The arithmetic of L GSNLayerTrainFn calls per stack -- the same kernels on the same
    numbers; where the library gives the calls fewer, larger row blocks to fit them side by side, the BatchNorm partial sums are merged
    in another blocking (last-bit differences).  K + L - 1 launches of T / K steps per direction instead of L of T steps.
    ``forward(ctx, meta, *xs, *flat)``: xs = n x [T, R_i, I_i]; flat = n x L x (w_ih, w_hh, bias, bn_w, bn_b); meta = dict(n, shared,
    stats [n][L] of (running_mean, running_var, num_batches_tracked) or None, momentum [n][L], eps [n][L], chunks K); returns the
    n x L spike tensors (stack-major)."""

    @staticmethod
    def forward(ctx, meta, *args):
        L_ = _lib.lib()
        _poll_pending(new_forward=True)
        n, K, shared = int(meta["n"]), int(meta["chunks"]), bool(meta["shared"])
        xs, flat = [a.contiguous().float() for a in args[:n]], args[n:]
        nl = len(flat) // (5 * n)
        T = xs[0].shape[0]
        GH, H = flat[1].shape
        Tc = -(-T // K)  # frames per chunk; the last one may be shorter (every call carries its own frame count)
        assert Tc * (K - 1) < T and nl >= 2
        clen = lambda ch: min(Tc, T - ch * Tc)
        dev = xs[0].device
        use_bn = flat[3] is not None
        f32 = dict(dtype=torch.float32, device=dev)
        sH = H * 4
        stk = []  # per stack: dict(R, I0, x, lay=[per layer dict])
        for i in range(n):
            x = xs[i]
            Ti, R, I0 = x.shape
            assert Ti == T
            nscr = L_.sfsn_train_seq_scratch_bytes(R, H) // 4      # words of one call's scratch (the error word: its last four)
            nscr_pad = (nscr + 63) // 64 * 64                      # (one zeroed buffer per layer, a 256-byte aligned row per chunk)
            lay = []
            for l in range(nl):
                w_ih, w_hh, bias, bn_w, bn_b = flat[5 * (i * nl + l):5 * (i * nl + l) + 5]
                stats = meta["stats"][i][l]
                assert tuple(w_hh.shape) == (GH, H) and (bn_w is not None) == use_bn
                if use_bn:
                    if R == 1:  # nn.BatchNorm1d in training mode (torch/nn/functional.py: _verify_batch_size)
                        raise ValueError(f"Expected more than 1 value per channel when training, got input size torch.Size([{R}, {H}])")
                    if stats is not None:
                        for nm, t_ in (("running_mean", stats[0]), ("running_var", stats[1])):
                            if t_.dtype != torch.float32 or not t_.is_contiguous() or t_.device != dev or t_.numel() != H:
                                raise TypeError(f"BatchNorm {nm} must be a contiguous float32 tensor of {H} elements on {dev}, got "
                                                f"{t_.dtype} {tuple(t_.shape)} on {t_.device}")
                d = dict(w_ih=w_ih.detach().contiguous().float(), w_hh=w_hh.detach().contiguous().float(), bias=bias.detach().contiguous().float(),
                         bw=bn_w.detach().contiguous().float() if use_bn else None, bb=bn_b.detach().contiguous().float() if use_bn else None,
                         stats=stats, spikes=torch.empty((T, R, H), **f32), u=torch.empty((T, R, H), **f32), f=torch.empty((T, R, H), **f32),
                         g=torch.empty((T, R, H), **f32), xhat=torch.empty((T, R, H), **f32) if use_bn else None,
                         invstd=torch.empty((T, H), **f32) if use_bn else None,
                         scr=torch.zeros((K, nscr_pad), dtype=torch.int32, device=dev))
                # layer 0's input term for all T at once; the deeper layers' chunk by chunk, behind the chunk's spikes
                d["z"] = torch.mm(x.reshape(T * R, I0), d["w_ih"].t()).view(T, R, GH) if l == 0 else torch.empty((T, R, GH), **f32)
                mom = meta["momentum"][i][l]
                d["n0"] = int(stats[2].item()) if (mom is None and use_bn and stats is not None and stats[2] is not None) else 0
                d["mom"], d["eps"] = mom, float(meta["eps"][i][l])
                lay.append(d)
            stk.append(dict(R=R, I0=I0, x=x, lay=lay, nscr=nscr))
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            for s_ in range(K + nl - 1):
                todo = [(i, l, s_ - l) for l in range(nl) if 0 <= s_ - l < K for i in range(n)]
                calls = (_lib.TrainSeqFwd * len(todo))()
                for c, (i, l, ch) in zip(calls, todo):
                    sk = stk[i]
                    d, t0, R = sk["lay"][l], ch * Tc, sk["R"]
                    sRH, sRG = R * H * 4, R * GH * 4
                    tn = clen(ch)
                    if l >= 1:
                        torch.mm(sk["lay"][l - 1]["spikes"][t0:t0 + tn].reshape(tn * R, H), d["w_ih"].t(), out=d["z"][t0:t0 + tn].view(tn * R, GH))
                    c.z, c.w_hh, c.bias, c.bn_w, c.bn_b = d["z"].data_ptr() + t0 * sRG, d["w_hh"].data_ptr(), d["bias"].data_ptr(), _dp(d["bw"]), _dp(d["bb"])
                    stats = d["stats"]
                    c.running_mean = _dp(stats[0]) if (use_bn and stats is not None) else None
                    c.running_var = _dp(stats[1]) if (use_bn and stats is not None) else None
                    c.momentum = float(-(d["n0"] + t0 + 1)) if d["mom"] is None else float(d["mom"])  # (cumulative average: batches before THIS call)
                    c.eps, c.R = d["eps"], R
                    c.spikes, c.u, c.f, c.g = (d[k].data_ptr() + t0 * sRH for k in ("spikes", "u", "f", "g"))
                    c.xhat = d["xhat"].data_ptr() + t0 * sRH if use_bn else None
                    c.invstd = d["invstd"].data_ptr() + t0 * sH if use_bn else None
                    c.scratch, c.T = d["scr"][ch].data_ptr(), tn
                    if ch > 0:
                        c.h0, c.c0 = d["spikes"].data_ptr() + (t0 - 1) * sRH, d["u"].data_ptr() + (t0 - 1) * sRH
                tmax = max(clen(ch) for _, _, ch in todo)
                with _Logged("fwd", tmax, [(stk[i]["R"], H, GH) for i, _, _ in todo]):
                    check(L_.sfsn_gsn_train_seq_fwd_multi(calls, len(todo), tmax, H, int(shared), st), "sfsn_gsn_train_seq_fwd_multi(stack)")
        errs = torch.stack([d["scr"][:, sk["nscr"] - 4:sk["nscr"]].max() for sk in stk for d in sk["lay"]]).max()
        if not any(ctx.needs_input_grad):
            if int(errs.item()) != 0:
                raise RuntimeError(_EXCHANGE_FAILED)
        else:
            _report(errs, dev, final=False)
        zero = torch.zeros((1,), **f32)
        saved = []
        for sk in stk:
            saved.append(sk["x"])
            for d in sk["lay"]:
                if use_bn and d["stats"] is not None and d["stats"][2] is not None:
                    d["stats"][2].add_(T)  # num_batches_tracked: one BatchNorm call per time step
                saved += [d["w_ih"], d["w_hh"], d["spikes"], d["u"], d["f"], d["g"], d["xhat"] if use_bn else zero, d["invstd"] if use_bn else zero,
                          d["bw"] if use_bn else zero]
        ctx.save_for_backward(*saved)
        ctx.fwd_err = errs
        # unused outputs arrive as None in backward() instead of materialised zero tensors (only the last layer's spikes are consumed in
        # the live recipe: without this every other layer paid a [T, R, H] buffer of zeros and a dh.add_(zeros) per chunk)
        ctx.set_materialize_grads(False)
        ctx.meta = (shared, use_bn, n, nl, K, T, H, GH, [(sk["R"], sk["I0"]) for sk in stk])
        return tuple(d["spikes"] for sk in stk for d in sk["lay"])

    @staticmethod
    def backward(ctx, *dys):
        shared, use_bn, n, nl, K, T, H, GH, geo = ctx.meta
        saved = ctx.saved_tensors
        L_ = _lib.lib()
        dev = saved[0].device
        f32 = dict(dtype=torch.float32, device=dev)
        Tc = -(-T // K)
        clen = lambda ch: min(Tc, T - ch * Tc)
        sH = H * 4
        per = 1 + 9 * nl
        stk = []
        for i in range(n):
            R, I0 = geo[i]
            x = saved[per * i]
            nscr = L_.sfsn_train_seq_scratch_bytes(R, H) // 4
            nscr_pad = (nscr + 63) // 64 * 64
            lay = []
            for l in range(nl):
                w_ih, w_hh, spikes, u, fg, gg, xhat, invstd, bw = saved[per * i + 1 + 9 * l:per * i + 10 + 9 * l]
                dy = dys[i * nl + l]
                d = dict(w_ih=w_ih, w_hh=w_hh, spikes=spikes, u=u, f=fg, g=gg, xhat=xhat, invstd=invstd, bw=bw,
                         d_gates=torch.empty((T, R, 2 * H), **f32), d_z=torch.empty((T, R, H), **f32) if shared else None,
                         d_bn_w=torch.zeros((H,), **f32) if use_bn else None, d_bn_b=torch.zeros((H,), **f32) if use_bn else None,
                         dc=torch.empty((2, R, H), **f32), scr=torch.zeros((K, nscr_pad), dtype=torch.int32, device=dev))
                # the gradient w.r.t. this layer's spikes: from the layer above (formed chunk by chunk, below) plus whatever reads the
                # returned tensor directly (all_layer_outputs)
                if l == nl - 1:
                    d["dh"] = torch.zeros((T, R, H), **f32) if dy is None else dy.contiguous().float()
                else:
                    d["dh"], d["dy"] = torch.empty((T, R, H), **f32), (None if dy is None else dy.contiguous().float())
                d["dzs"] = d["d_z"] if shared else d["d_gates"]  # the gradient of the (shared or per-gate) products: [T][R][G*H]
                lay.append(d)
            stk.append(dict(R=R, I0=I0, x=x, lay=lay, nscr=nscr))
        main = torch.cuda.current_stream(dev)
        st = ctypes.c_void_p(main.cuda_stream)
        side = None
        if STACK_SIDE_STREAM and not torch.cuda.is_current_stream_capturing():
            side = _side_streams.get((dev.index, main.cuda_stream))
            if side is None:
                side = _side_streams[(dev.index, main.cuda_stream)] = torch.cuda.Stream(device=dev)
        for sk in stk:  # accumulators of the per-chunk weight gradients
            for l, d in enumerate(sk["lay"]):
                d["dw_ih"] = torch.zeros((GH, sk["I0"] if l == 0 else H), **f32)
                d["dw_hh"], d["dbias"] = torch.zeros((GH, H), **f32), torch.zeros((2 * H,), **f32)
            sk["dx"] = torch.empty((T, sk["R"], sk["I0"]), **f32)
        if side is not None:
            ev0 = torch.cuda.Event()
            ev0.record(main)
            side.wait_event(ev0)  # (the zeroed accumulators)

        GB = max(1, -(-K // STACK_GRAD_BLOCKS))  # chunks per block of the weight-gradient GEMMs (a GEMM per chunk: +15 ms at B = 64)

        def chunk_grads(i, l, ch):
            """what the chunks [ch, ch + GB) of (stack i, layer l) add to the weight gradients, once the first of them (the last one
            processed) is done; layer 0: dL/dx of those frames"""
            if ch % GB:
                return
            sk = stk[i]
            d, t0, R, I0 = sk["lay"][l], ch * Tc, sk["R"], sk["I0"]
            Tb = min((min(ch + GB, K) - ch) * Tc, T - t0)
            dz = d["dzs"][t0:t0 + Tb].reshape(Tb * R, GH)
            inp = sk["x"][t0:t0 + Tb].reshape(Tb * R, I0) if l == 0 else sk["lay"][l - 1]["spikes"][t0:t0 + Tb].reshape(Tb * R, H)
            d["dw_ih"].add_(_tn_gemm(dz, inp))
            # dL/dW_hh = sum_t dz_t^T h_{t-1}, h_{-1} = 0: frames t0 .. t0 + Tb - 1 against spikes t0 - 1 .. (views: no shifted copy)
            a = 1 if t0 == 0 else 0
            if Tb - a > 0:
                d["dw_hh"].add_(_tn_gemm(d["dzs"][t0 + a:t0 + Tb].reshape((Tb - a) * R, GH), d["spikes"][t0 + a - 1:t0 + Tb - 1].reshape((Tb - a) * R, H)))
            d["dbias"].add_(d["d_gates"][t0:t0 + Tb].reshape(Tb * R, 2 * H).sum(0))
            if l == 0:
                torch.mm(dz, d["w_ih"], out=sk["dx"][t0:t0 + Tb].view(Tb * R, I0))

        with torch.cuda.device(dev):
            for s_ in range(K + nl - 1):
                todo = [(i, nl - 1 - k, K - 1 - (s_ - k)) for k in range(nl) if 0 <= s_ - k < K for i in range(n)]
                calls = (_lib.TrainSeqBwd * len(todo))()
                for c, (i, l, ch) in zip(calls, todo):
                    sk = stk[i]
                    d, t0, R = sk["lay"][l], ch * Tc, sk["R"]
                    sRH, s2H = R * H * 4, R * 2 * H * 4
                    tn = clen(ch)
                    if l < nl - 1:  # dL/d(spikes of layer l), chunk ch = d_z of layer l + 1 (made by the previous stage) . W_ih of layer l + 1
                        up = sk["lay"][l + 1]
                        torch.mm(up["dzs"][t0:t0 + tn].reshape(tn * R, GH), up["w_ih"], out=d["dh"][t0:t0 + tn].view(tn * R, H))
                        if d["dy"] is not None:
                            d["dh"][t0:t0 + tn].add_(d["dy"][t0:t0 + tn])
                    c.w_hh, c.dh_up, c.R = d["w_hh"].data_ptr(), d["dh"].data_ptr() + t0 * sRH, R
                    c.u, c.f, c.g = (d[k].data_ptr() + t0 * sRH for k in ("u", "f", "g"))
                    if use_bn:
                        c.xhat, c.invstd, c.bn_w = d["xhat"].data_ptr() + t0 * sRH, d["invstd"].data_ptr() + t0 * sH, d["bw"].data_ptr()
                    c.d_gates = d["d_gates"].data_ptr() + t0 * s2H
                    c.d_z = d["d_z"].data_ptr() + t0 * sRH if shared else None
                    c.d_bn_w, c.d_bn_b, c.scratch = _dp(d["d_bn_w"]), _dp(d["d_bn_b"]), d["scr"][ch].data_ptr()
                    c.dc_in = d["dc"][(ch + 1) & 1].data_ptr() if ch < K - 1 else None
                    c.dc_out = d["dc"][ch & 1].data_ptr() if ch > 0 else None
                    c.has_prev, c.T = int(ch > 0), tn
                tmax = max(clen(ch) for _, _, ch in todo)
                with _Logged("bwd", tmax, [(stk[i]["R"], H, GH) for i, _, _ in todo]):
                    check(L_.sfsn_gsn_train_seq_bwd_multi(calls, len(todo), tmax, H, int(shared), st), "sfsn_gsn_train_seq_bwd_multi(stack)")
                if side is not None:
                    ev = torch.cuda.Event()
                    ev.record(main)
                    side.wait_event(ev)
                with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                    for i, l, ch in todo:
                        chunk_grads(i, l, ch)
            if side is not None:
                ev = torch.cuda.Event()
                ev.record(side)
                main.wait_event(ev)
        # no host synchronisation here (see GSNLayerTrainFn.backward): NaN-poisoned gradients on a failed exchange, the error word to
        # pinned memory, check_pending() when the backward pass has finished
        errs = torch.maximum(torch.stack([d["scr"][:, sk["nscr"] - 4:sk["nscr"]].max() for sk in stk for d in sk["lay"]]).max(), ctx.fwd_err)
        poison = torch.where(errs > 0, torch.full((), float("nan"), **f32), torch.zeros((), **f32))
        _report(errs, dev, final=True)
        dxs, grads = [], []
        for sk in stk:
            for d in sk["lay"]:
                if use_bn:
                    d["d_bn_w"].add_(poison)
                    d["d_bn_b"].add_(poison)
                grads += [d["dw_ih"].add_(poison), d["dw_hh"].add_(poison), d["dbias"].add_(poison), d["d_bn_w"], d["d_bn_b"]]
            dxs.append(sk["dx"].add_(poison))
        return (None, *dxs, *grads)


def _dp(t):
    return None if t is None else t.data_ptr()


def _cell_args(cell):
    bn = getattr(cell, "batchnorm", None) if cell.use_bn else None
    stats, momentum, eps = None, 0.1, 1e-5
    if bn is not None:
        stats = (bn.running_mean, bn.running_var, bn.num_batches_tracked)
        momentum, eps = bn.momentum, bn.eps  # (momentum None = cumulative moving average: rejected in training mode)
    return bn, stats, momentum, eps


def gsn_stacks(xs, stacks, training: bool):
    """StackedGSU.forward of several independent stacks (the sub-band groups): layer l of all of them in one launch per direction when
    the launch can hold their workgroups together (same depth / hidden size / gate sharing / BatchNorm use, training-mode statistics
    or no BatchNorm); one stack after the other otherwise.  Returns a list of [x, S1, ..., SL] lists."""
    n = len(stacks)
    if GROUPS_TOGETHER and 1 < n:
        K = _stack_chunks(xs, stacks, training)
        if K > 1:
            return _pipelined_stacks([x.contiguous() for x in xs], stacks, K)
    def same(fn):
        return len({fn(st) for st in stacks}) == 1
    ok = (GROUPS_TOGETHER and not STEP_LAUNCHES and 1 < n <= _lib.TRAIN_MAX_CALLS and same(lambda st: len(st.layers))
          and all(x.is_cuda and x.shape[0] == xs[0].shape[0] for x in xs))
    if ok:
        for l in range(len(stacks[0].layers)):
            cells = [st.layers[l].cell for st in stacks]
            ok = ok and len({(tuple(c.weight_hh.shape), bool(c.shared_weights), bool(c.use_bn)) for c in cells}) == 1
            ok = ok and (not cells[0].use_bn or training)  # (eval-mode BatchNorm: the folded per-step path of GSNLayerTrainFn)
    if ok:
        L = _lib.lib()
        H = stacks[0].layers[0].cell.weight_hh.shape[1]
        Rs = (ctypes.c_int * n)(*[int(x.shape[1]) for x in xs])
        with torch.cuda.device(xs[0].device):
            ok = L.sfsn_gsn_train_multi_check(Rs, n, H, int(bool(stacks[0].layers[0].cell.shared_weights))) == _lib.SFSN_OK
    if not ok:
        return [gsn_stack(x, st, training) for x, st in zip(xs, stacks)]
    outs = [[x] for x in xs]
    cur = list(xs)
    for l in range(len(stacks[0].layers)):
        flat, stats, moms, epss = [], [], [], []
        for i, st in enumerate(stacks):
            cell = st.layers[l].cell
            bn, stt, mom, eps = _cell_args(cell)
            flat += [cur[i], cell.weight_ih, cell.weight_hh, cell.bias_ih, None if bn is None else bn.weight, None if bn is None else bn.bias]
            stats.append(stt); moms.append(mom); epss.append(eps)
        meta = dict(shared=bool(stacks[0].layers[l].cell.shared_weights), stats=stats, momentum=moms, eps=epss)
        cur = list(GSNLayersTrainFn.apply(meta, *flat))
        for i in range(n):
            outs[i].append(cur[i])
    return outs


def _stack_chunks(xs, stacks, training: bool) -> int:
    """Chunks of frames for GSNStackTrainFn, or 1: stacks of one depth >= 2 and one cell shape (hidden size, gate sharing, BatchNorm
    use) in every layer of every stack, the one-launch layer calls (training-mode BatchNorm or none), two layers of every stack resident
    TOGETHER (sfsn_gsn_train_multi_check with one call per stack and layer in flight -- the library gives them larger row blocks when
    it must), and chunks of at least STACK_MIN_FRAMES frames (the last chunk may be shorter: every call of a launch carries its own
    frame count)."""
    K = STACK_CHUNKS
    nl = len(stacks[0].layers)
    if K < 2 or STEP_LAUNCHES or nl < 2 or any(len(st.layers) != nl for st in stacks) or len(stacks) * nl > _lib.TRAIN_MAX_CALLS:
        return 1
    if not all(x.is_cuda and x.dim() == 3 and x.shape[0] == xs[0].shape[0] for x in xs):
        return 1
    cells = [layer.cell for st in stacks for layer in st.layers]
    if len({(tuple(c.weight_hh.shape), bool(c.shared_weights), bool(c.use_bn)) for c in cells}) != 1 or (cells[0].use_bn and not training):
        return 1
    H = cells[0].weight_hh.shape[1]
    if any(layer.cell.weight_ih.shape[1] != H for st in stacks for layer in st.layers[1:]):
        return 1
    T = int(xs[0].shape[0])
    K = min(K, T // max(1, STACK_MIN_FRAMES))  # chunks of ceil(T / K) frames, the last one shorter
    while K > 1 and -(-T // K) * (K - 1) >= T:  # (rounding up must not leave the last chunk empty)
        K -= 1
    if K < 2:
        return 1
    Rs = [int(x.shape[1]) for x in xs] * nl  # (every layer of every stack in flight at once: the pipeline's full stages)
    with torch.cuda.device(xs[0].device):
        if _lib.lib().sfsn_gsn_train_multi_check((ctypes.c_int * len(Rs))(*Rs), len(Rs), H, int(bool(cells[0].shared_weights))) != _lib.SFSN_OK:
            return 1
    return K


def _pipelined_stacks(xs, stacks, K):
    """[x, S1, ..., SL] per stack through GSNStackTrainFn."""
    global _STACK_CALLS
    nl = len(stacks[0].layers)
    flat, stats, moms, epss = [], [], [], []
    for st in stacks:
        s_, m_, e_ = [], [], []
        for layer in st.layers:
            bn, stt, mom, eps = _cell_args(layer.cell)
            flat += [layer.cell.weight_ih, layer.cell.weight_hh, layer.cell.bias_ih, None if bn is None else bn.weight, None if bn is None else bn.bias]
            s_.append(stt); m_.append(mom); e_.append(eps)
        stats.append(s_); moms.append(m_); epss.append(e_)
    meta = dict(n=len(stacks), shared=bool(stacks[0].layers[0].cell.shared_weights), stats=stats, momentum=moms, eps=epss, chunks=K)
    _STACK_CALLS += 1
    res = GSNStackTrainFn.apply(meta, *xs, *flat)
    return [[x] + list(res[i * nl:(i + 1) * nl]) for i, x in enumerate(xs)]


def gsn_stack(x: torch.Tensor, stack, training: bool) -> List[torch.Tensor]:
    """StackedGSU.forward (efficient_spiking_neuron.py:50-62) on the module's parameter containers: [x, S1, ..., SL]."""
    outs = [x]
    cur = x
    K = _stack_chunks([x], [stack], training)
    if K > 1:
        return _pipelined_stacks([x.contiguous()], [stack], K)[0]
    for layer in stack.layers:
        cell = layer.cell
        bn = getattr(cell, "batchnorm", None) if cell.use_bn else None
        stats = None
        momentum, eps = 0.1, 1e-5
        if bn is not None:
            stats = (bn.running_mean, bn.running_var, bn.num_batches_tracked)
            momentum = bn.momentum  # (None = cumulative moving average: rejected by GSNLayerTrainFn in training mode)
            eps = bn.eps
        cur = GSNLayerTrainFn.apply(cur, cell.weight_ih, cell.weight_hh, cell.bias_ih, None if bn is None else bn.weight,
                                    None if bn is None else bn.bias, stats, cell.shared_weights, bool(training and bn is not None),
                                    momentum, eps)
        outs.append(cur)
    return outs


def sequence_model(seq, x_bft: torch.Tensor, training: bool, time_major: bool = False):
    """SequenceModel.forward (modeling_spiking_fullsubnet.py:81-125; LSTM variant :68-79): [R, I, T] -> ([R, P, T], all_layer_outputs).
    time_major: the input is [T, R, I] already (forward_live builds the rows that way: no transposed copy of the feature tensor)."""
    if seq.sequence_model_name == "LSTM":
        x = x_bft.permute(1, 0, 2) if time_major else x_bft.permute(0, 2, 1)
        if seq.use_pre_layer_norm:
            x = seq.pre_layer_norm(x)
        y, _ = seq.sequence_model(x)
        y = seq.output_activate_function(seq.proj(y))
        return y.permute(0, 2, 1), []
    x = x_bft if time_major else x_bft.permute(2, 0, 1)  # time-major
    if seq.use_pre_layer_norm:
        x = _pre_ln(seq.pre_layer_norm, x)
    outs = gsn_stack(x.contiguous(), seq.sequence_model, training)
    y = _proj(seq.proj, outs[-1])
    outs = outs + [y]
    return seq.output_activate_function(y).permute(1, 2, 0), outs


def sequence_models(seqs, xs_bft, training: bool, time_major: bool = False):
    """SequenceModel.forward of several independent models (the sub-band groups): their cell stacks go through gsn_stacks (layer l of
    all of them in one launch per direction).  Returns [(y [R, P, T], all_layer_outputs), ...]."""
    if any(seq.sequence_model_name == "LSTM" for seq in seqs) or len(seqs) < 2:
        return [sequence_model(seq, x, training, time_major) for seq, x in zip(seqs, xs_bft)]
    xs = []
    for seq, x_bft in zip(seqs, xs_bft):
        x = x_bft if time_major else x_bft.permute(2, 0, 1)  # time-major
        if seq.use_pre_layer_norm:
            x = _pre_ln(seq.pre_layer_norm, x)
        xs.append(x.contiguous())
    res = []
    for seq, outs in zip(seqs, gsn_stacks(xs, [seq.sequence_model for seq in seqs], training)):
        y = _proj(seq.proj, outs[-1])
        res.append((seq.output_activate_function(y).permute(1, 2, 0), outs + [y]))
    return res


def _deep_filter(y: torch.Tensor, band: torch.Tensor, N: int, c: int, d: int, S: int) -> torch.Tensor:
    """deepfiltering (MODEL:315-346) of one sub-band group: y [B N, P, T] with channel p = ((ri c + fci) d + di) S + si (the projection's
    output), band [B, N c, T] complex (the noisy bins of the group) -> [B, S, N c, T] complex,
    Y[f, t] = sum_di X[f, t - (d - 1) + di] C[di, f, t] with zeros left of the first frame.  One complex product over all taps and one sum
    (round 6: the per-tap loop over real and imaginary planes was ~230 element-wise launches per training step, 3.2 -> 1.1 ms)."""
    B, T = band.shape[0], band.shape[2]
    coef = y.reshape(B, N, 2, c, d, S, T)
    if FAST_GLUE:
        cc = torch.complex(coef[:, :, 0], coef[:, :, 1]).permute(0, 3, 4, 1, 2, 5).reshape(B, d, S, N * c, T)
        taps = F.pad(band, (d - 1, 0)).unfold(2, T, 1).permute(0, 2, 1, 3)                                    # [B, d, N c, T] (a view)
        return (taps[:, :, None] * cc).sum(1)
    cre = coef[:, :, 0].permute(0, 3, 4, 1, 2, 5).reshape(B, d, S, N * c, T)
    cim = coef[:, :, 1].permute(0, 3, 4, 1, 2, 5).reshape(B, d, S, N * c, T)
    xp = F.pad(torch.view_as_real(band), (0, 0, d - 1, 0))                                                   # causal: zeros on the left of T
    xr, xi = xp[..., 0], xp[..., 1]
    yr = yi = 0
    for di in range(d):
        a, b = xr[:, None, :, di:di + T], xi[:, None, :, di:di + T]
        yr = yr + a * cre[:, di] - b * cim[:, di]
        yi = yi + a * cim[:, di] + b * cre[:, di]
    return torch.complex(yr, yi)


def _frozen_sequence_models(seqs, xs_bft, training: bool):
    """The frozen SequenceModel.forward (model_low_freq.py:100-139) of several independent models, cell stacks through gsn_stacks."""
    xs = [x_bft.permute(2, 0, 1).contiguous() for x_bft in xs_bft]  # [B, F, T] => [T, B, F]
    res = []
    for seq, outs in zip(seqs, gsn_stacks(xs, [seq.sequence_model for seq in seqs], training)):
        y = seq.fc_output_layer(outs[-1])
        res.append((y.permute(1, 2, 0), outs + [y]))
    return res


def _istft(spec: torch.Tensor, n_fft: int, hop: int, win_length: int, window: torch.Tensor, length) -> torch.Tensor:
    """torch.istft (center=True, onesided, not normalised: audio_feature.py:297-347) as differentiable ATen operations WITHOUT its
    host-side check of the window envelope (`window_envelop.abs().min() > 1e-11` is a device-to-host read: a synchronisation per training
    step, and not capturable in a HIP graph -- GraphedTrainStep).  The same arithmetic: inverse real FFT of every frame, times the
    window, overlap-add, divided by the overlap-added squared window; the envelope of a Hann window at hop <= n_fft / 2 is positive on
    every sample that is kept, the case the check exists for cannot occur."""
    B, Fq, T = spec.shape
    if win_length != n_fft:
        left = (n_fft - win_length) // 2
        window = F.pad(window, (left, n_fft - win_length - left))
    total = n_fft + hop * (T - 1)

    def overlap_add(fr):  # [b, T, n_fft] -> [b, total]
        if n_fft % hop == 0:  # frame t's k-th hop-sized piece lands on piece t + k: n_fft / hop shifted sums (every reference config)
            r = n_fft // hop
            pieces = fr.reshape(fr.shape[0], T, r, hop)
            acc = F.pad(pieces[:, :, 0], (0, 0, 0, r - 1))
            for k in range(1, r):
                acc = acc + F.pad(pieces[:, :, k], (0, 0, k, r - 1 - k))
            return acc.reshape(fr.shape[0], total)
        return F.fold(fr.transpose(1, 2), output_size=(1, total), kernel_size=(1, n_fft), stride=(1, hop)).reshape(fr.shape[0], total)

    frames = torch.fft.irfft(spec.transpose(1, 2).contiguous(), n=n_fft, dim=-1) * window                   # [B, T, n_fft]
    y = overlap_add(frames)
    env = overlap_add((window * window)[None, None, :].expand(1, T, n_fft))[0]
    start = n_fft // 2
    end = total - start if length is None else start + int(length)
    y = y[:, start:min(end, total)] / env[start:min(end, total)]
    return F.pad(y, (0, end - total)) if end > total else y


def _reflect(idx: torch.Tensor, nf: int) -> torch.Tensor:
    idx = torch.where(idx < 0, -idx, idx)
    return torch.where(idx > nf - 1, 2 * (nf - 1) - idx, idx)


def forward_live(model, wave: torch.Tensor):
    """SpikingFullSubNet.forward (modeling_spiking_fullsubnet.py:415-474) on differentiable operations."""
    assert wave.ndim == 2, f"Input tensor must be 2D, but got {wave.ndim}D."
    if not wave.is_cuda:
        raise RuntimeError("spiking_fullsubnet_amd has no CPU path: move the module and its input to a HIP device")
    B, length = wave.shape
    dev = wave.device
    window = torch.hann_window(model.n_fft, device=dev)
    noisy = torch.stft(wave, model.n_fft, model.hop_length, model.win_length, window=window, return_complex=True, pad_mode="constant")
    Fq, T = noisy.shape[1], noisy.shape[2]
    nf = Fq - 1
    mag = noisy.abs() ** model.fdrc
    mag = mag[:, :nf]                                           # the Nyquist bin is passed through untouched
    S = model.num_spks
    training = model.training
    fb_in = model.fb_input_size
    # round 6: the feature rows are assembled time-major ([T, B N, I]: what the cell stacks read) from a transposed copy of the 65 MB
    # magnitude instead of transposing the 230 MB of gathered rows afterwards
    tm = bool(FAST_GLUE)
    mag_t = mag.permute(2, 0, 1).contiguous() if tm else None                                                 # [T, B, nf]
    fb_out, fb_all = sequence_model(model.fb_model, mag_t[:, :, :fb_in] if tm else mag[:, :fb_in], training, tm)  # [B, P, T]
    fb_t = fb_out.permute(2, 0, 1) if tm else None                                                            # [T, B, P] (the projection's own layout)
    P_fb = fb_out.shape[1]
    sb = model.sb_model
    cut = list(sb.freq_cutoffs)
    enh_groups, sb_all = [], []
    for g in range(len(sb.sb_models)):  # (every group's ValueError before any group's launches)
        lo, hi, c = cut[g], cut[g + 1], sb.center_freq_sizes[g]
        if (hi - lo) % c != 0:
            raise ValueError(f"Number of frequency bins must be divisible by the center frequency.GOT: ctr_freq={c}, "
                             f"upper_cutoff_freq={hi}, lower_cutoff_freq={lo}")
    xs_g = []
    for g, seq in enumerate(sb.sb_models):
        lo, hi, c, n = cut[g], cut[g + 1], sb.center_freq_sizes[g], sb.neighbor_freq_sizes[g]
        N = (hi - lo) // c
        k = torch.arange(N, device=dev)
        idx_noisy = _reflect(lo + k[:, None] * c - n + torch.arange(c + 2 * n, device=dev)[None, :], nf)   # [N, c + 2n]
        idx_fb = (lo + k[:, None] * c + torch.arange(c, device=dev)[None, :]) % P_fb                        # [N, c] (tiled full-band output)
        if tm:
            x = torch.cat([mag_t[:, :, idx_noisy], fb_t[:, :, idx_fb]], dim=3)                               # [T, B, N, I]
            xs_g.append(x.reshape(T, B * N, x.shape[3]))
        else:
            x = torch.cat([mag[:, idx_noisy], fb_out[:, idx_fb]], dim=2)                                     # [B, N, I, T]
            xs_g.append(x.reshape(B * N, x.shape[2], T))
    ys_g = sequence_models(list(sb.sb_models), xs_g, training, tm)                                           # [(y [B N, P, T], outs)]
    for g, seq in enumerate(sb.sb_models):
        lo, hi, c, n, d = cut[g], cut[g + 1], sb.center_freq_sizes[g], sb.neighbor_freq_sizes[g], sb.df_orders[g]
        N = (hi - lo) // c
        y, outs = ys_g[g]
        sb_all.append(outs)
        # projection channel p = ((ri * c + fci) * d + di) * S + si  ->  coefficient [B, di, si, n * c + fci, T] (re, im)
        enh_groups.append(_deep_filter(y, noisy[:, lo:hi], N, c, d, S))                                      # [B, S, N c, T]
    enh = torch.cat(enh_groups, dim=2)
    enh_stft = torch.cat([enh, noisy[:, None, enh.shape[2]:].expand(B, S, Fq - enh.shape[2], T)], dim=2)     # bins past the groups pass through
    enh_y = _istft(enh_stft.reshape(B * S, Fq, T), model.n_fft, model.hop_length, model.win_length, window, length)
    if S > 1:
        return enh_y.reshape(B, S, -1), fb_all, sb_all
    return enh_y, enh_stft[:, 0].abs(), fb_all, sb_all


_LAPLACE_EPS = 2.220446049250313e-16  # audiozen/constant.py:11 (np.finfo(np.float32).eps is NOT what the reference adds)


def _frozen_sequence_model(seq, x_bft: torch.Tensor, training: bool):
    """The frozen SequenceModel.forward (model_low_freq.py:100-139): no LayerNorm, ``fc_output_layer``, no activation in any config."""
    x = x_bft.permute(2, 0, 1).contiguous()  # [B, F, T] => [T, B, F]
    outs = gsn_stack(x, seq.sequence_model, training)
    y = seq.fc_output_layer(outs[-1])
    return y.permute(1, 2, 0), outs + [y]


def forward_frozen(model, wave: torch.Tensor):
    """Separator.forward (recipes/intel_ndns/spiking_fullsubnet_freeze_phase/model_low_freq.py:561-618) on differentiable
    operations, for a module in training mode or an input that requires grad: the utterance-level ``offline_laplace_norm`` (:147-169:
    x / (mean over every non-batch dimension + EPSILON)) on the full-band input (:578) and on every group's concatenated sub-band
    input (:475), reflect-unfolded noisy AND full-band features (:350-431: the tiled full-band output has its own centre / neighbour
    sizes), the cell loop on the HIP training kernels (GSNLayerTrainFn / GSNLayersTrainFn), deep filtering and reconstruction as in forward_live."""
    ndim = wave.dim()
    assert ndim in (2, 3), "Input must be 2D (B, T) or 3D tensor (B, 1, T)"
    if ndim == 3:
        assert wave.size(1) == 1, "Input must be 2D (B, T) or 3D tensor (B, 1, T)"
        wave = wave.squeeze(1)
    if not wave.is_cuda:
        raise RuntimeError("spiking_fullsubnet_amd has no CPU path: move the module and its input to a HIP device")
    spec = model._spec()
    if spec.cum_laplace:
        raise NotImplementedError("cumulative_laplace_norm has no differentiable path (the reference's own Separator raises on it, "
                                  "model_low_freq.py:172-202); train with offline_laplace_norm")
    B, length = wave.shape
    dev = wave.device
    window = torch.hann_window(model.win_length, device=dev)
    noisy = torch.stft(wave, model.n_fft, model.hop_length, model.win_length, window=window, return_complex=True, pad_mode="constant")
    Fq, T = noisy.shape[1], noisy.shape[2]
    nf = Fq - 1
    mag = (noisy.abs() ** model.fdrc)[:, :nf]                      # [B, nf, T]: the Nyquist bin is passed through untouched
    training = model.training

    def laplace(x):  # one scalar per batch item over everything else (non-causal)
        dims = tuple(range(1, x.dim()))
        mu = x.mean(dim=dims, keepdim=True)
        if spec.gaussian:  # offline_gaussian_norm (:205-218): torch.std = the unbiased estimate
            return (x - mu) / (x.std(dim=dims, keepdim=True) + _LAPLACE_EPS)
        return x / (mu + _LAPLACE_EPS)
    fb_in = model.fb_freqs
    fb_out, fb_all = _frozen_sequence_model(model.fb_model, laplace(mag[:, :fb_in]), training)   # [B, P, T]
    P_fb = fb_out.shape[1]
    sbm = model.sb_model
    cut = [0] + list(sbm.freq_cutoffs) + [nf]
    enh_groups, sb_all = [], []
    for g in range(len(sbm.sb_models)):
        lo, hi, c, cf = cut[g], cut[g + 1], sbm.sb_num_center_freqs[g], sbm.fb_num_center_freqs[g]
        if (hi - lo) % c != 0 or (hi - lo) % cf != 0 or (hi - lo) // cf != (hi - lo) // c:
            raise ValueError(f"Number of frequency bins must be divisible by the center frequency.GOT: ctr_freq={c}, "
                             f"upper_cutoff_freq={hi}, lower_cutoff_freq={lo}")
    xs_g = []
    for g, seq in enumerate(sbm.sb_models):
        lo, hi = cut[g], cut[g + 1]
        c, n = sbm.sb_num_center_freqs[g], sbm.sb_num_neighbor_freqs[g]
        cf, nfb = sbm.fb_num_center_freqs[g], sbm.fb_num_neighbor_freqs[g]
        N = (hi - lo) // c
        k = torch.arange(N, device=dev)
        idx_noisy = _reflect(lo + k[:, None] * c - n + torch.arange(c + 2 * n, device=dev)[None, :], nf)        # [N, c + 2n]
        idx_fb = _reflect(lo + k[:, None] * cf - nfb + torch.arange(cf + 2 * nfb, device=dev)[None, :], nf) % P_fb  # tiled full-band output
        x = laplace(torch.cat([mag[:, idx_noisy], fb_out[:, idx_fb]], dim=2))                                    # [B, N, I, T]
        xs_g.append(x.reshape(B * N, x.shape[2], T))
    ys_g = _frozen_sequence_models(list(sbm.sb_models), xs_g, training)                                          # [(y [B N, P, T], outs)]
    for g, seq in enumerate(sbm.sb_models):
        lo, hi = cut[g], cut[g + 1]
        c = sbm.sb_num_center_freqs[g]
        d = seq.df_order
        N = (hi - lo) // c
        y, outs = ys_g[g]
        sb_all.append(outs)
        enh_groups.append(_deep_filter(y, noisy[:, lo:hi], N, c, d, 1)[:, 0])   # channel p = (ri * c + fci) * d + di  (:262-268)  [B, N c, T]
    enh = torch.cat(enh_groups, dim=1)
    enh_stft = torch.cat([enh, noisy[:, enh.shape[1]:]], dim=1)                                                  # bins past the groups pass through
    enh_y = _istft(enh_stft, model.n_fft, model.hop_length, model.win_length, window, length)
    return enh_y, enh_stft.abs(), fb_all, sb_all


class GraphedTrainStep:
    """One whole training step -- ``model(wave)``, ``loss_fn(outputs)``, ``loss.backward()`` and, when given, ``optimizer.step()`` --
    captured ONCE in a HIP graph and replayed for every batch of the same shape (PyTorch's whole-network capture recipe).

    Why: a step of the live baseline_m model at B = 64 is ~1000 kernels -- 84 resident layer-call launches with the library GEMMs and
    element-wise operations of the chunk pipeline between them -- of 44 ms device time together, and the host needs longer than that
    to enqueue them one by one (74.9 ms per step measured, the device idle 41 % of it: profiles/r06_training_kernel_stats.csv).
    Replayed from a graph the host's share is one call.

    The recipe's trainer (recipes/intel_ndns/spiking_fullsubnet/trainer.py:24-48) feeds fixed-length clips, so the shapes are static;
    a batch of another shape needs its own GraphedTrainStep (or the eager path: ``model(wave)`` as before).

    * ``example_wave`` [B, samples] fixes the shape; ``loss_fn(outputs) -> scalar``; the warm-up iterations (eager, on a side stream:
      library handles, occupancy queries, the allocator's pool) run forward + backward only and the module's parameters and buffers
      are put back afterwards (BatchNorm running statistics included), so capturing does not train.
    * ``step(wave)`` copies the batch into the graph's input, replays, and returns the loss tensor (static: overwritten by the next
      call); ``.outputs`` is the captured forward's return value, the gradients are in ``p.grad`` (static tensors written by every replay:
      they are re-attached if someone set them to None in between).
    * The layer calls' error words (a failed row-block exchange, see check_pending) are reduced INSIDE the graph and read after every
      replay (one host synchronisation per step, as on the eager path): ``step`` raises RuntimeError -- if an optimizer step is part of
      the graph the NaN-poisoned gradients have reached the weights by then, and the message says so.
    * BatchNorm with ``momentum=None`` (cumulative average: the factor depends on a host-side counter) cannot be captured."""

    def __init__(self, model, example_wave: torch.Tensor, loss_fn, optimizer=None, warmup: int = 2):
        global _capture_errs
        if not example_wave.is_cuda:
            raise RuntimeError("spiking_fullsubnet_amd has no CPU path: move the module and its input to a HIP device")
        for m_ in model.modules():
            if isinstance(m_, torch.nn.modules.batchnorm._BatchNorm) and m_.momentum is None and model.training:
                raise NotImplementedError("GraphedTrainStep: BatchNorm with momentum=None (cumulative moving average) needs a host-side "
                                          "counter per step and cannot be captured")
        self.model, self.loss_fn, self.optimizer = model, loss_fn, optimizer
        dev = example_wave.device
        self.static_wave = example_wave.detach().clone()
        self.params = [p for p in model.parameters() if p.requires_grad]
        check_pending()
        keep = {k: v.detach().clone() for k, v in model.state_dict().items()}
        main = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(main)
        stale = False
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._drop_grads()
                with warnings.catch_warnings(record=True) as seen:
                    warnings.simplefilter("always")
                    loss_fn(model(self.static_wave)).backward()
                stale = stale or any("AccumulateGrad node's stream does not match" in str(w_.message) for w_ in seen)
        main.wait_stream(side)
        if stale:
            # the parameters' gradient accumulators were created by an earlier EAGER step on another stream and something still holds
            # that step's autograd graph (a loss kept for logging, outputs): the captured backward would synchronise with that stream
            # -- on ROCm 7 the capture then dies inside hipStreamEndCapture (a segmentation fault, measured)
            raise RuntimeError("GraphedTrainStep: an autograd graph of an earlier eager step is still alive (a loss or output tensor kept "
                               "around?) and pins the parameters' gradient accumulators to another stream -- drop those tensors "
                               "(`del loss`, `loss = loss.item()`) before capturing")
        torch.cuda.synchronize(dev)
        check_pending()
        with torch.no_grad():
            for k, v in model.state_dict().items():
                v.copy_(keep[k])
        del keep
        self._drop_grads()  # (the captured backward then ASSIGNS the gradients: tensors of the graph's pool)
        self._pin = torch.zeros((1,), dtype=torch.int32, pin_memory=True)
        self._ev = torch.cuda.Event()
        self.graph = torch.cuda.CUDAGraph()
        _capture_errs = []
        try:
            with torch.cuda.graph(self.graph):
                self.outputs = model(self.static_wave)
                self.loss = loss_fn(self.outputs)
                self.loss.backward()
                if optimizer is not None:
                    optimizer.step()
                err = (torch.stack(_capture_errs).max() if _capture_errs else torch.zeros((), dtype=torch.int32, device=dev)).reshape(1)
                self._pin.copy_(err, non_blocking=True)
        finally:
            self.layer_calls_captured, _capture_errs = len(_capture_errs), None
        self.grads = [p.grad for p in self.params]

    def _drop_grads(self):
        for p in self.params:
            p.grad = None

    def step(self, wave: torch.Tensor) -> torch.Tensor:
        if wave.shape != self.static_wave.shape:
            raise ValueError(f"GraphedTrainStep was captured for waves of shape {tuple(self.static_wave.shape)}, got {tuple(wave.shape)}")
        for p, g in zip(self.params, self.grads):
            if p.grad is not g:
                p.grad = g
        self.static_wave.copy_(wave, non_blocking=True)
        self.graph.replay()
        self._ev.record()
        self._ev.synchronize()
        if int(self._pin[0]) != 0:
            raise RuntimeError(_EXCHANGE_FAILED + (" -- the optimizer step inside the graph has already applied the NaN gradients: restore "
                                                   "the weights from the last checkpoint" if self.optimizer is not None else ""))
        return self.loss

    __call__ = step
