"""Frame-by-frame (streaming) execution of the live model: BASELINE.json configs[4] / SURVEY.md section 8(d) "config 5".

The network is causal after the STFT: the features of frame t (compressed magnitude, sub-band unfold along frequency,
per-frame LayerNorm; modeling_spiking_fullsubnet.py:434-440, 239-258, 108-112) need frame t only, the recurrent cells carry
(h, c) (efficient_spiking_neuron.py:50-62 accepts and returns the states), and the deep filter reaches back ``df - 1``
frames of the *noisy* spectrum with zero padding before the first frame (modeling_spiking_fullsubnet.py:315-346).  A
session therefore keeps, on the device: the (h, c) state of every layer and ``max(df) - 1`` frames of input history, and
each ``step`` runs exactly the kernels of the offline forward on ``hop`` new frames.  Outputs are bit-identical to the
offline forward on the concatenated input (tested).

The ~25 launches of a step are captured once into a HIP graph and replayed per hop: at hop = 1 the step is launch-bound,
not compute-bound, so the graph is what sets the per-frame latency.

The frozen front-end (``model_low_freq.Separator``) normalises with utterance-level Laplace means
(model_low_freq.py:147-169), which are not causal: a session on it raises ``NotImplementedError``.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from ._lib import DfGroup, check
from .engine import Engine, _ptr


class StreamingSession:
    """``step(frames [B, F, hop] complex64) -> (enh_stft [B, S, F, hop], enh_mag [B, S, F, hop])`` with state carried."""

    def __init__(self, engine: Engine, batch: int = 1, hop: int = 1, graph: bool = True, rows_per_wg=None, owner=None):
        spec = engine.spec
        # the module the engine was packed from: reset() checks that its parameters have not changed since (the session's
        # captured graph holds pointers to THIS engine's packed weights)
        import weakref
        self._owner = weakref.ref(owner) if owner is not None else None
        if spec.laplace:
            raise NotImplementedError("the frozen front-end's offline Laplace normalisation (model_low_freq.py:147-169) needs the whole "
                                      "utterance; streaming is defined for the live (LayerNorm) front-end only")
        if batch < 1 or hop < 1:
            raise ValueError("batch and hop must be positive")
        self.eng, self.B, self.hop = engine, batch, hop
        self.rows_per_wg = rows_per_wg  # (full-band, sub-band) rows per scan workgroup; None = the engine's setting
        self.use_stack = True           # layer-pipelined stack launches where the engine's rule picks them (few rows: always)
        dev = self.dev = engine.device
        self.F = spec.n_fft // 2 + 1
        self.D = D = max(spec.df) - 1  # frames of input history the deep filter reaches back
        self.Th = Th = D + hop
        f32 = dict(dtype=torch.float32, device=dev)
        B, F, S, ng = batch, self.F, spec.num_spks, spec.n_groups
        self.inp = torch.zeros((B, F, hop), dtype=torch.complex64, device=dev)
        self.hist = torch.zeros((B, F, Th), dtype=torch.complex64, device=dev)
        self._tmp = torch.zeros((B, F, max(D, 1)), dtype=torch.complex64, device=dev)
        self.x_fb = torch.empty((Th, B, spec.fb_in), **f32)
        self.xs = [torch.empty((Th, B * spec.units(g), spec.sb_input_size(g)), **f32) for g in range(ng)]
        self.enh = torch.zeros((B, S, F, Th), dtype=torch.complex64, device=dev)
        self.enh_mag = torch.zeros((B, S, F, Th), **f32)

        def stack(seqs, Rs):
            H, G, nl = seqs[0].H, seqs[0].cells[0].G, len(seqs[0].cells)
            HP = (H + 63) // 64 * 64
            nb = engine.lib.sfsn_stack_scratch_bytes(nl, len(Rs), sum(Rs))
            return dict(spk=[[None] * len(Rs) for _ in range(nl)], scratch=torch.zeros((nb // 4 + 1,), dtype=torch.int32, device=dev),
                        zin=[[torch.empty((hop, R, G * H), **f32) for R in Rs] for _ in range(nl)],
                        s8=[[torch.zeros((Th, R, HP), dtype=torch.int8, device=dev) for R in Rs] for _ in range(nl)],
                        states=engine._zero_states(Rs, H, nl),
                        proj=[torch.empty((Th, R, seq.P), **f32) for seq, R in zip(seqs, Rs)], nl=nl)
        self.fb = stack([engine.fb], [B])
        self.sb = stack(engine.sb, [x.shape[1] for x in self.xs])
        self.dfg = (DfGroup * ng)()
        for g in range(ng):
            self.dfg[g].proj, self.dfg[g].n_units = _ptr(self.sb["proj"][g]), spec.units(g)
            self.dfg[g].fc, self.dfg[g].df = spec.ctr[g], spec.df[g]
        self.fg_fb = engine._feature_groups("fb", [self.x_fb], None)
        self.fg_sb = engine._feature_groups("sb", self.xs, None)
        self.frames_done = 0
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        if graph:
            self._capture()

    # -----------------------------------------------------------------------------------------------------------------
    def reset(self) -> None:
        """Back to the start of an utterance: zero (h, c) (modeling_spiking_fullsubnet.py:100-106) and zero history."""
        owner = self._owner() if self._owner is not None else None
        if owner is not None and owner.engine() is not self.eng:
            raise RuntimeError("the module's parameters (or device) changed after this streaming session was created: its packed "
                               "weights are stale -- create a new session with module.streaming(...)")
        for d in (self.fb, self.sb):
            for layer in d["states"]:
                for h, c in layer:
                    h.zero_()
                    c.zero_()
        self.hist.zero_()
        self.frames_done = 0

    def _enqueue(self) -> None:
        """One hop on torch's current stream: history shift, then the offline forward's kernels on frames [D, D+hop)."""
        with torch.cuda.device(self.dev):  # the C ABI launches on the calling thread's current device
            self._enqueue_on_device()

    def _enqueue_on_device(self) -> None:
        eng, spec, L = self.eng, self.eng.spec, self.eng.lib
        B, F, D, hop, Th, S, ng = self.B, self.F, self.D, self.hop, self.Th, spec.num_spks, spec.n_groups
        st = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        ri = torch.view_as_real(self.hist)
        if D + hop <= 16:  # history shift + append in one launch
            check(L.sfsn_hist_shift(_ptr(ri), _ptr(torch.view_as_real(self.inp)), B * F, D, hop, st), "sfsn_hist_shift")
        else:
            if D > 0:
                self._tmp[:, :, :D].copy_(self.hist[:, :, hop:hop + D])
                self.hist[:, :, :D].copy_(self._tmp[:, :, :D])
            self.hist[:, :, D:].copy_(self.inp)
        # default: the engine's full-band setting, 16 rows per workgroup for the sub-band stack -- few rows per hop anyway, and
        # that geometry lets layers >= 1 take their input product inside the scan (three launches less per hop)
        rpw_fb, rpw_sb = (eng.rows_per_wg[0], 16) if self.rows_per_wg is None else self.rows_per_wg

        def model(seqs, d, xs, tag, rpw):
            use_stack, wide, rpw_stack = eng._stack_choice(seqs, [x.shape[1] for x in xs], False)
            if use_stack and self.use_stack:
                # every layer of the stack in one launch (their prologues -- weights into registers / LDS -- run side by side
                # instead of one launch after the other; the layers hand each frame over inside the launch)
                eng._stage_input(seqs, 0, xs, d["zin"][0], D, hop, st, tag)
                eng._stage_stack(seqs, d, D, hop, st, tag, wide, rpw_stack, lag=0, scratch=d["scratch"])
                eng._stage_proj(seqs, d["s8"][-1], d["proj"], D, hop, st, tag)
                return
            fused = eng._fusable(seqs, rpw, False)  # layers >= 1: input product inside the scan (three launches less per hop)
            for l in range(d["nl"]):
                if l > 0 and fused:
                    eng._stage_scan_fused(seqs, l, d["states"][l], [None] * len(seqs), d["s8"], D, hop, st, tag)
                    continue
                # chunk-local zin holds frames [D, D+hop) at rows [0, hop): sources are offset by D inside _stage_input
                eng._stage_input(seqs, l, xs if l == 0 else d["s8"][l - 1], d["zin"][l], D, hop, st, tag)
                eng._stage_scan(seqs, l, d["zin"][l], d["states"][l], [None] * len(seqs), d["s8"][l], [None] * len(seqs), D, hop, st, tag, rpw)
            eng._stage_proj(seqs, d["s8"][-1], d["proj"], D, hop, st, tag)

        check(L.sfsn_features(_ptr(ri), None, B, F, Th, 0, spec.fdrc, self.fg_fb, 1, D, hop, st), "sfsn_features(fb)")
        model([eng.fb], self.fb, [self.x_fb], "fb", rpw_fb)
        check(L.sfsn_features(_ptr(ri), _ptr(self.fb["proj"][0]), B, F, Th, spec.fb_proj, spec.fdrc, self.fg_sb, ng, D, hop, st),
              "sfsn_features(sb)")
        model(eng.sb, self.sb, self.xs, "sb", rpw_sb)
        check(L.sfsn_deepfilter(_ptr(ri), B, F, Th, S, self.dfg, ng, _ptr(torch.view_as_real(self.enh)), _ptr(self.enh_mag), D, hop, st),
              "sfsn_deepfilter")

    def _capture(self) -> None:
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(2):  # module loading, function attributes and allocator warm-up happen outside the capture
                self._enqueue()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._enqueue()
        self._graph = g
        self.reset()

    def step(self, frames: torch.Tensor, copy: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
        """``frames``: complex64 [B, F, hop] on the device.  Returns the enhanced frames and their magnitudes; with
        ``copy=False`` the returned tensors are views of the session's buffers, valid until the next ``step``."""
        if frames.device != self.dev or frames.dtype != torch.complex64 or tuple(frames.shape) != (self.B, self.F, self.hop):
            raise RuntimeError(f"expected complex64 {(self.B, self.F, self.hop)} on {self.dev}, got {frames.dtype} {tuple(frames.shape)} "
                               f"on {frames.device}")
        self.inp.copy_(frames)
        if self._graph is not None:
            self._graph.replay()
        else:
            self._enqueue()
        self.frames_done += self.hop
        e, m = self.enh[..., self.D:], self.enh_mag[..., self.D:]
        return (e.clone(), m.clone()) if copy else (e, m)
