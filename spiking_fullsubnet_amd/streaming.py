"""Frame-by-frame (streaming) execution of the live model: BASELINE.json configs[4] / SURVEY.md section 8(d) "config 5".

The network is causal after the STFT: the features of frame t (compressed magnitude, sub-band unfold along frequency,
per-frame LayerNorm; modeling_spiking_fullsubnet.py:434-440, 239-258, 108-112) need frame t only, the recurrent cells carry
(h, c) (efficient_spiking_neuron.py:50-62 accepts and returns the states), and the deep filter reaches back ``df - 1``
frames of the *noisy* spectrum with zero padding before the first frame (modeling_spiking_fullsubnet.py:315-346).  A
session therefore keeps, on the device: the (h, c) state of every layer and ``max(df) - 1`` frames of input history, and
each ``step`` runs exactly the kernels of the offline forward on ``hop`` new frames.  Outputs are bit-identical to the
offline forward on the concatenated input (tested).

Two ways to run a hop.  ``one_launch`` (default where the library covers the model: shared or separate gate weights, at most
3 layers / 4 groups): ``sfsn_stream_hop`` -- the whole frame in ONE launch of a few dozen small workgroups whose waves hand
the frame from stage to stage through L2 (csrc/sfsn_hop.hip); its state is its own (double-buffered int8 spikes, membranes,
history).  Otherwise the ~15 launches of the offline kernels, captured once into a HIP graph and replayed per hop: at
hop = 1 that step is launch-bound, not compute-bound.  Both are bit-identical to the offline forward (tested).

The frozen front-end (``model_low_freq.Separator``) with ``offline_laplace_norm`` normalises with utterance-level means
(model_low_freq.py:147-169), which are not causal: a session on it raises ``NotImplementedError``.  With
``cumulative_laplace_norm`` (every row by its own running mean) it streams, through the one-launch hop only.
"""
from __future__ import annotations

import ctypes
import math
import os
import time
from typing import Optional, Tuple

import torch

from ._lib import DfGroup, HopDesc, HOP_MAX_GROUPS, HOP_MAX_LAYERS, NORM_CUMLAPLACE, check
from .engine import Engine, _ptr


def _raw_stream(device_index: int) -> int:
    """The HIP stream handle of torch's current stream on a device (without building a torch.cuda.Stream object per call)."""
    try:
        return torch._C._cuda_getCurrentRawStream(device_index)
    except AttributeError:  # pragma: no cover - older / newer torch without the private accessor
        return torch.cuda.current_stream(device_index).cuda_stream


class StreamingSession:
    """``step(frames [B, F, hop] complex64) -> (enh_stft [B, S, F, hop], enh_mag [B, S, F, hop])`` with state carried."""

    def __init__(self, engine: Engine, batch: int = 1, hop: int = 1, graph: bool = True, rows_per_wg=None, owner=None,
                 one_launch="auto", waveform: bool = False, host_io: bool = False, resident: bool = False, idle_ms: int = 1000):
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
        self._dev_index = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        self._hop = None
        self.waveform = bool(waveform)  # step_wave(): samples in, samples out (STFT and inverse STFT inside the launch)
        self._wave_calls = 0
        self.host_io = bool(host_io)  # waveform mode: samples come from and go to (pinned) host memory, no copy launches
        if self.host_io and not waveform:
            raise ValueError("host_io goes with waveform=True")
        # host_io only: ONE resident launch serves hop after hop (sfsn_stream_hop_resident), rung through a doorbell word in pinned
        # memory; it starts with the first hop that computes a frame and ends with reset(), close(), or idle_ms of silence
        self.resident = bool(resident)
        self.idle_ms = int(idle_ms)
        self._res = None
        if self.resident and not self.host_io:
            raise ValueError("resident goes with waveform=True, host_io=True")
        if spec.cum_laplace:
            one_launch = True  # the running means live in sfsn_stream_hop's state; the per-kernel sequence has no streaming form of it
        if self.waveform:
            if hop != 1 or spec.n_fft != 512:
                raise NotImplementedError("waveform streaming: one 128-sample hop per call, 512-point frames")
            one_launch = True
        if one_launch not in ("auto", True, False):
            raise ValueError("one_launch must be 'auto', True or False")
        if one_launch == "auto" and os.environ.get("SFSN_ONE_LAUNCH", "1") == "0":  # diagnostic: the per-kernel sequence everywhere
            one_launch = False
        if one_launch:
            self._hop = self._build_hop()
            if self._hop is None and one_launch is True:
                raise NotImplementedError("sfsn_stream_hop does not cover this model / batch (see include/sfsn.h)")
        if self._hop is None and graph:
            self._capture()

    # -----------------------------------------------------------------------------------------------------------------
    def _build_hop(self):
        """Descriptors + state of the one-launch hop (sfsn_stream_hop); None when the library does not cover this session.

        A launch wants every workgroup resident at once (one per compute unit), which bounds the clips per launch (32 at
        baseline_m on an MI355X); a bigger batch is cut into equal parts, one launch each, back to back on the stream -- every part
        has its own state, all parts share the weights and write into the same output tensors."""
        eng, spec, L = self.eng, self.eng.spec, self.eng.lib
        B, F, S, hop, D, ng, dev = self.B, self.F, spec.num_spks, self.hop, self.D, spec.n_groups, self.dev
        if ng > HOP_MAX_GROUPS or max(spec.fb_layers, spec.sb_layers) > HOP_MAX_LAYERS or D + hop > 32:
            return None

        # Arenas instead of ~80 separately allocated tensors: the weights a launch reads (2.9 MB at baseline_m) and the state it
        # reads and writes sit in contiguous ranges -- a launch that starts cold on every compute unit then misses in the TLBs
        # for a handful of pages, not for one page per tensor.
        class Pool:
            def __init__(self):
                self.size, self.buf = 0, None

            def take(self, nbytes):
                off = self.size
                self.size += (nbytes + 255) // 256 * 256
                return None if self.buf is None else self.buf[off:off + nbytes]

            def put(self, t):  # a copy of tensor t inside the pool -> its address (None in the sizing pass)
                v = self.take(t.numel() * t.element_size())
                if v is None:
                    return None
                v.copy_(t.contiguous().view(-1).view(torch.uint8))
                return v.data_ptr()

            def zeros(self, shape, dtype):
                v = self.take(math.prod(shape) * torch.empty((), dtype=dtype).element_size())
                return None if v is None else v.view(dtype).view(shape)

            def allocate(self):
                self.buf = torch.zeros((max(self.size, 256),), dtype=torch.uint8, device=dev)
                self.size = 0

        def ptr(t):
            return None if t is None else t.data_ptr()

        # ---- weights, once (two passes: size, then copy)
        wpool = Pool()
        seqs = [eng.fb] + list(eng.sb)

        def put_weights():
            w = []
            for seq in seqs:
                d = dict(p=(wpool.put(seq.proj_q), wpool.put(seq.proj_dq), wpool.put(seq.proj_b)), layers=[],
                         ln=(wpool.put(seq.ln_w), wpool.put(seq.ln_b)) if seq.ln_w is not None else None)
                for l, cell in enumerate(seq.cells):
                    e = dict(hh=(wpool.put(cell.w_hh_q), wpool.put(cell.w_hh_dq)))
                    if l == 0:  # fp32 input weights in MFMA fragment order (include/sfsn.h): one coalesced request per 16 columns
                        w0 = cell.w_ih_f32
                        kc = (w0.shape[1] + 15) // 16
                        w0 = torch.nn.functional.pad(w0, (0, kc * 16 - w0.shape[1]))
                        # (separate gate weights: 2H rows, the forget gate's tiles first -- tile NT + j is the cell gate's tile j)
                        e["ih"] = (wpool.put(w0.view(w0.shape[0] // 16, 16, kc, 4, 4).permute(0, 2, 3, 1, 4).contiguous()),)
                    elif len(cell.w_ih_q) == 1:
                        e["ih"] = (wpool.put(cell.w_ih_q[0][0]), wpool.put(cell.w_ih_q[0][1]))
                    else:
                        # separate gate weights: the engine packs each gate's [H, H] on its own ([3][NT][KS] KiB images); the hop reads ONE
                        # image of 2H rows, [3][2 NT][KS] with the forget gate's tiles first, and one dq vector [2H]
                        nt_ = seq.H // 16
                        pk = torch.cat([q_[0].reshape(3, nt_, -1) for q_ in cell.w_ih_q], dim=1).contiguous()
                        dq = torch.cat([q_[1][:seq.H] for q_ in cell.w_ih_q]).contiguous()
                        e["ih"] = (wpool.put(pk.reshape(-1)), wpool.put(dq))
                    e["c"] = (wpool.put(cell.bias), wpool.put(cell.alpha), wpool.put(cell.beta))
                    d["layers"].append(e)
                w.append(d)
            win = wpool.put(torch.hann_window(512, device=dev, dtype=torch.float32)) if self.waveform else None
            return w, win

        put_weights()
        wpool.allocate()
        weights, window = put_weights()

        # ---- one part = the clips [b0, b0 + nb) of the batch: descriptor + state
        def make_desc(nb, spool, a=None):
            """a: placeholder address for the sizing call (the plan never dereferences device pointers)."""
            desc = HopDesc()
            fgs = [self.fg_fb[0]] + [self.fg_sb[g] for g in range(ng)]
            dsts = [desc.fb] + [desc.sb[g] for g in range(ng)]
            for i, (dst, seq, fg, w) in enumerate(zip(dsts, seqs, fgs, weights)):
                R = nb * (1 if i == 0 else spec.units(i - 1))
                HP = (seq.H + 63) // 64 * 64
                dst.n_layers, dst.H, dst.P = len(seq.cells), seq.H, seq.P
                dst.df, dst.fc = (0, 0) if i == 0 else (spec.df[i - 1], spec.ctr[i - 1])
                dst.feat = fg
                if w["ln"] is not None:
                    dst.feat.ln_w, dst.feat.ln_b = w["ln"]
                if spec.cum_laplace:  # the rows' running sums travel with the session
                    dst.feat.norm = NORM_CUMLAPLACE
                    c0, c1 = spool.zeros((R,), torch.float32), spool.zeros((R,), torch.float32)
                    dst.cum[0], dst.cum[1] = (a, a) if a else (ptr(c0), ptr(c1))
                dst.w_p, dst.w_p_dq, dst.b_p = w["p"]
                for l, e in enumerate(w["layers"]):
                    o = dst.layer[l]
                    o.w_hh, o.w_hh_dq = e["hh"]
                    if l == 0:
                        o.w_ih_frag = e["ih"][0]
                    else:
                        o.w_ih, o.w_ih_dq = e["ih"]
                    o.bias, o.bn_alpha, o.bn_beta = e["c"]
                    h0, h1 = spool.zeros((R, HP), torch.int8), spool.zeros((R, HP), torch.int8)
                    c, spk = spool.zeros((R, seq.H), torch.float32), spool.zeros((hop, R, HP), torch.int8)
                    o.h[0], o.h[1], o.c, o.spikes = (a, a, a, a) if a else (ptr(h0), ptr(h1), ptr(c), ptr(spk))
            desc.n_groups, desc.B, desc.F, desc.S, desc.hop, desc.D, desc.fdrc = ng, nb, F, S, hop, D, spec.fdrc
            desc.unshared = 0 if spec.shared else 1
            st = dict(hist=spool.zeros((nb, F, max(D, 1), 2), torch.float32))
            if self.waveform:
                st.update(state=spool.zeros((nb, 512), torch.float32), ola=spool.zeros((nb, S, 512), torch.float32),
                          spec_g=spool.zeros((nb, F, 4), torch.float32), enh_g=spool.zeros((nb, S, F, 4), torch.float32))
            if a:
                desc.inp_ri = desc.hist_ri = desc.enh_ri = desc.enh_mag = a
                if self.waveform:
                    desc.wave_in = desc.wave_state = desc.ola_state = desc.wave_out = desc.window = desc.spec_g = desc.enh_g = a
            else:
                desc.hist_ri = ptr(st["hist"])
                if self.waveform:
                    desc.wave_state, desc.ola_state, desc.window = ptr(st["state"]), ptr(st["ola"]), window
                    desc.spec_g, desc.enh_g = ptr(st["spec_g"]), ptr(st["enh_g"])
            return desc, st

        probe = torch.zeros((16,), dtype=torch.uint8, device=dev)
        n_parts, nbytes = 0, 0
        for n in range(1, 9):  # equal parts: the fewest launches whose workgroups fit the chip
            if n > B:
                break
            nbytes = L.sfsn_hop_scratch_bytes(ctypes.byref(make_desc(-(-B // n), Pool(), probe.data_ptr())[0]))
            if nbytes:
                n_parts = n
                break
        if not n_parts:
            return None
        per = -(-B // n_parts)
        enh = torch.zeros((B, S, F, hop, 2), dtype=torch.float32, device=dev)
        mag = torch.zeros((B, S, F, hop), dtype=torch.float32, device=dev)
        wave_out = torch.zeros((B, S, 128), dtype=torch.float32, device=dev) if self.waveform else None
        host = None
        if self.host_io:  # pinned host memory the kernel reads / writes directly (hipHostMalloc: device-reachable at the same address)
            host = dict(inp=torch.zeros((B, 128), dtype=torch.float32).pin_memory(), out=torch.zeros((B, S, 128), dtype=torch.float32).pin_memory(),
                        done=torch.zeros((B * S,), dtype=torch.int32).pin_memory())
            host["done_np"] = host["done"].numpy()
            host["bell"] = torch.zeros((16,), dtype=torch.int32).pin_memory()  # the resident kernel's doorbell (word 0)
            host["bell_np"] = host["bell"].numpy().view("uint32")
        parts = []
        for b0 in range(0, B, per):
            nb = min(per, B - b0)
            spool = Pool()
            make_desc(nb, spool, probe.data_ptr())
            spool.allocate()
            desc, st = make_desc(nb, spool)
            desc.inp_ri = _ptr(self.inp)
            desc.enh_ri, desc.enh_mag = ptr(enh[b0:]), ptr(mag[b0:])
            if self.waveform:
                desc.wave_in, desc.wave_out = probe.data_ptr(), ptr(wave_out[b0:])  # (wave_in: set per call)
                if host is not None:
                    desc.wave_out, desc.done = ptr(host["out"][b0:]), ptr(host["done"][b0 * S:])
            nbytes = L.sfsn_hop_scratch_bytes(ctypes.byref(desc))
            assert nbytes, "the sizing call accepted this geometry"
            scratch = torch.zeros((nbytes // 4 + 1,), dtype=torch.int32, device=dev)  # word 0: the error flag
            desc.scratch, desc.scratch_bytes = ptr(scratch), nbytes
            # the error word of a launch is looked at, without blocking, at a later step (pinned copy behind the launch)
            parts.append(dict(desc=desc, ref=ctypes.byref(desc), b0=b0, nb=nb, st=st, spool=spool.buf, scratch=scratch,
                              err=torch.zeros((1,), dtype=torch.int32).pin_memory(), err_pending=False))
        return dict(parts=parts, desc=parts[0]["desc"], scratch=parts[0]["scratch"], wpool=wpool.buf, enh=torch.view_as_complex(enh),
                    mag=mag, wave_out=wave_out, host=host)

    def _launch_hops(self, base_ptr: int, stride_bytes: int, field: str, frame_index=None) -> None:
        """One sfsn_stream_hop launch per part, back to back on torch's current stream; `field` of each descriptor is pointed at
        the part's slice of the caller's input (clips are the slowest axis: a part is a contiguous range)."""
        idx = self._dev_index
        st = ctypes.c_void_p(_raw_stream(idx))
        same = torch.cuda.current_device() == idx  # the C ABI launches on the calling thread's current device
        for part in self._hop["parts"]:
            if part["err_pending"] and int(part["err"][0]) != 0:  # written behind an earlier launch; no blocking here
                self.check_errors()  # (synchronises, clears the sticky words and raises)
            d = part["desc"]
            d.frames_before = self.frames_done
            setattr(d, field, base_ptr + part["b0"] * stride_bytes)
            if frame_index is not None:
                d.frame_index = frame_index
            if same:
                rc = self.eng.lib.sfsn_stream_hop(part["ref"], st)
            else:
                with torch.cuda.device(self.dev):
                    rc = self.eng.lib.sfsn_stream_hop(part["ref"], st)
            if rc:
                check(rc, "sfsn_stream_hop")
            d.launch_index += 1
        self.frames_done += self.hop
        if self.frames_done % 256 < self.hop:  # every ~256 frames the error words follow the launches into pinned memory
            for part in self._hop["parts"]:
                part["err"].copy_(part["scratch"][:1], non_blocking=True)
                part["err_pending"] = True

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
        self._stop_resident()
        if self._hop is not None:
            self.check_errors()
            for part in self._hop["parts"]:
                part["spool"].zero_()  # (h, c), tagged spike buffers, history, waveform state: one fill per part
            self._wave_calls = 0
        self.frames_done = 0

    def check_errors(self) -> None:
        """Raise if a hand-off wait of an earlier one-launch hop expired (blocks until the hops enqueued so far have finished)."""
        if self._hop is None:
            return
        torch.cuda.current_stream(self.dev).synchronize()
        bad = False
        for part in self._hop["parts"]:
            if int(part["scratch"][0].item()) != 0:
                # the word is sticky on the device: clear it (and the pinned copy) when reporting, or every later check --
                # reset() included -- would report this failure again (round-2 advisor finding)
                part["scratch"][:1].zero_()
                part["err"].zero_()
                part["err_pending"] = False
                bad = True
        if bad:
            torch.cuda.current_stream(self.dev).synchronize()
            raise RuntimeError("sfsn_stream_hop: a bounded hand-off wait expired inside a launch (results invalid)")

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
        if self.waveform:
            raise RuntimeError("this session was opened with waveform=True: use step_wave(samples)")
        if self._hop is not None:
            h = self._hop
            if not (frames.is_contiguous() and frames.data_ptr() % 8 == 0):
                self.inp.copy_(frames)
                frames = self.inp
            self._launch_hops(frames.data_ptr(), self.F * self.hop * 8, "inp_ri")  # read in place: no staging copy in front
            return (h["enh"].clone(), h["mag"].clone()) if copy else (h["enh"], h["mag"])
        self.inp.copy_(frames)
        if self._graph is not None:
            self._graph.replay()
        else:
            self._enqueue()
        self.frames_done += self.hop
        e, m = self.enh[..., self.D:], self.enh_mag[..., self.D:]
        return (e.clone(), m.clone()) if copy else (e, m)

    def step_wave(self, samples: torch.Tensor, copy: bool = True) -> torch.Tensor:
        """Waveform streaming (``waveform=True``): ``samples`` float32 [B, 128] on the device -- the next 8 ms of every clip.
        Returns enhanced samples [B, S, 128]: call c returns the samples that entered with call c - 3 (the framing of
        torch.stft(center=True) needs 256 samples of look-ahead, the overlap-add another hop: 24 ms of algorithmic delay, as for
        any causal use of the reference's 32 ms / 8 ms analysis); the first three calls of an utterance return zeros.
        One launch per call: the frame's STFT, the whole model, the inverse STFT with its overlap-add state."""
        if not self.waveform:
            raise RuntimeError("open the session with waveform=True")
        if samples.device != self.dev or samples.dtype != torch.float32 or tuple(samples.shape) != (self.B, 128) or not samples.is_contiguous():
            raise RuntimeError(f"expected contiguous float32 {(self.B, 128)} on {self.dev}, got {samples.dtype} {tuple(samples.shape)}")
        h = self._hop
        c = self._wave_calls
        self._wave_calls += 1
        out = h["wave_out"]
        if c == 0:  # no frame ends here yet (frame 0 covers samples [-256, 256)): the samples only enter the state
            for part in h["parts"]:
                part["st"]["state"][:, 384:].copy_(samples[part["b0"]:part["b0"] + part["nb"]])
            return torch.zeros_like(out)
        self._launch_hops(samples.data_ptr(), 128 * 4, "wave_in", frame_index=c - 1)
        if c < 3:  # padded positions torch.istft trims
            return torch.zeros_like(out)
        return out.clone() if copy else out


    # ---- the resident launch (host_io, resident=True) ---------------------------------------------------------------------------
    def _start_resident(self, c: int) -> None:
        h = self._hop
        if len(h["parts"]) != 1:
            raise NotImplementedError("resident streaming: batches one launch covers (one part)")
        part, host = h["parts"][0], h["host"]
        d = part["desc"]
        d.frames_before, d.frame_index, d.wave_in = self.frames_done, c - 1, host["inp"].data_ptr()
        host["bell_np"][:2] = 0  # word 0: the doorbell; word 1: set by the kernel when it leaves
        with torch.cuda.device(self.dev):
            side = torch.cuda.Stream(self.dev)  # non-blocking: the kernel stays resident behind everything else the caller does
            side.wait_stream(torch.cuda.current_stream(self.dev))  # (state fills / the first call's copy come first)
            rc = self.eng.lib.sfsn_stream_hop_resident(part["ref"], ctypes.c_void_p(host["bell"].data_ptr()), self.idle_ms,
                                                       ctypes.c_void_p(side.cuda_stream))
        if rc:
            check(rc, "sfsn_stream_hop_resident")
        self._res = dict(k=0, stream=side)

    def _stop_resident(self) -> None:
        """End the resident launch (no-op without one) and fold the hops it served into the descriptor's launch counter."""
        r = self._res
        if r is None:
            return
        self._res = None
        self._hop["host"]["bell_np"][0] = 0xFFFFFFFF
        r["stream"].synchronize()
        self._hop["parts"][0]["desc"].launch_index += r["k"]

    def _ring_resident(self, c: int) -> int:
        r = self._res
        if r is not None and self._hop["host"]["bell_np"][1] != 0:
            self._stop_resident()  # the doorbell stayed silent long enough for the kernel to leave: start another
            r = None
        if r is None:
            self._start_resident(c)
            r = self._res
        r["k"] += 1
        self._hop["host"]["bell_np"][0] = r["k"]
        self.frames_done += self.hop
        return self._hop["parts"][0]["desc"].launch_index + r["k"]

    def close(self) -> None:
        """End a resident launch, if one is running (reset() does the same)."""
        self._stop_resident()

    def __del__(self):
        try:
            self._stop_resident()
        except Exception:
            pass

    def step_wave_host(self, samples, timeout_s: float = 2.0) -> torch.Tensor:
        """Waveform streaming with the samples on the HOST (``waveform=True, host_io=True``): ``samples`` = float32 [B, 128] CPU
        tensor (or anything ``torch.as_tensor`` takes).  The launch reads them from pinned host memory and writes the enhanced
        samples and a completion word per (clip, speaker) back into pinned host memory; the caller's thread spins on the words --
        no copy launch, no stream synchronisation.  Returns a CPU tensor [B, S, 128] (a view of the session's pinned buffer, valid
        until the next call); same three-hop delay as ``step_wave``."""
        h = self._hop
        if not self.host_io or h is None:
            raise RuntimeError("open the session with waveform=True, host_io=True")
        host = h["host"]
        host["inp"].copy_(torch.as_tensor(samples, dtype=torch.float32).reshape(self.B, 128))
        c = self._wave_calls
        self._wave_calls += 1
        if c == 0:
            for part in h["parts"]:
                # (blocking: the next call overwrites the pinned buffer right away)
                part["st"]["state"][:, 384:].copy_(host["inp"][part["b0"]:part["b0"] + part["nb"]])
            torch.cuda.current_stream(self.dev).synchronize()
            return torch.zeros_like(host["out"])
        if self.resident:
            target = self._ring_resident(c)
        else:
            target = h["parts"][0]["desc"].launch_index + 1
            self._launch_hops(host["inp"].data_ptr(), 128 * 4, "wave_in", frame_index=c - 1)
        done, t_end, t_soft = host["done_np"], None, None
        target &= 0xFFFFFFFF
        while (int(done.min()) & 0xFFFFFFFF) != target or (int(done.max()) & 0xFFFFFFFF) != target:  # (all parts carry the same launch index; uint32 compare: the index wraps)
            if t_end is None:
                t_end = time.perf_counter() + timeout_s
                t_soft = t_end - timeout_s + 0.02
            elif t_soft is not None and not self.resident and time.perf_counter() > t_soft:
                # 20 ms without the words (a hop takes tens of microseconds): a hand-off wait inside the launch has probably
                # expired -- the launch has ended by now, its error word says so, and check_errors() raises it (round-2 advisor:
                # a failed launch used to show up only as the 2 s timeout)
                t_soft = None
                self.check_errors()
            elif time.perf_counter() > t_end:
                if self.resident:
                    self._stop_resident()
                raise RuntimeError("sfsn_stream_hop: no completion word from the launch (see check_errors())")
            if self._res is not None and host["bell_np"][1] != 0 and (int(done.min()) & 0xFFFFFFFF) != target:
                # the resident kernel's watchdog fired just as this hop was rung: the hop was not served -- start another kernel on it
                self.frames_done -= self.hop
                self._res["k"] -= 1
                self._stop_resident()
                self.check_errors()
                target = self._ring_resident(c) & 0xFFFFFFFF
        return torch.zeros_like(host["out"]) if c < 3 else host["out"]
