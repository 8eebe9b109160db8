#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X -- the metric BASELINE.json names.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM:
complex noisy STFT [B=64, 257, T=1000] -> enhanced STFT + enhanced magnitude, INCLUDING every per-layer
tensor the reference module returns (API-faithful: layer inputs, fp32 spike tensors, projections).
Workload = BASELINE.json configs[2] (single MI355X, full model: full-band + sub-band groups, B=64, T=1000)
with the live `baseline_m` sizes, in the fp32 parity mode (the mode the parity tests gate).
For N > 1 every rank owns 64 clips (weak scaling: the path shards over independent clips, no data-path
collective) and each step ends with the RCCL all-gather of the enhanced magnitudes -- the analogue of the
reference's `accelerator.gather_for_metrics` (audiozen/trainer.py:511,555).

`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run (one rank per GPU).

Phases: S -- one forward at a time (THE STRICT B=64 FIGURE, `config.single_stream`), then the same forward as whole-sequence
launches with per-kernel HIP-event timers (the roofline figures of the strict schedule); K -- the scan kernels of the timed
region's geometry, each alone on the chip; B -- the timed region: `--inflight` independent batches in flight, each lane on its
OWN input batch (seeded per lane), EXACTLY K steps between barrier + synchronize on both sides, max over ranks -> `value`;
then 2,000 one-frame hops of a B=1 streaming session (BASELINE configs[4]) -> `config.streaming`.

Prints ONE JSON line on rank 0 (see the task contract), carrying
  roofline      -- top level: the JOB -- algorithmic bytes of one forward (whole path) / ms_per_step of the timed region,
                   against the 8 TB/s HBM3E peak: a figure whose time fits in ms_per_step by construction.  Sub-fields, each with the
                   time it was measured over: `single_stream_job` (the same bytes / the strict ms per forward),
                   `sub_band_scan_single_forward` (the sub-band scan kernel of the strict schedule: algorithmic bytes per launch /
                   HIP-event launch duration -- the north star's "HBM roofline on the sub-band scan"), `dominant_kernel_timed_region`
                   (the fused sub-band scan of the timed region's geometry, alone on the chip and as a time share inside the
                   region), `full_band_stack` (MFMA utilisation).  PMC-derived fields (`traffic`, `mfma.pmc`) come from
                   profiles/r06_pmc.json and are attached only when that file was taken with the library build that is running
                   (source hash) on this workload (B, T, geometry, forwards in flight);
  cpu_baseline  -- the CPU oracle (oracle/, a C restatement of the reference) timed on this host's cores on the whole workload
                   (all B clips x all T frames, groups of clips side by side; rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")  # HIP maps streams onto this many hardware queues (default 4): the forwards in
# flight each need their own, or independent batches serialise behind each other (12 lanes: 16 queues 37.3, 24 queues 38.2 M frames/s)

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

# The contract is ONE JSON line on stdout.  RCCL (and anything else below us) writes banners to file descriptor 1 through C stdio:
# keep the real stdout for the line and send every other writer of fd 1 to stderr.
JSON_OUT = os.fdopen(os.dup(1), "w")
os.dup2(2, 1)


def _emit(text: str) -> None:
    JSON_OUT.write(text + "\n")
    JSON_OUT.flush()


HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
INT8_PEAK_TOPS = 5000.0  # dense int8 MFMA peak = the dense fp8 figure of that guide (2x bf16): the SPEC number the fractions are priced against
INT8_MEASURED_TOPS = 3944.0  # ... and the ceiling a micro-benchmark reaches on this part (same guide): printed beside it
# API-faithful algorithmic bytes per clip-frame of the sub-band scan, baseline_m sizes:
#   read 256 (noisy_mag) + 64 (fb_out) floats, write 1,152 coefficients and both layers' fp32 spikes
#   (13 rows x 2 x 224) = 29,184 B.  The scan kernel is launched once per layer, so one launch is charged half.
SB_SCAN_BYTES_PER_FRAME_PER_LAUNCH = 29184 / 2
# SURVEY.md 8(d), whole path, API-faithful, baseline_m: complex in + complex out + magnitude out (5,140 B) + every
# all_layer_outputs tensor (8,646 floats) = 39,724 B per clip-frame; minimal (no layer outputs): 5,140 B
JOB_BYTES_PER_FRAME_API = 39724
JOB_BYTES_PER_FRAME_MIN = 5140
SB_SCAN_BYTES_PER_FRAME_MIN = 5888  # sub-band scan, minimal (M): read 256 + 64 floats, write 1,152 coefficients


def _self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU (the driver's
    own invocation does exactly this; with SFSN_BENCH_BACKEND=gloo the ranks may share GPUs -- a plumbing check)."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env, stdout=JSON_OUT.fileno()))  # (the ranks get the real stdout as their fd 1)


PROFILE_JSON = os.path.join(ROOT, "profiles", "r06_pmc.json")


def _pmc_profile():
    """PMC-derived figures (HBM bytes per launch, MFMA busy cycles) from the committed rocprofv3 passes -- used only when they
    were taken with THIS library build (source hash) and workload; otherwise the fields stay null (bench.py cannot read
    hardware counters itself)."""
    try:
        from spiking_fullsubnet_amd import _lib
        pj = json.load(open(PROFILE_JSON))
        if pj.get("source_hash") != _lib.source_hash():
            return None
        return pj  # (the caller also matches the workload: B, T, geometry -- see _pmc_matches)
    except Exception:
        return None


def _pmc_matches(pj, B, T, geom, n_lanes):
    """PMC figures are attached only to the workload they were taken on (round-2 advisor finding: the source hash alone let a run
    with other arguments report mislabelled traffic)."""
    w = (pj or {}).get("workload") or {}
    # (the traffic of one forward does not depend on how many forwards are in flight: matched on the batch, the sequence length
    #  and the launch geometry of the timed region)
    return bool(pj) and w.get("B") == B and w.get("T") == T and list(w.get("timed_region_rows_per_wg", [])) == list(geom)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-layer-outputs", action="store_true", help="skip the fp32 spike tensors of the module API (reported in config)")
    ap.add_argument("--sequential", action="store_true", help="the timed region runs one forward at a time (= --inflight 1)")
    ap.add_argument("--time-all", action="store_true", help="HIP-event time every launch group, not only the scans")
    ap.add_argument("--time-region", action="store_true", help="HIP events around the fused sub-band scan inside the timed region too (off: nothing but the forwards is enqueued there)")
    ap.add_argument("--rpw", type=str, default="", help="rows per scan workgroup 'fb,sb' of the per-layer launches (0 = auto)")
    ap.add_argument("--inflight", type=int, default=12, help="forwards in flight on separate HIP streams (batch-level pipelining)")
    ap.add_argument("--no-phase-a", action="store_true", help="skip the untimed single-forward phases (no roofline object): profiling runs of the timed region alone")
    ap.add_argument("--no-saturated", action="store_true", help="(kept for old command lines; the saturated-launch leg is gone)")
    ap.add_argument("--stack", type=str, default="auto", help="stack scan policy: auto | 0 | 1")
    ap.add_argument("--streaming", action="store_true", help="BASELINE configs[4]: frame-by-frame session, per-call latency (own JSON line)")
    ap.add_argument("--hop", type=int, default=1, help="frames per streaming call")
    ap.add_argument("--no-graph", action="store_true", help="streaming: launch the kernels one by one instead of replaying the HIP graph")
    ap.add_argument("--no-one-launch", action="store_true", help="streaming: the offline kernels per hop (HIP graph) instead of sfsn_stream_hop")
    ap.add_argument("--waveform", action="store_true", help="streaming: samples in, samples out (STFT and inverse STFT inside the launch)")
    ap.add_argument("--resident", action="store_true", help="with --host-io: one resident launch serves every hop (doorbell in pinned memory) instead of one launch per hop")
    ap.add_argument("--host-io", action="store_true", help="waveform streaming with the samples in host memory on both sides (pinned, read / written by the launch)")
    ap.add_argument("--training", action="store_true", help="SURVEY 8f-4: one training step (forward in train() mode + backward) of the live model, own JSON line")
    ap.add_argument("--no-training-leg", action="store_true", help="skip the three training steps (config.training) behind the timed region")
    ap.add_argument("--no-w16-leg", action="store_true", help="skip the 16-bit-weight report leg (config.w16) behind the timed region")
    ap.add_argument("--weight-bits", type=int, default=24, choices=(24, 16), help="16: the WHOLE line in the 16-bit-weight mode (module.weight_bits = 16; a report mode, not the parity gate)")
    ap.add_argument("--no-streaming-leg", action="store_true", help="skip the 2,000-hop streaming measurement (config.streaming) behind the timed region")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _self_launch(args)
    args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    # one rank per GPU.  (SFSN_BENCH_BACKEND=gloo lets several ranks share one GPU: a plumbing check of the multi-rank
    # code path on a single-GPU box, not a measurement.)
    backend = os.environ.get("SFSN_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or os.environ.get("SFSN_BENCH_FORCE_DIST"):  # (forced: the RCCL all-gather with a single rank, a plumbing check)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            # one rank per GPU: a rank needs a GPU of its own (RCCL over xGMI); more ranks than visible GPUs cannot be one node's job
            assert int(os.environ["WORLD_SIZE"]) <= torch.cuda.device_count() and local_rank < torch.cuda.device_count(), \
                f"WORLD_SIZE={os.environ['WORLD_SIZE']} ranks but {torch.cuda.device_count()} visible GPU(s): one rank per GPU under nccl"
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == int(os.environ["WORLD_SIZE"]) == max(world, 1)

    import refweights as rw
    import spiking_fullsubnet_amd as pkg
    from spiking_fullsubnet_amd import _lib

    kw = rw.LIVE_M
    B, T = args.batch, args.frames
    sd = rw.live_state_dict(kw, 21)
    model = pkg.SpikingFullSubNet(**kw)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model = model.eval().to(dev)
    model.weight_bits = args.weight_bits
    def make_input(lane):
        """One batch per lane, seeded per (rank, lane): the forwards in flight read DIFFERENT inputs, as a serving loop does."""
        wave = torch.from_numpy(rw.synth_wave(B, T, seed=1000 * rank + lane)).to(dev)
        x = model._stft(wave).contiguous()  # untimed: the STFT is the edge of the path
        assert x.shape == (B, 257, T)
        return x
    if args.training:
        return training_bench(args, model, dev, world, rank, rw, B, T)
    stft = make_input(0)
    eng = model.engine()
    eng.stack_scan = "auto" if args.stack == "auto" else bool(int(args.stack))
    if args.streaming:
        return streaming_bench(args, model, dev, world, rank)
    want_layers = not args.no_layer_outputs
    if args.no_layer_outputs:
        eng.lean_skips_proj = True  # (the lean modes of the bench: the sub-band coefficient rows stay in LDS too, see Engine.lean_skips_proj)
    gathered = {}  # per HIP stream (lane): the all-gathered magnitudes of that lane's batch

    def forward(x=None):
        res = eng.forward_stft(stft if x is None else x, want_layers=want_layers, pipeline=False)
        if dist is not None:
            # the one exchange of the path (the analogue of accelerator.gather_for_metrics, audiozen/trainer.py:511,555): enqueued on
            # the forward's own stream, so with several forwards in flight it overlaps the scans of the other lanes
            key = torch.cuda.current_stream(dev).cuda_stream
            if key not in gathered:
                gathered[key] = torch.empty((world * B, 1, 257, T), dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(gathered[key], res["enh_mag"])
        return res

    def timed_region(step_fn, steps, warmup):
        for _ in range(warmup):
            step_fn()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    # ---- the roof as this box delivers it: a 1 GiB device-to-device copy (read + write), next to the 8 TB/s spec number
    src_ = torch.empty((1 << 28,), dtype=torch.float32, device=dev)
    dst_ = torch.empty_like(src_)
    for _ in range(3):
        dst_.copy_(src_)
    torch.cuda.synchronize()
    t_c = time.perf_counter()
    for _ in range(10):
        dst_.copy_(src_)
    torch.cuda.synchronize()
    copy_gbps = 10 * 2 * src_.numel() * 4 / (time.perf_counter() - t_c) / 1e9
    del src_, dst_

    scan_tags = None if args.time_all else {"scan:sb", "scan:fb", "stack:sb", "stack:fb", "scanx:sb", "scanf:sb"}
    geom_b = tuple(int(v) for v in args.rpw.split(",")) if args.rpw else (8, 16)

    def set_geometry(g):
        """(full-band, sub-band) rows per scan workgroup; the full-band entry also sets the rows per workgroup of the full-band
        stack launch.  (0, 0) = the engine's defaults for a forward alone (4 rows: shortest chain); the timed region runs
        (8, 16): half / a quarter of the workgroups per forward, so that more forwards' scans fit side by side."""
        eng.rows_per_wg = g
        eng.stack_rows_fb_auto = g[0] if g[0] in (4, 8, 16) else 4
    n_lanes = 1 if args.sequential else max(1, min(args.inflight, args.steps))  # (--steps 20: 12 lanes measured 36-37 M, 10 lanes 34.5-35.8 M frames/s)
    # The timed region deals its steps round-robin over the lanes: a step count that is not a multiple of the lanes leaves some lanes one
    # forward short, i.e. the region ends in a drain where part of the chip idles (--steps 20 on 12 lanes = 12 + 8: the driver's line
    # sat 2.4 % under the 60-step one, round-5 review).  The count is rounded UP to a multiple of the lanes; the line says so
    # (`steps` = the steps actually timed, `config.steps_requested` = the command line's) and `value` counts exactly the timed steps.
    steps_requested = args.steps
    if n_lanes > 1 and args.steps % n_lanes:
        args.steps = -(-args.steps // n_lanes) * n_lanes
        print(f"bench.py: --steps {steps_requested} rounded up to {args.steps} (a multiple of the {n_lanes} lanes in flight)", file=sys.stderr)

    # ---- phase S (untimed for `value`): THE STRICT NUMBER -- one forward at a time on one stream, B clips x T frames per step,
    #      nothing else in flight.  Launch geometry = the engine's default for a forward alone (full-band stack in one
    #      layer-pipelined launch, sub-band layers as full-chip launches at 4 rows per workgroup).
    single, t_s, t_k = None, {}, {}
    if not args.no_phase_a:
        set_geometry((0, 0))
        ka = max(2, min(args.steps, 8))
        dt_a = timed_region(forward, ka, min(args.warmup, 2) + 1)
        eng.check_stack_errors()
        # the same forward with every scan as ONE whole-sequence launch (no chunk overlap), HIP events around the scans: the
        # kernels' own durations for the roofline figures
        ov = eng.overlap_chunks
        eng.overlap_chunks = 0
        eng.timers, eng.timer_tags = {}, scan_tags
        for _ in range(4):
            forward()
        t_s = eng.timer_summary()
        eng.timers = None
        eng.overlap_chunks = ov
        sb_how = ("sub-band layers side by side in one launch (layer 2 forms its input product inside the scan)" if "stack:sb" in t_s and "scan:sb" not in t_s
                  else "sub-band layers as full-chip launches")
        single = dict(ms_per_step=round(1e3 * dt_a / ka, 4), value=round(world * B * T * ka / dt_a, 1), steps=ka, in_flight=1,
                      schedule=f"full-band stack in one layer-pipelined launch; {sb_how}; the sequence in "
                               f"{eng.overlap_chunks} chunks with the sub-band models one chunk behind the full-band model on a second stream"
                      if eng.overlap_chunks > 1 else f"full-band stack in one layer-pipelined launch; {sb_how}")
        # ---- phase K (untimed): the scan kernels of the timed region's geometry, each alone on the chip (one forward at a time)
        if n_lanes > 1:
            set_geometry(geom_b)
            ov = eng.overlap_chunks
            eng.overlap_chunks = 0
            eng.timers, eng.timer_tags = {}, scan_tags
            for _ in range(4):
                forward()
            t_k = eng.timer_summary()
            eng.timers = None
            eng.overlap_chunks = ov

    # ---- phase B (THE timed region): `--inflight` forwards in flight on as many HIP streams -- batch-level pipelining of
    #      independent batches, as a serving loop runs them.  The recurrent scans are latency-bound chains that occupy a
    #      fraction of the CUs (16 rows per sub-band workgroup here, so that several scans fit side by side); the next batches'
    #      scans and time-parallel kernels fill the rest of the chip.  Every step is one complete pass over one batch.
    t_b = {}
    ov_default = eng.overlap_chunks
    if n_lanes > 1:
        eng.overlap_chunks = 0  # the full-band / sub-band overlap of ONE forward only pays when nothing else fills the chip
        set_geometry(geom_b)
        lanes = [torch.cuda.Stream(device=dev) for _ in range(n_lanes)]
        lane_in = [stft] + [make_input(i) for i in range(1, n_lanes)]  # every lane its own batch (131 MB each at B=64, T=1000)
        counter = [0]
        for s_, x_ in zip(lanes, lane_in):  # untimed: first use of a lane allocates its scratch buffers and warms its memory pool
            with torch.cuda.stream(s_):
                forward(x_)
        torch.cuda.synchronize()
        # (round 5: no HIP-event timers inside the timed region -- each timed group costs a pair of event records on the lane's stream;
        #  `--time-region` arms them for the dominant kernel's in-region time share, which the default line now takes from phase K alone)
        if args.time_region:
            eng.timers, eng.timer_tags = {}, {"scanf:sb"}

        def step():
            k_ = counter[0] % len(lanes)
            counter[0] += 1
            with torch.cuda.stream(lanes[k_]):
                return forward(lane_in[k_])
    else:
        set_geometry(tuple(int(v) for v in args.rpw.split(",")) if args.rpw else (0, 0))
        step = forward
    dt = timed_region(step, args.steps, args.warmup)
    if eng.timers is not None:
        t_b = eng.timer_summary()
        eng.timers = None
    eng.check_stack_errors()

    # ---- the mode the reference's LIVE recipe runs (recipes/intel_ndns/spiking_fullsubnet/trainer.py:31,52 star-unpack and discard the
    #      per-layer lists; only the frozen trainer reads them, and only to reduce them to SynOPs / NeuronOPs, audiozen/metric.py:303-340):
    #      layer_outputs="counts" -- no fp32 spike tensor is written, every scan counts the spikes it flushes, nothing extra
    #      is launched.  Behind the timed region of `value` (which stays API-faithful), same steps: the strict forward, the sub-band
    #      launch alone, and the region with the same lanes.  -> config.no_layer_outputs
    lean = None
    if want_layers and not args.no_phase_a and not args.sequential:
        eng.lean_skips_proj = True  # (this leg only: the coefficient rows stay in LDS -- sfsn_proj_deepfilter with proj = NULL)

        def forward_lean(x=None):
            res = eng.forward_stft(stft if x is None else x, want_layers=False, want_counts=True, pipeline=False)
            if dist is not None:
                key = torch.cuda.current_stream(dev).cuda_stream
                dist.all_gather_into_tensor(gathered[key], res["enh_mag"])
            return res
        set_geometry((0, 0))
        eng.overlap_chunks = ov_default
        ka = max(2, min(args.steps, 8))
        dt_ls = timed_region(forward_lean, ka, 3)
        eng.check_stack_errors()
        eng.overlap_chunks = 0
        eng.timers, eng.timer_tags = {}, scan_tags
        for _ in range(4):
            forward_lean()
        t_l = eng.timer_summary()
        eng.timers = None
        lean = dict(single_stream=dict(ms_per_step=round(1e3 * dt_ls / ka, 4), value=round(world * B * T * ka / dt_ls, 1), steps=ka, in_flight=1),
                    scan_groups_whole_launch_ms={k: round(v["mean_ms"], 4) for k, v in t_l.items()})
        if n_lanes > 1:
            set_geometry(geom_b)
            counter[0] = 0

            def step_lean():
                k_ = counter[0] % len(lanes)
                counter[0] += 1
                with torch.cuda.stream(lanes[k_]):
                    return forward_lean(lane_in[k_])
            dt_lr = timed_region(step_lean, args.steps, args.warmup)
            eng.check_stack_errors()
            lean["timed_region"] = dict(ms_per_step=round(1e3 * dt_lr / args.steps, 4), value=round(world * B * T * args.steps / dt_lr, 1),
                                        steps=args.steps, warmup=args.warmup, in_flight=n_lanes, scan_rows_per_workgroup=list(geom_b))
        eng.overlap_chunks = ov_default
        eng.lean_skips_proj = False

    # ---- BASELINE configs[2] words its mode "bf16": the 16-bit-weight mode (module.weight_bits = 16: recurrent, spike-input and projection
    #      weights rounded to 16 significant bits of their row grid = two int8 digit planes instead of three; activations are spikes and
    #      fp32 as before).  SURVEY 0: it cannot meet the 1e-4 gate -- this is a REPORT leg behind the timed region (the parity mode stays
    #      `value`): the strict forward, the sub-band launch and the per-layer sub-band scan in both modes, and what the rounding does to
    #      the spike trains and the output on the same input.  -> config.w16
    w16 = None
    if rank == 0 and want_layers and not args.no_phase_a and not args.no_w16_leg and args.weight_bits == 24:
        try:
            m16 = pkg.SpikingFullSubNet(**kw)
            m16.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
            m16 = m16.eval().to(dev)
            m16.weight_bits = 16
            e16 = m16.engine()
            e16.stack_scan = eng.stack_scan
            # (the second engine shares the first one's side streams: streams created THIS late in the process -- behind the twelve lanes'
            #  and the sessions' -- landed on one hardware queue and the full-band / sub-band overlap of a forward was gone: 2.77 ms against 2.36)
            e16._ov_streams, e16._err_stream = eng._ov_streams, eng._err_stream
            set_geometry((0, 0))
            e16.rows_per_wg, e16.stack_rows_fb_auto = eng.rows_per_wg, eng.stack_rows_fb_auto
            eng.overlap_chunks = e16.overlap_chunks = ov_default
            ka = max(2, min(args.steps, 8))

            def run16(x=None):
                return e16.forward_stft(stft, want_layers=True, pipeline=False)

            def run24(x=None):
                return eng.forward_stft(stft, want_layers=True, pipeline=False)
            # the two modes alternately at THIS point of the process (the strict figure of phase S was taken minutes ago, before the timed
            # region's allocations): two rounds each, the better one reported for both
            dt16 = dt24 = 1e9
            for _ in range(2):
                dt24 = min(dt24, timed_region(run24, ka, 3))
                dt16 = min(dt16, timed_region(run16, ka, 3))
            e16.check_stack_errors()

            def whole_launch(e_, pair):
                ov_, ps_ = e_.overlap_chunks, e_.pair_scan
                e_.overlap_chunks, e_.pair_scan = 0, pair
                e_.timers, e_.timer_tags = {}, scan_tags
                for _ in range(4):
                    e_.forward_stft(stft, want_layers=True, pipeline=False)
                t_ = e_.timer_summary()
                e_.timers = None
                e_.overlap_chunks, e_.pair_scan = ov_, ps_
                return {k: round(v["mean_ms"], 4) for k, v in t_.items()}
            t16_pair, t16_layer, t24_layer = whole_launch(e16, True), whole_launch(e16, False), whole_launch(eng, False)
            ref = eng.forward_stft(stft, want_layers=True, pipeline=False)
            got = e16.forward_stft(stft, want_layers=True, pipeline=False)
            torch.cuda.synchronize()
            agree = {}
            for name, a_, b_ in ([("fb", ref["fb_all"], got["fb_all"])] + [(f"sb{g}", ref["sb_all"][g], got["sb_all"][g]) for g in range(len(ref["sb_all"]))]):
                for l in range(1, len(a_) - 1):
                    agree[f"{name}/layer{l}"] = round(float((a_[l] == b_[l]).float().mean().item()), 5)
            rel = float((torch.linalg.vector_norm(got["enh_mag"] - ref["enh_mag"]) / torch.linalg.vector_norm(ref["enh_mag"])).item())
            w16 = dict(mode="module.weight_bits = 16 (sfsn_w3_pack_bits: two int8 digit planes; the real-valued layer-0 input product stays exact); "
                            "a report, not the parity mode (SURVEY 0: 16-bit weights cannot meet 1e-4 against the fp32 reference)",
                       single_stream=dict(ms_per_step=round(1e3 * dt16 / ka, 4), value=round(world * B * T * ka / dt16, 1), steps=ka, in_flight=1,
                                          fp32_mode_ms_per_step=round(1e3 * dt24 / ka, 4)),
                       scan_groups_whole_launch_ms=dict(w16_pair_launch=t16_pair, w16_per_layer_two_plane_scans=t16_layer,
                                                        fp32_mode_per_layer_three_plane_scans=t24_layer,
                                                        fp32_mode_pair_launch={k: round(v["mean_ms"], 4) for k, v in t_s.items()}),
                       sub_band_pair_launch_hbm_frac=(round(kw["sb_num_layers"] * SB_SCAN_BYTES_PER_FRAME_PER_LAUNCH * B * T
                                                            / (t16_pair["stack:sb"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if "stack:sb" in t16_pair else None),
                       spike_agreement_with_fp32_mode=agree, enh_mag_rel_l2_vs_fp32_mode=round(rel, 5),
                       note="same input batch, B x T as the headline; spike agreement = fraction of equal entries of the fp32 spike tensors per "
                            "layer; a 2^-16 weight perturbation decorrelates the spike trains like any other perturbation of this recurrence")
            del m16, e16, ref, got
            torch.cuda.empty_cache()
        except Exception as e:  # reported, never required for the headline
            w16 = dict(error=repr(e))
        eng.overlap_chunks = ov_default

    # ---- BASELINE configs[4] behind the timed region: 2,000 one-frame hops of a B=1 streaming session (about 70 ms), so that the
    #      driver's record of the default command carries the streaming latency too (python bench.py --streaming prints the full line)
    streaming = None
    if rank == 0 and not args.no_streaming_leg and not args.no_phase_a:
        try:
            streaming = streaming_measure(model, dev, 1, 1, 2000, 200, None, True, "auto")
        except Exception as e:  # the streaming leg is reported, never required for the headline
            streaming = dict(error=repr(e))

    # ---- SURVEY 8f-4 behind the timed region too: three training steps (forward in train() mode + backward) of a COPY of the model at
    #      this batch, so that the driver's record of the default command carries the figure (python bench.py --training prints the line)
    training_leg = None
    if rank == 0 and not args.no_training_leg and not args.no_phase_a:
        try:
            m2 = pkg.SpikingFullSubNet(**kw)  # (a fresh module with the same weights: the measured model's engine and buffers stay untouched)
            m2.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
            m2 = m2.to(dev).train()
            w2 = torch.from_numpy(rw.synth_wave(B, T, seed=7)).to(dev)

            def tstep():
                for p_ in m2.parameters():
                    p_.grad = None
                out = m2(w2)
                (out[0].pow(2).mean() + out[1].mean()).backward()
            tstep()
            torch.cuda.synchronize()
            t_tr = time.perf_counter()
            for _ in range(3):
                tstep()
            torch.cuda.synchronize()
            training_leg = dict(ms_per_step=round((time.perf_counter() - t_tr) / 3 * 1e3, 2), steps=3, clips=B, frames=T,
                                what="forward in train() mode + backward of the whole live model (per-step batch-statistics BatchNorm, triangle "
                                     "surrogate; no optimiser step); the layers of every stack pipelined over chunks of frames (training.GSNStackTrainFn)")
            del m2, w2
            torch.cuda.empty_cache()
        except Exception as e:  # reported, never required for the headline
            training_leg = dict(error=repr(e))

    if rank == 0:
        frames = world * B * T * args.steps
        ms_per_step = 1e3 * dt / args.steps
        alg = SB_SCAN_BYTES_PER_FRAME_PER_LAUNCH * B * T  # algorithmic bytes of one sub-band layer launch
        job_alg = (JOB_BYTES_PER_FRAME_API if want_layers else JOB_BYTES_PER_FRAME_MIN) * B * T  # ... of one whole forward
        pj = _pmc_profile()
        if not _pmc_matches(pj, B, T, geom_b, n_lanes):
            pj = None
        spec = eng.spec
        sb_rows = [B * spec.units(g) for g in range(spec.n_groups)]
        Hs = kw["sb_hidden_size"]

        def wgs(rows, rpw):
            return sum(-(-r // rpw) for r in rows)

        def hbm(ms, nbytes=alg):
            a = nbytes / (ms * 1e-3) / 1e9
            return dict(launch_ms=round(ms, 4), achieved=round(a, 1), unit="GB/s", frac=round(a / HBM_PEAK_GBPS, 4))

        roofline = None
        if single is not None:
            # --- the sub-band scan of a forward ALONE (phase S): one launch per layer, every group in it
            ss = t_s.get("scan:sb")
            strict = None
            if ss:
                rp_s = 4 if wgs(sb_rows, 8) < 200 else (8 if wgs(sb_rows, 16) < 200 else 16)  # the library's rule (sfsn_gsn_layer_scan)
                body = ("gsn_scan3_kernel (IO-specialised waves: %d compute + loader + storer, %d-row re-dealt epilogue)" % (Hs // 16, rp_s)
                        if Hs // 16 <= 14 and rp_s <= 8 and spec.shared else "gsn_scan_kernel (round 2's body)")
                strict = dict(kernel=f"{body}, KS={(Hs + 63) // 64}, OUT=fp32+int8 spikes, {spec.n_groups} sub-band groups in one launch per layer, "
                                     f"{wgs(sb_rows, rp_s)} workgroups of {rp_s} rows",
                              **hbm(ss["mean_ms"]), per_step_us=round(1e3 * ss["mean_ms"] / T, 3),
                              launches=ss["n"], algorithmic_bytes_per_launch=int(alg),
                              traffic=(pj or {}).get("sb_scan_single_hbm_bytes_per_launch"),
                              measured_in="one forward at a time, the whole sequence in one launch per layer (HIP events on the launch stream)")
            # --- round 4: the sub-band LAYERS SIDE BY SIDE in one launch (sfsn_gsn_stack_scan, layer 1 = the IO-wave scan at 8 rows
            #     publishing its int8 spikes, layer 2 = the FUSED3 role that forms its input product inside the scan): the
            #     algorithmic bytes of BOTH layers over the launch's time
            sp = t_s.get("stack:sb")
            if sp and not ss:
                nl_sb = kw["sb_num_layers"]
                strict = dict(kernel=f"gsn_stack_wide_kernel<KS={(Hs + 63) // 64}> (all {nl_sb} sub-band layers in ONE launch: layer 1 = IO-wave scan role, "
                                     f"8 rows per workgroup, int8 spikes handed to layer 2 inside the launch; layers >= 2 = FUSED3 role, input "
                                     f"product inside the scan, two frames per matrix instruction), {nl_sb * wgs(sb_rows, 8)} workgroups of 8 rows",
                              **hbm(sp["mean_ms"], nl_sb * alg), per_step_us=round(1e3 * sp["mean_ms"] / T, 3),
                              launches=sp["n"], layers_per_launch=nl_sb, algorithmic_bytes_per_launch=int(nl_sb * alg),
                              traffic=(pj or {}).get("sb_pair_hbm_bytes_per_launch"),
                              measured_in="one forward at a time, the whole sequence in one launch for all layers (HIP events on the launch stream)",
                              floor_note="the north star's 0.40 is NOT reachable with exact 24-bit weights in this structure: a frame of the layer-2 "
                                         "(FUSED3) role is 72 matrix instructions x 16 clk = 1,152 clk on the SIMDs that carry four tiles PLUS ~780 clk of "
                                         "epilogue VALU -- on gfx950 the two do not overlap within a SIMD (scripts/micro/pingpong_step.hip, round 6: two "
                                         "INDEPENDENT 8-row blocks per workgroup, half a step apart, 1,409 against 1,401 clk per 8 row-frames; staggered "
                                         "wave slots 1,444) -- i.e. ~1,930 clk = 0.80 us per frame = 0.29 of the roof if the IO side were free; measured "
                                         "2,040-2,180 clk with it (profiles/r05_stall_ledger.txt).  0.40 needs <= 1,400 clk per frame")
            # --- the full-band stack (phase S): both layers + the layer-2 input product in ONE layer-pipelined launch
            fb = t_s.get("stack:fb")
            full_band = None

            def proj_wgs(rows):  # 16-row blocks x the column parts that fit the role's padding to eight blocks (sfsn_stack.hip, proj_split_host)
                nb = (rows + 15) // 16
                room, sp = ((nb + 7) // 8 * 8) // nb, 1
                while sp * 2 <= room and sp * 2 <= 4:
                    sp *= 2
                return nb * sp
            if fb:
                Hf, nl = kw["fb_hidden_size"], kw["fb_num_layers"]
                steps_ms = fb["mean_ms"]
                useful = 2.0 * B * Hf * Hf * (2 * nl - 1) * T  # int8 MACs x 2 of the recurrent + layer>=1 input products, one digit plane
                # executed by the matrix cores: x3 digit planes, 16-column MFMA tiles for 4 (scan) / 16 (input product) rows
                executed = 2.0 * Hf * Hf * 3 * T * (nl * (B / 4) * 16 + (nl - 1) * B)
                full_band = dict(kernel=f"gsn_stack_fb_kernel (round 5: IO-specialised waves -- ten compute waves x two tiles + loader + storer, W_hh two digit planes in "
                                        f"registers + one in LDS; PROJ role on twelve waves, split by columns over its padding workgroups, feeds layer 2): this "
                                        f"whole-sequence launch and the chunks of the strict schedule alike (DESIGN 5.6b)",
                                 launch_ms=round(steps_ms, 4), per_step_us=round(1e3 * steps_ms / T, 3),
                                 workgroups=nl * ((B + 3) // 4) + (nl - 1) * proj_wgs(B), launches_per_forward=1,
                                 mfma=dict(useful_TOPS=round(useful / (steps_ms * 1e-3) / 1e12, 2), executed_TOPS=round(executed / (steps_ms * 1e-3) / 1e12, 2),
                                           peak_TOPS=INT8_PEAK_TOPS, useful_frac_of_peak=round(useful / (steps_ms * 1e-3) / 1e12 / INT8_PEAK_TOPS, 5),
                                           measured_ceiling_TOPS=INT8_MEASURED_TOPS,
                                           useful_frac_of_measured_ceiling=round(useful / (steps_ms * 1e-3) / 1e12 / INT8_MEASURED_TOPS, 5),
                                           pmc=(pj or {}).get("full_band_stack_mfma"),
                                           note="useful = 2*B*H*H ops per recurrent / input product per frame; executed counts the three int8 digit "
                                                "planes and the 16-column MFMA tiles; pmc = SQ_VALU_MFMA_BUSY_CYCLES based utilisation from "
                                                "profiles/ (null unless taken with this build and workload)"))
            # --- TOP LEVEL: the job.  Algorithmic bytes of one forward (whole path) / time per step of the timed region:
            #     a figure whose time fits in ms_per_step by construction.  Per-kernel figures are sub-fields, each with the time it
            #     was measured over.
            ja = job_alg / (ms_per_step * 1e-3) / 1e9
            roofline = dict(bound="hbm", kernel="the whole forward (all kernels of one pass over one batch)",
                            achieved=round(ja, 1), peak=HBM_PEAK_GBPS, unit="GB/s", frac=round(ja / HBM_PEAK_GBPS, 4),
                            traffic=(pj or {}).get("forward_hbm_bytes"),
                            algorithmic_bytes_per_step=int(job_alg), ms_per_step=round(ms_per_step, 4),
                            bytes_definition=("SURVEY 8d whole path, API-faithful: 39,724 B per clip-frame" if want_layers else
                                              "SURVEY 8d whole path, minimal: 5,140 B per clip-frame (fp32 spike tensors skipped)"),
                            measured_in=f"the timed region: {n_lanes} forward(s) in flight, each on its own input batch")
            if single is not None:
                sa = job_alg / (single["ms_per_step"] * 1e-3) / 1e9
                roofline["single_stream_job"] = dict(achieved=round(sa, 1), unit="GB/s", frac=round(sa / HBM_PEAK_GBPS, 4), ms_per_step=single["ms_per_step"])
            if n_lanes > 1 and t_k.get("scanf:sb"):
                # the kernel that dominates the TIMED region by CU-time: the fused-input sub-band layer-2 scan at 16 rows per
                # workgroup; alone on the chip (phase K), and as it ran inside the region (HIP events on the lane streams: other
                # forwards' kernels run beside it, so that wall time is a time share and may exceed ms_per_step)
                kf, kx, kp = t_k["scanf:sb"], t_k.get("scanx:sb"), t_k.get("scan:sb")
                roofline["dominant_kernel_timed_region"] = dict(
                    kernel=(f"gsn_scan_fused3_kernel<KS={(Hs + 63) // 64}> (round 6: IO-specialised waves -- 14 compute waves + loader + storer; " if Hs // 16 <= 14
                            else f"gsn_scan_fused_kernel<KS={(Hs + 63) // 64}> (") +
                           f"sub-band layer 2, input product inside, {geom_b[1]} rows per workgroup, {wgs(sb_rows, geom_b[1])} workgroups, "
                           f"{spec.n_groups} groups in one launch, OUT=fp32+int8 spikes)",
                    alone_on_chip=hbm(kf["mean_ms"]), per_step_us=round(1e3 * kf["mean_ms"] / T, 3),
                    in_region_time_share=hbm(t_b["scanf:sb"]["mean_ms"]) if t_b.get("scanf:sb") else None,  # (--time-region; profiles/r05_region_wg_residency.json has the exact per-workgroup figures)
                    launches=(t_b.get("scanf:sb") or kf)["n"],
                    algorithmic_bytes_per_launch=int(alg), frames_per_launch=B * T,
                    traffic=(pj or {}).get("sb_fused_hbm_bytes_per_launch"),
                    other_scan_kernels_alone_ms={k: round(v["mean_ms"], 4) for k, v in dict(layer1_fused_x=kx, layer1_plain=kp).items() if v})
            roofline["measured_copy_GBps"] = round(copy_gbps, 1)
            roofline["sub_band_scan_single_forward"] = strict
            roofline["full_band_stack"] = full_band
            roofline["profiles"] = PROFILE_JSON.replace(ROOT + os.sep, "") if pj else None
        lean_obj = None
        if lean is not None:
            # rooflines on the MINIMAL byte variants of SURVEY 8d (no layer outputs): whole path 5,140 B, sub-band scan 5,888 B per clip-frame
            mn_job, mn_sb = JOB_BYTES_PER_FRAME_MIN * B * T, SB_SCAN_BYTES_PER_FRAME_MIN * B * T
            def frac(nbytes, ms):
                a = nbytes / (ms * 1e-3) / 1e9
                return dict(achieved=round(a, 1), unit="GB/s", frac=round(a / HBM_PEAK_GBPS, 4), ms=round(ms, 4), algorithmic_bytes=int(nbytes))
            lp = lean["scan_groups_whole_launch_ms"].get("stack:sb")
            lean_obj = dict(
                mode='layer_outputs="counts": no fp32 spike tensors; SpikeSummary (exact spike count + shape) per layer, counted inside the scans '
                     '(sfsn_scan_segment.spike_count); since round 6 the sub-band coefficient rows are not written either (Engine.lean_skips_proj: they '
                     'stay in LDS inside sfsn_proj_deepfilter) -- what recipes/intel_ndns/spiking_fullsubnet/trainer.py:31,52 needs (it discards the lists) '
                     'and what audiozen/metric.py:303-340 reads',
                single_stream=lean["single_stream"], timed_region=lean.get("timed_region"),
                scan_groups_whole_launch_ms=lean["scan_groups_whole_launch_ms"],
                roofline=dict(job_minimal_bytes_single_stream=frac(mn_job, lean["single_stream"]["ms_per_step"]),
                              job_minimal_bytes_timed_region=frac(mn_job, lean["timed_region"]["ms_per_step"]) if lean.get("timed_region") else None,
                              sub_band_scan_minimal_bytes=frac(mn_sb, lp) if lp else None,
                              sub_band_scan_api_bytes_for_comparison=frac(kw["sb_num_layers"] * alg, lp) if lp else None,
                              traffic=(pj or {}).get("forward_hbm_bytes_no_layer_outputs"),
                              note="SURVEY 8d minimal variants: the scan still reads its input terms / features and writes the int8 spike rows the "
                                   "next product reads -- bytes the minimal figure does not count; the fraction is small because the mode is "
                                   "bound by the T-step dependency chain, not by these bytes (per-step us are the figures to read)"))
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(kw, sd, stft)
        line = dict(metric="STFT frames/sec at B=64 T=1000 F=257 (hot path: noisy STFT -> enhanced STFT + magnitude)",
                    value=round(frames / dt, 1), unit="frames/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                    ms_per_step=round(ms_per_step, 4), higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32",
                    data="synthetic (0.05*randn waveform -> Hann-512/128 STFT, resident in HBM; seeded random weights, randomised BN stats)",
                    config=dict(workload=("configs[2]: single MI355X" if world == 1 else
                                          f"configs[3]: {world} x MI355X, clips sharded over ranks ({world * B} per step), configs[2] per GPU") +
                                         ", full model (full-band + 3 sub-band groups / 13 units), live baseline_m sizes, fp32 parity mode",
                                clips_per_gpu=B, frames=T, bins=257,
                                layer_outputs="api-faithful (fp32 spikes returned)" if want_layers else "skipped",
                                in_flight=n_lanes, scan_rows_per_workgroup=list(eng.rows_per_wg), steps_requested=steps_requested,
                                single_stream=single, streaming=streaming, training=training_leg, no_layer_outputs=lean_obj, w16=w16,
                                weight_bits=args.weight_bits,
                                visible_gpus=torch.cuda.device_count(),
                                world_size=(dist.get_world_size() if dist is not None else 1), backend=(backend if dist is not None else None),
                                library=dict(abi=_lib.ABI_VERSION, source_hash=_lib.source_hash(), stack_scan=str(eng.stack_scan)),
                                parallelism=f"clip-sharded x{world}" + (" + RCCL all_gather(enh_mag) per step, on the forward's stream" if dist is not None else "")),
                    roofline=roofline, cpu_baseline=cpu)
        JSON_OUT.write(json.dumps(line) + "\n")
        JSON_OUT.flush()
    if dist is not None:
        dist.destroy_process_group()


def streaming_measure(model, dev, B, hop, steps, warmup, rpw, graph, one_launch):
    """Per-call latency of a streaming session: host wall time from handing over the frame(s) to the enhanced frame(s) being complete
    (synchronised), `steps` calls after `warmup`; then the unsynchronised call rate."""
    sess = model.streaming(batch=B, hop=hop, graph=graph, rows_per_wg=rpw, one_launch=one_launch)
    is_one = sess._hop is not None
    g = torch.Generator(device="cpu").manual_seed(3)
    frames = (0.05 * torch.randn((steps + warmup, B, 257, hop, 2), generator=g)).to(dev)
    frames = torch.view_as_complex(frames)
    lat, enq = [], []
    for i in range(steps + warmup):
        x = frames[i]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sess.step(x, copy=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        if i >= warmup:
            lat.append(time.perf_counter() - t0)
            enq.append(t1 - t0)
    lat = np.sort(np.asarray(lat)) * 1e6
    enq = np.sort(np.asarray(enq)) * 1e6
    sess.check_errors()
    # throughput of back-to-back calls without a host sync per call (the graph replays queue up)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        sess.step(frames[warmup + i], copy=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(workload="configs[4]: streaming, live baseline_m sizes, fp32 parity mode", clips_per_gpu=B, hop_frames=hop, calls=steps, warmup=warmup,
                schedule=("one launch per hop (sfsn_stream_hop: a wave per layer tile, frame handed from stage to stage through L2)"
                          if is_one else "the offline kernels per hop, replayed from a HIP graph" if graph
                          else "the offline kernels per hop, launched one by one"),
                one_launch=is_one, hip_graph=bool(graph) and not is_one,
                p50_us=round(float(lat[len(lat) // 2]), 1), p99_us=round(float(lat[int(len(lat) * 0.99)]), 1), min_us=round(float(lat[0]), 1),
                mean_us=round(float(lat.mean()), 2), host_enqueue_p50_us=round(float(enq[len(enq) // 2]), 1),
                unsynchronised_calls_per_s=round(steps / dt, 1), real_time_factor_at_8ms_hop=round(8e3 * hop / float(lat[len(lat) // 2]), 1))


def streaming_bench(args, model, dev, world, rank):
    """BASELINE.json configs[4]: B clips per GPU (default run: --batch 1), ``hop`` frames per call, state carried on the device;
    per-call latency = host wall time from handing over the frame(s) to the enhanced frame(s) being complete (synchronised)."""
    B = args.batch if args.batch != 64 else 1
    if args.waveform:
        return waveform_streaming_bench(args, model, dev, world, rank, B)
    hop, steps, warmup = args.hop, max(args.steps, 2000), max(args.warmup, 200)
    rpw = tuple(int(v) for v in args.rpw.split(",")) if args.rpw else None
    m = streaming_measure(model, dev, B, hop, steps, warmup, rpw, not args.no_graph, False if args.no_one_launch else "auto")
    if rank == 0:
        cfg = dict(m)
        for k in ("p50_us", "mean_us", "calls", "warmup"):
            cfg.pop(k)
        _emit(json.dumps({
            "metric": "streaming per-call latency p50 (BASELINE configs[4]: state carried on the device, hop frames per call)",
            "value": m["p50_us"], "unit": "us", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(m["mean_us"] / 1e3, 4), "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (0.05*randn complex frames; seeded random weights, randomised BN stats)",
            "config": cfg}))


def training_bench(args, model, dev, world, rank, rw, B, T):
    """a training step of the live baseline_m model -- forward in train() mode (BatchNorm on the batch statistics
    of every time step inside every cell, efficient_spiking_neuron.py:123,149-150) and backward through the triangle surrogate
    (:94-101) -- on this package's differentiable path (training.py: the layers of a model's stacks pipelined over chunks of frames in one grid per stage and direction, library GEMMs for the
    time-parallel products).  The recipe's batch is 64 clips (baseline_m.toml:72); the loss here is a stand-in of the same shape
    class (a mean over the enhanced waveform and magnitude).  Wall time per step, synchronised, no optimiser step (the optimiser,
    losses and trainer stay the reference's: out of scope)."""
    steps, warmup = min(args.steps, 5), min(args.warmup, 2)
    model.train()
    wave = torch.from_numpy(rw.synth_wave(B, T, seed=1000 * rank + 7)).to(dev)

    def one():
        for p_ in model.parameters():
            p_.grad = None
        out = model(wave)
        loss = out[0].pow(2).mean() + out[1].mean()
        loss.backward()
        return loss

    for _ in range(max(warmup, 1)):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    gn = float(torch.sqrt(sum((p_.grad.float() ** 2).sum() for p_ in model.parameters() if p_.grad is not None)))
    loss = float(loss.detach())  # (and no tensor of the eager steps' autograd graphs alive: GraphedTrainStep below refuses otherwise)
    from spiking_fullsubnet_amd import training as _tr
    # the same step captured once in a HIP graph and replayed (training.GraphedTrainStep): what the host's share of the eager figure is
    graphed = None
    try:
        gs = _tr.GraphedTrainStep(model, wave, lambda out: out[0].pow(2).mean() + out[1].mean())
        gs(wave)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            gs(wave)
        torch.cuda.synchronize()
        graphed = dict(ms_per_step=round((time.perf_counter() - t0) / steps * 1e3, 2), layer_call_launches_captured=gs.layer_calls_captured,
                       note="forward + loss + backward replayed from one HIP graph: gradients bit-identical to the eager step "
                            "(tests/test_training.py); at B = 64 the device is busy for the whole eager step (kernel time 69.5 of 70 ms, "
                            "profiles/r06_training_b64_kernel_stats.csv), so the graph buys ~1 %; at small batches the host's share is "
                            "larger (B = 8, T = 200: 12.2 -> 10.0 ms)")
        del gs
    except Exception as e:  # reported, never required for the line
        graphed = dict(error=repr(e))
    torch.cuda.empty_cache()
    # one more step (untimed) with HIP events around the layer-call launches: where the step's time is, per recurrent step
    _tr.launch_log = []
    one()
    torch.cuda.synchronize()
    log, _tr.launch_log = _tr.launch_log, None
    roof = None
    if log:
        per = {}
        for kind, T_, shapes, e0, e1 in log:
            d = per.setdefault(kind, dict(launches=0, ms=0.0, steps=0, flop=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["steps"] += T_
            d["flop"] += sum(2.0 * T_ * R * H * GH for R, H, GH in shapes)  # the recurrent product of every layer call of the launch
        tot_ms = sum(d["ms"] for d in per.values())
        flop = sum(d["flop"] for d in per.values())
        roof = {"bound": "latency (two grid-wide exchanges per recurrent step: BatchNorm partial sums, then the new spikes / d_z)",
                "kernel": "gsn_train_seq_fwd_kernel / gsn_train_seq_bwd_kernel (the layers of a model's stacks in one grid, layer l+1 a chunk of frames behind layer l: training.GSNStackTrainFn; the sub-band groups share the grid)",
                "layer_call_launches_ms": round(tot_ms, 2), "share_of_step": round(tot_ms / ms, 3),
                "forward": {"launches": per.get("fwd", {}).get("launches"), "ms": round(per.get("fwd", {}).get("ms", 0.0), 2),
                            "us_per_recurrent_step": round(1e3 * per["fwd"]["ms"] / per["fwd"]["steps"], 2) if "fwd" in per else None},
                "backward": {"launches": per.get("bwd", {}).get("launches"), "ms": round(per.get("bwd", {}).get("ms", 0.0), 2),
                             "us_per_recurrent_step": round(1e3 * per["bwd"]["ms"] / per["bwd"]["steps"], 2) if "bwd" in per else None},
                "mfma": {"instruction": "v_mfma_f32_16x16x4_f32", "useful_TFLOPs": round(flop / (tot_ms * 1e-3) / 1e12, 2), "peak_TFLOPs": 157.3,
                         "frac": round(flop / (tot_ms * 1e-3) / 157.3e12, 4),
                         "note": "recurrent products only (h.W_hh^T forward, dz.W_hh backward); the time-parallel products are library GEMMs outside these launches"}}
    if rank == 0:
        _emit(json.dumps({
            "metric": "training step wall time (forward in train() mode + backward), live baseline_m", "value": round(ms, 2), "unit": "ms",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 2), "higher_is_better": False, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (0.05*randn waveform; seeded random weights, randomised BN stats)",
            "config": {"workload": "SURVEY 8f-4: training-mode forward + backward, per-step batch-statistics BatchNorm, triangle surrogate",
                       "clips_per_gpu": B, "frames": T, "clip_frames_per_s": round(B * T / (ms / 1e3), 1), "loss": loss, "hip_graph_replay": graphed,
                       "grad_norm": gn, "optimizer_step": "not included (the reference's optimiser; out of scope)",
                       "cell_steps_per_training_step": 2 * 4 * T,
                       "note": "the same loop written as ATen operations per cell step (the reference's structure) takes 2.65 s at B=16 and "
                               "2.8 s at B=64 (scripts/exp_train.py); round 3 (one launch per cell step and direction): 315 ms at B=64; round 4 (one launch per layer call): 117 ms; round 5 (the layers of a stack pipelined over chunks of frames in one grid): 75 ms; round 6 (LayerNorm, feature assembly and deep filter of the differentiable path rewritten): 69-70 ms"},
            "roofline": roof}))


def waveform_streaming_bench(args, model, dev, world, rank, B):
    """Streaming on waveforms: 128 new samples (8 ms) per call, enhanced samples back (three hops late: the look-ahead of the
    centred 32 ms analysis + the overlap-add); per-call latency as in streaming_bench."""
    steps, warmup = max(args.steps, 2000), max(args.warmup, 200)
    sess = model.streaming(batch=B, waveform=True, host_io=args.host_io, resident=args.resident)
    g = torch.Generator(device="cpu").manual_seed(3)
    chunks = 0.05 * torch.randn((steps + warmup, B, 128), generator=g)
    if not args.host_io:
        chunks = chunks.to(dev)
    lat, enq = [], []
    for i in range(steps + warmup):
        x = chunks[i]
        if not args.resident:  # (a device-wide synchronise would wait for the resident kernel's watchdog)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        if args.host_io:  # samples in a CPU tensor -> enhanced samples in a CPU tensor (the call returns when they are there)
            sess.step_wave_host(x)
            t1 = time.perf_counter()
        else:
            sess.step_wave(x, copy=False)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
        if i >= warmup:
            lat.append(time.perf_counter() - t0)
            enq.append(t1 - t0)
    sess.close()
    sess.check_errors()
    lat = np.sort(np.asarray(lat)) * 1e6
    enq = np.sort(np.asarray(enq)) * 1e6
    if rank == 0:
        _emit(json.dumps({
            "metric": "waveform streaming per-call latency p50 (BASELINE configs[4] end to end: 128 samples in, 128 enhanced samples out" +
                      ("; host memory to host memory)" if args.host_io else ")"),
            "value": round(float(lat[len(lat) // 2]), 1), "unit": "us", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(float(lat.mean()) / 1e3, 4), "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (0.05*randn samples; seeded random weights, randomised BN stats)",
            "config": {"workload": "configs[4]: streaming on waveforms, live baseline_m sizes, fp32 parity mode", "clips_per_gpu": B,
                       "hop_samples": 128, "algorithmic_delay_samples": 384, "host_io": bool(args.host_io), "resident": bool(args.resident),
                       "schedule": ("one resident launch, a doorbell per hop" if args.resident else "one launch per hop") +
                                   ": STFT of the new frame, the whole model, inverse STFT with overlap-add state",
                       "host_enqueue_p50_us": round(float(enq[len(enq) // 2]), 1), "p99_us": round(float(lat[int(len(lat) * 0.99)]), 1),
                       "min_us": round(float(lat[0]), 1), "real_time_factor_at_8ms_hop": round(8e3 / float(lat[len(lat) // 2]), 1)}}))


def cpu_baseline(kw, sd, stft, weight_seed=21, workers=None, threads=None):
    """The CPU oracle (oracle/, the C restatement of the reference) on the host cores of this box, on the SAME workload: the B clips x
    all T frames of the timed region's first input batch, over and over for ~12 s.  Clips are independent: `workers` processes
    (oracle/cpu_bench_worker.py; default: as many as the cgroup CPU quota grants, else one per two hardware threads) each run whole clips
    (clip w, w + workers, ... of as many copies of the batch as it takes to give every worker one), single-threaded, so no core
    waits for another during the four sequence models; inside a worker the oracle takes the rows of a clip through the recurrence in
    blocks of 8 with the inner loops compiled for AVX-512 / AVX2 (sfsn_oracle.c gsn_layer; same additions in the same order).
    (Round 2: OpenMP over the rows of the whole batch, 3.6 x one core on a 256-thread host -- 64 full-band rows, four models one
    after the other, every row streaming both weight matrices per step.)  Same arithmetic as the parity oracle (double accumulation,
    rounded once); a stated baseline, not the target."""
    import ctypes
    import subprocess
    import tempfile
    from oracle import model as omodel
    spec = omodel.spec_from_live_kwargs(kw)
    full = stft.cpu().numpy()
    B, _, T = full.shape
    ncpu = os.cpu_count() or 1
    # what this process may actually use: the scheduler affinity and the cgroup CPU quota (the GPU boxes of this pool show 256
    # hardware threads and a cpu.max of 16 CPUs: more runnable workers than that only time-slice -- 64 / 128 / 256 workers gave
    # 13.7 / 10.2 / 5.3 x one core)
    avail, quota = ncpu, None
    try:
        avail = min(avail, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
            avail = max(1, min(avail, int(quota + 0.5)))
    except (OSError, ValueError):
        pass
    workers = int(os.environ.get("SFSN_CPU_WORKERS", workers or (avail if quota is not None else max(1, avail // 2))))
    threads = int(os.environ.get("SFSN_CPU_THREADS", threads or 1))
    worker = os.path.join(ROOT, "oracle", "cpu_bench_worker.py")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "stft.npy")
        np.save(path, full)
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="false")
        start = time.time() + 8.0 + 0.03 * workers     # every worker has imported numpy, built its weights and warmed up by then
        deadline = start + 12.0
        procs = [subprocess.Popen([sys.executable, worker, path, str(w % B), str(w % B + 1), repr(start), repr(deadline), str(weight_seed)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for w in range(workers)]
        outs = []
        for p in procs:
            o, _ = p.communicate(timeout=600)
            outs.append(json.loads(o.strip().splitlines()[-1]))
        el = max(o["t_end"] for o in outs) - start
        n_total = sum(o["forwards"] for o in outs)
        value = round(n_total * T / el, 1)
    # one core, for calibration: one clip, all T frames, one thread (in this process)
    single = cpu_model = None
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(1)
        omodel.forward_from_stft(spec, sd, full[:1, :, :16], "f32")
        t1 = time.perf_counter()
        omodel.forward_from_stft(spec, sd, full[:1], "f32")
        single = round(T / (time.perf_counter() - t1), 1)
        gomp.omp_set_num_threads(ncpu)
        with open("/proc/cpuinfo") as fh:
            cpu_model = next((l.split(":", 1)[1].strip() for l in fh if l.startswith("model name")), None)
    except Exception:  # the baseline is reported, never required
        pass
    return dict(value=value, unit="frames/s", cores=workers * threads, hardware_threads=ncpu, cgroup_cpu_quota=quota, kind="port", single_core_value=single, cpu_model=cpu_model,
                all_cores_over_one_core=(round(value / single, 1) if single and value else None),
                scaling_note=f"{workers} single-clip worker processes side by side, {threads} thread(s) each; rows of a clip in blocks of 8 through "
                             "the recurrence, AVX-512 / AVX2 inner loops (sfsn_oracle.c gsn_layer)",
                sample=f"{n_total} clip-forwards of T={T} frames (clips of the timed region's first input batch, B={B}, cycled over the workers) in "
                       f"{el:.1f} s of wall time, fp32 oracle (double accumulation, rounded once), all layer outputs produced")


if __name__ == "__main__":
    main()
