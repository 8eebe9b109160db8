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

Prints ONE JSON line on rank 0 (see the task contract), carrying
  roofline      -- the dominant kernel (the fused sub-band GSN scan): algorithmic bytes per launch / HIP-event
                   measured launch duration, against the 8 TB/s HBM3E peak;
  cpu_baseline  -- the CPU oracle (oracle/, a C restatement of the reference, OpenMP over rows) timed on this
                   host on a bounded sample of the same workload (rank 0, N=1 only).
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

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# API-faithful algorithmic bytes per clip-frame of the sub-band scan, baseline_m sizes:
#   read 256 (noisy_mag) + 64 (fb_out) floats, write 1,152 coefficients and both layers' fp32 spikes
#   (13 rows x 2 x 224) = 29,184 B.  The scan kernel is launched once per layer, so one launch is charged half.
SB_SCAN_BYTES_PER_FRAME_PER_LAUNCH = 29184 / 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-layer-outputs", action="store_true", help="skip the fp32 spike tensors of the module API (reported in config)")
    ap.add_argument("--sequential", action="store_true", help="force the sequential single-stream schedule")
    ap.add_argument("--pipeline", action="store_true", help="force the time-pipelined multi-stream schedule")
    ap.add_argument("--time-all", action="store_true", help="HIP-event time every launch group, not only the dominant kernel")
    ap.add_argument("--seq-chunk", type=int, default=0, help="frames per chunk of the single-stream schedule")
    ap.add_argument("--rpw", type=str, default="", help="rows per scan workgroup 'fb,sb' (0 = auto)")
    ap.add_argument("--inflight", type=int, default=12, help="forwards in flight on separate HIP streams (batch-level pipelining)")
    ap.add_argument("--chunk", type=int, default=0, help="frames per pipeline chunk (default: engine default)")
    ap.add_argument("--no-phase-a", action="store_true", help="skip the single-stream phase (no roofline object): profiling runs of the timed region alone")
    ap.add_argument("--no-saturated", action="store_true", help="skip the 4x-rows launch of the dominant kernel (roofline.saturated)")
    ap.add_argument("--streaming", action="store_true", help="BASELINE configs[4]: frame-by-frame session, per-call latency (own JSON line)")
    ap.add_argument("--hop", type=int, default=1, help="frames per streaming call")
    ap.add_argument("--no-graph", action="store_true", help="streaming: launch the kernels one by one instead of replaying the HIP graph")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    # one rank per GPU.  (SFSN_BENCH_BACKEND=gloo lets several ranks share one GPU: a plumbing check of the multi-rank
    # code path on a single-GPU box, not a measurement.)
    backend = os.environ.get("SFSN_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    import refweights as rw
    import spiking_fullsubnet_amd as pkg

    kw = rw.LIVE_M
    B, T = args.batch, args.frames
    sd = rw.live_state_dict(kw, 21)
    model = pkg.SpikingFullSubNet(**kw)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model = model.eval().to(dev)
    wave = torch.from_numpy(rw.synth_wave(B, T, seed=rank)).to(dev)
    stft = model.stft(wave).contiguous()  # untimed: the STFT is the edge of the path
    assert stft.shape == (B, 257, T)
    eng = model.engine()
    if args.streaming:
        return streaming_bench(args, model, dev, world, rank)
    if args.chunk:
        eng.pipeline_chunk = args.chunk
    if args.seq_chunk:
        eng.seq_chunk = args.seq_chunk
    want_layers = not args.no_layer_outputs
    gathered = {}  # per HIP stream (lane): the all-gathered magnitudes of that lane's batch

    info = {}

    def forward():
        res = eng.forward_stft(stft, want_layers=want_layers, pipeline=False if args.sequential else (True if args.pipeline else None))
        info.update(pipelined=res["pipelined"], n_chunks=res["n_chunks"])
        if world > 1:
            key = torch.cuda.current_stream(dev).cuda_stream
            if key not in gathered:
                gathered[key] = torch.empty((world * B, 1, 257, T), dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(gathered[key], res["enh_mag"])
        return res

    def timed_region(step_fn, steps, warmup):
        for _ in range(warmup):
            step_fn()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
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

    # ---- phase A (untimed for `value`): one forward at a time on one stream, the latency-optimal launch geometry.  The
    #      dominant kernel runs alone here, so its HIP-event duration is the kernel's own (roofline), not a time share.
    eng.rows_per_wg = (0, 0)
    eng.timers, eng.timer_tags = {}, (None if args.time_all else {"scan:sb", "scan:fb"})
    ka = max(2, min(args.steps, 8))
    if args.no_phase_a:
        scan_ms, single = {}, None
        args.no_saturated = True
    else:
        dt_a = timed_region(forward, ka, min(args.warmup, 2) + 1)
        scan_ms = eng.timer_summary()
        single = dict(ms_per_step=round(1e3 * dt_a / ka, 4), value=round(world * B * T * ka / dt_a, 1), steps=ka)
    eng.timers = None

    # ---- the same kernel with the chip full (untimed for `value`): at B=64 a sub-band scan launch is 208 workgroups of 4 rows,
    #      a latency-bound chain per workgroup; four times the rows (16 per workgroup, same 208 workgroups) shows what the
    #      kernel moves per second when every CU has a full tile -- which is also how it runs in the timed region below,
    #      where the scans of several forwards share the chip
    saturated = None
    if world == 1 and not args.no_saturated and B * 4 * T <= 256 * 1000:
        stft4 = stft.repeat(4, 1, 1)
        eng.timers, eng.timer_tags = {}, {"scan:sb"}
        for _ in range(3):
            eng.forward_stft(stft4, want_layers=want_layers, pipeline=False)
        sat = eng.timer_summary().get("scan:sb")
        eng.timers = None
        eng._ws.clear()
        del stft4
        torch.cuda.empty_cache()
        if sat:
            ach = SB_SCAN_BYTES_PER_FRAME_PER_LAUNCH * 4 * B * T / (sat["min_ms"] * 1e-3) / 1e9
            saturated = dict(clips=4 * B, launch_ms=round(sat["min_ms"], 4), achieved=round(ach, 1), unit="GB/s", frac=round(ach / HBM_PEAK_GBPS, 4),
                             note="same kernel, 4x the rows in one launch (16 rows per workgroup): not the bench workload, shown to separate "
                                  "the kernel's efficiency from the occupancy of a B=64 launch")

    # ---- phase B (THE timed region): `--inflight` forwards in flight on as many HIP streams -- batch-level pipelining of
    #      independent batches, as a serving loop runs them.  The recurrent scans are latency-bound chains that occupy a
    #      fraction of the CUs (16 rows per workgroup here, so that several scans fit side by side); the next batches'
    #      scans and time-parallel kernels fill the rest of the chip.  Every step is one complete pass over one batch.
    n_lanes = 1
    if args.inflight > 1:
        eng.rows_per_wg = tuple(int(v) for v in args.rpw.split(",")) if args.rpw else (4, 16)
        n_lanes = max(1, min(args.inflight, args.steps // 2))  # a short run cannot amortise the fill / drain of many lanes
        lanes = [torch.cuda.Stream(device=dev) for _ in range(n_lanes)]
        counter = [0]
        for s_ in lanes:  # untimed: first use of a lane allocates its scratch buffers and warms its memory pool
            with torch.cuda.stream(s_):
                forward()
        torch.cuda.synchronize()

        def step():
            s_ = lanes[counter[0] % len(lanes)]
            counter[0] += 1
            with torch.cuda.stream(s_):
                return forward()
    else:
        if args.rpw:
            eng.rows_per_wg = tuple(int(v) for v in args.rpw.split(","))
        step = forward
    dt = timed_region(step, args.steps, args.warmup)

    if rank == 0:
        frames = world * B * T * args.steps
        ms_per_step = 1e3 * dt / args.steps
        sb_ms = scan_ms.get("scan:sb")
        roofline = None
        if sb_ms:
            # one launch of the scan kernel covers T / n_chunks frames of every clip (time-pipelined schedule)
            frames_per_launch = B * T / info["n_chunks"]
            bytes_per_launch = SB_SCAN_BYTES_PER_FRAME_PER_LAUNCH * frames_per_launch
            achieved = bytes_per_launch / (sb_ms["mean_ms"] * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj.get("B") == B and tj.get("T") == T:  # measured on one whole-sequence launch; scale to this launch's frames
                    traffic = int(tj.get("sb_scan_hbm_bytes_per_launch") / info["n_chunks"])
            roofline = dict(bound="hbm", kernel="gsn_scan_kernel<G=1,KS=4,NW=16,TPW=1,OUT=fp32+int8 spikes,4-row repacked epilogue> (3 sub-band groups in one launch, one launch per layer)",
                            achieved=round(achieved, 1), peak=HBM_PEAK_GBPS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBPS, 4),
                            traffic=traffic, launch_ms=round(sb_ms["mean_ms"], 4), launches=sb_ms["n"],
                            algorithmic_bytes_per_launch=int(bytes_per_launch), measured_copy_GBps=round(copy_gbps, 1),
                            per_step_us=round(1e3 * sb_ms["mean_ms"] / (T / info["n_chunks"]), 3),
                            frames_per_launch=int(frames_per_launch), schedule=("time-pipelined x%d chunks on %d streams" % (info["n_chunks"], 4)) if info["pipelined"] else "sequential",
                            measured_in="phase A: single stream, one forward at a time (the kernel runs alone; in the timed region "
                                        "several forwards share the chip and a launch's wall time is a time share, not the kernel's own)",
                            other_kernels_ms={k: round(v["mean_ms"], 4) for k, v in scan_ms.items() if k not in ("scan:sb", "scan:fb")})
            fb_ms = scan_ms.get("scan:fb")
            if fb_ms:
                # the other recurrent kernel: one launch per full-band layer, B rows x H=320 on B/4 CUs -- a pure dependency
                # chain (its algorithmic traffic is ~1 % of the sub-band scan's): reported as time per step
                roofline["full_band_scan"] = dict(kernel="gsn_scan_kernel<G=1,KS=5,NW=8,TPW=3,LP=1> (W_hh: two digit planes in registers, one in LDS)",
                                                  launch_ms=round(fb_ms["mean_ms"], 4), per_step_us=round(1e3 * fb_ms["mean_ms"] / (T / info["n_chunks"]), 3),
                                                  workgroups=(B + 3) // 4)
            if traffic is not None and tj.get("forward_hbm_bytes") and want_layers:
                # the whole job against the same roof: PMC-measured HBM bytes of one forward (all kernels) / time per step of
                # the timed region (several forwards in flight)
                jb = float(tj["forward_hbm_bytes"])
                if saturated:
                    roofline["saturated"] = saturated
                roofline["job"] = dict(hbm_bytes_per_step=int(jb), achieved=round(jb / (ms_per_step * 1e-3) / 1e9, 1), unit="GB/s",
                                       frac=round(jb / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4))
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
                                in_flight=(n_lanes if args.inflight > 1 else 1), scan_rows_per_workgroup=list(eng.rows_per_wg),
                                single_stream=single,
                                parallelism=f"clip-sharded x{world}" + (" + RCCL all_gather(enh_mag)" if world > 1 else "")),
                    roofline=roofline, cpu_baseline=cpu)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def streaming_bench(args, model, dev, world, rank):
    """BASELINE.json configs[4]: B clips per GPU (default run: --batch 1), ``hop`` frames per call, state carried on the device;
    per-call latency = host wall time from handing over the frame(s) to the enhanced frame(s) being complete (synchronised)."""
    B = args.batch if args.batch != 64 else 1
    hop, steps, warmup = args.hop, max(args.steps, 2000), max(args.warmup, 200)
    rpw = tuple(int(v) for v in args.rpw.split(",")) if args.rpw else None
    sess = model.streaming(batch=B, hop=hop, graph=not args.no_graph, rows_per_wg=rpw)
    g = torch.Generator(device="cpu").manual_seed(3)
    frames = (0.05 * torch.randn((steps + warmup, B, 257, hop, 2), generator=g)).to(dev)
    frames = torch.view_as_complex(frames)
    lat = []
    for i in range(steps + warmup):
        x = frames[i]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sess.step(x, copy=False)
        torch.cuda.synchronize()
        if i >= warmup:
            lat.append(time.perf_counter() - t0)
    lat = np.sort(np.asarray(lat)) * 1e6
    # throughput of back-to-back calls without a host sync per call (the graph replays queue up)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        sess.step(frames[warmup + i], copy=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({
            "metric": "streaming per-call latency p50 (BASELINE configs[4]: state carried on the device, hop frames per call)",
            "value": round(float(lat[len(lat) // 2]), 1), "unit": "us", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(float(lat.mean()) / 1e3, 4), "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (0.05*randn complex frames; seeded random weights, randomised BN stats)",
            "config": {"workload": "configs[4]: streaming, live baseline_m sizes, fp32 parity mode", "clips_per_gpu": B, "hop_frames": hop,
                       "hip_graph": not args.no_graph, "p99_us": round(float(lat[int(len(lat) * 0.99)]), 1),
                       "min_us": round(float(lat[0]), 1), "unsynchronised_calls_per_s": round(steps / dt, 1),
                       "real_time_factor_at_8ms_hop": round(8e3 * hop / float(lat[len(lat) // 2]), 1)}}))


def cpu_baseline(kw, sd, stft):
    """The CPU oracle (C restatement of the reference, OpenMP over rows) on a bounded sample of the same workload."""
    from oracle import model as omodel
    spec = omodel.spec_from_live_kwargs(kw)
    Ts = 128
    sample = stft[:, :, :Ts].cpu().numpy()  # all 64 clips x 128 frames (the model is causal: a prefix is a valid workload)
    omodel.forward_from_stft(spec, sd, sample[:4], "f32")  # warm the OpenMP pool / page in
    n, t0 = 0, time.perf_counter()
    while True:
        omodel.forward_from_stft(spec, sd, sample, "f32")
        n += 1
        el = time.perf_counter() - t0
        if el > 12.0 or n >= 20:
            break
    frames = n * sample.shape[0] * Ts
    # one core, for calibration: a smaller sample of the same batch, OpenMP pinned to one thread
    single = cpu_model = None
    try:
        import ctypes
        gomp = ctypes.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(1)
        small = sample[:2, :, :32]
        t1 = time.perf_counter()
        omodel.forward_from_stft(spec, sd, small, "f32")
        single = round(small.shape[0] * small.shape[2] / (time.perf_counter() - t1), 1)
        gomp.omp_set_num_threads(os.cpu_count())
        with open("/proc/cpuinfo") as fh:
            cpu_model = next((l.split(":", 1)[1].strip() for l in fh if l.startswith("model name")), None)
    except Exception:  # the baseline is reported, never required
        pass
    return dict(value=round(frames / el, 1), unit="frames/s", cores=os.cpu_count(), kind="port", single_core_value=single, cpu_model=cpu_model,
                sample=f"{n} x (B={sample.shape[0]}, T={Ts} prefix of the same synthetic batch), {el:.1f} s of wall time, fp32 oracle "
                       f"(oracle/sfsn_oracle.c via oracle.model), OpenMP threads = all {os.cpu_count()} host cores",
                reference_pytorch_cpu="3,265 frames/s for the reference's own PyTorch forward at B=64,T=1000 on 8 vCPU (BASELINE.md section 2, survey container)")


if __name__ == "__main__":
    main()
