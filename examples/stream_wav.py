#!/usr/bin/env python3
"""Real-time style enhancement of a 16 kHz mono WAV file, 8 ms at a time, on one MI355X.

    python examples/stream_wav.py noisy.wav enhanced.wav [--ckpt path/to/checkpoints/best] [--config m|s]

Every 128-sample hop goes through ONE launch (sfsn_stream_hop in waveform mode: STFT of the new frame, the whole
Spiking-FullSubNet, inverse STFT with its overlap-add state); samples are read from and written to pinned host memory by the
launch itself.  The output lags the input by 384 samples (24 ms: the look-ahead of the reference's centred 32 ms analysis plus the
overlap-add) -- the script drops that lead-in and flushes the tail with zeros, so the file lengths match.
Without --ckpt the weights are the reference's random initialisation (useful as a latency demo only).
"""
import argparse
import os
import sys
import time
import wave

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spiking_fullsubnet_amd as pkg  # noqa: E402

BASELINE_M = dict(  # recipes/intel_ndns/spiking_fullsubnet/baseline_m.toml [model.args]
    n_fft=512, hop_length=128, win_length=512, fdrc=0.5, fb_input_size=64, fb_hidden_size=320, fb_num_layers=2, fb_proj_size=64,
    fb_output_activate_function=False, sb_hidden_size=224, sb_num_layers=2, freq_cutoffs=[0, 32, 128, 256], df_orders=[5, 3, 1],
    center_freq_sizes=[4, 32, 64], neighbor_freq_sizes=[15, 15, 15], use_pre_layer_norm_fb=True, use_pre_layer_norm_sb=True, bn=True,
    shared_weights=True, sequence_model="GSN", num_spks=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("inp")
    ap.add_argument("out")
    ap.add_argument("--ckpt", default=None, help="an Accelerate checkpoint directory of the live recipe (pytorch_model.bin / model.safetensors)")
    args = ap.parse_args()
    with wave.open(args.inp, "rb") as w:
        assert w.getnchannels() == 1 and w.getsampwidth() == 2 and w.getframerate() == 16000, "16 kHz mono 16-bit PCM expected"
        x = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
    model = pkg.SpikingFullSubNet(**BASELINE_M)
    if args.ckpt:
        from spiking_fullsubnet_amd.checkpoint import load_checkpoint
        load_checkpoint(model, args.ckpt)
    model = model.to("cuda").eval()
    sess = model.streaming(batch=1, waveform=True, host_io=True)
    n_hops = -(-len(x) // 128) + 3  # + the 3 hops of algorithmic delay
    xp = np.zeros(n_hops * 128, np.float32)
    xp[:len(x)] = x
    y = np.zeros(n_hops * 128, np.float32)
    lat = []
    for c in range(n_hops):
        t0 = time.perf_counter()
        o = sess.step_wave_host(torch.from_numpy(xp[128 * c:128 * (c + 1)]).reshape(1, 128))
        lat.append(time.perf_counter() - t0)
        y[128 * c:128 * (c + 1)] = o[0, 0].numpy()
    sess.check_errors()
    y = y[384:384 + len(x)]  # drop the lead-in
    with wave.open(args.out, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes((np.clip(y, -1.0, 1.0) * 32767.0).astype(np.int16).tobytes())
    lat = np.sort(np.asarray(lat[10:])) * 1e6
    print(f"{len(x) / 16000:.2f} s of audio, {n_hops} hops: per-hop latency p50 {lat[len(lat) // 2]:.1f} us, p99 {lat[int(len(lat) * 0.99)]:.1f} us "
          f"({8000.0 / lat[len(lat) // 2]:.0f} x real time)")


if __name__ == "__main__":
    main()
