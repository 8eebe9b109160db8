#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Run in the build container only (``/root/reference`` is mounted there and nowhere else):

    python tests/golden/make_golden.py

The reference is pure Python; it is imported from /root/reference with empty stub modules for its
module-level third-party imports that are absent here (librosa, soundfile, onnxruntime, pesq, pystoi
-- none is touched on the model path).  Nothing of the reference's source is copied: the
fixtures hold inputs, weights (synthetic, or the MIT-licensed baseline_s checkpoint re-serialised)
and the reference's outputs.  Per-step membranes are captured with forward hooks on the reference's
own ``GSUCell`` modules, the enhanced spectrum by wrapping the module's ``istft`` attribute.

Fixture metadata records the torch version that served as the oracle's oracle.
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))  # tests/
REF = "/root/reference"
FROZEN_DIR = f"{REF}/recipes/intel_ndns/spiking_fullsubnet_freeze_phase"


def import_reference():
    for name in ("librosa", "soundfile", "onnxruntime", "pesq", "pystoi"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            sys.modules[name] = m
    sys.modules["pesq"].pesq = None
    sys.modules["pystoi"].stoi = None
    sys.path.insert(0, REF)
    sys.path.insert(0, FROZEN_DIR)
    import torch  # noqa
    from audiozen.models.spiking_fullsubnet import efficient_spiking_neuron as neuron
    from audiozen.models.spiking_fullsubnet import modeling_spiking_fullsubnet as live
    import model_low_freq as frozen
    from audiozen import metric
    return neuron, live, frozen, metric


def pack(spk: np.ndarray) -> np.ndarray:
    return np.packbits(spk.astype(np.uint8).reshape(-1))


def to_torch_sd(sd):
    import torch
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


class CellTap:
    """Forward hooks on every GSUCell: records the post-BN membrane cy of every step, per cell."""

    def __init__(self, model, neuron):
        self.mem = {}
        self.handles = []
        for name, mod in model.named_modules():
            if isinstance(mod, neuron.GSUCell) or type(mod).__name__ == "GSUCell":
                self.mem[name] = []
                self.handles.append(mod.register_forward_hook(self._mk(name)))

    def _mk(self, name):
        def hook(_m, _inp, out):
            self.mem[name].append(out[1][1].detach().numpy().copy())
        return hook

    def stacked(self):
        return {k: np.stack(v) for k, v in self.mem.items()}

    def close(self):
        for h in self.handles:
            h.remove()


def gsn_cases(neuron, out):
    import torch
    cases = [  # name, I, H, L, R, T, shared, bn, nonzero_state
        ("tiny_shared_bn", 12, 32, 2, 5, 40, True, True, False),
        ("tiny_unshared_nobn", 9, 16, 2, 4, 30, False, False, False),
        ("tiny_unshared_bn_state", 20, 48, 1, 3, 25, False, True, True),
        ("sb_m_shape", 38, 224, 2, 3, 24, True, True, False),
        ("fb_m_shape_state", 64, 320, 1, 2, 16, True, True, True),
    ]
    meta = []
    for ci, (name, I, H, L, R, T, shared, bn, st) in enumerate(cases):
        rng = np.random.default_rng(100 + ci)
        import refweights
        sd = {}
        for l in range(L):
            refweights._cell(rng, f"layers.{l}.cell.", I if l == 0 else H, H, shared, bn, sd)
        net = neuron.efficient_spiking_neuron(I, H, L, shared_weights=shared, bn=bn).eval()
        net.load_state_dict(to_torch_sd(sd), strict=True)
        x = rng.standard_normal((T, R, I)).astype(np.float32)
        if st:
            h0 = [(rng.random((R, H)) > 0.5).astype(np.float32) for _ in range(L)]
            c0 = [rng.standard_normal((R, H)).astype(np.float32) for _ in range(L)]
        else:
            h0 = [np.zeros((R, H), np.float32) for _ in range(L)]
            c0 = [np.zeros((R, H), np.float32) for _ in range(L)]
        tap = CellTap(net, neuron)
        with torch.no_grad():
            states = [neuron.MemoryState(torch.from_numpy(h0[l].copy()), torch.from_numpy(c0[l].copy())) for l in range(L)]
            y, out_states, all_out = net(torch.from_numpy(x), states)
        mems = tap.stacked()
        tap.close()
        out[f"{name}/x"] = x
        for k, v in sd.items():
            out[f"{name}/sd/{k}"] = v
        for l in range(L):
            out[f"{name}/h0/{l}"] = h0[l]
            out[f"{name}/c0/{l}"] = c0[l]
            out[f"{name}/spikes/{l}"] = all_out[l + 1].numpy()
            out[f"{name}/membrane/{l}"] = mems[f"layers.{l}.cell"]
            out[f"{name}/hT/{l}"] = out_states[l][0].numpy()
            out[f"{name}/cT/{l}"] = out_states[l][1].numpy()
        meta.append((name, I, H, L, R, T, int(shared), int(bn)))
    out["cases"] = np.array([m[0] for m in meta])
    out["dims"] = np.array([m[1:] for m in meta], dtype=np.int64)



def gsn_train_cases(neuron, out):
    """The reference's StackedGSU in TRAINING mode (per-time-step batch-statistics BatchNorm that updates the running statistics T
    times per forward, efficient_spiking_neuron.py:149-150) and the backward pass through the triangle surrogate (:94-101): a fixed
    random cotangent `gy` on the last layer's spike train gives dL/dy = gy; recorded are the spike trains, the final states, the
    BatchNorm buffers after the forward and the gradients of the input and of every parameter."""
    import torch
    import refweights
    cases = [  # name, I, H, L, R, T, shared, bn
        ("train_tiny_shared_bn", 12, 32, 2, 6, 10, True, True),
        ("train_tiny_unshared_bn", 9, 16, 2, 5, 8, False, True),
        ("train_tiny_shared_nobn", 10, 16, 1, 4, 9, True, False),
        ("train_sb_shape", 38, 224, 2, 16, 6, True, True),
        # recipe scale (round-3 review): the sub-band stack of group 0 at the recipe's batch of 64 (baseline_m.toml:72: 512 rows ->
        # 11 row blocks per neuron tile exchange their BatchNorm partial sums inside every step launch) and the full-band stack
        # (64 rows, H = 320: 4 row blocks).  Spike trains stored packed, the cotangent regenerated from its seed, no membranes.
        ("train_sb_recipe", 38, 224, 2, 512, 24, True, True),
        ("train_fb_recipe", 64, 320, 2, 64, 24, True, True),
    ]
    meta = []
    for ci, (name, I, H, L, R, T, shared, bn) in enumerate(cases):
        big = T * R * H > 200000
        rng = np.random.default_rng(300 + ci)
        sd = {}
        for l in range(L):
            refweights._cell(rng, f"layers.{l}.cell.", I if l == 0 else H, H, shared, bn, sd)
        net = neuron.efficient_spiking_neuron(I, H, L, shared_weights=shared, bn=bn).train()
        net.load_state_dict(to_torch_sd(sd), strict=True)
        x = torch.from_numpy(rng.standard_normal((T, R, I)).astype(np.float32)).requires_grad_(True)
        gy_np = (np.random.default_rng(7000 + ci).standard_normal((T, R, H)).astype(np.float32) if big
                 else rng.standard_normal((T, R, H)).astype(np.float32))
        gy = torch.from_numpy(gy_np)
        tap = CellTap(net, neuron)
        states = [neuron.MemoryState(torch.zeros(R, H), torch.zeros(R, H)) for _ in range(L)]
        y, out_states, all_out = net(x, states)
        (y * gy).sum().backward()
        mems = tap.stacked()
        tap.close()
        out[f"{name}/x"] = x.detach().numpy()
        if big:
            out[f"{name}/gy_seed"] = np.asarray(7000 + ci)
        else:
            out[f"{name}/gy"] = gy.numpy()
        for k, v in sd.items():
            out[f"{name}/sd/{k}"] = v
        for l in range(L):
            spk = all_out[l + 1].detach().numpy()
            if big:
                assert set(np.unique(spk)) <= {0.0, 1.0}
                out[f"{name}/spikes_packed/{l}"] = np.packbits(spk.astype(np.uint8).reshape(-1))
                out[f"{name}/spike_rate/{l}"] = np.asarray(float(spk.mean()))
                # where the reference's own (post-BatchNorm) membrane is within 1e-4 of the threshold: a first spike disagreement is
                # only acceptable there (two correct fp32 evaluations of the layer's input product differ in the last bits)
                out[f"{name}/near1e-4/{l}"] = np.packbits((np.abs(mems[f"layers.{l}.cell"]) < 1e-4).reshape(-1))
            else:
                out[f"{name}/spikes/{l}"] = spk
                out[f"{name}/membrane/{l}"] = mems[f"layers.{l}.cell"]
                out[f"{name}/hT/{l}"] = out_states[l][0].detach().numpy()
                out[f"{name}/cT/{l}"] = out_states[l][1].detach().numpy()
        out[f"{name}/grad/x"] = x.grad.numpy()
        for k, p in net.named_parameters():
            out[f"{name}/grad/{k}"] = p.grad.numpy()
        for k, b in net.named_buffers():
            out[f"{name}/buf/{k}"] = b.detach().numpy()
        meta.append((name, I, H, L, R, T, int(shared), int(bn)))
    out["cases"] = np.array([m[0] for m in meta])
    out["dims"] = np.array([m[1:] for m in meta], dtype=np.int64)


def run_model(model, neuron, wave, frozen_front):
    """Run a reference model on a waveform; return dict of path inputs/outputs."""
    import torch
    captured = {}
    tap = CellTap(model, neuron)
    if frozen_front:
        real_istft = torch.istft

        def tap_istft(x, *a, **k):
            captured["enh_stft"] = x.detach().numpy().copy()
            return real_istft(x, *a, **k)
        torch.istft, saved = tap_istft, real_istft
    else:
        inner = model.istft

        def tap_istft(x, *a, **k):
            captured["enh_stft"] = x.detach().numpy().copy()
            return inner(x, *a, **k)
        model.istft = tap_istft
    try:
        with torch.no_grad():
            outs = model(torch.from_numpy(wave))
            stft = torch.stft(torch.from_numpy(wave), 512, 128, 512, window=torch.hann_window(512), return_complex=True,
                              pad_mode="constant")
    finally:
        if frozen_front:
            torch.istft = saved
    mems = tap.stacked()
    tap.close()
    res = dict(wave=wave, stft=stft.numpy(), enh_stft=captured["enh_stft"], enh_y=outs[0].numpy())
    if len(outs) == 4:
        res["enh_mag"] = outs[1].numpy()
        fb_all, sb_all = outs[2], outs[3]
    else:
        fb_all, sb_all = outs[1], outs[2]
    res["fb_all"] = [a.numpy() for a in fb_all]
    res["sb_all"] = [[a.numpy() for a in lst] for lst in sb_all]
    res["mem"] = mems
    return res


def store_model_case(out, res, store_membranes: bool, near_tau=(1e-4, 1e-3)):
    for k in ("wave", "stft", "enh_stft", "enh_y"):
        out[k] = res[k]
    if "enh_mag" in res:
        out["enh_mag"] = res["enh_mag"]

    def put(prefix, lst, mem_prefix):
        out[f"{prefix}/x"] = lst[0]
        out[f"{prefix}/proj"] = lst[-1]
        for l, spk in enumerate(lst[1:-1]):
            out[f"{prefix}/spikes_packed/{l}"] = pack(spk)
            out[f"{prefix}/spikes_shape/{l}"] = np.array(spk.shape, dtype=np.int64)
            mem = res["mem"][f"{mem_prefix}sequence_model.layers.{l}.cell"]
            if store_membranes:
                out[f"{prefix}/membrane/{l}"] = mem
            for tau in near_tau:
                out[f"{prefix}/near{tau:g}/{l}"] = pack(np.abs(mem) < tau)

    put("fb", res["fb_all"], "fb_model.")
    for g, lst in enumerate(res["sb_all"]):
        put(f"sb{g}", lst, f"sb_model.sb_models.{g}.")
    out["n_groups"] = np.asarray(len(res["sb_all"]))


def main():
    import torch
    import refweights as rw
    neuron, live, frozen, metric = import_reference()
    torch.set_num_threads(4)
    meta = dict(torch_version=torch.__version__, numpy_version=np.__version__)

    only = set(sys.argv[1:])  # e.g. `make_golden.py frozen_m_zoo` regenerates that fixture alone
    if not only:
        out = {}
        gsn_cases(neuron, out)
        np.savez_compressed(os.path.join(HERE, "gsn_cells.npz"), **out, **{f"meta/{k}": np.asarray(v) for k, v in meta.items()})
        print("gsn_cells.npz", len(out))

    if not only or "gsn_train" in only:
        out = {}
        gsn_train_cases(neuron, out)
        np.savez_compressed(os.path.join(HERE, "gsn_train_cells.npz"), **out, **{f"meta/{k}": np.asarray(v) for k, v in meta.items()})
        print("gsn_train_cells.npz", len(out))

    def live_train_case(fname, kw, seed, B, T, wave_seed=0, pack=False):
        """A whole live model in TRAINING mode on a waveform: forward outputs, a scalar loss (mean square of the enhanced waveform
        + mean of the enhanced magnitude: both outputs of forward() carry gradient), every parameter's gradient, BatchNorm buffers
        after the step's forward (the recipe's training step: recipes/intel_ndns/spiking_fullsubnet/trainer.py:24-48)."""
        sd = rw.live_state_dict(kw, seed)
        model = live.SpikingFullSubNet(**kw).train()
        model.load_state_dict(to_torch_sd(sd), strict=True)
        wave = torch.from_numpy(rw.synth_wave(B, T, wave_seed))
        outs = model(wave)
        enh_y, enh_mag = outs[0], outs[1]
        loss = enh_y.pow(2).mean() + enh_mag.mean()
        loss.backward()
        out = dict(wave=wave.numpy(), enh_y=enh_y.detach().numpy(), enh_mag=enh_mag.detach().numpy(), loss=np.asarray(float(loss)))
        def put(key, a, is_spikes):
            a = a.detach().numpy()
            if pack and is_spikes:
                assert set(np.unique(a)) <= {0.0, 1.0}
                out[key + "/packed"] = np.packbits(a.astype(np.uint8).reshape(-1))
                out[key + "/shape"] = np.asarray(a.shape, dtype=np.int64)
            elif pack:
                out[key + "/head"] = a[:4].copy()  # (layer inputs / projections at this size: the first four frames only)
            else:
                out[key] = a
        for i, a in enumerate(outs[2]):
            put(f"fb_all/{i}", a, 0 < i < len(outs[2]) - 1)
        for g, lst in enumerate(outs[3]):
            for i, a in enumerate(lst):
                put(f"sb_all/{g}/{i}", a, 0 < i < len(lst) - 1)
        for k, p in model.named_parameters():
            out[f"grad/{k}"] = p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        for k, b in model.named_buffers():
            out[f"buf/{k}"] = b.detach().numpy()
        out["weight_seed"] = np.asarray(seed)
        np.savez_compressed(os.path.join(HERE, fname), **out, **{f"meta/{k}": np.asarray(v) for k, v in meta.items()})
        print(fname, float(loss))

    if not only or "live_tiny_train" in only:
        live_train_case("live_tiny_train.npz", rw.LIVE_TINY, 11, 3, 12)
    if not only or "live_m_train" in only:
        # one training step of the whole model at baseline_m sizes, B = 16, T = 32 (sub-band rows 128 / 48 / 32, H = 224; full band 16
        # rows, H = 320): spike trains packed, layer inputs / projections by their first frames, every parameter's gradient
        live_train_case("live_m_train.npz", rw.LIVE_M, 21, 16, 32, wave_seed=4, pack=True)

    def live_case(fname, kw, seed, B, T, store_mem, wave_seed=0, modulated=False):
        sd = rw.live_state_dict(kw, seed)
        model = live.SpikingFullSubNet(**kw).eval()
        model.load_state_dict(to_torch_sd(sd), strict=True)
        res = run_model(model, neuron, rw.synth_wave(B, T, wave_seed, modulated=modulated), frozen_front=False)
        out = {}
        store_model_case(out, res, store_mem)
        out["weight_seed"] = np.asarray(seed)
        if kw.get("num_spks", 1) == 1:
            import contextlib, io
            with contextlib.redirect_stdout(io.StringIO()):
                out["synops"] = np.asarray(metric.compute_synops([torch.from_numpy(a) for a in res["fb_all"]],
                                           [[torch.from_numpy(a) for a in l] for l in res["sb_all"]], kw["shared_weights"]))
                out["neuronops"] = np.asarray(metric.compute_neuronops([torch.from_numpy(a) for a in res["fb_all"]],
                                              [[torch.from_numpy(a) for a in l] for l in res["sb_all"]]))
        np.savez_compressed(os.path.join(HERE, fname), **out, **{f"meta/{k}": np.asarray(v) for k, v in meta.items()})
        print(fname, {k: v.shape for k, v in out.items() if k in ("stft", "enh_stft")})

    if not only:
        live_case("live_tiny.npz", rw.LIVE_TINY, 11, 2, 24, True)
        live_case("live_tiny_2spk.npz", rw.LIVE_TINY_2SPK, 12, 2, 20, True)
        live_case("live_tiny_unshared.npz", rw.LIVE_TINY_UNSHARED, 13, 1, 20, True)
        live_case("live_m.npz", rw.LIVE_M, 21, 1, 40, False)
    if not only or "live_m_am" in only:
        # the sizes bench.py runs (BASELINE configs[2]), two clips x 200 frames of the amplitude-modulated noise SURVEY 8d names
        # (0.5 (1 + sin 2 pi 3 Hz t)): the spike rates swing with the envelope
        live_case("live_m_am.npz", rw.LIVE_M, 21, 2, 200, False, wave_seed=4, modulated=True)

    def frozen_case(fname, kw, sd, B, T, store_mem, store_weights, module=None):
        model = (module or frozen).Separator(**kw).eval()
        model.load_state_dict(to_torch_sd(sd), strict=True)
        res = run_model(model, neuron, rw.synth_wave(B, T, 1), frozen_front=True)
        out = {}
        store_model_case(out, res, store_mem)
        if store_weights:
            for k, v in sd.items():
                out[f"sd/{k}"] = np.asarray(v)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            out["synops"] = np.asarray(metric.compute_synops([torch.from_numpy(a) for a in res["fb_all"]],
                                       [[torch.from_numpy(a) for a in l] for l in res["sb_all"]], kw["shared_weights"]))
            out["neuronops"] = np.asarray(metric.compute_neuronops([torch.from_numpy(a) for a in res["fb_all"]],
                                          [[torch.from_numpy(a) for a in l] for l in res["sb_all"]]))
        np.savez_compressed(os.path.join(HERE, fname), **out, **{f"meta/{k}": np.asarray(v) for k, v in meta.items()})
        print(fname)

    def zoo_weights(name):
        zoo = torch.load(f"{REF}/model_zoo/intel_ndns/spike_fsb/{name}/checkpoints/best/pytorch_model.bin", map_location="cpu")
        return {k: v.numpy() for k, v in zoo.items()}

    if not only:
        frozen_case("frozen_tiny.npz", rw.FROZEN_TINY, rw.frozen_state_dict(rw.FROZEN_TINY, 31), 2, 24, True, False)
        frozen_case("frozen_s_zoo.npz", rw.FROZEN_S, zoo_weights("baseline_s"), 1, 126, False, True)
    if not only or "frozen_tiny_train" in only:
        # the frozen Separator as an ordinary trainable module (model_low_freq.py:485-618): one training step in .train() mode
        sd = rw.frozen_state_dict(rw.FROZEN_TINY, 31)
        model = frozen.Separator(**rw.FROZEN_TINY).train()
        model.load_state_dict(to_torch_sd(sd), strict=True)
        wave = torch.from_numpy(rw.synth_wave(3, 14, 2))
        outs = model(wave)
        loss = outs[0].pow(2).mean() + outs[1].mean()
        loss.backward()
        out = dict(wave=wave.numpy(), enh_y=outs[0].detach().numpy(), enh_mag=outs[1].detach().numpy(), loss=np.asarray(float(loss.detach())))
        for i, a in enumerate(outs[2]):
            out[f"fb_all/{i}"] = a.detach().numpy()
        for g_, lst in enumerate(outs[3]):
            for i, a in enumerate(lst):
                out[f"sb_all/{g_}/{i}"] = a.detach().numpy()
        for k, p_ in model.named_parameters():
            out[f"grad/{k}"] = p_.grad.numpy() if p_.grad is not None else np.zeros(tuple(p_.shape), np.float32)
        for k, b in model.named_buffers():
            out[f"buf/{k}"] = b.detach().numpy()
        out["weight_seed"] = np.asarray(31)
        np.savez_compressed(os.path.join(HERE, "frozen_tiny_train.npz"), **out, **{f"meta/{k}": np.asarray(v) for k, v in meta.items()})
        print("frozen_tiny_train.npz", float(loss))
    if not only or "frozen_s_zoo_4s" in only:
        # BASELINE configs[0] as written: the trained baseline_s generator on ONE 4 s clip (T = 501 frames at 16 kHz / hop 128);
        # the weights are those of frozen_s_zoo.npz (not stored twice)
        frozen_case("frozen_s_zoo_4s.npz", rw.FROZEN_S, zoo_weights("baseline_s"), 1, 501, False, False)
    if not only or "frozen_cum" in only:
        # cumulative_laplace_norm (recipes/.../baseline_m_cumulative_laplace_norm.toml): model_low_freq.Separator raises on the 5-D
        # sub-band tensor (model_low_freq.py:172-202 unpacks four dimensions); the same class in model_low_freq_count_time.py
        # carries the form that accepts it (:182-204) -- that one makes the fixture (it prints stage timings: silenced)
        import importlib
        frozen_ct = importlib.import_module("model_low_freq_count_time")
        kw_c = dict(rw.FROZEN_TINY, norm_type="cumulative_laplace_norm")
        _print = print
        try:
            import builtins
            builtins.print = lambda *a, **k: None
            frozen_case("frozen_tiny_cum.npz", kw_c, rw.frozen_state_dict(rw.FROZEN_TINY, 35), 3, 40, True, False, module=frozen_ct)
            kw_m = dict(rw.FROZEN_M, norm_type="cumulative_laplace_norm")
            frozen_case("frozen_m_cum.npz", kw_m, rw.frozen_state_dict(rw.FROZEN_M, 36), 1, 48, False, False, module=frozen_ct)
        finally:
            builtins.print = _print
        print("frozen_tiny_cum.npz frozen_m_cum.npz")
    if not only or "frozen_gauss" in only:
        # offline_gaussian_norm (model_low_freq.py:205-218: (x - mean) / (std + eps) per clip, torch.std = unbiased): no recipe uses it
        frozen_case("frozen_tiny_gauss.npz", dict(rw.FROZEN_TINY, norm_type="offline_gaussian_norm"), rw.frozen_state_dict(rw.FROZEN_TINY, 37), 3, 40, True, False)
    if not only or "frozen_l" in only:
        # baseline_l sizes: four sub-band groups (16 + 24 + 2 + 1 units), sub-band hidden size 256
        frozen_case("frozen_l.npz", rw.FROZEN_L, rw.frozen_state_dict(rw.FROZEN_L, 33), 1, 24, False, False)
    if not only or "frozen_xl" in only:
        # baseline_xl sizes: separate forget / cell gate weights at full-band hidden size 320 (W_hh streamed from L2)
        frozen_case("frozen_xl.npz", rw.FROZEN_XL, rw.frozen_state_dict(rw.FROZEN_XL, 34), 2, 64, False, False)  # (round 6: two clips x 64 frames -- the split scan and the G = 2 IO-wave scan meet a reference fixture with more than one row)
    if not only or "frozen_m_zoo" in only:
        # the trained baseline_m generator (the sizes bench.py runs: full-band 320, sub-band 224, deep-filter orders 5/3/1)
        frozen_case("frozen_m_zoo.npz", rw.FROZEN_M, zoo_weights("baseline_m"), 1, 100, False, True)


if __name__ == "__main__":
    main()
