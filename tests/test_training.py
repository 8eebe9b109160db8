"""Training-mode path: the HIP training-step kernels behind GSNLayerTrainFn and the differentiable forward of
the live module, against fixtures made by the REFERENCE in .train() mode with loss.backward() (tests/golden/make_golden.py:
gsn_train_cells.npz = StackedGSU alone, live_tiny_train.npz = the whole tiny model)."""
import os

import numpy as np
import pytest
import torch

import refweights as rw

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _close(a, b, name, rtol=2e-3, atol_frac=2e-4):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    tol = atol_frac * max(1e-12, np.abs(b).max()) + rtol * np.abs(b)
    bad = np.abs(a - b) > tol
    assert not bad.any(), f"{name}: {int(bad.sum())} of {a.size} outside tolerance, max abs err {np.abs(a - b).max():.3g} (scale {np.abs(b).max():.3g})"


@pytest.mark.parametrize("ci", range(4))
def test_gsn_stack_training_forward_and_backward_match_the_reference(ci):
    """StackedGSU in training mode: spike trains of every layer, BatchNorm buffers after the forward (running statistics updated once
    per time step), and the gradients of the input and of every parameter for a fixed cotangent on the last layer's spikes."""
    import spiking_fullsubnet_amd.modeling_spiking_fullsubnet as M
    from spiking_fullsubnet_amd import training
    g = np.load(os.path.join(GOLD, "gsn_train_cells.npz"))
    name = str(g["cases"][ci])
    I, H, L, R, T, shared, bn = [int(v) for v in g["dims"][ci]]
    stack = M.StackedGSU(I, H, L, bool(shared), bool(bn))
    sd = {k[len(name) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(name + "/sd/")}
    stack.load_state_dict(sd, strict=True)
    stack = stack.to(DEV).train()
    x = _t(g[f"{name}/x"]).requires_grad_(True)
    outs = training.gsn_stack(x, stack, training=True)
    for l in range(L):
        ref = g[f"{name}/spikes/{l}"]
        got = outs[l + 1].detach().cpu().numpy()
        assert got.shape == ref.shape
        assert (got == ref).all(), f"{name} layer {l}: {(got != ref).sum()} spikes differ from the reference's training-mode forward"
    (outs[-1] * _t(g[f"{name}/gy"])).sum().backward()
    _close(x.grad.cpu().numpy(), g[f"{name}/grad/x"], f"{name}: dL/dx")
    for k, p in stack.named_parameters():
        _close(p.grad.cpu().numpy(), g[f"{name}/grad/{k}"], f"{name}: grad {k}")
    for k, b in stack.named_buffers():
        ref = g[f"{name}/buf/{k}"]
        if k.endswith("num_batches_tracked"):
            assert int(b) == int(ref), (k, int(b), int(ref))
        else:
            _close(b.cpu().numpy(), ref, f"{name}: buffer {k}", rtol=1e-4, atol_frac=1e-5)


def test_live_module_training_step_matches_the_reference():
    """The whole tiny live model in .train() mode on a waveform: forward outputs, the loss of the fixture (mean square of the enhanced
    waveform + mean enhanced magnitude), every parameter's gradient, BatchNorm buffers -- against the reference's own training step."""
    import spiking_fullsubnet_amd as pkg
    g = np.load(os.path.join(GOLD, "live_tiny_train.npz"))
    kw = rw.LIVE_TINY
    sd = rw.live_state_dict(kw, int(g["weight_seed"]))
    m = pkg.SpikingFullSubNet(**kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to(DEV).train()
    outs = m(_t(g["wave"]))
    assert len(outs) == 4
    enh_y, enh_mag = outs[0], outs[1]
    for i, a in enumerate(outs[2]):
        ref = g[f"fb_all/{i}"]
        if 0 < i < len(outs[2]) - 1:
            assert (a.detach().cpu().numpy() == ref).all(), f"full-band spike tensor {i} differs"
        else:
            _close(a.detach().cpu().numpy(), ref, f"fb_all[{i}]", rtol=1e-4, atol_frac=1e-5)
    for gi, lst in enumerate(outs[3]):
        for i, a in enumerate(lst):
            ref = g[f"sb_all/{gi}/{i}"]
            if 0 < i < len(lst) - 1:
                assert (a.detach().cpu().numpy() == ref).all(), f"sub-band {gi} spike tensor {i} differs"
            else:
                _close(a.detach().cpu().numpy(), ref, f"sb_all[{gi}][{i}]", rtol=1e-4, atol_frac=1e-5)
    _close(enh_mag.detach().cpu().numpy(), g["enh_mag"], "enh_mag", rtol=1e-4, atol_frac=1e-5)
    _close(enh_y.detach().cpu().numpy(), g["enh_y"], "enh_y", rtol=1e-3, atol_frac=1e-4)
    loss = enh_y.pow(2).mean() + enh_mag.mean()
    assert float(loss) == pytest.approx(float(g["loss"]), rel=1e-5)
    loss.backward()
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        _close(p.grad.cpu().numpy(), g[f"grad/{k}"], f"grad {k}")
    for k, b in m.named_buffers():
        if f"buf/{k}" not in g.files:
            continue
        ref = g[f"buf/{k}"]
        if k.endswith("num_batches_tracked"):
            assert int(b) == int(ref), k
        else:
            _close(b.cpu().numpy(), ref, f"buffer {k}", rtol=1e-4, atol_frac=1e-5)
    # the module still serves inference through the kernels after .eval(), with the statistics the training step has just updated
    m.eval()
    with torch.no_grad():
        y = m(_t(g["wave"]))
    assert y[0].shape == enh_y.shape and torch.isfinite(y[0]).all()


def test_lstm_and_output_activation_options_run_on_the_aten_path():
    """sequence_model="LSTM" (the reference's nn.LSTM ablation, modeling_spiking_fullsubnet.py:38-45,68-79) and an output activation:
    constructor options the inference kernels do not cover are served by the differentiable path, in eval mode too."""
    import spiking_fullsubnet_amd as pkg
    kw = dict(rw.LIVE_TINY, sequence_model="LSTM", fb_output_activate_function="relu")
    torch.manual_seed(0)
    m = pkg.SpikingFullSubNet(**kw).to(DEV).eval()
    wave = _t(rw.synth_wave(2, 16, 5))
    with torch.no_grad():
        outs = m(wave)
    assert outs[0].shape == wave.shape and outs[1].shape[1] == kw["n_fft"] // 2 + 1 and outs[2] == [] and outs[3] == [[], [], []][:len(outs[3])]
    m.train()
    outs = m(wave)
    outs[0].pow(2).mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_inference_entry_points_refuse_training_mode():
    import spiking_fullsubnet_amd as pkg
    kw = rw.LIVE_TINY
    m = pkg.SpikingFullSubNet(**kw).to(DEV).train()
    stft = torch.zeros((1, kw["n_fft"] // 2 + 1, 8), dtype=torch.complex64, device=DEV)
    with pytest.raises(RuntimeError):
        m.forward_stft(stft)
    with pytest.raises(RuntimeError):
        m.streaming(batch=1)
