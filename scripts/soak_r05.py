"""Round-5 soak: the split scan (baseline_xl forwards) and the pipelined training step, repeated; every error word checked.
usage: python scripts/soak_r05.py [xl forwards] [training steps]"""
import sys, os, time, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import refweights as rw
import spiking_fullsubnet_amd as pkg
from spiking_fullsubnet_amd import training
dev = torch.device("cuda")
n_fwd = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_tr = int(sys.argv[2]) if len(sys.argv) > 2 else 60
wave = torch.from_numpy(rw.synth_wave(64, 1000, 0)).to(dev)
kw = rw.FROZEN_XL
m = pkg.Separator(**kw); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in rw.frozen_state_dict(kw, 1).items()}); m = m.eval().to(dev)
stft = m._stft(wave); eng = m.engine()
ref = eng.forward_stft(stft); torch.cuda.synchronize(); eng.check_stack_errors()
t0 = time.perf_counter(); bad = 0
for i in range(n_fwd):
    out = eng.forward_stft(stft)
    if i % 25 == 24:
        eng.check_stack_errors()
        bad += int(not torch.equal(torch.view_as_real(ref["enh_stft"]), torch.view_as_real(out["enh_stft"])))
torch.cuda.synchronize(); eng.check_stack_errors()
print(f"baseline_xl: {n_fwd} forwards in {time.perf_counter() - t0:.1f} s, split-scan launches {eng.launches.get('split_scan')}, mismatching checks {bad}", flush=True)
del m, eng, ref, out
kw = rw.LIVE_M
m = pkg.SpikingFullSubNet(**kw); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in rw.live_state_dict(kw, 21).items()}); m = m.to(dev).train()
losses = []
t0 = time.perf_counter()
for i in range(n_tr):
    for p_ in m.parameters(): p_.grad = None
    o = m(wave); loss = o[0].pow(2).mean() + o[1].mean(); loss.backward()
    training.check_pending()
    losses.append(float(loss))
torch.cuda.synchronize()
fin = all(bool(torch.isfinite(p_.grad).all()) for p_ in m.parameters() if p_.grad is not None)
print(f"training: {n_tr} steps in {time.perf_counter() - t0:.1f} s, stacks through GSNStackTrainFn {training._STACK_CALLS}, finite gradients {fin}, loss {losses[0]:.5f} .. {losses[-1]:.5f}", flush=True)
