"""Round 4 soak: the product paths that wait inside a launch, many times over -- (1) the strict forward (pair launch + full-band stack on
two streams, B = 64, T = 1000), error words checked every 100 forwards; (2) a whole training step (one-launch layer calls, the
sub-band groups in one grid, B = 16, T = 250).  Prints a line per 100 iterations; any hang shows as a missing line (run under timeout)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
from spiking_fullsubnet_amd import training
DEV = "cuda:0"
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 300
kw = rw.LIVE_M; sd = rw.live_state_dict(kw, 21)
m = pkg.SpikingFullSubNet(**kw); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True); m = m.eval().to(DEV)
stft = m._stft(torch.from_numpy(rw.synth_wave(64, 1000, 3)).to(DEV)); eng = m.engine()
ref = eng.forward_stft(stft); torch.cuda.synchronize()
t0 = time.time()
for i in range(NF):
    out = eng.forward_stft(stft)
    if i % 100 == 99:
        eng.check_stack_errors()
        same = torch.equal(torch.view_as_real(out["enh_stft"]), torch.view_as_real(ref["enh_stft"]))
        print(f"forward {i + 1}: {'bit-identical to the first' if same else 'DIFFERENT'}  {(time.time() - t0) / (i + 1) * 1e3:.3f} ms each", flush=True)
        assert same
m.train()
wave = torch.from_numpy(rw.synth_wave(16, 250, 5)).to(DEV)
t0 = time.time()
first = None
for i in range(NT):
    for p_ in m.parameters():
        p_.grad = None
    out = m(wave)
    loss = out[0].pow(2).mean() + out[1].mean()
    loss.backward()
    if i % 50 == 49:
        training.check_pending()
        torch.cuda.synchronize()
        gn = float(torch.sqrt(sum((p_.grad.float() ** 2).sum() for p_ in m.parameters() if p_.grad is not None)))
        print(f"training step {i + 1}: loss {float(loss):.6f} grad norm {gn:.6f}  {(time.time() - t0) / (i + 1) * 1e3:.1f} ms each", flush=True)
        assert np.isfinite(gn)
print("soak done", flush=True)
