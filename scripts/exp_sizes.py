"""Single-stream forward time of every reference model size at B=64, T=1000 (seeded random weights)."""
import sys, os, time, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import refweights as rw
import spiking_fullsubnet_amd as pkg
dev = torch.device("cuda")
wave = torch.from_numpy(rw.synth_wave(64, 1000, 0)).to(dev)
for name, front, kw in (("baseline_s (frozen)", "frozen", rw.FROZEN_S), ("baseline_m (frozen)", "frozen", rw.FROZEN_M), ("baseline_m (live)", "live", rw.LIVE_M),
                        ("baseline_l (frozen)", "frozen", rw.FROZEN_L), ("baseline_xl (frozen)", "frozen", rw.FROZEN_XL)):
    sd = rw.live_state_dict(kw, 1) if front == "live" else rw.frozen_state_dict(kw, 1)
    m = (pkg.SpikingFullSubNet if front == "live" else pkg.Separator)(**kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.eval().to(dev)
    stft = m._stft(wave)
    for _ in range(2): m.forward_stft(stft)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m.forward_stft(stft)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("%-22s %7.2f ms per forward  %6.2f M frames/s" % (name, dt * 1e3, 64000 / dt / 1e6))
