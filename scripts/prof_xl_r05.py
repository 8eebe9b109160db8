import sys, os, time, numpy as np, torch
R = "/root/repo"
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import refweights as rw
import spiking_fullsubnet_amd as pkg
dev = torch.device("cuda")
wave = torch.from_numpy(rw.synth_wave(64, 1000, 0)).to(dev)
kw = rw.FROZEN_XL
sd = rw.frozen_state_dict(kw, 1)
m = pkg.Separator(**kw); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.eval().to(dev)
stft = m._stft(wave)
for _ in range(4): m.forward_stft(stft)
torch.cuda.synchronize()
print(m.engine().launches)
