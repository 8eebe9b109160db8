import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
DEV="cuda:0"
kw=rw.LIVE_M; sd=rw.live_state_dict(kw,21)
m=pkg.SpikingFullSubNet(**kw); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k,v in sd.items()}, strict=True); m=m.eval().to(DEV)
stft=m._stft(torch.from_numpy(rw.synth_wave(64,1000,3)).to(DEV))
eng=m.engine(); eng.stack_scan=False
for fuse in (False, True):
  for rp in ((4,8),(4,16)):
    eng.fuse_input=fuse; eng.rows_per_wg=rp
    for _ in range(2): eng.forward_stft(stft)
    eng.timers, eng.timer_tags = {}, None
    for _ in range(3): eng.forward_stft(stft)
    torch.cuda.synchronize(); eng_t=eng.timers; eng.timers=None
    ev=eng_t['scan:sb']; print(fuse, rp, 'scan:sb launches', [round(a.elapsed_time(b),3) for a,b in ev][-4:])
