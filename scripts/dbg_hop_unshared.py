"""Debugging aid: the one-launch hop with separate gate weights against the offline forward, with the cell gate's weights made equal to
the forget gate's (layer by layer, weight by weight) to localise a mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
DEV = "cuda:0"
kw = rw.LIVE_TINY_UNSHARED
base = rw.live_state_dict(kw, 7)

def run(sd, tag):
    m = pkg.SpikingFullSubNet(**kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.eval().to(DEV)
    T = 12
    wave = torch.from_numpy(rw.synth_wave(1, T, 7)).to(DEV)
    stft = torch.stft(wave, kw["n_fft"], kw["hop_length"], kw["win_length"], window=torch.hann_window(kw["win_length"], device=DEV),
                      return_complex=True, pad_mode="constant")[..., :T].contiguous()
    off = m.engine().forward_stft(stft, want_layers=False)
    sess = m.streaming(batch=1, hop=1)
    outs = [sess.step(stft[..., t:t + 1].contiguous())[0] for t in range(T)]
    sess.check_errors()
    e = torch.cat(outs, -1)
    d = (torch.view_as_real(e) - torch.view_as_real(off["enh_stft"])).abs()
    first = [int(t) for t in range(T) if float(d[..., t, :].max()) > 0]
    print(f"{tag}: one launch = {sess._hop is not None}; max abs diff {float(d.max()):.3e}; frames that differ {first[:6]}", flush=True)

run(base, "as is")
for which in ("weight_hh", "weight_ih", "bias_ih", "all"):
    sd = dict(base)
    for k, v in base.items():
        v = np.asarray(v)
        if (which in k or which == "all") and any(s_ in k for s_ in ("weight_hh", "weight_ih", "bias_ih")):
            H2 = v.shape[0]
            v = v.copy(); v[H2 // 2:] = v[:H2 // 2]
            sd[k] = v
    run(sd, f"cell gate := forget gate in {which}")
