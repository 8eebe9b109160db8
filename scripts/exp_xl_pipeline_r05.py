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
eng = m.engine()
ref = eng.forward_stft(stft); torch.cuda.synchronize()
for pipe, chunk in ((False, 0), (True, 250), (True, 125), (True, 500)):
    if chunk: eng.pipeline_chunk = chunk
    try:
        out = eng.forward_stft(stft, pipeline=pipe); torch.cuda.synchronize(); eng.check_stack_errors()
        ok = torch.equal(torch.view_as_real(ref["enh_stft"]), torch.view_as_real(out["enh_stft"]))
        for _ in range(2): eng.forward_stft(stft, pipeline=pipe)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): eng.forward_stft(stft, pipeline=pipe)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"pipeline={pipe} chunk={chunk}: {dt*1e3:.2f} ms  identical={ok} launches={eng.launches.get('split_scan')}", flush=True)
    except Exception as e:
        print("pipeline", pipe, chunk, "failed:", repr(e)[:200], flush=True)
