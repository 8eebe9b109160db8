"""Round 3 re-run of the time-pipelined schedule (stages on their own streams, chained by events) with the scan3 kernels:
sub-band scans at 4 / 8 rows per workgroup, chunk 125 / 250 / 500, against the default schedule (3 overlap chunks)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
dev = torch.device("cuda", 0)
kw = rw.LIVE_M
model = pkg.SpikingFullSubNet(**kw)
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in rw.live_state_dict(kw, 21).items()})
model = model.eval().to(dev)
eng = model.engine()
stft = model._stft(torch.from_numpy(rw.synth_wave(64, 1000, 0)).to(dev)).contiguous()
ref = eng.forward_stft(stft)
torch.cuda.synchronize()
def run(pipe, chunk, n=6):
    eng.pipeline_chunk = chunk
    out = eng.forward_stft(stft, pipeline=pipe); torch.cuda.synchronize()
    same = torch.equal(torch.view_as_real(out["enh_stft"]), torch.view_as_real(ref["enh_stft"]))
    t0 = time.perf_counter()
    for _ in range(n):
        eng.forward_stft(stft, pipeline=pipe)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, same
print("default schedule            %.2f ms  identical %s" % run(None, 128))
for two in (1, 0):
    eng.pipeline_two_phase = bool(two)
    for rpw in ((0, 0), (4, 8), (4, 4)):
        eng.rows_per_wg = rpw
        for c in (500, 250, 125):
            try:
                print("two_phase %d rpw %s chunk %3d : %.2f ms  identical %s" % ((two, rpw, c) + run(True, c)), flush=True)
            except Exception as e:
                print("two_phase %d rpw %s chunk %3d : %r" % (two, rpw, c, e), flush=True)
eng.check_stack_errors()
