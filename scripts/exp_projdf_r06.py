"""Round 6: the sub-band epilogue as one launch (sfsn_proj_deepfilter) against sfsn_spike_proj_multi + sfsn_deepfilter.
B = 64, T = 1000, live baseline_m: HIP-event time of the launch group(s) per forward (whole sequence and the strict forward's
three chunks), and the strict forward itself, alternating.  SFSN_PDF_FT / SFSN_PDF_WGS / SFSN_PDF_LDS_KB vary the new kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg

B, T = int(os.environ.get("B", 64)), int(os.environ.get("T", 1000))
dev = torch.device("cuda:0")
kw = rw.LIVE_M
sd = rw.live_state_dict(kw, 21)
m = pkg.SpikingFullSubNet(**kw)
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
m = m.eval().to(dev)
x = m._stft(torch.from_numpy(rw.synth_wave(B, T, seed=0)).to(dev)).contiguous()
eng = m.engine()

def strict(n=8):
    for _ in range(3): eng.forward_stft(x, pipeline=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): eng.forward_stft(x, pipeline=False)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def groups(chunks, lean=False):
    ov = eng.overlap_chunks; eng.overlap_chunks = chunks
    eng.timers, eng.timer_tags = {}, {"proj:sb", "deepfilter", "projdf"}
    for _ in range(4): eng.forward_stft(x, pipeline=False, want_layers=not lean, want_counts=lean)
    s = eng.timer_summary(); eng.timers = None; eng.overlap_chunks = ov
    return {k: (round(v["mean_ms"] * 1e3, 1), v["n"]) for k, v in s.items()}

for rnd in range(3):
    for fused in (False, True):
        eng.fuse_projdf = fused
        print(f"round {rnd} fused={fused}: strict {strict():.3f} ms | whole-seq us {groups(0)} | 3 chunks us {groups(3)} | lean whole {groups(0, True)}", flush=True)
eng.check_stack_errors()
