"""Round-6 soak: the new kernels (sfsn_proj_deepfilter, gsn_scan_fused3 / fusedx3) under load, bit for bit against the paths they
replace.  Phase 1: the timed region's geometry, LANES forwards in flight, each lane's enh_stft / coefficient rows / last-layer spikes
hashed every iteration and compared with a reference computed once with SFSN_FUSED_V2=1 + the two-launch epilogue.  Phase 2: the strict
forward (pair launch, three chunks).  usage: python scripts/soak_r06.py [region iterations] [strict forwards]"""
import sys, os, time, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import refweights as rw
import spiking_fullsubnet_amd as pkg
dev = torch.device("cuda")
n_reg = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n_str = int(sys.argv[2]) if len(sys.argv) > 2 else 200
LANES, B, T = 12, 64, 1000
kw = rw.LIVE_M
m = pkg.SpikingFullSubNet(**kw); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in rw.live_state_dict(kw, 21).items()}); m = m.eval().to(dev)
eng = m.engine()
xs = [m._stft(torch.from_numpy(rw.synth_wave(B, T, seed=100 + i)).to(dev)).contiguous() for i in range(LANES)]

def digest(res):
    ts = [torch.view_as_real(res["enh_stft"]), res["enh_mag"]] + [t for l in [res["fb_all"]] + res["sb_all"] for t in l if torch.is_tensor(t) and t.device.type != "meta"]
    return torch.stack([t.view(torch.int32).to(torch.int64).sum() for t in ts])

def geometry(region):
    eng.rows_per_wg = (8, 16) if region else (0, 0)
    eng.stack_rows_fb_auto = 8 if region else 4
    eng.overlap_chunks = 0 if region else 3

# ---- references: the paths of round 5 (round 2's 16-row bodies, projection + deep filter as two launches)
os.environ["SFSN_FUSED_V2"] = "1"; eng.fuse_projdf = False
geometry(True)
ref_reg = [digest(eng.forward_stft(x, pipeline=False)) for x in xs]
geometry(False)
ref_str = digest(eng.forward_stft(xs[0], pipeline=False))
torch.cuda.synchronize(); eng.check_stack_errors()
del os.environ["SFSN_FUSED_V2"]; eng.fuse_projdf = True

geometry(True)
lanes = [torch.cuda.Stream(device=dev) for _ in range(LANES)]
bad, t0 = 0, time.perf_counter()
for it in range(n_reg):
    outs = []
    for s_, x in zip(lanes, xs):
        with torch.cuda.stream(s_):
            outs.append(digest(eng.forward_stft(x, pipeline=False)))
    torch.cuda.synchronize()
    bad += sum(int(not torch.equal(a, b)) for a, b in zip(outs, ref_reg))
eng.check_stack_errors()
print(f"region geometry: {n_reg} x {LANES} forwards in flight in {time.perf_counter() - t0:.1f} s, launches {dict(eng.launches)}, mismatching lanes {bad}", flush=True)
geometry(False)
bad2, t0 = 0, time.perf_counter()
for it in range(n_str):
    d = digest(eng.forward_stft(xs[0], pipeline=False))
    if it % 10 == 9:
        bad2 += int(not torch.equal(d, ref_str))
torch.cuda.synchronize(); eng.check_stack_errors()
print(f"strict forward: {n_str} forwards in {time.perf_counter() - t0:.1f} s, mismatching checks {bad2}", flush=True)
assert bad == 0 and bad2 == 0
