"""Round 4: streaming latency of a model with separate forget / cell gate weights (baseline_xl's sizes: full band 320, sub band 224,
cumulative Laplace norm) -- one launch per hop (sfsn_stream_hop, G = 2 kernels) against the offline kernels replayed from a HIP graph
-- and of the shared-weight baseline_m sizes beside it (must be unchanged)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
import bench
DEV = torch.device("cuda:0")
def live(kw, seed):
    m = pkg.SpikingFullSubNet(**kw); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in rw.live_state_dict(kw, seed).items()}, strict=True)
    return m.eval().to(DEV)
cases = [("live baseline_m (shared)", live(rw.LIVE_M, 5)), ("live baseline_m sizes, separate gate weights", live(dict(rw.LIVE_M, shared_weights=False), 8))]
kx = dict(rw.FROZEN_XL, norm_type="cumulative_laplace_norm")
mx = pkg.Separator(**kx); mx.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in rw.frozen_state_dict(kx, 34).items()}, strict=True)
cases.append(("frozen baseline_xl.toml as written (separate weights, cumulative Laplace norm)", mx.eval().to(DEV)))
for name, m in cases:
    for one in (("auto", False) if "xl" not in name else ("auto",)):
        r = bench.streaming_measure(m, DEV, 1, 1, 2000, 200, None, True, one)
        print(f"{name}: one_launch={r.get('one_launch')} p50 {r['p50_us']:.1f} us  p99 {r['p99_us']:.1f} us", flush=True)
