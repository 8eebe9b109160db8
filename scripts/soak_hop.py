"""Soak test of the one-launch streaming hop: 100k hops against the graph-replay session on the same frames (tag wrap-around,
state parity, history) -- outputs compared every 500 hops, error words checked at the end.  Run on the MI355X box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import refweights as rw
from test_hip_parity import build_module
DEV = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
kw = dict(rw.LIVE_M, shared_weights=False) if os.environ.get("UNSHARED") else rw.LIVE_M  # UNSHARED=1: separate gate weights (the G = 2 hop kernels)
model = build_module("live", kw, rw.live_state_dict(kw, 5))
a, b = model.streaming(batch=B, one_launch=True), model.streaming(batch=B, one_launch=False)
g = torch.Generator(device="cpu").manual_seed(1)
pool = torch.view_as_complex((0.05 * torch.randn((64, B, 257, 1, 2), generator=g)).to(DEV))
idx = torch.randint(0, 64, (n,), generator=g).tolist()
t0 = time.perf_counter()
bad = 0
for i in range(n):
    x = pool[idx[i]]
    ea, ma = a.step(x, copy=False)
    eb, mb = b.step(x, copy=False)
    if i % 500 == 499 or i < 300:
        if not (torch.equal(torch.view_as_real(ea), torch.view_as_real(eb)) and torch.equal(ma, mb)):
            bad += 1
            print("mismatch at hop", i)
            if bad > 5: break
a.check_errors()
print("hops", n, "mismatches", bad, "seconds %.1f" % (time.perf_counter() - t0))
