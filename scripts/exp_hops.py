"""Diagnostic: cost of a cross-stream dependency hop (event record + wait) with tiny kernels, plain vs CU-masked streams."""
import os, sys, time, ctypes
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[1] if len(sys.argv) > 1 else "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
dev = torch.device("cuda", 0)
m = pkg.SpikingFullSubNet(**rw.LIVE_TINY).eval().to(dev)
eng = m.engine()
x = torch.zeros(1024, device=dev)

def pingpong(sa, sb, n=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(sa):
            x.add_(1.0)
            e = torch.cuda.Event(); e.record(sa)
        sb.wait_event(e)
        with torch.cuda.stream(sb):
            x.add_(1.0)
            e2 = torch.cuda.Event(); e2.record(sb)
        sa.wait_event(e2)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (2 * n) * 1e6

def same(sa, n=400):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(sa):
        for i in range(n): x.add_(1.0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

p1, p2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
m1 = eng._masked_stream([w * 32 + b for w in range(8) for b in range(0, 8)])
m2 = eng._masked_stream([w * 32 + b for w in range(8) for b in range(8, 32)])
print("HWQ", os.environ["GPU_MAX_HW_QUEUES"])
print("same-stream tiny kernel      %.1f us" % same(p1))
print("same-stream masked           %.1f us" % same(m1))
print("hop plain <-> plain          %.1f us" % pingpong(p1, p2))
print("hop masked <-> masked        %.1f us" % pingpong(m1, m2))
print("hop plain <-> masked         %.1f us" % pingpong(p1, m1))
