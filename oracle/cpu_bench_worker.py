"""Worker process of bench.py's cpu_baseline leg -- TEST / MEASUREMENT INFRASTRUCTURE (see oracle/__init__.py).

Runs the CPU oracle's whole-model forward on clips [lo, hi) of a batch stored as .npy, over and over between a common
start time and a deadline, with its own OpenMP team (OMP_NUM_THREADS from the environment), and prints one JSON line:
how many forwards it completed and when the last one finished.  One process per group of clips: no interpreter lock is
shared between groups (64 Python threads in one process reached 9 x one core on a 256-thread host, the numpy glue of the
oracle's model composition serialised them).

python oracle/cpu_bench_worker.py <stft.npy> <lo> <hi> <start_epoch_s> <deadline_epoch_s> <weight_seed>
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

import refweights as rw  # noqa: E402
from oracle import model as omodel  # noqa: E402


def main():
    path, lo, hi, start, deadline, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6])
    kw = rw.LIVE_M
    sd = rw.live_state_dict(kw, seed)
    spec = omodel.spec_from_live_kwargs(kw)
    x = np.load(path, mmap_mode="r")[lo:hi]
    x = np.ascontiguousarray(x)
    omodel.forward_from_stft(spec, sd, x[:1, :, :16], "f32")  # build the library's state, start the OpenMP team, page in
    while time.time() < start:
        time.sleep(0.002)
    n = 0
    t_end = time.time()
    while True:
        omodel.forward_from_stft(spec, sd, x, "f32")
        n += 1
        t_end = time.time()
        if t_end >= deadline:
            break
    print(json.dumps(dict(lo=lo, hi=hi, forwards=n, t_end=t_end)), flush=True)


if __name__ == "__main__":
    main()
