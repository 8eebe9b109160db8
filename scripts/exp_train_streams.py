"""Round 4: the one-launch training layer calls of the three sub-band groups of baseline_m (R = 512 / 192 / 128 at B = 64, H = 224),
one after the other and together in one launch per layer and direction (training.gsn_stacks); forward and backward timed separately.
(STREAMS=1: the groups on three streams instead -- the arrangement that stopped dispatching, see scripts/dbg_train_hang.py.)
usage: python scripts/exp_train_streams.py [T] [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spiking_fullsubnet_amd import training
from spiking_fullsubnet_amd import modeling_spiking_fullsubnet as M
DEV = "cuda:0"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
torch.manual_seed(0)
groups = [(8 * B, 38), (3 * B, 94), (2 * B, 158)]
stacks = [M.StackedGSU(I, 224, 2, True, True).to(DEV).train() for _, I in groups]
xs = [torch.randn(T, R, I, device=DEV, requires_grad=True) for R, I in groups]
streams = [torch.cuda.Stream() for _ in groups]

def run(par, which=(0, 1, 2)):
    torch.cuda.synchronize()
    main = torch.cuda.current_stream()
    t0 = time.perf_counter()
    outs = []
    use_streams = bool(os.environ.get("STREAMS"))
    if par and not use_streams:
        outs = [o[-1].sum() for o in training.gsn_stacks([xs[g] for g in which], [stacks[g] for g in which], True)]
    for g in (which if not (par and not use_streams) else ()):
        if par:
            streams[g].wait_stream(main)
            with torch.cuda.stream(streams[g]):
                outs.append(training.gsn_stack(xs[g], stacks[g], True)[-1].sum())
        else:
            outs.append(training.gsn_stack(xs[g], stacks[g], True)[-1].sum())
    if par and use_streams:
        for g in which:
            main.wait_stream(streams[g])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    loss = sum(outs)
    loss.backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    training.check_pending()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3

configs = ((0, 1, 2),) if os.environ.get("ONLY_ALL") else ((0,), (1,), (2,), (0, 1, 2))
for which in configs:
    for par in ((False, True) if len(which) > 1 else (False,)):
        for _ in range(int(os.environ.get("REPEAT", "1"))):
            run(par, which)
        f, b = run(par, which)
        n = 2 * T * len(which)
        print(f"groups {which} {'together    ' if par else 'one by one  '}: forward {f:8.2f} ms ({f * 1e3 / n:6.2f} us per step and layer), "
              f"backward {b:8.2f} ms ({b * 1e3 / n:6.2f})", flush=True)
