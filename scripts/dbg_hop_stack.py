"""Repro hunt: does a streaming hop launch earlier in the process make the full-size all-stacks forward time out?
python scripts/dbg_hop_stack.py <variant>   variant: none | one (hop=1 kernel) | multi (hop>1 kernel) | graph (graph-path session)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import refweights as rw
from test_hip_parity import build_module
DEV = torch.device("cuda:0")
variant = sys.argv[1]
kwm = rw.LIVE_M
if variant != "none":
    m = build_module("live", kwm, rw.live_state_dict(kwm, 5))
    B, hop = (1, 1) if variant in ("one", "graph") else (3, 4)
    sess = m.streaming(batch=B, hop=hop, one_launch=(variant != "graph"))
    x = torch.view_as_complex(0.05 * torch.randn((B, 257, hop, 2), device=DEV))
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 50):
        sess.step(x)
    sess.check_errors()
    torch.cuda.synchronize()
    print("streamed", variant)
model = build_module("live", kwm, rw.live_state_dict(kwm, 21))
eng = model.engine()
stft = model._stft(torch.from_numpy(rw.synth_wave(64, 1000, 3)).to(DEV))
for rep in range(3):
    for mode in (False, "auto", True):
        eng.stack_scan = mode
        b = eng.forward_stft(stft)
        try:
            eng.check_stack_errors()
            print(rep, mode, "ok")
        except RuntimeError as e:
            print(rep, mode, "FAILED", str(e)[-90:])
