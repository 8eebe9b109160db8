"""Phase ledger of featproj_kernel (round 5): sfsn_featproj.hip compiled ALONE with -DFP_STAMPS into a small library
(scripts/micro/featproj_stamps.sh); wave 0 of each job's first workgroup sums shader clocks per phase.
phases: 0 prologue (W pieces, table, first tile), 1 rows, 2 wait at barrier 1, 3 product, 4 park, 5 wait at barrier 2, 6 stores"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spiking_fullsubnet_amd import _lib
from spiking_fullsubnet_amd._lib import FeatProjJob
DEV = "cuda:0"
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfp_stamps.so"))
_P, _I, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
L.sfsn_features_proj.argtypes = [_P, _P, _I, _I, _I, _I, _F, ctypes.POINTER(FeatProjJob), _I, _I, _I, _P, ctypes.c_size_t, _P]
B, T, NT, F, FB = 64, 1000, int(os.environ.get("NT", 380)), 257, 64
ri = torch.randn((B, F, T, 2), device=DEV) * 3
fbp = torch.randn((T, B, FB), device=DEV)
def run(name, groups, Hs, fb, need_x):
    n = len(groups); jobs = (FeatProjJob * n)(); keep = []
    for i, ((lo, nu, ctr, nbr, cfb, nfb), H) in enumerate(zip(groups, Hs)):
        I = ctr + 2 * nbr + (cfb + 2 * nfb if cfb else 0)
        x = torch.empty((T, B * nu, I), device=DEV); lw, lb = torch.rand(I, device=DEV) + 0.5, torch.randn(I, device=DEV) * 0.1
        g = jobs[i].feat
        g.x, g.lo, g.n_units, g.ctr, g.nbr, g.ctr_fb, g.nbr_fb, g.norm, g.ln_eps = x.data_ptr(), lo, nu, ctr, nbr, cfb, nfb, _lib.NORM_LAYERNORM, 1e-5
        g.ln_w, g.ln_b = lw.data_ptr(), lb.data_ptr(); keep += [x, lw, lb]
        if H is not None:
            w, bias, z = torch.randn((H, I), device=DEV) * 0.1, torch.randn(H, device=DEV), torch.empty((NT, B * nu, H), device=DEV)
            keep += [w, bias, z]
            jobs[i].w, jobs[i].bias, jobs[i].z, jobs[i].H, jobs[i].ldz = w.data_ptr(), bias.data_ptr(), z.data_ptr(), H, H
            if not need_x: g.x = None
    def one():
        rc = L.sfsn_features_proj(ri.data_ptr(), fbp.data_ptr() if fb else None, B, F, T, FB if fb else 0, 0.5, jobs, n, 100, NT, None, 0, None)
        assert rc == 0, rc
    for _ in range(3): one()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): one()
    b.record(); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * (8 * 8))()
    assert L.sfsn_fp_debug(out) == 0
    print(f"{name}: {a.elapsed_time(b) / 10 * 1e3:.1f} us per launch")
    for i in range(n):
        print("   job", i, " ".join(f"{out[i * 8 + k]:>8d}" for k in range(7)), " sum", sum(out[i * 8 + k] for k in range(7)))
sb = [(0, 8, 4, 15, 4, 0), (32, 3, 32, 15, 32, 0), (128, 2, 64, 15, 64, 0)]
run("sub-band chunk", sb, (None, 224, 224), True, True)
run("full-band chunk", [(0, 1, 64, 0, 0, 0)], (320,), False, True)
run("group 2 alone", sb[2:], (224,), True, False)
