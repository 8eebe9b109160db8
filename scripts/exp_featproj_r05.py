"""Round 5: features + layer-0 input products in one launch (sfsn_features_proj) against the two launches.
(1) the launches alone on the chip at baseline_m's chunk sizes (sub-band chunk of NT frames, full-band chunk), HIP-event timed;
(2) one forward alone (strict schedule) and with layer_outputs="counts", fused on / off, interleaved.
usage: python scripts/exp_featproj_r05.py [B] [T] [NT]     (SFSN_FP_BLOCKS=n: workgroups of the fused launch)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import refweights as rw
import spiking_fullsubnet_amd as pkg
from spiking_fullsubnet_amd import _lib
from spiking_fullsubnet_amd._lib import FeatProjJob, FeatureGroup, InProjJob, check
DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
NT = int(sys.argv[3]) if len(sys.argv) > 3 else 380
L = _lib.lib()
rng = np.random.default_rng(1)
F, FB = 257, 64
ri = torch.randn((B, F, T, 2), device=DEV) * 3
fbp = torch.randn((T, B, FB), device=DEV)


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def case(name, groups, Hs, fb, need_x):
    n = len(groups)
    fg, jobs, ij, keep = (FeatureGroup * n)(), (FeatProjJob * n)(), [], []
    for i, ((lo, nu, ctr, nbr, cfb, nfb), H) in enumerate(zip(groups, Hs)):
        I = ctr + 2 * nbr + (cfb + 2 * nfb if cfb else 0)
        x = torch.empty((T, B * nu, I), device=DEV)
        lw, lb = torch.rand(I, device=DEV) + 0.5, torch.randn(I, device=DEV) * 0.1
        g = fg[i]
        g.x, g.lo, g.n_units, g.ctr, g.nbr, g.ctr_fb, g.nbr_fb, g.norm, g.ln_eps = x.data_ptr(), lo, nu, ctr, nbr, cfb, nfb, _lib.NORM_LAYERNORM, 1e-5
        g.ln_w, g.ln_b = lw.data_ptr(), lb.data_ptr()
        jobs[i].feat = g
        keep += [x, lw, lb]
        if H is not None:
            w, bias, z = torch.randn((H, I), device=DEV) * 0.1, torch.randn(H, device=DEV), torch.empty((NT, B * nu, H), device=DEV)
            keep += [w, bias, z]
            jobs[i].w, jobs[i].bias, jobs[i].z, jobs[i].H, jobs[i].ldz = w.data_ptr(), bias.data_ptr(), z.data_ptr(), H, H
            if not need_x:
                jobs[i].feat.x = None
            a = InProjJob()
            a.x, a.w, a.bias, a.z, a.M, a.K, a.N, a.ldz = x.data_ptr() + 100 * B * nu * I * 4, w.data_ptr(), bias.data_ptr(), z.data_ptr(), NT * B * nu, I, H, H
            ij.append(a)
    arr = (InProjJob * len(ij))(*ij)
    fbptr = fbp.data_ptr() if fb else None
    def two():
        check(L.sfsn_features(ri.data_ptr(), fbptr, B, F, T, FB if fb else 0, 0.5, fg, n, 100, NT, None), "f")
        if len(ij) > 1:
            check(L.sfsn_input_proj_f32_multi(arr, len(ij), None), "m")
        else:
            a = ij[0]
            check(L.sfsn_input_proj_f32(a.x, a.w, a.bias, a.z, a.M, a.K, a.N, a.ldz, None), "s")
    def feat():
        check(L.sfsn_features(ri.data_ptr(), fbptr, B, F, T, FB if fb else 0, 0.5, fg, n, 100, NT, None), "f")
    def one():
        check(L.sfsn_features_proj(ri.data_ptr(), fbptr, B, F, T, FB if fb else 0, 0.5, jobs, n, 100, NT, None, 0, None), "fp")
    print(f"{name} (B={B}, {NT} frames, rows {'written' if need_x else 'not written'}): features alone {timed(feat):.1f} us, two launches {timed(two):.1f} us, "
          f"one launch {timed(one):.1f} us", flush=True)


sb = [(0, 8, 4, 15, 4, 0), (32, 3, 32, 15, 32, 0), (128, 2, 64, 15, 64, 0)]
for need_x in (True, False):
    case("sub-band chunk", sb, (None, 224, 224), True, need_x)
    case("full-band chunk", [(0, 1, 64, 0, 0, 0)], (320,), False, need_x)
    case("sub-band chunk, group 0 too", sb, (224, 224, 224), True, need_x)

if os.environ.get("FORWARD", "1") != "0":
    kw = rw.LIVE_M; sd = rw.live_state_dict(kw, 21)
    m = pkg.SpikingFullSubNet(**kw); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True); m = m.eval().to(DEV)
    stft = m._stft(torch.from_numpy(rw.synth_wave(B, T, 3)).to(DEV)); eng = m.engine()
    eng.fuse_featproj = False
    ref = eng.forward_stft(stft); torch.cuda.synchronize()
    eng.fuse_featproj = True
    out = eng.forward_stft(stft); torch.cuda.synchronize(); eng.check_stack_errors()
    ok = torch.equal(torch.view_as_real(ref["enh_stft"]), torch.view_as_real(out["enh_stft"])) and all(
        torch.equal(x, y) for x, y in zip(ref["fb_all"] + sum(ref["sb_all"], []), out["fb_all"] + sum(out["sb_all"], [])))
    print("forward with the fused launch:", "bit-identical" if ok else "MISMATCH", eng.launches, flush=True)
    for rep in range(3):
        for fused in (False, True):
            for layers in (True, False):
                eng.fuse_featproj = fused
                kwargs = dict(want_layers=layers, want_counts=not layers)
                for _ in range(3): eng.forward_stft(stft, **kwargs)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10): eng.forward_stft(stft, **kwargs)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
                print(f"rep {rep} fused={int(fused)} layer_outputs={'tensors' if layers else 'counts'}: {dt*1e3:.3f} ms per forward", flush=True)
    eng.check_stack_errors()
