"""Direct timing of sfsn_gsn_stack_scan on synthetic data: layers x rows-per-workgroup x lag, H=224 sub-band geometry
(rows 512 + 192 + 128) or H=320 (rows 64).  python scripts/exp_stack_direct.py H nl rpw lag [T]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spiking_fullsubnet_amd import _lib
from spiking_fullsubnet_amd._lib import ScanSegment, FusedInput, check
from spiking_fullsubnet_amd.engine import pack_w3

def run(H, nl, rpw, lag, T=1000, Rs=None, reps=5):
    if os.environ.get("ROWS"): Rs = [int(v) for v in os.environ["ROWS"].split(",")]
    L = _lib.lib(); dev = "cuda:0"
    Rs = Rs or ([512, 192, 128] if H <= 256 else [64])
    ns = len(Rs); HP = (H + 63) // 64 * 64
    rng = np.random.default_rng(0)
    keep = []
    def dv(a): t = torch.from_numpy(np.ascontiguousarray(a)).to(dev); keep.append(t); return t
    segs = (ScanSegment * (nl * ns))(); fin = (FusedInput * (nl * ns))()
    s8 = [[torch.zeros((T, R, HP), dtype=torch.int8, device=dev) for R in Rs] for _ in range(nl)]
    for l in range(nl):
        for i, R in enumerate(Rs):
            w = (rng.standard_normal((H, H)) / np.sqrt(H)).astype(np.float32)
            pk, dq = pack_w3(w); pk, dq = dv(pk), dv(dq)
            sg = segs[l * ns + i]
            zin = dv((rng.standard_normal((T, R, H)) * 0.5).astype(np.float32)) if (l == 0 or H > 256 or not os.environ.get('NOZIN')) else None
            sg.zin = None if zin is None else zin.data_ptr()
            sg.w_hh, sg.w_dq = pk.data_ptr(), dq.data_ptr()
            sg.bias = dv((rng.standard_normal(2 * H) * 0.1).astype(np.float32)).data_ptr()
            sg.bn_alpha = dv(np.ones(H, np.float32)).data_ptr(); sg.bn_beta = dv(np.zeros(H, np.float32)).data_ptr()
            sg.h_state = dv(np.zeros((R, H), np.float32)).data_ptr(); sg.c_state = dv(np.zeros((R, H), np.float32)).data_ptr()
            spk = torch.empty((T, R, H), dtype=torch.float32, device=dev); keep.append(spk)
            sg.spikes_f32 = spk.data_ptr(); sg.spikes_i8 = s8[l][i].data_ptr(); sg.membrane = None; sg.R = R
            if l > 0:
                wi = (rng.standard_normal((H, H)) / np.sqrt(H)).astype(np.float32)
                pk2, dq2 = pack_w3(wi); pk2, dq2 = dv(pk2), dv(dq2)
                fin[l * ns + i].spikes_in = s8[l - 1][i].data_ptr(); fin[l * ns + i].w_ih = pk2.data_ptr(); fin[l * ns + i].w_ih_dq = dq2.data_ptr()
    nb = L.sfsn_stack_scratch_bytes(nl, ns, sum(Rs))
    scratch = torch.zeros((nb // 4,), dtype=torch.int32, device=dev)
    rp = (ctypes.c_int * nl)(*([rpw] * nl))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def go(): check(L.sfsn_gsn_stack_scan(segs, fin, nl, ns, T, H, rp, lag, ctypes.c_void_p(scratch.data_ptr()), nb, st), "stack")
    go(); torch.cuda.synchronize()
    assert int(scratch[0].item()) == 0, "hand-off wait expired"
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); go(); b.record()
    torch.cuda.synchronize()
    ms = min(a.elapsed_time(b) for a, b in ev)
    rate = [float((s8[l][0] != 0).float().mean().item()) for l in range(nl)]
    if os.environ.get("SFSN_STACK_DEBUG"):
        w = scratch.cpu().numpy().astype(np.int64)
        # block layout: per layer [PROJ roles (if any)] then scan roles, each role padded to a multiple of 8 blocks
        blk, roles = 0, []
        for l in range(nl):
            for i, R in enumerate(Rs):
                if l > 0 and not os.environ.get('NOZIN'):
                    n = (R + 31) // 32; roles.append((f"PROJ l{l} seg{i}", blk, blk + n)); blk = (blk + n + 7) & ~7
                n = (R + rpw - 1) // rpw; roles.append((f"scan l{l} seg{i}", blk, blk + n)); blk = (blk + n + 7) & ~7
        dbg = w[blk + 2: blk + 2 + 4 * blk].reshape(-1, 4); t0 = min(dbg[b0:b1, 2].min() for _, b0, b1 in roles)
        for (name, b0, b1) in roles:
            d = dbg[b0:b1]
            print(f"   {name:14s} waits/WG {d[:,0].mean():7.1f}  polls/WG {d[:,1].mean():8.1f}  start us [{(d[:,2].min()-t0)/100:8.1f}, {(d[:,2].max()-t0)/100:8.1f}]  end us [{(d[:,3].min()-t0)/100:8.1f}, {(d[:,3].max()-t0)/100:8.1f}]")
    print(f"H={H} layers={nl} rpw={rpw} lag={lag} T={T} rows={Rs}: {ms:.3f} ms = {1e3*ms/T:.3f} us/step; spike rates {['%.2f' % r for r in rate]}", flush=True)

if __name__ == "__main__":
    a = sys.argv[1:]
    if a:
        run(int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]) if len(a) > 4 else 1000)
    else:
        for (H, nl, rpw, lag) in [(224, 1, 8, 16), (224, 1, 4, 16), (224, 2, 8, 16), (224, 2, 8, 4), (224, 2, 8, 64), (224, 3, 8, 16),
                                  (320, 1, 4, 16), (320, 2, 4, 16), (320, 2, 4, 4), (320, 2, 8, 16)]:
            run(H, nl, rpw, lag)
        run(224, 2, 8, 16, Rs=[512])
        run(224, 2, 8, 16, Rs=[64])
