"""Phase ledger of input_proj_bf3_kernel (round 5): sfsn_kernels.hip compiled ALONE with -DIP_STAMPS (scripts/micro/inproj_stamps.sh);
wave 0 of the launch's first workgroup sums shader clocks per phase: 0 prologue (W pieces, first tile), 1 product, 2 wait at the
barrier, 3 park (split of the prefetched tile), 4 stores issued, 5 wait at the closing barrier."""
import ctypes, os, sys
import torch
DEV = "cuda:0"
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libip_stamps.so"))
_P, _I = ctypes.c_void_p, ctypes.c_int
L.sfsn_input_proj_f32.argtypes = [_P, _P, _P, _P, _I, _I, _I, _I, _P]
for name, M, K, N in (("group 2 chunk", 64 * 380 * 2, 158, 224), ("group 1 chunk", 64 * 380 * 3, 94, 224), ("full band chunk", 64 * 380, 64, 320)):
    x, w, b, z = torch.randn((M, K), device=DEV), torch.randn((N, K), device=DEV) * 0.1, torch.randn(N, device=DEV), torch.empty((M, N), device=DEV)
    def one():
        assert L.sfsn_input_proj_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), z.data_ptr(), M, K, N, N, None) == 0
    for _ in range(3): one()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): one()
    e.record(); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    assert L.sfsn_ip_debug(out) == 0
    tiles = -(-M // 64)
    print(f"{name} (M={M}, K={K}, N={N}; {tiles} tiles, {tiles / 256:.1f} per workgroup): {a.elapsed_time(e) / 10 * 1e3:.1f} us per launch; clk per phase",
          " ".join(f"{out[k]:>7d}" for k in range(6)), " sum", sum(out[k] for k in range(6)), f"= {sum(out[k] for k in range(6)) / 2400:.1f} us at 2.4 GHz")
