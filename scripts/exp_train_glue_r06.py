"""Round 6: the ATen glue of the training step in isolation (forward + backward, ms): LayerNorm over short rows and the deep filter,
as training.py writes them today against cheaper formulations."""
import time, torch, torch.nn.functional as F
dev = torch.device("cuda:0")
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def ln_manual(x, w, b, eps=1e-5):
    var, mean = torch.var_mean(x, dim=-1, unbiased=False, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * w + b

class LNFn(torch.autograd.Function):
    """LayerNorm with a hand-written backward out of a few fused-by-shape ATen ops"""
    @staticmethod
    def forward(ctx, x, w, b, eps):
        var, mean = torch.var_mean(x, dim=-1, unbiased=False, keepdim=True)
        rstd = torch.rsqrt(var + eps)
        xhat = (x - mean) * rstd
        ctx.save_for_backward(xhat, rstd, w)
        return torch.addcmul(b, xhat, w)
    @staticmethod
    def backward(ctx, dy):
        xhat, rstd, w = ctx.saved_tensors
        g = dy * w
        m1 = g.mean(-1, keepdim=True)
        m2 = (g * xhat).mean(-1, keepdim=True)
        dx = (g - m1 - xhat * m2) * rstd
        dyf = dy.reshape(-1, dy.shape[-1]); xf = xhat.reshape(-1, dy.shape[-1])
        return dx, (dyf * xf).sum(0), dyf.sum(0), None

tot = {"aten": 0.0, "manual": 0.0, "fn": 0.0}
for (T, R, I) in [(1000, 512, 38), (1000, 192, 94), (1000, 128, 158)]:
    x = torch.randn(T, R, I, device=dev, requires_grad=True)
    w = torch.randn(I, device=dev, requires_grad=True); b = torch.randn(I, device=dev, requires_grad=True)
    gy = torch.randn(T, R, I, device=dev)
    def run(f):
        def go():
            y = f()
            torch.autograd.grad(y, (x, w, b), gy)
        return go
    a = bench(run(lambda: F.layer_norm(x, (I,), w, b)))
    m = bench(run(lambda: ln_manual(x, w, b)))
    f = bench(run(lambda: LNFn.apply(x, w, b, 1e-5)))
    y0 = F.layer_norm(x, (I,), w, b); y1 = LNFn.apply(x, w, b, 1e-5)
    g0 = torch.autograd.grad(y0, (x, w, b), gy); g1 = torch.autograd.grad(y1, (x, w, b), gy)
    print(f"LN {T}x{R}x{I}: aten {a:.3f} ms, var_mean autograd {m:.3f}, custom Function {f:.3f}; max diff y {float((y0-y1).abs().max()):.2e} dx {float((g0[0]-g1[0]).abs().max()):.2e} dw rel {float(((g0[1]-g1[1]).abs().max())/g0[1].abs().max()):.2e}")
    tot["aten"] += a; tot["manual"] += m; tot["fn"] += f
print("LN total fwd+bwd:", {k: round(v, 3) for k, v in tot.items()})

# deep filter: today's loop (training.forward_live) against a complex formulation
B, S, T = 64, 1, 1000
def df_loop(y, noisy, N, c, d):
    coef = y.reshape(B, N, 2, c, d, S, T)
    cre = coef[:, :, 0].permute(0, 3, 4, 1, 2, 5).reshape(B, d, S, N * c, T)
    cim = coef[:, :, 1].permute(0, 3, 4, 1, 2, 5).reshape(B, d, S, N * c, T)
    xp = F.pad(torch.view_as_real(noisy), (0, 0, d - 1, 0))
    xr, xi = xp[..., 0], xp[..., 1]
    yr = yi = 0
    for di in range(d):
        a, b = xr[:, None, :, di:di + T], xi[:, None, :, di:di + T]
        yr = yr + a * cre[:, di] - b * cim[:, di]
        yi = yi + a * cim[:, di] + b * cre[:, di]
    return torch.complex(yr, yi)
def df_cplx(y, noisy, N, c, d):
    coef = y.reshape(B, N, 2, c, d, S, T)
    cc = torch.complex(coef[:, :, 0], coef[:, :, 1])                    # [B, N, c, d, S, T]
    cc = cc.permute(0, 3, 4, 1, 2, 5).reshape(B, d, S, N * c, T)
    xp = F.pad(noisy, (d - 1, 0))                                       # [B, Nc, T + d - 1]
    xt = xp.unfold(2, T, 1).permute(0, 2, 1, 3)                         # [B, d, Nc, T] (a view)
    return (xt[:, :, None] * cc).sum(1)
tl = tc = 0.0
for (N, c, d) in [(8, 4, 5), (3, 32, 3), (2, 64, 1)]:
    P = 2 * c * d * S
    y = torch.randn(B * N, P, T, device=dev, requires_grad=True)
    noisy = torch.randn(B, N * c, T, dtype=torch.complex64, device=dev)
    ge = torch.randn(B, S, N * c, T, dtype=torch.complex64, device=dev)
    def run(f):
        def go():
            e = f(y, noisy, N, c, d)
            torch.autograd.grad(e, y, ge)
        return go
    a = bench(run(df_loop)); b_ = bench(run(df_cplx))
    e0 = df_loop(y, noisy, N, c, d); e1 = df_cplx(y, noisy, N, c, d)
    g0, = torch.autograd.grad(e0, y, ge); g1, = torch.autograd.grad(e1, y, ge)
    print(f"DF N={N} c={c} d={d}: loop {a:.3f} ms, complex {b_:.3f}; max diff {float((e0-e1).abs().max()):.2e} grad {float((g0-g1).abs().max()):.2e}")
    tl += a; tc += b_
print("DF total fwd+bwd: loop", round(tl, 3), "complex", round(tc, 3))
