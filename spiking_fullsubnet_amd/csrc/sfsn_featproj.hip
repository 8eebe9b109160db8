// sfsn_featproj.hip -- the feature prologue and the layer-0 input product of a chunk in ONE launch (sfsn_features_proj, ABI 17).
// gfx950 only.
//
// What it replaces: sfsn_features (one launch: gather + normalise, rows written to HBM) followed by sfsn_input_proj_f32(_multi)
// (a second launch that reads those rows back, splits them into three bf16 pieces and forms x.W_ih^T + b on the bf16 matrix
// cores) -- MODEL:434-440,239-258,108-112 + NEURON:141-142.  Here a workgroup keeps the rows it has just normalised: they go
// from registers straight into the three bf16 planes the MFMA B fragments are read from (no fp32 staging, no round trip), and
// only when somebody reads them afterwards (the API's all_layer_outputs[0], or a scan that forms its product itself) are they
// also written to HBM.  Per job (= one feature group) a range of persistent workgroups; a workgroup walks (clip, 32-frame) tiles:
//
//   prefetch of the NEXT tile's STFT bins / full-band columns into registers (in flight during everything below)
//   for every 32-row sub-tile (rows = (frame, unit) pairs, frame-major):
//       eight waves x four rows: gather from the LDS tiles, normalise (the expressions of features_kernel, same lane
//       assignment and wave reductions: bit-identical rows), neighbouring lanes pair up by DPP, split3 -> planes in LDS
//       barrier;  2 x TPW x KS x 6 bf16 MFMAs per wave (the six products and two accumulators of input_proj_bf3_kernel,
//       same order: bit-identical input terms) -> output tile in LDS;  barrier;  whole-row stores of the tile
//   compress + park the prefetched tile
//
// W_ih pieces live in registers for the whole launch (TPW x KS x 3 x 4 VGPRs).  A job without weights is the feature kernel
// alone (rows written, no product): group 0 of baseline_m, whose product the scan forms itself (FUSEDX3 / fused-x).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "sfsn.h"
#include "sfsn_scan_dev.h"  // v4f, bf8, split3
#include "sfsn_feat_dev.h"  // reflect_bin, compress_mag, wave_sum

typedef int v2i __attribute__((ext_vector_type(2)));

#define FP_TT 32    // frames per tile
#define FP_ROWS 32  // rows per sub-tile (two 16-row MFMA blocks)
#define FP_RPW 4    // rows per wave and sub-tile (independent chains: the row arithmetic is latency bound)
#define FP_NPRE 9   // magnitude bins per thread and tile: f_cnt <= 16 * FP_NPRE
#define FP_NFB 4    // full-band values per thread and tile: 32 * FB <= 512 * FP_NFB, 512 % FB == 0
#define FP_MAX_JOBS SFSN_MAX_GROUPS

struct FpJobDev {
    float* x;  // nullable when w is given
    const float* ln_w;
    const float* ln_b;
    const float* mu;
    const float* w;  // [H][I], NULL = rows only
    const float* bias;
    float* z;  // [nt][B*N][ldz], frame t0 first
    int lo, N, ctr, nbr, ctr_fb, nbr_fb, I1, I, norm;
    float eps;
    int f_lo, f_cnt;
    int H, ldz, NT, tpw, ks, nu;
    int block0, nblocks;
    int off_fbT, off_offs, off_planes, off_obuf;  // LDS layout (bytes; the magnitude tile sits at 0)
};
struct FpParams {
    FpJobDev job[FP_MAX_JOBS];
    int n, B, F, T, FB, t0, t1;
    float fdrc;
    float* zero_ptr;
    size_t zero_n16;
    int zero_block0;
};

// A copy the optimiser cannot see through: what is derived from it inside a loop is recomputed there (two or three VALU
// instructions) instead of being hoisted out of the tile loop into registers the W pieces need (LICM kept ~60 VGPRs of lane
// indices, compare masks and LDS addresses alive across the loop; the product's kernels spilled).
__device__ __forceinline__ int fp_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

#ifdef FP_STAMPS  // scripts/micro/featproj_stamps.sh: shader-clock sums per phase as wave 0 of each job's first workgroup sees them
__device__ unsigned long long fp_dbg[FP_MAX_JOBS][8];
#define FP_T(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); dbg_[k] += now_ - last_; last_ = now_; } while (0)
#else
#define FP_T(k) do {} while (0)
#endif

__device__ __forceinline__ void fp_barrier() {
    // LDS traffic only: a raw barrier behind lgkmcnt(0) (__syncthreads() would also drain every row store in flight)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- the product of one 32-row sub-tile: 2 row blocks x TPW column tiles, the six piece products of input_proj_bf3_kernel in
//      its order (KSJ = the job's k-steps <= KSM, the kernel's: W holds zeros beyond)
template <int TPW, int KSM, int KSJ>
__device__ __forceinline__ void fp_product(const bf8 (&W)[TPW][KSM][3], const v4f (&bv)[TPW], const int (&col)[TPW], const unsigned* xb,
                                           float* obuf, const int H, const int tid) {
    constexpr int LDX = KSJ * 32 + 8, PLANE = FP_ROWS * LDX / 2;
    const int NP = H + 4;
#pragma unroll 1
    for (int mi = 0; mi < FP_ROWS / 16; ++mi) {
        const int lo_ = fp_opaque(tid) & 63, n = lo_ & 15, q = lo_ >> 4;
        v4f hi[TPW], lo[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) hi[i] = lo[i] = v4f{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KSJ; ++ks) {
            const unsigned* src = xb + (((mi * 16 + n) * LDX + ks * 32 + q * 8) >> 1);
            const bf8 b1 = *reinterpret_cast<const bf8*>(src), b2 = *reinterpret_cast<const bf8*>(src + PLANE),
                      b3 = *reinterpret_cast<const bf8*>(src + 2 * PLANE);
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                if (col[i] < 0) continue;
                lo[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[i][ks][2], b1, lo[i], 0, 0, 0);
                hi[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[i][ks][0], b1, hi[i], 0, 0, 0);
                lo[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[i][ks][1], b2, lo[i], 0, 0, 0);
                lo[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[i][ks][0], b3, lo[i], 0, 0, 0);
                lo[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[i][ks][1], b1, lo[i], 0, 0, 0);
                lo[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[i][ks][0], b2, lo[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            if (col[i] < 0) continue;
            v4f acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = (hi[i][r] + lo[i][r]) + bv[i][r];
            if (col[i] + 3 < H) {
                *reinterpret_cast<v4f*>(&obuf[(mi * 16 + n) * NP + col[i]]) = acc;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col[i] + r < H) obuf[(mi * 16 + n) * NP + col[i] + r] = acc[r];
            }
        }
    }
}

// One kernel per (TPW, KSM) = the LARGEST column-tile count per wave and k-step count among the launch's jobs; a job with fewer
// runs the same code with its own k-step count (a switch around the whole product phase) and skips absent column tiles.  (One
// kernel with a switch over every (TPW, KS, NU) instantiation spilled ~100 VGPRs to scratch where each instantiation alone
// spilled none, and a kernel with scratch is throttled chip-wide: 370 us instead of 140 for the two launches.)
template <int TPW, int KSM>
__global__ __launch_bounds__(512) void featproj_kernel(const float* __restrict__ stft, const float* __restrict__ fb, const FpParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x >= p.zero_block0) {  // the zero initial state of the forward's scans (as sfsn_features_z)
        const size_t blk = (size_t)((int)blockIdx.x - p.zero_block0);
        v4f* dst = reinterpret_cast<v4f*>(p.zero_ptr);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const size_t k = (blk * 8 + i) * 512 + threadIdx.x;
            if (k < p.zero_n16) dst[k] = v4f{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    int ji = 0;
    for (int i = 1; i < p.n; ++i)
        if ((int)blockIdx.x >= p.job[i].block0) ji = i;
    const FpJobDev& j = p.job[ji];
    const int blk = (int)blockIdx.x - j.block0;

#ifdef FP_STAMPS
    unsigned long long dbg_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
    constexpr int NU = (KSM * 32 + 63) / 64;  // feature slots per lane (I <= 64 NU: host)
    constexpr int RB = NU >= 3 ? 2 : 4;       // rows a wave has in flight (independent chains: the row arithmetic is latency bound)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int B = p.B, T = p.T, FB = p.FB, N = j.N, I = j.I, H = j.H, nf = p.F - 1;
    const bool prod = j.w != nullptr;
    const int ksj = j.ks;
    const int LDXj = ksj * 32 + 8, PLANEj = FP_ROWS * LDXj / 2;  // the job's plane geometry (dwords per plane)
    float* magT = reinterpret_cast<float*>(smem);                     // [f_cnt][33]
    float* fbT = reinterpret_cast<float*>(smem + j.off_fbT);          // [32][FB]
    v2i* offs = reinterpret_cast<v2i*>(smem + j.off_offs);            // [N][I] (element offset from magT, frame stride)
    unsigned* xb = reinterpret_cast<unsigned*>(smem + j.off_planes);  // [3][32][LDXj] bf16
    float* obuf = reinterpret_cast<float*>(smem + j.off_obuf);        // [32][H + 4]
    const int ntt = (p.t1 - p.t0 + FP_TT - 1) / FP_TT;
    const int ntile = B * ntt;
    const bool has_fb = j.ctr_fb > 0;

    // ---- W_ih pieces (A fragments: lane holds 8 consecutive k of weight row ct*16 + n), bias
    bf8 W[TPW][KSM][3];
    v4f bv[TPW];
    int col[TPW];
    {
        const int lane = tid & 63, n = lane & 15, q = lane >> 4;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ct = wave + 8 * i;
            const bool have = prod && ct < j.NT;
            col[i] = have ? ct * 16 + q * 4 : -1;
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[i][r] = (j.bias && have && col[i] + r < H) ? j.bias[col[i] + r] : 0.0f;
            const int wr = ct * 16 + n;
#pragma unroll
            for (int ks = 0; ks < KSM; ++ks) {
                unsigned pw[3][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = ks * 32 + q * 8 + 2 * e;
                    const float a = (have && wr < H && k < I) ? j.w[(size_t)wr * I + k] : 0.0f;
                    const float b = (have && wr < H && k + 1 < I) ? j.w[(size_t)wr * I + k + 1] : 0.0f;
                    split3(a, b, pw[0][e], pw[1][e], pw[2][e]);
                }
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) W[i][ks][pl] = *reinterpret_cast<const bf8*>(pw[pl]);
            }
        }
    }
    // ---- gather table of the group (the same for every tile): entry >= 0 = offset of the bin's row in magT, < 0 = -1 - full-band column
    for (int idx = tid; idx < N * I; idx += 512) {
        const int k = idx / I, jj = idx - k * I;
        offs[idx] = jj < j.I1 ? v2i{(reflect_bin(j.lo + k * j.ctr - j.nbr + jj, nf) - j.f_lo) * 33, 1}
                              : v2i{j.off_fbT / 4 + reflect_bin(j.lo + k * j.ctr_fb - j.nbr_fb + (jj - j.I1), nf) % FB, FB};
    }
    if (prod) {  // columns I..KQ-1 of the planes are never written: zero once
        for (int i = tid; i < 3 * PLANEj; i += 512) xb[i] = 0u;
    }
    float lw[NU], lb[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int jj = (tid & 63) + 64 * u;
        const bool in = jj < I && j.norm == SFSN_NORM_LAYERNORM;
        lw[u] = in ? j.ln_w[jj] : 0.0f;
        lb[u] = in ? j.ln_b[jj] : 0.0f;
    }
    const float inv_I = 1.0f / (float)I;

    // ---- tile prefetch: thread = (frame ptt, bin lane pf0), bins pf0 + 16 i; full-band values tid + 512 i.  Addresses are a
    // uniform base + ONE 32-bit per-thread offset + a uniform stride per i (64-bit per-i addresses hoisted out of the tile loop cost
    // 34 registers)
    v2f pre[FP_NPRE];
    float pfb[FP_NFB];
    const int nfbi = has_fb ? FP_TT * FB / 512 : 0;  // 512 % FB == 0 (host): frames per i = 512 / FB
    const int fb_dt = has_fb ? 512 / FB : 0;
    auto fetch = [&](int st) __attribute__((always_inline)) {
        const int b = st / ntt, ti = st - b * ntt;
        const int tido = fp_opaque(tid), ptt = tido & 31, pf0 = tido >> 5;
        const int t = p.t0 + ti * FP_TT + ptt;
        const int tc = t < p.t1 ? t : p.t1 - 1;
        const float* base = stft + ((size_t)b * p.F + j.f_lo) * T * 2;  // uniform
        const unsigned vo = ((unsigned)pf0 * (unsigned)T + (unsigned)tc) * 2u;
        const unsigned stride = 32u * (unsigned)T;  // 16 bins
#pragma unroll
        for (int i = 0; i < FP_NPRE; ++i)
            if (pf0 + 16 * i < j.f_cnt) pre[i] = *reinterpret_cast<const v2f*>(base + (size_t)(i * stride) + vo);
        const float* fbase = fb + (size_t)b * FB;  // uniform
        const int fb_tt0 = has_fb ? tido / FB : 0, fb_f = has_fb ? tido - fb_tt0 * FB : 0;
#pragma unroll
        for (int i = 0; i < FP_NFB; ++i)
            if (i < nfbi) {
                int tf = p.t0 + ti * FP_TT + fb_tt0 + i * fb_dt;
                if (tf > p.t1 - 1) tf = p.t1 - 1;
                pfb[i] = fbase[(unsigned)tf * (unsigned)(B * FB) + (unsigned)fb_f];
            }
    };
    auto park = [&](int st) __attribute__((always_inline)) {
        const int b = st / ntt, ti = st - b * ntt;
        const int t0t = p.t0 + ti * FP_TT;
        const int tido = fp_opaque(tid), ptt = tido & 31, pf0 = tido >> 5;
        const bool live = t0t + ptt < p.t1;
        float* mrow = magT + pf0 * 33 + ptt;
#pragma unroll
        for (int i = 0; i < FP_NPRE; ++i)
            if (pf0 + 16 * i < j.f_cnt) mrow[i * 16 * 33] = live ? compress_mag(pre[i][0], pre[i][1], p.fdrc) : 0.0f;
        float* frow = fbT + tido;
        const int fb_tt0 = has_fb ? tido / FB : 0;
#pragma unroll
        for (int i = 0; i < FP_NFB; ++i)
            if (i < nfbi) frow[i * 512] = (t0t + fb_tt0 + i * fb_dt < p.t1) ? pfb[i] : 0.0f;
    };

    if (blk < ntile) {
        fetch(blk);
        park(blk);
    }
    __syncthreads();
    FP_T(0);
    for (int st = blk; st < ntile; st += j.nblocks) {
        const int nxt = st + j.nblocks;
        if (nxt < ntile) fetch(nxt);
        const int b = st / ntt, ti = st - b * ntt;
        const int t0t = p.t0 + ti * FP_TT;
        const float lap_den = j.norm == SFSN_NORM_LAPLACE ? j.mu[b] + 2.220446049250313e-16f : 1.0f;
        const float gau_mu = j.norm == SFSN_NORM_GAUSSIAN ? j.mu[b] : 0.0f;
        const float gau_den = j.norm == SFSN_NORM_GAUSSIAN ? j.ln_w[b] + 2.220446049250313e-16f : 1.0f;
        for (int s = 0; s < N; ++s) {  // 32 frames x N units = N sub-tiles of 32 rows, frame-major
            // ---- four rows per wave, RB at a time: gather, normalise (features_kernel's expressions), planes / x
#pragma unroll 1
            for (int rb = 0; rb < FP_RPW; rb += RB) {
                const int lane = fp_opaque(tid) & 63;
                float v[RB][NU];
                int rtt[RB], rk[RB];
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const int gr = s * FP_ROWS + wave * FP_RPW + rb + r;
                    rtt[r] = gr / N;
                    rk[r] = gr - rtt[r] * N;
                }
                // branchless gather: every table read of the batch, then every value read (two LDS round trips per batch)
                v2i e[RB][NU];
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int u = 0; u < NU; ++u) e[r][u] = offs[lane + 64 * u < I ? rk[r] * I + lane + 64 * u : 0];
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const float val = magT[e[r][u][0] + rtt[r] * e[r][u][1]];
                        v[r][u] = lane + 64 * u < I ? val : 0.0f;
                    }
                float y[RB][NU];
                if (j.norm == SFSN_NORM_LAYERNORM) {
                    float mean[RB], rstd[RB];
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        float sum = 0.0f;
#pragma unroll
                        for (int u = 0; u < NU; ++u) sum += v[r][u];
                        mean[r] = wave_sum(sum) * inv_I;
                    }
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        float ss = 0.0f;
#pragma unroll
                        for (int u = 0; u < NU; ++u) {
                            const float d = v[r][u] - mean[r];
                            if (lane + 64 * u < I) ss += d * d;
                        }
                        rstd[r] = __builtin_amdgcn_rsqf(wave_sum(ss) * inv_I + j.eps);
                    }
#pragma unroll
                    for (int r = 0; r < RB; ++r)
#pragma unroll
                        for (int u = 0; u < NU; ++u) y[r][u] = ((v[r][u] - mean[r]) * rstd[r]) * lw[u] + lb[u];
                } else if (j.norm == SFSN_NORM_LAPLACE) {
#pragma unroll
                    for (int r = 0; r < RB; ++r)
#pragma unroll
                        for (int u = 0; u < NU; ++u) y[r][u] = v[r][u] / lap_den;
                } else if (j.norm == SFSN_NORM_GAUSSIAN) {
#pragma unroll
                    for (int r = 0; r < RB; ++r)
#pragma unroll
                        for (int u = 0; u < NU; ++u) y[r][u] = (v[r][u] - gau_mu) / gau_den;
                } else {
#pragma unroll
                    for (int r = 0; r < RB; ++r)
#pragma unroll
                        for (int u = 0; u < NU; ++u) y[r][u] = v[r][u];
                }
                if (j.x) {
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        const int t = t0t + rtt[r];
                        if (t < p.t1) {  // wave-uniform
                            float* out = j.x + (((size_t)t * B + b) * N + rk[r]) * I;
#pragma unroll
                            for (int u = 0; u < NU; ++u)
                                if (lane + 64 * u < I) out[lane + 64 * u] = y[r][u];
                        }
                    }
                }
                if (prod) {
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        const int lr = wave * FP_RPW + rb + r;
#pragma unroll
                        for (int u = 0; u < NU; ++u) {
                            // the odd neighbour's value (quad_perm [1,0,3,2]): even lanes hold the pair (k, k + 1)
                            const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(y[r][u]), 0xB1, 0xf, 0xf, true));
                            unsigned p1, p2, p3;
                            split3(y[r][u], other, p1, p2, p3);
                            const int jj = lane + 64 * u;
                            if (!(lane & 1) && jj < I) {  // I is even
                                const int o = (lr * LDXj + jj) >> 1;
                                xb[o] = p1;
                                xb[PLANEj + o] = p2;
                                xb[2 * PLANEj + o] = p3;
                            }
                        }
                    }
                }
            }
            FP_T(1);
            if (prod) {
                fp_barrier();  // planes complete; every thread's reads of the previous output tile are done
                FP_T(2);
                if (ksj == 2) fp_product<TPW, KSM, 2>(W, bv, col, xb, obuf, H, tid);
                if constexpr (KSM >= 3) { if (ksj == 3) fp_product<TPW, KSM, 3>(W, bv, col, xb, obuf, H, tid); }
                if constexpr (KSM >= 5) { if (ksj == 5) fp_product<TPW, KSM, 5>(W, bv, col, xb, obuf, H, tid); }
                if constexpr (KSM >= 6) { if (ksj == 6) fp_product<TPW, KSM, 6>(W, bv, col, xb, obuf, H, tid); }
                // the prefetched tile is parked behind the last product of this tile (every wave is past the rows of this tile) and
                // BEFORE this sub-tile's stores are issued: vmcnt retires in order
                FP_T(3);
                if (s == N - 1 && nxt < ntile) park(nxt);
                FP_T(4);
                fp_barrier();  // output tile complete (and the parked tiles)
                FP_T(5);
                const int n4 = H >> 2, NP = H + 4;
                for (int idx = fp_opaque(tid); idx < FP_ROWS * n4; idx += 512) {
                    const int r = idx / n4, c4 = idx - r * n4;
                    const int gr = s * FP_ROWS + r;
                    const int tt = gr / N, k = gr - tt * N;
                    const int t = t0t + tt;
                    if (t < p.t1)
                        *reinterpret_cast<v4f*>(j.z + (((size_t)(t - p.t0) * B + b) * N + k) * j.ldz + c4 * 4) =
                            *reinterpret_cast<const v4f*>(&obuf[r * NP + c4 * 4]);
                }
                FP_T(6);
            }
        }
        if (!prod) {
            fp_barrier();  // every wave is done with the tiles
            if (nxt < ntile) park(nxt);
            fp_barrier();
            FP_T(4);
        }
    }
#ifdef FP_STAMPS
    if (blk == 0 && tid == 0)
        for (int k = 0; k < 8; ++k) fp_dbg[ji][k] = dbg_[k];
#endif
}

// =====================================================================================================
// host
// =====================================================================================================
static int fp_cu_count() {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
}

extern "C" int sfsn_features_proj(const float* stft_ri, const float* fb_tbf, int B, int F, int T, int FB, float fdrc,
                                  const sfsn_featproj_job* jobs, int n_jobs, int t0, int nt, float* zero_ptr, size_t zero_bytes,
                                  void* stream) {
    if (!stft_ri || !jobs || n_jobs <= 0 || n_jobs > FP_MAX_JOBS || B <= 0 || F < 2 || T <= 0 || FB < 0) return SFSN_EINVAL;
    if (t0 < 0 || nt <= 0 || t0 + nt > T) return SFSN_EINVAL;
    if ((zero_bytes != 0 && !zero_ptr) || (zero_bytes & 15) || (reinterpret_cast<uintptr_t>(zero_ptr) & 15)) return SFSN_EINVAL;
    static const bool no_bf3 = getenv("SFSN_INPROJ_F32") != nullptr;  // (the diagnostic switch of sfsn_input_proj_f32)
    if (no_bf3) return SFSN_EUNSUPPORTED;
    if (FB > 0 && (FP_TT * FB > 512 * FP_NFB || 512 % FB != 0)) return SFSN_EUNSUPPORTED;
    if ((double)B * F * T * 2 >= 2147483648.0 || (double)T * B * (FB > 0 ? FB : 1) >= 2147483648.0) return SFSN_EUNSUPPORTED;  // 32-bit offsets
    const int nf = F - 1;
    FpParams p;
    p.n = n_jobs; p.B = B; p.F = F; p.T = T; p.FB = FB; p.t0 = t0; p.t1 = t0 + nt; p.fdrc = fdrc;
    const int ntile = B * ((nt + FP_TT - 1) / FP_TT);
    double wt[FP_MAX_JOBS];
    size_t lds = 0;
    int tpw_max = 0, ks_max = 0, nu_max = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const sfsn_featproj_job& jb = jobs[i];
        const sfsn_feature_group& g = jb.feat;
        if (g.n_units <= 0 || g.ctr <= 0 || g.nbr < 0 || g.ctr_fb < 0 || g.nbr_fb < 0 || g.lo < 0) return SFSN_EINVAL;
        const int I1 = g.ctr + 2 * g.nbr, I2 = g.ctr_fb > 0 ? g.ctr_fb + 2 * g.nbr_fb : 0, I = I1 + I2;
        if (I > 256) return SFSN_EUNSUPPORTED;
        if (g.lo + g.n_units * g.ctr > nf || g.nbr >= nf || (I2 && (FB <= 0 || g.nbr_fb >= nf || !fb_tbf))) return SFSN_EINVAL;
        if ((g.norm == SFSN_NORM_LAYERNORM && (!g.ln_w || !g.ln_b)) || (g.norm == SFSN_NORM_LAPLACE && !g.mu) ||
            (g.norm == SFSN_NORM_GAUSSIAN && (!g.mu || !g.ln_w)))
            return SFSN_EINVAL;
        if (g.norm != SFSN_NORM_NONE && g.norm != SFSN_NORM_LAYERNORM && g.norm != SFSN_NORM_LAPLACE && g.norm != SFSN_NORM_GAUSSIAN)
            return SFSN_EINVAL;
        if (!jb.w && !g.x) return SFSN_EINVAL;  // a job must produce something
        FpJobDev& d = p.job[i];
        d.x = g.x; d.ln_w = g.ln_w; d.ln_b = g.ln_b; d.mu = g.mu; d.lo = g.lo; d.N = g.n_units; d.ctr = g.ctr; d.nbr = g.nbr;
        d.ctr_fb = g.ctr_fb; d.nbr_fb = g.nbr_fb; d.I1 = I1; d.I = I; d.norm = g.norm; d.eps = g.ln_eps;
        d.w = jb.w; d.bias = jb.bias; d.z = jb.z; d.H = jb.H; d.ldz = jb.ldz;
        // magnitude bins the group reads (reflected at both ends exactly as the kernel does)
        int fmin = nf, fmax = -1;
        const int ends[2] = {g.lo - g.nbr, g.lo + g.n_units * g.ctr - 1 + g.nbr};
        for (int e = 0; e < 2; ++e) {
            const int f = ends[e], r = f < 0 ? -f : (f > nf - 1 ? 2 * (nf - 1) - f : f);
            const int c = f < 0 ? 0 : (f > nf - 1 ? nf - 1 : f);
            const int lo_ = r < c ? r : c, hi_ = r > c ? r : c;
            if (lo_ < fmin) fmin = lo_;
            if (hi_ > fmax) fmax = hi_;
        }
        d.f_lo = fmin; d.f_cnt = fmax - fmin + 1;
        if (d.f_cnt > 16 * FP_NPRE) return SFSN_EUNSUPPORTED;
        size_t off = ((size_t)d.f_cnt * 33 * 4 + 15) & ~(size_t)15;
        d.off_fbT = (int)off;
        off += ((size_t)FP_TT * (FB > 0 ? FB : 1) * 4 + 15) & ~(size_t)15;
        d.off_offs = (int)off;
        off += ((size_t)d.N * I * 8 + 15) & ~(size_t)15;
        d.off_planes = d.off_obuf = (int)off;
        if (jb.w) {
            // what sfsn_input_proj_f32 runs on input_proj_bf3_kernel (the other shapes take its fp32-MFMA form: a different rounding)
            if (!jb.z || jb.H <= 0 || jb.ldz < jb.H || (reinterpret_cast<uintptr_t>(jb.z) & 15)) return SFSN_EINVAL;
            const int NT = (jb.H + 15) / 16, TPW = (NT + 7) / 8;
            const int KSB = I <= 64 ? 2 : (I <= 96 ? 3 : (I <= 160 ? 5 : 6));
            const long long M = (long long)nt * B * d.N;
            const bool inst = (TPW <= 3 && KSB <= 3) || TPW == 1 || (TPW == 2 && KSB == 5);  // input_proj_bf3_kernel's instantiations
            if (!inst || (I % 2) || I > 192 || (jb.H % 4) || (jb.ldz % 4) || M < 64 || TPW * KSB > 12) return SFSN_EUNSUPPORTED;
            d.NT = NT; d.tpw = TPW; d.ks = KSB; d.nu = (KSB * 32 + 63) / 64;
            const int LDX = KSB * 32 + 8;
            off += (size_t)3 * FP_ROWS * LDX * 2;
            d.off_obuf = (int)off;
            off += (size_t)FP_ROWS * (jb.H + 4) * 4;
            wt[i] = (double)ntile * d.N * (2.0 * TPW * KSB * 6 * 16 * 2 + 1800.0);
        } else {
            d.NT = 0; d.tpw = 0; d.ks = 2; d.nu = (I + 63) / 64;
            wt[i] = (double)ntile * d.N * 900.0;
        }
        if (off > lds) lds = off;
        if (d.tpw > tpw_max) tpw_max = d.tpw;
        if (jb.w && d.ks > ks_max) ks_max = d.ks;
        if (d.nu > nu_max) nu_max = d.nu;
    }
    // the kernel: the largest (column tiles per wave, k-steps) among the jobs; every job's rows must fit its feature slots
    if (tpw_max < 1) tpw_max = 1;
    if (ks_max < 2) ks_max = 2;
    if (nu_max == 2 && ks_max < 3) ks_max = 3;
    if (nu_max == 3 && ks_max < 5) ks_max = 5;
    if (nu_max > 3 || tpw_max * ks_max > 12 || (ks_max == 6 && tpw_max > 1) || (ks_max == 5 && tpw_max > 2)) return SFSN_EUNSUPPORTED;
    if (lds > 150 * 1024) return SFSN_EUNSUPPORTED;
    // persistent workgroups dealt in proportion to the jobs' estimated cycles
    static const int blocks_env = getenv("SFSN_FP_BLOCKS") ? atoi(getenv("SFSN_FP_BLOCKS")) : 0;
    const int per_cu = lds <= 78 * 1024 ? 2 : 1;
    const int total = blocks_env > 0 ? blocks_env : fp_cu_count() * per_cu;
    double sum = 0;
    for (int i = 0; i < n_jobs; ++i) sum += wt[i];
    int blocks = 0;
    for (int i = 0; i < n_jobs; ++i) {
        int nb = (int)(total * wt[i] / sum + 0.5);
        if (nb < 1) nb = 1;
        if (nb > ntile) nb = ntile;
        p.job[i].block0 = blocks;
        p.job[i].nblocks = nb;
        blocks += nb;
    }
    p.zero_block0 = blocks;
    p.zero_ptr = zero_ptr;
    p.zero_n16 = zero_bytes / 16;
    blocks += (int)((p.zero_n16 + 8 * 512 - 1) / (8 * 512));
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define FP_LAUNCH(TPW_, KSM_)                                                                                                    \
    if (tpw_max == TPW_ && ks_max == KSM_) {                                                                                     \
        static int lds_seen[64] = {0};                                                                                           \
        auto kern = featproj_kernel<TPW_, KSM_>;                                                                                 \
        if ((int)lds > lds_seen[dev]) {                                                                                          \
            if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                      \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)      \
                return SFSN_EHIP;                                                                                                \
            lds_seen[dev] = (int)lds;                                                                                            \
        }                                                                                                                        \
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, st, stft_ri, fb_tbf, p);                                \
        return hipGetLastError() == hipSuccess ? SFSN_OK : SFSN_EHIP;                                                            \
    }
    FP_LAUNCH(1, 2) FP_LAUNCH(2, 2) FP_LAUNCH(3, 2) FP_LAUNCH(1, 3) FP_LAUNCH(2, 3) FP_LAUNCH(3, 3) FP_LAUNCH(1, 5) FP_LAUNCH(2, 5)
    FP_LAUNCH(1, 6)
#undef FP_LAUNCH
    return SFSN_EUNSUPPORTED;
}

#ifdef FP_STAMPS
extern "C" int sfsn_fp_debug(unsigned long long* out /* [FP_MAX_JOBS][8] host */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fp_dbg), sizeof(unsigned long long) * FP_MAX_JOBS * 8) == hipSuccess ? 0 : -1;
}
#endif
