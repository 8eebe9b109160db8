// sfsn_projdf.hip -- the sub-band epilogue in ONE launch (round 6): projection S2 . W_p^T + b (MODEL:118 / FROZEN:125), output
// re-index + deep filter + Nyquist pass-through + |.| (MODEL:160-167,315-346,450-474; FROZEN:15-39,259-265,588-607).  gfx950 only.
//
// Why: sfsn_spike_proj_multi wrote the coefficient rows (295 MB per forward at B = 64, T = 1000, baseline_m) and sfsn_deepfilter read
// them straight back (674 MB per dispatch, the fattest time-parallel kernel); inside bench.py's timed region the time-parallel half of
// a forward is HBM-bound (profiles/r05_region_marginal_ledger.json), so bytes are time there.  Here the coefficient tile of a
// (clip, frame block, unit range) is formed on the int8 matrix cores, stays in LDS, is written out ONCE (it is `all_layer_outputs[-1]`
// of the module API; skipped when nobody reads it) and is applied to the noisy spectrum from LDS.
//
// Round 3 built this fusion with serial phases behind __syncthreads() in small non-persistent workgroups and measured it slower than the
// two launches (profiles/EXPERIMENTS.md).  What is different: persistent workgroups (the W_p tiles of a job stay in registers for the
// launch, as in spike_proj_fast_body), two LOADER waves per workgroup that keep the tile after next in flight for a whole tile's time
// (see projdf_body), raw barriers that do not drain stores, and a tile order that puts the two 16-frame halves of a 256-byte output
// run back to back in the same workgroup (the L2 merges them).
//
// Arithmetic: the projection is spike_proj_fast_body's instruction for instruction (three exact int8 digit products, recombine3,
// `* dq + bias`), the filter deepfilter_pass_kernel's expression for expression on the same fp32 coefficients: bit-identical to the
// two launches (tests/test_hip_parity.py::test_projection_and_deep_filter_in_one_launch_*).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "sfsn_feat_dev.h"
#include "sfsn_scan_dev.h"

#define PDF_MAX_JOBS 16
#define PDF_THREADS 512
#define PDF_CW 6     // compute waves (matrix phase, coefficient write-out, filter); waves 6 and 7 are the two loader waves
#define PDF_NV 16    // 16-byte spike vectors per LOADER lane and tile (two loaders: <= 2048 per tile)
#define PDF_NX 16    // spectrum elements per loader lane and tile (<= 2048 per tile)
#define PDF_RING 3   // LDS slots of (spike tile, spectrum tile): tile j + 2 is written while tile j is multiplied and filtered

struct PdfJobDev {
    const int8_t* s;   // [T][R][KP] int8 spikes of the group's last layer (frame 0 of the sequence)
    const int8_t* w;   // packed W_p
    const float* dq;
    const float* bias;
    float* y;          // [T][R][P] coefficient rows (API tensor) or nullptr
    int R, N;          // rows per frame of the group (B * N), units per clip
    int k0, U;         // this job's units [k0, k0 + U) of every clip
    int fc, df, lo;    // bins per unit, filter order, first bin of the GROUP
    int P, NT, NWN, tpw;
    int FT;            // frames per tile (16)
    int block0, nblocks;
    int kind;          // 0: projection + filter, 1: pass-through bins [fcov, F)
};
struct PdfParams {
    PdfJobDev job[PDF_MAX_JOBS];
    int n, B, F, T, S, t0, t1, fcov;
    const float* stft;
    float* enh;
    float* mag;
};

// Workgroup = six compute waves + two LOADER waves.  What the first form of this kernel measured (round 6, B = 64, T = 1000: 324 us per
// forward, the same as the two launches it replaces although it moves 0.3 GB less): every wave requested the next tile's operands at
// the top of a tile and parked them behind its matrix phase -- ~1 us later, against 2-3 us of HBM latency under load: 15 k clk per tile
// for ~5 k clk of work.  A wave that does nothing but load has a vmcnt queue of loads only (whatever the compiler's waits look like,
// they never wait for a store) and 200 registers to hold half a tile in flight for a whole tile's time: loader k requests its half of
// tile j + 2 at the top of tile j, crosses the tile's first barrier with the loads in flight, and writes them into ring slot (j + 2) % 3
// (free since tile j - 1's last barrier) before the tile's second barrier.
template <int TPW, int KS>
__device__ __forceinline__ void projdf_body(const PdfParams& p, const PdfJobDev& jb, int blk, char* smem) {
    constexpr int KP = KS * 64, SROW = KP + 16, C16 = KP / 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int FT = jb.FT, U = jb.U, P = jb.P, N = P, NT = jb.NT, NWN = jb.NWN;
    const int B = p.B, F = p.F, T = p.T, S = p.S, t1 = p.t1;
    const int MR = FT * U, MRT = (MR + 15) >> 4;
    const int LDF = U * P + 1;                       // odd: the 16 frames of a filter wave hit distinct banks
    const int nb = U * jb.fc, XW = FT + jb.df - 1, NXT = nb * XW, NVT = MR * C16;
    const int SBUF = MRT * 16 * SROW, XB = (NXT * 8 + 15) & ~15, SLOTB = SBUF + XB;
    float* obuf = reinterpret_cast<float*>(smem + PDF_RING * SLOTB);                   // [FT][LDF]
    int* rowoff = reinterpret_cast<int*>(obuf + ((FT * LDF + 3) & ~3));                // [MRT*16]: obuf offset of tile row m (or -1)

    const int ntile = (t1 - p.t0 + FT - 1) / FT;     // frame tiles per clip
    const int tiles = B * ntile;
    // a contiguous range of tiles per workgroup: consecutive frame tiles of one clip follow each other in the same workgroup (the two
    // 16-frame halves of a 256-byte run of the enhanced spectrum are written within microseconds of each other: the L2 merges them)
    const int tl0 = (int)((long long)tiles * blk / jb.nblocks), tl1 = (int)((long long)tiles * (blk + 1) / jb.nblocks);
    const size_t frame_s = (size_t)jb.R * KP;
    const int fbin0 = jb.lo + jb.k0 * jb.fc;

    for (int m = tid; m < MRT * 16; m += PDF_THREADS) {
        const int fl = m / U, u = m - fl * U;
        rowoff[m] = m < MR ? fl * LDF + u * P : -1;
    }

    if (wave >= PDF_CW) {
        // ================================================= loader wave k =================================================
        const int k = wave - PDF_CW;
        // my vectors of a tile: piece index 2 i + k (a piece = 64 consecutive 16-byte vectors / spectrum elements)
        int s_fl[PDF_NV], s_in[PDF_NV], s_lds[PDF_NV];
#pragma unroll
        for (int i = 0; i < PDF_NV; ++i) {
            int v = (2 * i + k) * 64 + lane;
            const bool have = v < NVT;
            if (v > NVT - 1) v = NVT - 1;
            const int r = v / C16, c16 = v - r * C16;
            const int fl = r / U, u = r - fl * U;
            s_fl[i] = fl;
            s_in[i] = u * KP + c16 * 16;
            s_lds[i] = have ? r * SROW + c16 * 16 : -1;
        }
        int x_jc[PDF_NX];                            // (bin row << 8) | column, or -1: no element
#pragma unroll
        for (int i = 0; i < PDF_NX; ++i) {
            const int e = (2 * i + k) * 64 + lane;
            const int ec = e < NXT ? e : NXT - 1;
            const int j = ec / XW, c = ec - j * XW;
            x_jc[i] = e < NXT ? (j << 8) | c : -1;
        }
        v4i pre[PDF_NV];
        float2 xpre[PDF_NX];
        bool x_ok[PDF_NX];
        auto fetch = [&](int tl) __attribute__((always_inline)) {
            const int b = tl / ntile, tb = p.t0 + (tl - b * ntile) * FT;
            const int flmax = t1 - 1 - tb;           // frames past the chunk re-read its last frame (their rows are never stored)
            const int8_t* sb = jb.s + (size_t)tb * frame_s + ((size_t)b * jb.N + jb.k0) * KP;
#pragma unroll
            for (int i = 0; i < PDF_NV; ++i) {
                // (no branch around a load: hipcc drains vmcnt(0) at every control-flow merge, which would leave ONE load in flight at a
                //  time -- slots beyond the tile re-read its last vector: same line for all 64 lanes, an issue slot and nothing else)
                const int fl = s_fl[i] < flmax ? s_fl[i] : flmax;
                pre[i] = *reinterpret_cast<const v4i*>(sb + (size_t)fl * frame_s + s_in[i]);
            }
            const float* xb = p.stft + ((size_t)b * F + fbin0) * T * 2;
            const int ts0 = tb - (jb.df - 1);
#pragma unroll
            for (int i = 0; i < PDF_NX; ++i) {
                const int jc = x_jc[i] < 0 ? 0 : x_jc[i];
                const int ts = ts0 + (jc & 255);
                const int tsc = ts < 0 ? 0 : (ts > T - 1 ? T - 1 : ts);
                xpre[i] = *reinterpret_cast<const float2*>(xb + ((size_t)(jc >> 8) * T + tsc) * 2);
                x_ok[i] = ts >= 0 && ts < T;
            }
        };
        auto park = [&](int slot) __attribute__((always_inline)) {
            char* sl = smem + slot * SLOTB;
#pragma unroll
            for (int i = 0; i < PDF_NV; ++i)
                if (s_lds[i] >= 0) *reinterpret_cast<v4i*>(sl + s_lds[i]) = pre[i];
            float2* xt = reinterpret_cast<float2*>(sl + SBUF);
#pragma unroll
            for (int i = 0; i < PDF_NX; ++i)
                if (x_jc[i] >= 0) xt[(2 * i + k) * 64 + lane] = x_ok[i] ? xpre[i] : make_float2(0.0f, 0.0f);
        };
        // prologue: the first two tiles
        for (int d = 0; d < 2; ++d)
            if (tl0 + d < tl1) {
                fetch(tl0 + d);
                park(d);
            }
        __syncthreads();
        int slot = 0;
        for (int tl = tl0; tl < tl1; ++tl) {
            const bool more = tl + 2 < tl1;
            if (more) fetch(tl + 2);
            __builtin_amdgcn_s_barrier();            // (the tile's first barrier: my loads stay in flight across it)
            if (more) park(slot == 0 ? 2 : slot - 1);  // = (slot + 2) % 3: free since the last barrier of tile tl - 1
            slot = slot == 2 ? 0 : slot + 1;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ================================================= compute waves =================================================
    const int MW = PDF_CW / NWN;
    const int cg = wave % NWN, mw = wave / NWN;
    const bool worker = mw < MW;
    // ---- W_p tiles of this wave -> registers, once (spike_proj_fast_body's deal over six waves)
    v4i W[TPW][KS][3];
    v4f dqv[TPW], bv[TPW];
    int col[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int ct = cg + NWN * i;
        const bool have = worker && ct < NT;
        col[i] = have ? ct * 16 + q * 4 : -1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const size_t tile = (size_t)d * NT + (have ? ct : 0);
                W[i][ks][d] = *reinterpret_cast<const v4i*>(jb.w + ((tile * KS + ks) * 64 + lane) * 16);
            }
        dqv[i] = *reinterpret_cast<const v4f*>(jb.dq + (have ? col[i] : 0));
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[i][r] = (jb.bias && have && col[i] + r < N) ? jb.bias[col[i] + r] : 0.0f;
    }
    __syncthreads();
    constexpr int CT = PDF_CW * 64;                  // compute threads
    const int tt = tid & (FT - 1), slot0 = tid / FT, nslot = CT / FT;
    const int Q = (U * P) >> 2;                      // 16-byte units of a frame's coefficient rows
    int slot = 0;
    for (int tl = tl0; tl < tl1; ++tl) {
        const int b = tl / ntile, tb = p.t0 + (tl - b * ntile) * FT;
        const char* sl = smem + slot * SLOTB;
        const int8_t* sbuf = reinterpret_cast<const int8_t*>(sl);
        const float2* xc = reinterpret_cast<const float2*>(sl + SBUF);
        // ---- matrix phase: coefficient tile -> obuf
        if (worker) {
            for (int mi = mw; mi < MRT; mi += MW) {
                const int8_t* sr = sbuf + (mi * 16 + n) * SROW + q * 16;
                v4i bfr[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) bfr[ks] = *reinterpret_cast<const v4i*>(sr + ks * 64);
                const int ro = rowoff[mi * 16 + n];
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    if (col[i] < 0) continue;
                    v4i a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[i][ks][0], bfr[ks], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[i][ks][1], bfr[ks], a1, 0, 0, 0);
                        a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[i][ks][2], bfr[ks], a2, 0, 0, 0);
                    }
                    if (ro >= 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (col[i] + r < N) obuf[ro + col[i] + r] = recombine3(a0[r], a1[r], a2[r]) * dqv[i][r] + bv[i][r];
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        // ---- the coefficient rows leave once (the module API's last `all_layer_outputs` entry): a wave per frame, whole rows
        if (jb.y) {
            for (int fl = wave; fl < FT; fl += PDF_CW) {
                const int t = tb + fl;
                if (t >= t1) continue;
                float* yp = jb.y + (((size_t)t * jb.R) + (size_t)b * jb.N + jb.k0) * P;
                const float* op = obuf + fl * LDF;
                for (int c4 = lane; c4 < Q; c4 += 64) {
                    const v4f o = {op[4 * c4], op[4 * c4 + 1], op[4 * c4 + 2], op[4 * c4 + 3]};
#if SFSN_NT_OUT
                    __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(yp + 4 * c4));
#else
                    *reinterpret_cast<v4f*>(yp + 4 * c4) = o;
#endif
                }
            }
        }
        // ---- deep filter from LDS (deepfilter_pass_kernel's expressions)
        {
            const int t = tb + tt;
            if (t < t1) {
                const float* pr = obuf + tt * LDF;
                int u = 0, fci = slot0;
                while (fci >= jb.fc) { fci -= jb.fc; ++u; }
                for (int j = slot0; j < nb; j += nslot) {
                    const int f = fbin0 + u * jb.fc + fci;
                    const float* pu = pr + u * P;
                    for (int s_ = 0; s_ < S; ++s_) {
                        float yr = 0.0f, yi = 0.0f;
                        const float2* xr_ = xc + j * XW + tt;
                        for (int d = 0; d < jb.df; ++d) {
                            const float2 xv = xr_[d];
                            const float cr = pu[((0 * jb.fc + fci) * jb.df + d) * S + s_];
                            const float ci = pu[((1 * jb.fc + fci) * jb.df + d) * S + s_];
                            yr += xv.x * cr - xv.y * ci;
                            yi += xv.x * ci + xv.y * cr;
                        }
                        const size_t o = (((size_t)b * S + s_) * F + f) * T + t;
                        // (plain stores: the two 16-frame halves of a 256-byte run are written by consecutive tiles and merge in the L2 --
                        //  as non-temporal stores the lean strict forward measured 1 % slower)
                        *reinterpret_cast<float2*>(p.enh + 2 * o) = make_float2(yr, yi);
                        if (p.mag) p.mag[o] = fast_abs2(yr, yi);
                    }
                    fci += nslot;
                    while (fci >= jb.fc) { fci -= jb.fc; ++u; }
                }
            }
        }
        slot = slot == 2 ? 0 : slot + 1;
        // obuf and this tile's ring slot are rewritten next: LDS reads done, stores stay in flight (raw barrier)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    }
}

// bins no group covers (at least Nyquist) pass through untouched (MODEL:461-470): (clip, 64-frame tile) units over the job's blocks
__device__ __forceinline__ void passthrough_body(const PdfParams& p, const PdfJobDev& jb, int blk) {
    const int tid = threadIdx.x, tt = tid & 63, fs = tid >> 6;
    const int ntile = (p.t1 - p.t0 + 63) / 64, tiles = p.B * ntile;
    for (int tl = blk; tl < tiles; tl += jb.nblocks) {
        const int b = tl / ntile, t = p.t0 + (tl - b * ntile) * 64 + tt;
        if (t >= p.t1) continue;
        for (int f = p.fcov + fs; f < p.F; f += PDF_THREADS / 64) {
            const float2 xv = *reinterpret_cast<const float2*>(p.stft + (((size_t)b * p.F + f) * p.T + t) * 2);
            for (int s_ = 0; s_ < p.S; ++s_) {
                const size_t o = (((size_t)b * p.S + s_) * p.F + f) * p.T + t;
                *reinterpret_cast<float2*>(p.enh + 2 * o) = xv;
                if (p.mag) p.mag[o] = fast_abs2(xv.x, xv.y);
            }
        }
    }
}

template <int KS>
__global__ __launch_bounds__(PDF_THREADS) void projdf_kernel(const PdfParams p) {
    extern __shared__ __attribute__((aligned(16))) char pdf_smem[];
    int j = 0;
    for (int i = 1; i < p.n; ++i)
        if ((int)blockIdx.x >= p.job[i].block0) j = i;
    const PdfJobDev& jb = p.job[j];
    const int blk = (int)blockIdx.x - jb.block0;
    if (jb.kind == 1) passthrough_body(p, jb, blk);
    else if (jb.tpw == 1) projdf_body<1, KS>(p, jb, blk, pdf_smem);
    else if (jb.tpw == 2) projdf_body<2, KS>(p, jb, blk, pdf_smem);
    else projdf_body<3, KS>(p, jb, blk, pdf_smem);
}

// =====================================================================================================
// host side
// =====================================================================================================
static inline bool pdf_aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }
static int pdf_cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
}

static size_t pdf_lds_bytes(int KS, int FT, int U, int P, int fc, int df) {
    const int KP = KS * 64, SROW = KP + 16, MRT = (FT * U + 15) / 16;
    const size_t slot = (size_t)MRT * 16 * SROW + (((size_t)U * fc * (FT + df - 1) * 8 + 15) & ~(size_t)15);
    const size_t obuf = (size_t)((FT * (U * P + 1) + 3) & ~3) * 4;
    const size_t rows = (size_t)MRT * 16 * 4;
    return PDF_RING * slot + obuf + rows;
}

extern "C" int sfsn_proj_deepfilter(const float* stft_ri, int B, int F, int T, int S, int H, const sfsn_projdf_group* groups, int n_groups,
                                    float* enh_ri, float* enh_mag, int t0, int nt, void* stream) {
    if (!stft_ri || !enh_ri || !groups || n_groups <= 0 || n_groups > SFSN_MAX_GROUPS || B <= 0 || F < 2 || T <= 0 || S <= 0 || H <= 0)
        return SFSN_EINVAL;
    if (t0 < 0 || nt <= 0 || t0 + nt > T) return SFSN_EINVAL;
    const int KS = (H + 63) / 64;
    if (KS > 4) return SFSN_EUNSUPPORTED;
    static const int ft_env = getenv("SFSN_PDF_FT") ? atoi(getenv("SFSN_PDF_FT")) : 0;        // A/B runs: frames per tile
    static const int wgs_env = getenv("SFSN_PDF_WGS") ? atoi(getenv("SFSN_PDF_WGS")) : 0;     // A/B runs: workgroups of the launch
    static const size_t lds_cap_env = getenv("SFSN_PDF_LDS_KB") ? (size_t)atoi(getenv("SFSN_PDF_LDS_KB")) * 1024 : (size_t)150 * 1024;
    PdfParams p;
    p.n = 0; p.B = B; p.F = F; p.T = T; p.S = S; p.t0 = t0; p.t1 = t0 + nt; p.stft = stft_ri; p.enh = enh_ri; p.mag = enh_mag;
    double wt[PDF_MAX_JOBS];
    int tiles_of[PDF_MAX_JOBS];
    size_t lds = 0;
    int lo = 0;
    for (int i = 0; i < n_groups; ++i) {
        const sfsn_projdf_group& g = groups[i];
        if (!g.spikes_i8 || !g.w_packed || !g.w_dq || g.n_units <= 0 || g.fc <= 0 || g.df <= 0) return SFSN_EINVAL;
        if (!pdf_aligned16(g.spikes_i8) || !pdf_aligned16(g.w_packed) || !pdf_aligned16(g.w_dq) || !pdf_aligned16(g.proj)) return SFSN_EINVAL;
        const int P = 2 * g.fc * g.df * S, NT = (P + 15) / 16;
        if (P % 4) return SFSN_EUNSUPPORTED;
        int TPW = (NT + PDF_CW - 1) / PDF_CW;
        if (TPW > 3) return SFSN_EUNSUPPORTED;  // (P <= 288)
        const int NWN = (NT + TPW - 1) / TPW;
        // units per job and frames per tile: the largest that fit the per-thread prefetch slots and the LDS budget
        int FT = 16, U = g.n_units;
        (void)ft_env;
        auto fits = [&](int ft, int u) {
            return ft * u * (KS * 4) <= PDF_NV * 128 && u * g.fc * (ft + g.df - 1) <= PDF_NX * 128 && g.df <= 200 &&
                   pdf_lds_bytes(KS, ft, u, P, g.fc, g.df) <= lds_cap_env;
        };
        while (U > 1 && !fits(FT, U)) {
            const int np = (g.n_units + U - 1) / U + 1;  // one more pass
            U = (g.n_units + np - 1) / np;
        }
        if (!fits(FT, U)) return SFSN_EUNSUPPORTED;
        for (int k0 = 0; k0 < g.n_units; k0 += U) {
            if (p.n >= PDF_MAX_JOBS - 1) return SFSN_EUNSUPPORTED;
            PdfJobDev& d = p.job[p.n];
            d.s = g.spikes_i8; d.w = g.w_packed; d.dq = g.w_dq; d.bias = g.bias; d.y = g.proj;
            d.R = B * g.n_units; d.N = g.n_units; d.k0 = k0; d.U = (g.n_units - k0 < U) ? g.n_units - k0 : U;
            d.fc = g.fc; d.df = g.df; d.lo = lo; d.P = P; d.NT = NT; d.NWN = NWN; d.tpw = TPW; d.FT = FT; d.kind = 0;
            const size_t l = pdf_lds_bytes(KS, FT, d.U, P, g.fc, g.df);
            if (l > lds) lds = l;
            tiles_of[p.n] = B * ((nt + FT - 1) / FT);
            // bytes a tile moves: spikes in, coefficient rows out, spectrum in, enhanced spectrum + magnitude out
            wt[p.n] = (double)tiles_of[p.n] * ((double)FT * d.U * (KS * 64 + (d.y ? P * 4.0 : 0.0)) + (double)d.U * g.fc * FT * (8.0 + S * 12.0));
            ++p.n;
        }
        lo += g.n_units * g.fc;
    }
    if (lo > F) return SFSN_EINVAL;
    p.fcov = lo;
    const int n_cu = pdf_cu_count();
    // one persistent workgroup per compute unit (the ring and the coefficient tile take most of a unit's LDS).  Measured with the loader
    // waves, B = 64, T = 1000: 256 / 512 / 1024 workgroups 238 / 246 / 296 us per forward (every workgroup reloads its W_p tiles)
    int total = wgs_env > 0 ? wgs_env : n_cu;
    int n_pass = 0;
    if (lo < F) {  // the pass-through bins: a few workgroups of their own
        PdfJobDev& d = p.job[p.n];
        d = PdfJobDev{};
        d.kind = 1;
        const int tiles = B * ((nt + 63) / 64);
        n_pass = tiles < 8 ? tiles : 8;
        tiles_of[p.n] = tiles;
        wt[p.n] = 0;
        ++p.n;
    }
    // block ranges in proportion to the jobs' bytes, at least one and at most `tiles` each
    {
        double sum = 0;
        for (int i = 0; i < p.n; ++i) sum += wt[i];
        int blocks = 0;
        for (int i = 0; i < p.n; ++i) {
            int nbk = p.job[i].kind == 1 ? n_pass : (int)((total - n_pass) * wt[i] / sum + 0.5);
            if (nbk < 1) nbk = 1;
            if (nbk > tiles_of[i]) nbk = tiles_of[i];
            p.job[i].block0 = blocks; p.job[i].nblocks = nbk;
            blocks += nbk;
        }
        total = blocks;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
#define PDF_CASE(KS_)                                                                                                      \
    if (KS == KS_) {                                                                                                       \
        auto kern = projdf_kernel<KS_>;                                                                                    \
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                    \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)    \
            return SFSN_EHIP;                                                                                              \
        hipLaunchKernelGGL(kern, dim3(total), dim3(PDF_THREADS), lds, st, p);                                              \
        return hipGetLastError() == hipSuccess ? SFSN_OK : SFSN_EHIP;                                                      \
    }
    PDF_CASE(1) PDF_CASE(2) PDF_CASE(3) PDF_CASE(4)
#undef PDF_CASE
    return SFSN_EUNSUPPORTED;
}
