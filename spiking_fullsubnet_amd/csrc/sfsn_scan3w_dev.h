// sfsn_scan3w_dev.h -- the IO-wave scan for hidden sizes that need TWO output tiles per compute wave (256 < H <= 320: the full-band
// model of baseline_m / l / xl, 17..20 tiles), round 5, gfx950 only.
//
// Round 2's scan_body runs H = 320 as 8 waves x 3 / 2 tiles with every wave fetching its own input term, flushing its share of the
// spikes and waiting on one in-order vmcnt queue: 1.28-1.30 us per step at 4 rows per workgroup (3050 clk) although the matrix pipe
// holds only 20 tiles x 15 instructions x 16 clk / 4 SIMDs = 1200 clk of it.  This is sfsn_scan3_dev.h's structure for that size:
//   compute waves (waves 0..9): tile `wave` and, for wave < NT - 10, tile `10 + wave`; digit planes 1 and 2 of W_hh register
//       resident (80 VGPRs for two tiles), plane 0 in LDS as A fragments (NT x KS KiB, 1 KiB contiguous per wave instruction: conflict
//       free); B fragments from the int8 state buffer, 15 matrix instructions per tile, the live values re-dealt with DPP row shifts
//       (1 / 2 values per lane at 4 / 8 rows), per-neuron constants in registers, new spikes to LDS, one barrier per step.  No
//       global memory instruction in the loop;
//   the loader wave (wave 10): the input term of frame t + D - 1 -> LDS ring (LDS-DMA, only loads -- and, at 8 rows, its share of
//       the fp32 spike stores -- in its vmcnt queue); the only wave that polls the producers' counters in a stack launch;
//   the storer wave (wave 11): spikes of frame t - 1, LDS -> global as whole contiguous blocks; write-through + progress counter
//       when another role of the launch reads them;
//   (SFSN_S3W_IOTILES = 1, measured and OFF: both IO waves also own one output tile, 5 / 5 / 5 / 5 tiles per SIMD instead of
//       6 / 6 / 4 / 4 -- slower, see the switch below.)
// 12 waves = 768 threads (three per SIMD: 170 registers each).  Arithmetic is scan_body's value for value (three exact integer
// accumulators recombined by recombine3 -- the integer form (a2 << 16) + (a1 << 8) + a0 of the H <= 256 role could leave int32 at
// K = 320 --, same fma / exp2 / rcp sequence): bit-identical outputs (tests/test_stack_scan.py).  4 or 8 rows per workgroup, shared
// gate weights.
#ifndef SFSN_SCAN3W_DEV_H
#define SFSN_SCAN3W_DEV_H
#include "sfsn_scan3_dev.h"

#ifndef SFSN_S3W_IOTILES
// 1: the two IO waves also own an output tile each (5 / 5 / 5 / 5 tiles per SIMD instead of 6 / 6 / 4 / 4).  Measured, H = 320, B = 64,
// T = 1000, the stack alone: 1.21 / 1.48 ms at 4 / 8 rows against 1.16 / 1.34 without (round 2's body: 1.30-1.32 / 1.60-1.62) -- an IO
// wave's chain (its DMAs / stores, then its tile, then its waits) becomes the step.  Off.
#define SFSN_S3W_IOTILES 0
#endif
// Ring depth (frames) of a gated / publishing role and the frames of a publishing storer's stores that may be in flight: both are part
// of the lag between linked layers (a consumer starts D - 1 + lag frames behind what its producer has PUBLISHED, and a producer
// publishes PF frames behind what it has stored).  Measured (scripts/exp_fb3_r05.py, B = 64, T = 1000, the stack alone at 4 rows):
// (9, 8) 1.157 ms, (5, 8) 1.167, (5, 4) 1.152, (4, 3) 1.088 -- the shallow ring wins even alone on the chip.
#ifndef SFSN_S3W_DG
#define SFSN_S3W_DG 4
#endif
#ifndef SFSN_S3W_PF
#define SFSN_S3W_PF 3
#endif
#ifndef SFSN_S3W_DP
#define SFSN_S3W_DP SFSN_S3W_DG  // ring of a role that publishes but is not gated itself (its depth is no part of any lag)
#endif
template <int KS, int RPW, int FLG = 0>
struct Scan3wCfg {
    static constexpr int NTHR = 768, NWAVES = 12, NTMAX = 20;
    static constexpr int HP = KS * 64, LDH = HP + 32;
    __host__ __device__ static constexpr int chunks(int NT) { return RPW * NT * 4; }
    __host__ __device__ static constexpr int pieces(int NT) { return (chunks(NT) + 63) / 64; }
    __host__ __device__ static constexpr int slot_bytes(int NT) { return pieces(NT) * 1024; }
    static constexpr int MAXP = (RPW * NTMAX * 4 + 63) / 64;  // pieces at NT = 20: 5 / 10 at 4 / 8 rows
    static constexpr bool GATED = (FLG & 1) != 0, PUB = (FLG & 2) != 0;
    static constexpr int DWANT = GATED ? SFSN_S3W_DG : (PUB ? SFSN_S3W_DP : 6);
    // what the 160 KiB leave beside the digit plane (NTMAX x KS KiB) and the state buffers; LDS-DMA destinations stay below 64 KiB
    static constexpr int ROOM = 160 * 1024 - 512 - NTMAX * KS * 1024 - 2 * 16 * LDH;
    static constexpr int DFIT = (ROOM < 65536 ? ROOM : 65536) / (MAXP * 1024);
    static constexpr int D = DWANT < DFIT ? DWANT : DFIT;  // 9 at 4 rows, 4 at 8 rows (H = 320)
    static_assert(D >= 3, "ring");
    __host__ __device__ static constexpr int hbuf_off(int NT) { return D * slot_bytes(NT); }
    __host__ __device__ static constexpr int plane_off(int NT) { return hbuf_off(NT) + 2 * 16 * LDH; }
    __host__ __device__ static constexpr int flag_off(int NT) { return plane_off(NT) + NT * KS * 1024; }
    __host__ __device__ static constexpr int lds_bytes(int NT) { return flag_off(NT) + 16; }
    static constexpr int NV = RPW == 8 ? 2 : 1;  // live values per lane and tile
};

// fp32 spikes of a frame for roles with up to 20 tiles: S3FlushF with its store-instruction count sized for NTMAX tiles
template <int RPW, int LDH>
struct S3wFlushF {
    static constexpr int MAXF = (RPW * 20 * 4 + 63) / 64;
    int lf[MAXF];
    unsigned okf;
    int nsf;
    __device__ __forceinline__ void init(int lane, int row0, int R, int H, int k_lo = 0, int k_hi = MAXF) {
        const int q4 = H / 4;
        const int rows_live = (R - row0 < RPW) ? R - row0 : RPW;
        const int nall = (rows_live * q4 + 63) / 64;
        const int hi = k_hi < nall ? k_hi : nall, lo = k_lo < hi ? k_lo : hi;
        nsf = hi - lo;
        okf = 0;
#pragma unroll
        for (int k = 0; k < MAXF; ++k) {
            const int u = 64 * k + lane, rr = u / q4, c4 = u - rr * q4;
            lf[k] = rr * LDH + c4 * 4;
            if (k >= lo && k < hi && u < RPW * q4 && row0 + rr < R) okf |= 1u << k;
        }
    }
    __device__ __forceinline__ void run(const int8_t* hsrc, float* pf, int lane) const {
#pragma unroll
        for (int k = 0; k < MAXF; ++k) {
            if ((okf >> k) & 1u) {
                const unsigned pk = *reinterpret_cast<const unsigned*>(hsrc + lf[k]);
                const v4f sp = {(float)(pk & 0xffu), (float)((pk >> 8) & 0xffu), (float)((pk >> 16) & 0xffu), (float)(pk >> 24)};
#if SFSN_S3_NT
                __builtin_nontemporal_store(sp, reinterpret_cast<v4f*>(pf + (size_t)(64 * k + lane) * 4));
#else
                *reinterpret_cast<v4f*>(pf + (size_t)(64 * k + lane) * 4) = sp;
#endif
            }
        }
    }
};

// The tiles of one wave: NTL (1 or 2) output tiles with digit planes 1 and 2 of their W_hh rows in registers, their slice of the
// membrane and their per-neuron constants.  `step` = the matrix instructions and the epilogue of one time step for these tiles
// (straight-line code: NTL is compile-time).  Used by the compute waves (two tiles, or one) AND by the two IO waves (one tile each:
// with 20 tiles on 10 compute waves two SIMDs carried six tiles and two four; matrix pipe and VALU of a SIMD largely serialise, so
// the six-tile SIMDs were the step -- 5 / 5 / 5 / 5 with the IO waves' tiles).
template <int KS, int RPW, int FLG, int NTL>
struct S3wTiles {
    using C = Scan3wCfg<KS, RPW, FLG>;
    static constexpr int LDH = C::LDH, NV = C::NV;
    v4i W[NTL][KS][2];
    float c[NTL][NV], dq[NTL][NV], db[NTL][NV], al[NTL][NV], be[NTL][NV];
    unsigned zoff[NTL], hoff[NTL], woff[NTL];
    int cj[NTL];
    unsigned boff;
    int grow;
    bool live;
    __device__ __forceinline__ void init(const Scan3Role& rl, int H, int NT, const int (&tiles)[NTL], int lane) {
        const int n = lane & 15, q = lane >> 4;
        const int row = RPW == 8 ? (n & 7) : (n & 3);
        const int sub = RPW == 8 ? 2 * (n >> 3) : (n >> 2);  // first of my NV neurons within the 4q group
        live = rl.row0 + row < rl.R;
        grow = live ? rl.row0 + row : rl.R - 1;
        boff = (unsigned)(n * LDH + q * 16);
#pragma unroll
        for (int i = 0; i < NTL; ++i) {
            const int ct = tiles[i];
            cj[i] = ct * 16 + q * 4 + sub;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int d = 0; d < 2; ++d)
                    W[i][ks][d] = *reinterpret_cast<const v4i*>(rl.w_hh + ((((size_t)(d + 1) * NT + ct) * KS + ks) * 64 + lane) * 16);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                c[i][j] = rl.c_state[(size_t)grow * H + cj[i] + j];
                dq[i][j] = rl.w_dq[cj[i] + j];
                db[i][j] = rl.bias[H + cj[i] + j] - rl.bias[cj[i] + j];
                al[i][j] = rl.bn_alpha[cj[i] + j];
                be[i][j] = rl.bn_beta[cj[i] + j];
            }
            zoff[i] = (unsigned)(((ct * 4 + q) * RPW + row) * 16 + sub * 4);  // my input-term bytes within a ring slot
            hoff[i] = (unsigned)(row * LDH + cj[i]);
            woff[i] = (unsigned)(((ct * KS) * 64 + lane) * 16);               // my fragment of the LDS digit plane, k-step 0
        }
    }
    // one time step: state fragments from hc, input term from the ring slot zs, digit plane 0 from wplane; new spikes to hn
    __device__ __forceinline__ void step(const int8_t* hc, int8_t* hn, const char* zs, const char* wplane) {
        v4i b[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b[ks] = *reinterpret_cast<const v4i*>(hc + boff + ks * 64);
#pragma unroll
        for (int i = 0; i < NTL; ++i) {
            float z[NV];
            if constexpr (NV == 2) {
                typedef float v2f_ __attribute__((ext_vector_type(2)));
                const v2f_ zz = *reinterpret_cast<const v2f_*>(zs + zoff[i]);
                z[0] = zz.x; z[1] = zz.y;
            } else {
                z[0] = *reinterpret_cast<const float*>(zs + zoff[i]);
            }
            v4i a[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v4i w0 = *reinterpret_cast<const v4i*>(wplane + woff[i] + ks * 1024);
                a[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[i][ks][0], b[ks], a[1], 0, 0, 0);
                a[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[i][ks][1], b[ks], a[2], 0, 0, 0);
                a[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, b[ks], a[0], 0, 0, 0);
            }
            int v[3][NV];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if constexpr (RPW == 8) {
                    // columns 0..7 are live: lanes 8..15 of a row of 16 take elements 2, 3 of the lane 8 below them
                    v[d][0] = __builtin_amdgcn_update_dpp(a[d][0], a[d][2], 0x118, 0xf, 0xC, false);
                    v[d][1] = __builtin_amdgcn_update_dpp(a[d][1], a[d][3], 0x118, 0xf, 0xC, false);
                } else {
                    int x = a[d][0];
                    x = __builtin_amdgcn_update_dpp(x, a[d][1], 0x114, 0xf, 0x2, false);  // row_shr:4  -> lanes 4..7
                    x = __builtin_amdgcn_update_dpp(x, a[d][2], 0x118, 0xf, 0x4, false);  // row_shr:8  -> lanes 8..11
                    x = __builtin_amdgcn_update_dpp(x, a[d][3], 0x11C, 0xf, 0x8, false);  // row_shr:12 -> lanes 12..15
                    v[d][0] = x;
                }
            }
            unsigned pk = 0;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                // K up to 320: |a1 * 256 + a0| < 2^24 is exact as a float, |a2| < 2^16 too, the fma rounds the sum once (recombine3:
                // the integer form (a2 << 16) + (a1 << 8) + a0 of the H <= 256 role could overflow int32 here)
                const float rec = recombine3(v[0][j], v[1][j], v[2][j]);
                const float pre_f = __builtin_fmaf(rec, dq[i][j], z[j]);
                const float pre_g = pre_f + db[i][j];
                const float f = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre_f * -1.44269504088896341f));
                const float m = __builtin_fmaf(f, c[i][j] - pre_g, pre_g);
                const float y = __builtin_fmaf(m, al[i][j], be[i][j]);
                c[i][j] = y;
                pk |= (y >= 0.0f) ? (1u << (8 * j)) : 0u;
            }
            if constexpr (NV == 2) *reinterpret_cast<unsigned short*>(hn + hoff[i]) = (unsigned short)pk;
            else hn[hoff[i]] = (int8_t)pk;
        }
    }
    // final state of my neurons (hl = h_{T-1})
    __device__ __forceinline__ void finish(const Scan3Role& rl, int H, const int8_t* hl) const {
        if (live) {
#pragma unroll
            for (int i = 0; i < NTL; ++i)
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    rl.c_state[(size_t)grow * H + cj[i] + j] = c[i][j];
                    rl.h_state[(size_t)grow * H + cj[i] + j] = (float)hl[hoff[i] + j];
                }
        }
    }
};

// The compute wave's loop for NTL (1 or 2) tiles.
template <int KS, int RPW, int FLG, int NTL>
__device__ __forceinline__ void scan3w_compute(const Scan3Role& rl, char* smem, int T, int H, int NT, const int (&tiles)[NTL], int lane) {
    using C = Scan3wCfg<KS, RPW, FLG>;
    constexpr int LDH = C::LDH, D = C::D;
    constexpr bool GATED = C::GATED;
    const int SLOT = C::slot_bytes(NT);
    int8_t* hbuf = reinterpret_cast<int8_t*>(smem + C::hbuf_off(NT));
    const char* wplane = smem + C::plane_off(NT);
    volatile int* flag = reinterpret_cast<volatile int*>(smem + C::flag_off(NT));
    S3wTiles<KS, RPW, FLG, NTL> ts;
    ts.init(rl, H, NT, tiles, lane);
    __syncthreads();                       // initial state in hbuf[0], digit plane 0 in LDS
    __builtin_amdgcn_s_barrier();          // the loader's prologue frames have landed
    int stop = 0;
    S3_PB_DECL();
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        if constexpr (GATED) stop = flag[t & 1];  // written by the loader during step t-1 (or before)
        ts.step(hbuf + (t & 1) * 16 * LDH, hbuf + ((t & 1) ^ 1) * 16 * LDH, smem + (t % D) * SLOT, wplane);
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        S3_PB_TIC();
        __builtin_amdgcn_s_barrier();
        S3_PB_TOC(0);
        if constexpr (GATED) if (__builtin_amdgcn_readfirstlane(stop)) break;
    }
    S3_PB_OUT(rl, tiles[0], lane);
    ts.finish(rl, H, hbuf + (T & 1) * 16 * LDH);
}

// FLG bit 0: the input term is written by other workgroups of this launch (gated on lk.in, sc1 loads); bit 1: the int8 spikes feed
// other workgroups of this launch (sc1 stores, progress in lk.out).  OUT bit 0: fp32 spikes, bit 1: int8 spikes.  768 threads.
template <int KS, int RPW, int OUT, int FLG>
__device__ __forceinline__ void scan3w_role(const Scan3Role& rl, const StackLink& lk, char* smem, int T, int H, int NT) {
    using C = Scan3wCfg<KS, RPW, FLG>;
    constexpr int LDH = C::LDH, HP = C::HP, D = C::D, NTHR = C::NTHR;
    constexpr bool GATED = C::GATED, PUB = C::PUB;
    // the loader wave writes fp32 spikes too: all of them in a publishing 4-row role (round 3), its share of every frame at 8 rows
#ifndef SFSN_S3W_LSPLIT4
// 1: the fp32 stores are split between the IO waves at 4 rows too (instead of all on the loader wave of a publishing role, round 3's
// rule): the strict forward 2.93-2.97 -> 2.86-2.89 ms (the loader wave of a gated / publishing role is what suffers beside the pair launch)
#define SFSN_S3W_LSPLIT4 1
#endif
    constexpr bool LSPLIT = (RPW == 8 || SFSN_S3W_LSPLIT4) && (OUT & 1);
    constexpr bool LSF = (OUT & 1) && ((SFSN_S3_LSF && PUB) || LSPLIT);
    const int ltake = LSPLIT ? rl.lsplit : 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int R = rl.R, row0 = rl.row0;
    // tiles: compute wave w (0..9) owns tile w and, for w < NT - 10, tile 10 + w (SFSN_S3W_IOTILES: 12 + w, tiles 10 / 11 with the IO waves)
    constexpr int NCW = 10;
    const int SLOT = C::slot_bytes(NT);
    const char* wplane = smem + C::plane_off(NT);
    int8_t* hbuf = reinterpret_cast<int8_t*>(smem + C::hbuf_off(NT));
    volatile int* flag = reinterpret_cast<volatile int*>(smem + C::flag_off(NT));

    // ---- set-up by all threads: state buffers zeroed (pad rows / columns must read as 0 spikes), digit plane 0 -> LDS, h_{-1} -> hbuf[0]
    for (int i = tid; i < 2 * 16 * LDH / 4; i += NTHR) reinterpret_cast<int*>(hbuf)[i] = 0;
    if (tid == 0) { flag[0] = 0; flag[1] = 0; }
    {
        v4i* dst = reinterpret_cast<v4i*>(smem + C::plane_off(NT));
        const v4i* src = reinterpret_cast<const v4i*>(rl.w_hh);  // plane 0 = the first NT x KS KiB of the packed array
        for (int i = tid; i < NT * KS * 64; i += NTHR) dst[i] = src[i];
    }
    __syncthreads();
    for (int idx = tid; idx < RPW * (H / 4); idx += NTHR) {
        const int rr = idx / (H / 4), j4 = (idx - rr * (H / 4)) * 4;
        const int rsrc = row0 + rr < R ? row0 + rr : R - 1;  // (rows past R duplicate row R-1 in every value: see scan3_role)
        const v4f h = *reinterpret_cast<const v4f*>(rl.h_state + (size_t)rsrc * H + j4);
        const unsigned pk = (h.x > 0.5f ? 1u : 0u) | (h.y > 0.5f ? 0x100u : 0u) | (h.z > 0.5f ? 0x10000u : 0u) |
                            (h.w > 0.5f ? 0x1000000u : 0u);
        *reinterpret_cast<unsigned*>(hbuf + rr * LDH + j4) = pk;
    }

    if (wave < NCW) {
        if (wave < NT - (SFSN_S3W_IOTILES ? 12 : 10)) {
            const int tiles[2] = {wave, (SFSN_S3W_IOTILES ? 12 : 10) + wave};
            scan3w_compute<KS, RPW, FLG, 2>(rl, smem, T, H, NT, tiles, lane);
        } else {
            const int tiles[1] = {wave};
            scan3w_compute<KS, RPW, FLG, 1>(rl, smem, T, H, NT, tiles, lane);
        }
        return;
    }

    if (wave == NCW) {
        // ================================================= loader wave =================================================
        const int np = C::pieces(NT), nch = C::chunks(NT);
        unsigned goff[C::MAXP];
#pragma unroll
        for (int p = 0; p < C::MAXP; ++p) {
            int e = 64 * p + lane;
            if (e > nch - 1) e = nch - 1;  // surplus lanes of the last piece re-fetch the last chunk
            const int cidx = e / RPW, r = e - cidx * RPW;
            const int grow = (row0 + r < R) ? row0 + r : R - 1;
            goff[p] = (unsigned)((grow * H + cidx * 4) * 4);
        }
        const size_t frame = (size_t)R * H;
        int avail = GATED ? 0 : T;
        int failed = 0;
        S3wFlushF<RPW, LDH> ff;
        if constexpr (LSF) ff.init(lane, row0, R, H, 0, ltake);
        int allow = (D - 2) * np;
        if constexpr (LSF) allow = (D - 2) * (np + ff.nsf) + ff.nsf;
        if (allow > 62) allow = 62;
        auto ensure = [&](int need) __attribute__((always_inline)) {
            if constexpr (GATED) s3_ensure(lk, need, T, avail, failed, lane);
        };
        auto issue = [&](int slot, int td) __attribute__((always_inline)) {
            const float* zt = rl.zin + (size_t)td * frame;
#pragma unroll
            for (int p = 0; p < C::MAXP; ++p)
                if (p < np) dma16_to_lds<GATED>(__builtin_amdgcn_readfirstlane((unsigned)(slot * SLOT + p * 1024)), zt, goff[p]);
        };
        S3wTiles<KS, RPW, FLG, 1> ts;  // (SFSN_S3W_IOTILES) my own output tile: its step runs between this step's IO and its waits
        if constexpr (SFSN_S3W_IOTILES) {
            const int tiles[1] = {NCW};
            ts.init(rl, H, NT, tiles, lane);
        }
        __syncthreads();
        ensure(D - 1 < T ? D - 1 : T);
        if (!failed)
            for (int s0 = 0; s0 < D - 1; ++s0) issue(s0, s0 < T ? s0 : (T > 0 ? T - 1 : 0));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (GATED) if (failed && lane == 0) flag[0] = 1;  // read during step 0
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        int stop = 0;
        S3_PB_DECL();
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            if constexpr (GATED) stop = failed;
            const int td = (t + D - 1 < T) ? t + D - 1 : T - 1;
            S3_PB_TIC();
            ensure(td + 1);
#ifndef SFSN_PB_ISSUE  // (-DSFSN_PB_ISSUE: the DMA issue is counted with the polls)
            S3_PB_TOC(2);
#endif
            if (!failed) issue((t + D - 1) % D, td);
#ifdef SFSN_PB_ISSUE
            S3_PB_TOC(2);
#endif
            if constexpr (GATED) if (failed && lane == 0) flag[(t + 1) & 1] = 1;  // read during step t+1 (see scan3_role)
            if constexpr (LSF) if (t > 0) ff.run(hbuf + (t & 1) * 16 * LDH, rl.spikes_f32 + ((size_t)(t - 1) * R + row0) * H, lane);
            if constexpr (SFSN_S3W_IOTILES) ts.step(hbuf + (t & 1) * 16 * LDH, hbuf + ((t & 1) ^ 1) * 16 * LDH, smem + (t % D) * SLOT, wplane);
            S3_PB_TIC();
            wait_vmcnt_n(allow);
            S3_PB_TOC(1);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            S3_PB_TIC();
            __builtin_amdgcn_s_barrier();
            S3_PB_TOC(0);
            if constexpr (GATED) if (stop) break;
        }
        S3_PB_OUT(rl, NCW, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // DMAs past the end are invisible to the compiler
        if constexpr (LSF) if (T > 0 && !(GATED && stop)) ff.run(hbuf + (T & 1) * 16 * LDH, rl.spikes_f32 + ((size_t)(T - 1) * R + row0) * H, lane);
        if constexpr (SFSN_S3W_IOTILES) ts.finish(rl, H, hbuf + (T & 1) * 16 * LDH);
        return;
    }

    if (wave == NCW + 1) {
        // ================================================= storer wave =================================================
        constexpr int MAX8 = (RPW * KS * 4 + 63) / 64;
        constexpr int nu8 = RPW * (HP / 16), ns8 = (nu8 + 63) / 64;
        constexpr bool F32 = (OUT & 1) && (!LSF || LSPLIT);
        S3wFlushF<RPW, LDH> ff;
        if constexpr (F32) ff.init(lane, row0, R, H, LSPLIT ? ltake : 0);
        int l8[MAX8];
        unsigned ok8 = 0;
        unsigned cnt = 0;  // spikes flushed by this lane (roles without an fp32 spike tensor: rl.count)
#pragma unroll
        for (int k = 0; k < MAX8; ++k) {
            const int u = 64 * k + lane, rr = u / (HP / 16), c16 = u - rr * (HP / 16);
            l8[k] = rr * LDH + c16 * 16;
            if (k < ns8 && u < nu8 && row0 + rr < R) ok8 |= 1u << k;
        }
        auto flushf = [&](const int8_t* hsrc, int ts) __attribute__((always_inline)) {
            if constexpr (F32) ff.run(hsrc, rl.spikes_f32 + ((size_t)ts * R + row0) * H, lane);
        };
        auto flush8 = [&](const int8_t* hsrc, int ts) __attribute__((always_inline)) {
            if constexpr (OUT & 2) {
                int8_t* p8 = rl.spikes_i8 + ((size_t)ts * R + row0) * HP;
#pragma unroll
                for (int k = 0; k < MAX8; ++k) {
                    if ((ok8 >> k) & 1u) {
                        const v4i d = *reinterpret_cast<const v4i*>(hsrc + l8[k]);
                        if (PUB) store16_sc1(p8, (unsigned)((64 * k + lane) * 16), d);
                        else *reinterpret_cast<v4i*>(p8 + (size_t)(64 * k + lane) * 16) = d;
                        if constexpr (!(OUT & 1)) cnt += popc16(d);
                    }
                }
            }
        };
        const int rows_live = (R - row0 < RPW) ? R - row0 : RPW;
        const int spf = (F32 ? ff.nsf : 0) + ((OUT & 2) ? (rows_live * (HP / 16) + 63) / 64 : 0);
        const int pf = spf > 0 ? (62 / spf < SFSN_S3W_PF ? 62 / spf : SFSN_S3W_PF) : 8;  // frames of my stores that may be in flight
        S3wTiles<KS, RPW, FLG, 1> ts;  // (SFSN_S3W_IOTILES) my own output tile
        if constexpr (SFSN_S3W_IOTILES) {
            const int tiles[1] = {NCW + 1};
            ts.init(rl, H, NT, tiles, lane);
        }
        __syncthreads();
        __builtin_amdgcn_s_barrier();
        int stop = 0;
        S3_PB_DECL();
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            if constexpr (GATED) stop = flag[t & 1];
            if (t > 0) {
                const int8_t* hc = hbuf + (t & 1) * 16 * LDH;  // = h_{t-1}
                flush8(hc, t - 1);
                flushf(hc, t - 1);
            }
            if constexpr (SFSN_S3W_IOTILES) ts.step(hbuf + (t & 1) * 16 * LDH, hbuf + ((t & 1) ^ 1) * 16 * LDH, smem + (t % D) * SLOT, wplane);
            if constexpr (PUB) {
                if (t > 0) {
                    S3_PB_TIC();
                    wait_vmcnt_n(pf * spf);
                    S3_PB_TOC(1);
                    if (lane == 0 && t - pf > 0) stack_publish(lk, t - pf);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // my LDS reads are done before the buffer is rewritten (step t+1)
            S3_PB_TIC();
            __builtin_amdgcn_s_barrier();
            S3_PB_TOC(0);
            if constexpr (GATED) if (__builtin_amdgcn_readfirstlane(stop)) break;
        }
        S3_PB_OUT(rl, NCW + 1, lane);
        if (T > 0 && !(GATED && __builtin_amdgcn_readfirstlane(stop))) {
            const int8_t* hl = hbuf + (T & 1) * 16 * LDH;
            flush8(hl, T - 1);
            flushf(hl, T - 1);
        }
        if constexpr (SFSN_S3W_IOTILES) ts.finish(rl, H, hbuf + (T & 1) * 16 * LDH);
        if constexpr (PUB) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) stack_publish(lk, T);  // (also after an expired spin: consumers must not wait for us)
        }
        if constexpr (!(OUT & 1)) wave_count_add(rl.count, cnt);
        return;
    }

    // ================================================= spare waves (NT < 20): keep the barrier count =================================================
    __syncthreads();
    __builtin_amdgcn_s_barrier();
    int stop = 0;
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        if constexpr (GATED) stop = flag[t & 1];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        if constexpr (GATED) if (__builtin_amdgcn_readfirstlane(stop)) break;
    }
}

#endif
