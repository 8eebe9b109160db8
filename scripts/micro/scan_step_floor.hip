// scan_step_floor.hip -- what does ONE step of the recurrent scan cost on gfx950, and how low can its structure go?
//
// Stand-alone (no torch, no library).  H = 224 neurons (14 output tiles of 16), K padded to 256, three int8 digit planes of
// W_hh register resident, hidden state as int8 in LDS (double buffered), one workgroup per CU, T steps.  No global memory
// traffic inside the loop unless IO=1: this is the floor of the dependency chain
//     barrier -> B fragments from LDS -> MFMA chains -> epilogue (recombine, sigmoid, lerp, BN, threshold) -> LDS -> barrier
// Variants:
//   mfma_rate     : back-to-back v_mfma_i32_16x16x64_i8 / 16x16x32_i8 per SIMD (1 or 4 waves per SIMD)
//   lds_roundtrip : ds_write_b8 -> lgkmcnt(0) -> s_barrier -> 4 x ds_read_b128 -> lgkmcnt(0), NW waves
//   step_v0       : round 2's structure: NW=16 waves x 1 tile, all MFMAs then the epilogue, one barrier per step
//   step_v1       : 4 (or 8) waves x up to 4 (2) tiles, the epilogue of tile i-1 interleaved with the MFMAs of tile i
//                   inside the wave (sched_group_barrier pipeline), RPW = 4 / 8 / 16 rows per workgroup
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scan_step_floor.bin scan_step_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdint.h>
#include <type_traits>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int H = 224, NT = 14, KS = 4, HP = 256, LDH = HP + 32;

// ---------------------------------------------------------------------------------------------------------------------
template <int SHAPE, int CHAINS>
__global__ __launch_bounds__(1024) void mfma_rate_kernel(int iters, int* out, long long* clk) {
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)threadIdx.x};
    v4i acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = v4i{c, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 48 / CHAINS; ++j)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if constexpr (SHAPE == 64) acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[c], 0, 0, 0);
                else {
                    const long la = ((long)a.y << 32) | (unsigned)a.x, lb = ((long)b.y << 32) | (unsigned)b.x;
                    acc[c] = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, acc[c], 0, 0, 0);
                }
            }
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) asm volatile("" : "+v"(acc[c]));
    }
    const long long t1 = __builtin_readcyclecounter();
    int s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c].x + acc[c].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void lds_roundtrip_kernel(int iters, int* out, long long* clk) {
    __shared__ __attribute__((aligned(16))) int8_t hbuf[2 * 16 * LDH];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    for (int i = tid; i < 2 * 16 * LDH / 4; i += NW * 64) reinterpret_cast<int*>(hbuf)[i] = 0;
    __syncthreads();
    int s = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < iters; ++t) {
        const int8_t* hc = hbuf + (t & 1) * 16 * LDH;
        int8_t* hn = hbuf + ((t & 1) ^ 1) * 16 * LDH;
        v4i b[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b[ks] = *reinterpret_cast<const v4i*>(hc + n * LDH + ks * 64 + q * 16);
        s += b[0].x + b[1].y + b[2].z + b[3].w;
        hn[(n & 7) * LDH + ((tid >> 4) & 255)] = (int8_t)(s & 1);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float cell(float rec, float dq, float z, float db, float& c, float alpha, float beta) {
    const float pre_f = __builtin_fmaf(rec, dq, z);
    const float pre_g = pre_f + db;
    const float f = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre_f * -1.44269504088896341f));
    const float m = __builtin_fmaf(f, c - pre_g, pre_g);
    const float y = __builtin_fmaf(m, alpha, beta);
    c = y;
    return y;
}
__device__ __forceinline__ float recomb(int a0, int a1, int a2) {  // exact int32 sum (|sum| < 2^31), one rounding
    return (float)((a2 << 16) + (a1 << 8) + a0);
}

struct StepArgs {
    const int8_t* w;      // packed digits [3][NT][KS][64][16]
    const float* cst;     // [4][HP]: dq, db, alpha, beta
    const float* zin;     // [T][R][H] (IO) or [R][H]
    float* spikes;        // [T][R][H] (IO)
    int8_t* spikes8;      // [T][R][HP] (IO)
    float* cout;          // [R][H]
    long long* clk;
    long long* stamps;  // [NW][8]
    int T, R;
};

// round 2's structure (sfsn_scan_dev.h scan_body, RP4 / plain epilogue), no global traffic in the loop
template <int RPW>
__global__ __launch_bounds__(1024) void step_v0_kernel(const StepArgs p) {
    constexpr int NW = 16;
    __shared__ __attribute__((aligned(16))) int8_t hbuf[2 * 16 * LDH];
    __shared__ float cst[4][HP];
    __shared__ float zl[16][HP];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 16 * LDH / 4; i += NW * 64) reinterpret_cast<int*>(hbuf)[i] = 0;
    for (int i = tid; i < 4 * HP; i += NW * 64) (&cst[0][0])[i] = (i % HP) < H ? p.cst[i] : 0.f;
    for (int i = tid; i < 16 * HP; i += NW * 64) zl[i / HP][i % HP] = (i % HP) < H ? p.zin[(size_t)(((int)blockIdx.x * RPW + (i / HP) % RPW) % p.R) * H + i % HP] : 0.f;
    __syncthreads();
    const bool have = wave < NT;
    const int ct = have ? wave : 0;
    v4i W[KS][3];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int d = 0; d < 3; ++d) W[ks][d] = *reinterpret_cast<const v4i*>(p.w + ((((size_t)d * NT + ct) * KS + ks) * 64 + lane) * 16);
    const int cc = ct * 16 + q * 4;
    float c[4] = {0, 0, 0, 0};
    const int r4 = n >> 2, row4 = n & 3;
    const long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < p.T; ++t) {
        const int8_t* hc = hbuf + (t & 1) * 16 * LDH;
        int8_t* hn = hbuf + ((t & 1) ^ 1) * 16 * LDH;
        if (have) {
            v4i b[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) b[ks] = *reinterpret_cast<const v4i*>(hc + n * LDH + ks * 64 + q * 16);
            v4i a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[ks][0], b[ks], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[ks][1], b[ks], a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[ks][2], b[ks], a2, 0, 0, 0);
            }
            if constexpr (RPW == 4) {
                auto pick = [&](const v4i& a) __attribute__((always_inline)) {
                    int v = a[0];
                    v = __builtin_amdgcn_update_dpp(v, a[1], 0x114, 0xf, 0x2, false);
                    v = __builtin_amdgcn_update_dpp(v, a[2], 0x118, 0xf, 0x4, false);
                    v = __builtin_amdgcn_update_dpp(v, a[3], 0x11C, 0xf, 0x8, false);
                    return v;
                };
                const int cj = cc + r4;
                const float rec = recomb(pick(a0), pick(a1), pick(a2));
                const float y = cell(rec, cst[0][cj], zl[row4][cj], cst[1][cj], c[0], cst[2][cj], cst[3][cj]);
                hn[row4 * LDH + cj] = (y >= 0.f) ? 1 : 0;
            } else {
                unsigned pk = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y = cell(recomb(a0[r], a1[r], a2[r]), cst[0][cc + r], zl[n][cc + r], cst[1][cc + r], c[r], cst[2][cc + r], cst[3][cc + r]);
                    pk |= (y >= 0.f) ? (1u << (8 * r)) : 0u;
                }
                *reinterpret_cast<unsigned*>(hn + n * LDH + cc) = pk;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    }
    const long long t1 = __builtin_readcyclecounter();
    if (have) *reinterpret_cast<v4f*>(p.cout + (size_t)(((int)blockIdx.x * 16 + n) % p.R) * H + cc) = v4f{c[0], c[1], c[2], c[3]};
    if (tid == 0 && blockIdx.x == 0) p.clk[0] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------
// v1: NW waves (4: one per SIMD, 8: two), wave w owns tiles w, w+NW, ... ; the epilogue of tile i-1 is interleaved with the
// MFMAs of tile i (same wave: the matrix pipe takes 16 clk per instruction, the VALU work goes into the gaps).
// RPW rows per workgroup: 16 (4 values per lane), 8 (re-dealt: 2 values per lane), 4 (re-dealt: 1 value per lane).
// KT = 1: the k tail (neurons 192..223) as one 16x16x32 step instead of a padded 16x16x64 one.
// SCHED = 0: source order only; 1: sched_group_barrier pipeline (1 MFMA : VPM VALU)
template <int RPW>
struct Deal {
    static constexpr int NV = RPW == 16 ? 4 : RPW == 8 ? 2 : 1;  // values per lane and tile after the re-deal
};

template <int NW, int RPW, int KT, int SCHED, int VPM, int PRIO = 0, int STAMP = 0, int ORDER = 0>
__global__ __launch_bounds__(NW * 64) void step_v1_kernel(const StepArgs p) {
    constexpr int NTL = (NT + NW - 1) / NW, NV = Deal<RPW>::NV;
    __shared__ __attribute__((aligned(16))) int8_t hbuf[2 * 16 * LDH];
    __shared__ float zl[16][HP];
    __shared__ __attribute__((aligned(16))) int cntw[4];  // ORDER == 2: tiles completed per k-slice, counted up over the steps
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 16 * LDH / 4; i += NW * 64) reinterpret_cast<int*>(hbuf)[i] = 0;
    for (int i = tid; i < 16 * HP; i += NW * 64) zl[i / HP][i % HP] = (i % HP) < H ? p.zin[(size_t)(((int)blockIdx.x * RPW + (i / HP) % RPW) % p.R) * H + i % HP] : 0.f;
    if (tid < 4) cntw[tid] = 0;
    __syncthreads();
    // my row and my neurons within a tile after the re-deal
    const int row = RPW == 16 ? n : RPW == 8 ? (n & 7) : (n & 3);
    int nof[NV];  // neuron offset within the tile's 4q group
#pragma unroll
    for (int j = 0; j < NV; ++j) nof[j] = RPW == 16 ? j : RPW == 8 ? (n >> 3) + 2 * j : (n >> 2);
    v4i W[NTL][KS][3];
    float c[NTL][NV], dq[NTL][NV], db[NTL][NV], al[NTL][NV], be[NTL][NV];
    int cj[NTL][NV];
    int ntl = 0;
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        const int ct0 = wave + NW * i;
        if (ct0 < NT) ntl = i + 1;
        const int ct = ct0 < NT ? ct0 : NT - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int d = 0; d < 3; ++d) W[i][ks][d] = *reinterpret_cast<const v4i*>(p.w + ((((size_t)d * NT + ct) * KS + ks) * 64 + lane) * 16);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            cj[i][j] = ct * 16 + q * 4 + nof[j];
            c[i][j] = 0.f;
            dq[i][j] = p.cst[0 * HP + cj[i][j]];
            db[i][j] = p.cst[1 * HP + cj[i][j]];
            al[i][j] = p.cst[2 * HP + cj[i][j]];
            be[i][j] = p.cst[3 * HP + cj[i][j]];
        }
    }
    ntl = __builtin_amdgcn_readfirstlane(ntl);
    if constexpr (PRIO == 1) {  // waves w, w+4, w+8, w+12 share a SIMD: the earlier slot wins the matrix pipe, the later ones trail
        const int slot = wave >> 2;
        if (slot == 0) __builtin_amdgcn_s_setprio(3);
        else if (slot == 1) __builtin_amdgcn_s_setprio(2);
        else if (slot == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    } else if constexpr (PRIO == 2) {  // reversed
        const int slot = wave >> 2;
        if (slot == 0) __builtin_amdgcn_s_setprio(0);
        else if (slot == 1) __builtin_amdgcn_s_setprio(1);
        else if (slot == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
    }

    auto mfma_tile = [&](int i, const v4i (&b)[KS], v4i (&a)[3]) __attribute__((always_inline)) {
        a[0] = a[1] = a[2] = v4i{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (KT && ks == KS - 1) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const long la = ((long)W[i][ks][d].y << 32) | (unsigned)W[i][ks][d].x, lb = ((long)b[ks].y << 32) | (unsigned)b[ks].x;
                    a[d] = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, a[d], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int d = 0; d < 3; ++d) a[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[i][ks][d], b[ks], a[d], 0, 0, 0);
            }
        }
    };
    auto epi_tile = [&](int i, const v4i (&a)[3], int8_t* hn) __attribute__((always_inline)) {
        int v[3][NV];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if constexpr (RPW == 16) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[d][j] = a[d][j];
            } else if constexpr (RPW == 8) {
                v[d][0] = __builtin_amdgcn_update_dpp(a[d][0], a[d][1], 0x118, 0xf, 0xC, false);  // lanes 8..15 <- a[1] of lane-8
                v[d][1] = __builtin_amdgcn_update_dpp(a[d][2], a[d][3], 0x118, 0xf, 0xC, false);
            } else {
                int x = a[d][0];
                x = __builtin_amdgcn_update_dpp(x, a[d][1], 0x114, 0xf, 0x2, false);
                x = __builtin_amdgcn_update_dpp(x, a[d][2], 0x118, 0xf, 0x4, false);
                x = __builtin_amdgcn_update_dpp(x, a[d][3], 0x11C, 0xf, 0x8, false);
                v[d][0] = x;
            }
        }
        if constexpr (RPW == 16) {
            unsigned pk = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = cell(recomb(v[0][j], v[1][j], v[2][j]), dq[i][j], zl[row][cj[i][j]], db[i][j], c[i][j], al[i][j], be[i][j]);
                pk |= (y >= 0.f) ? (1u << (8 * j)) : 0u;
            }
            *reinterpret_cast<unsigned*>(hn + row * LDH + cj[i][0]) = pk;
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const float y = cell(recomb(v[0][j], v[1][j], v[2][j]), dq[i][j], zl[row][cj[i][j]], db[i][j], c[i][j], al[i][j], be[i][j]);
                hn[row * LDH + cj[i][j]] = (y >= 0.f) ? 1 : 0;
            }
        }
    };

    const long long t0 = __builtin_readcyclecounter();
    long long st_last[4] = {0, 0, 0, 0}, st_sum[3] = {0, 0, 0}, st_early = 0;
    v4i eacc[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
    unsigned edm = 0;
    auto loop = [&](auto ntag) __attribute__((always_inline)) {
        constexpr int NTW = decltype(ntag)::value;  // tiles of THIS wave: compile time, so that a step is one basic block
#pragma unroll 1
        for (int t = 0; t < p.T; ++t) {
            const int8_t* hc = hbuf + (t & 1) * 16 * LDH;
            int8_t* hn = hbuf + ((t & 1) ^ 1) * 16 * LDH;
            long long s0 = 0, s1 = 0, s2 = 0, s2b = 0, s3 = 0;
            if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(s0)); __builtin_amdgcn_sched_barrier(0); }
            if constexpr (NTW == 1 && ORDER == 2) {
                // EARLY: the k-slices of h(t) whose four tiles have finished go through the products of step t+1 before the barrier
                constexpr int LAST = (NT - 1) >> 2;
                const int mys = wave >> 2;
                v4i b[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    if (!((edm >> ks) & 1u)) b[ks] = *reinterpret_cast<const v4i*>(hc + n * LDH + ks * 64 + q * 16);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if ((edm >> ks) & 1u) continue;
#pragma unroll
                    for (int d = 0; d < 3; ++d) eacc[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[0][ks][d], b[ks], eacc[d], 0, 0, 0);
                }
                if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(s1)); __builtin_amdgcn_sched_barrier(0); }
                epi_tile(0, eacc, hn);
                eacc[0] = eacc[1] = eacc[2] = v4i{0, 0, 0, 0};
                edm = 0;
                asm volatile("" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(&cntw[mys], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(s2)); __builtin_amdgcn_sched_barrier(0); }
                if (mys < LAST && t + 1 < p.T) {
                    const unsigned want = (1u << LAST) - 1u;
                    const int need = 4 * (t + 1), need_last = (NT - 4 * LAST) * (t + 1);
                    const unsigned caddr = (unsigned)(size_t)cntw;
#pragma unroll 1
                    for (int spins = 0; spins < 64; ++spins) {
                        v4i cn;
                        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(cn) : "v"(caddr) : "memory");
#pragma unroll
                        for (int ks = 0; ks < LAST; ++ks) {
                            if (!((edm >> ks) & 1u) && __builtin_amdgcn_readfirstlane(cn[ks]) >= need) {
                                const v4i bb = *reinterpret_cast<const v4i*>(hn + n * LDH + ks * 64 + q * 16);
#pragma unroll
                                for (int d = 0; d < 3; ++d) eacc[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[0][ks][d], bb, eacc[d], 0, 0, 0);
                                edm |= 1u << ks;
                            }
                        }
                        if (edm == want) break;
                        if (__builtin_amdgcn_readfirstlane(cn[LAST]) >= need_last) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(s2b)); __builtin_amdgcn_sched_barrier(0); }
            } else if constexpr (NTW > 0) {
                v4i b[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) b[ks] = *reinterpret_cast<const v4i*>(hc + n * LDH + ks * 64 + q * 16);
                v4i acc[2][3];
                mfma_tile(0, b, acc[0]);
                if constexpr (SCHED || ORDER == 1) __builtin_amdgcn_sched_barrier(0);
                if constexpr (STAMP && NTW == 1) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(s1)); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int i = 1; i < NTW; ++i) {
                    if constexpr (ORDER == 1) {  // M(i-1) E(i-1) M(i) E(i): no pipelining inside the wave, the OTHER waves of the SIMD fill in
                        epi_tile(i - 1, acc[(i - 1) & 1], hn);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_tile(i, b, acc[i & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        mfma_tile(i, b, acc[i & 1]);
                        epi_tile(i - 1, acc[(i - 1) & 1], hn);
                    }
                    if constexpr (SCHED) {
#pragma unroll
                        for (int m = 0; m < 12; ++m) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
                            __builtin_amdgcn_sched_group_barrier(0x006, VPM, 0);  // VPM VALU / SALU
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (STAMP) if (i == NTW - 1) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(s1)); __builtin_amdgcn_sched_barrier(0); }
                }
                epi_tile(NTW - 1, acc[(NTW - 1) & 1], hn);
            }
            if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(s2)); __builtin_amdgcn_sched_barrier(0); }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            if constexpr (STAMP) {
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s3));
                st_last[0] = s0; st_last[1] = s1; st_last[2] = s2; st_last[3] = s3; st_early = s2b;
                st_sum[0] += s1 - s0; st_sum[1] += s2 - s0; st_sum[2] += s3 - s0;
            }
        }
    };
    if (ntl == NTL) loop(std::integral_constant<int, NTL>{});
    else if (ntl == NTL - 1) loop(std::integral_constant<int, NTL - 1>{});
    else loop(std::integral_constant<int, 0>{});
    const long long t1 = __builtin_readcyclecounter();
    if constexpr (STAMP) if (blockIdx.x == 0 && lane == 0) {
        for (int k = 0; k < 4; ++k) p.stamps[wave * 8 + k] = st_last[k];
        for (int k = 0; k < 3; ++k) p.stamps[wave * 8 + 4 + k] = st_sum[k];
        p.stamps[wave * 8 + 7] = st_early;
    }
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (i < ntl) p.cout[(size_t)(((int)blockIdx.x * RPW + row) % p.R) * H + cj[i][j]] = c[i][j];
    if (tid == 0 && blockIdx.x == 0) p.clk[0] = t1 - t0;
}


// ---------------------------------------------------------------------------------------------------------------------
// v3: two groups of waves half a step apart.  Group X owns the tiles that produce k-chunks 0,1 of the hidden state (7 tiles,
// 112 neurons in 128 columns), group Y the tiles of k-chunks 2,3.  Two barriers per step:
//   half 1: X = [6 MFMAs on h(t-1)[Y]] + epilogue -> h(t)[X] ;  Y = [6 MFMAs on h(t-1)[Y]]               (Y's other 6 were issued in half 2)
//   half 2: X = [6 MFMAs of step t+1 on h(t)[X]]            ;  Y = epilogue -> h(t)[Y], [6 MFMAs of step t+1 on h(t)[X]]
// so that on every SIMD one group's VALU phase runs beside the other group's (and its own next) matrix work.
// 16 waves: X = waves 0..6, Y = waves 8..14 (waves w and w+8 share a SIMD); waves 7, 15 only keep the barrier count.
template <int RPW, int PRIO, int YORDER>
__global__ __launch_bounds__(1024) void step_v3_kernel(const StepArgs p) {
    constexpr int NW = 16, NV = Deal<RPW>::NV;
    __shared__ __attribute__((aligned(16))) int8_t hbuf[2 * 16 * LDH];
    __shared__ float zl[16][HP];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 16 * LDH / 4; i += NW * 64) reinterpret_cast<int*>(hbuf)[i] = 0;
    for (int i = tid; i < 16 * HP; i += NW * 64) zl[i / HP][i % HP] = (i % HP) < H ? p.zin[(size_t)(((int)blockIdx.x * RPW + (i / HP) % RPW) % p.R) * H + i % HP] : 0.f;
    __syncthreads();
    const bool isY = wave >= 8;
    const int wt = wave & 7;  // tile within the group
    const bool have = wt < 7;
    const int ct = have ? (isY ? 7 + wt : wt) : 0;
    const int colbase = (isY ? 128 : 0) + (have ? wt : 0) * 16;  // LDS column of the tile's first neuron (k order = [X | pad | Y | pad])
    const int row = RPW == 16 ? n : RPW == 8 ? (n & 7) : (n & 3);
    int nof[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) nof[j] = RPW == 16 ? j : RPW == 8 ? (n >> 3) + 2 * j : (n >> 2);
    v4i W[KS][3];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int d = 0; d < 3; ++d) W[ks][d] = *reinterpret_cast<const v4i*>(p.w + ((((size_t)d * NT + ct) * KS + ks) * 64 + lane) * 16);
    float c[NV], dq[NV], db[NV], al[NV], be[NV];
    int cj[NV], lc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        cj[j] = ct * 16 + q * 4 + nof[j];
        lc[j] = colbase + q * 4 + nof[j];
        c[j] = 0.f; dq[j] = p.cst[cj[j]]; db[j] = p.cst[HP + cj[j]]; al[j] = p.cst[2 * HP + cj[j]]; be[j] = p.cst[3 * HP + cj[j]];
    }
    if constexpr (PRIO == 1) { if (!isY) __builtin_amdgcn_s_setprio(2); }
    if constexpr (PRIO == 2) { if (isY) __builtin_amdgcn_s_setprio(2); }

    auto mfma2 = [&](int k0, const int8_t* hb, v4i (&a)[3]) __attribute__((always_inline)) {
        v4i b0 = *reinterpret_cast<const v4i*>(hb + n * LDH + k0 * 64 + q * 16);
        v4i b1 = *reinterpret_cast<const v4i*>(hb + n * LDH + (k0 + 1) * 64 + q * 16);
#pragma unroll
        for (int d = 0; d < 3; ++d) a[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[k0][d], b0, a[d], 0, 0, 0);
#pragma unroll
        for (int d = 0; d < 3; ++d) a[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[k0 + 1][d], b1, a[d], 0, 0, 0);
    };
    auto epi = [&](const v4i (&a)[3], int8_t* hn) __attribute__((always_inline)) {
        int v[3][NV];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if constexpr (RPW == 16) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[d][j] = a[d][j];
            } else if constexpr (RPW == 8) {
                v[d][0] = __builtin_amdgcn_update_dpp(a[d][0], a[d][1], 0x118, 0xf, 0xC, false);
                v[d][1] = __builtin_amdgcn_update_dpp(a[d][2], a[d][3], 0x118, 0xf, 0xC, false);
            } else {
                int x = a[d][0];
                x = __builtin_amdgcn_update_dpp(x, a[d][1], 0x114, 0xf, 0x2, false);
                x = __builtin_amdgcn_update_dpp(x, a[d][2], 0x118, 0xf, 0x4, false);
                x = __builtin_amdgcn_update_dpp(x, a[d][3], 0x11C, 0xf, 0x8, false);
                v[d][0] = x;
            }
        }
        if constexpr (RPW == 16) {
            unsigned pk = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = cell(recomb(v[0][j], v[1][j], v[2][j]), dq[j], zl[row][cj[j]], db[j], c[j], al[j], be[j]);
                pk |= (y >= 0.f) ? (1u << (8 * j)) : 0u;
            }
            *reinterpret_cast<unsigned*>(hn + row * LDH + lc[0]) = pk;
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const float y = cell(recomb(v[0][j], v[1][j], v[2][j]), dq[j], zl[row][cj[j]], db[j], c[j], al[j], be[j]);
                hn[row * LDH + lc[j]] = (y >= 0.f) ? 1 : 0;
            }
        }
    };
    const long long t0 = __builtin_readcyclecounter();
    v4i acc[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
    if (!have) {
        for (int t = 0; t < p.T; ++t) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
    } else if (!isY) {
#pragma unroll 1
        for (int t = 0; t < p.T; ++t) {
            const int8_t* hp = hbuf + ((t & 1) ^ 1) * 16 * LDH;  // h(t-1)
            int8_t* hc = hbuf + (t & 1) * 16 * LDH;              // h(t)
            mfma2(2, hp, acc);                                   // on h(t-1)[Y]
            epi(acc, hc);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();                        // h(t)[X] published
            acc[0] = acc[1] = acc[2] = v4i{0, 0, 0, 0};
            mfma2(0, hc, acc);                                   // step t+1 on h(t)[X]
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();                        // h(t)[Y] published
        }
    } else {
#pragma unroll 1
        for (int t = 0; t < p.T; ++t) {
            const int8_t* hp = hbuf + ((t & 1) ^ 1) * 16 * LDH;
            int8_t* hc = hbuf + (t & 1) * 16 * LDH;
            mfma2(2, hp, acc);                                   // on h(t-1)[Y]
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            v4i nxt[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
            if constexpr (YORDER == 0) {
                mfma2(0, hc, nxt);                               // step t+1 on h(t)[X], issued first: runs under the epilogue
                epi(acc, hc);
            } else {
                epi(acc, hc);
                mfma2(0, hc, nxt);
            }
            acc[0] = nxt[0]; acc[1] = nxt[1]; acc[2] = nxt[2];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (have)
#pragma unroll
        for (int j = 0; j < NV; ++j) p.cout[(size_t)(((int)blockIdx.x * RPW + row) % p.R) * H + cj[j]] = c[j] + acc[0][0];
    if (tid == 0 && blockIdx.x == 0) p.clk[0] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------
// f8: the FUSED layer (layer >= 1 of a stack): recurrent product + the layer's own input product S1.W_ih^T, the latter batched
// over two frames in the idle MFMA columns (columns 0..7 = the 8 rows at frame f, columns 8..15 = the same rows at frame f+1).
// 8 waves x 2 tiles, every weight fragment register resident (no LDS weight planes); a wave runs the input product of its
// tile 0 on even steps and of its tile 1 on odd steps (12 extra MFMAs per step).  rows/wg = 8.
// ORD = 0: rec tile0, rec tile1, input, epilogues;  1: rec0, epi0 || rec1, epi1 || input (source order, compiler schedules)
template <int ORD, int BATCH>
__global__ __launch_bounds__(512) void step_f8_kernel(const StepArgs p) {
    constexpr int NW = 8, NTL = 2, NV = 2;
    __shared__ __attribute__((aligned(16))) int8_t hbuf[2 * 16 * LDH];
    __shared__ __attribute__((aligned(16))) int8_t sring[4 * 16 * LDH];  // input spikes of 4 frames (static here)
    __shared__ float zl[NT * 64 * 2];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 16 * LDH / 4; i += NW * 64) reinterpret_cast<int*>(hbuf)[i] = 0;
    for (int i = tid; i < 4 * 16 * LDH / 4; i += NW * 64) reinterpret_cast<int*>(sring)[i] = (i * 2654435761u >> 7) & 0x01010101;
    __syncthreads();
    const int row = n & 7;
    int nof[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) nof[j] = (n >> 3) + 2 * j;
    v4i Wh[NTL][KS - 1][3], Wi[NTL][KS - 1][3];
    long Wht[NTL][3], Wit[NTL][3];  // k tail (neurons 192..223): 8 bytes per lane, one 16x16x32 step
    float c[NTL][NV], dqh[NTL][NV], db[NTL][NV], al[NTL][NV], be[NTL][NV], zc[NTL][2][NV];
    v4f dqi4[NTL], bf4[NTL];
    int cj[NTL][NV];
    const bool two = wave + NW < NT;
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        const int ct0 = wave + NW * i;
        const int ct = ct0 < NT ? ct0 : NT - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (ks < KS - 1) {
                    Wh[i][ks][d] = *reinterpret_cast<const v4i*>(p.w + ((((size_t)d * NT + ct) * KS + ks) * 64 + lane) * 16);
                    Wi[i][ks][d] = *reinterpret_cast<const v4i*>(p.w + ((((size_t)d * NT + (NT - 1 - ct)) * KS + ks) * 64 + lane) * 16);
                } else {
                    Wht[i][d] = *reinterpret_cast<const long*>(p.w + ((((size_t)d * NT + ct) * KS + ks) * 64 + lane) * 16);
                    Wit[i][d] = *reinterpret_cast<const long*>(p.w + ((((size_t)d * NT + (NT - 1 - ct)) * KS + ks) * 64 + lane) * 16);
                }
            }
        dqi4[i] = v4f{1.0f / 8388608.f, 1.0f / 8388608.f, 1.0f / 8388608.f, 1.0f / 8388608.f};
        bf4[i] = v4f{0.1f, -0.1f, 0.05f, 0.f};
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            cj[i][j] = ct * 16 + q * 4 + nof[j];
            c[i][j] = 0.f; zc[i][0][j] = 0.f; zc[i][1][j] = 0.f;
            dqh[i][j] = p.cst[cj[i][j]]; db[i][j] = p.cst[HP + cj[i][j]]; al[i][j] = p.cst[2 * HP + cj[i][j]]; be[i][j] = p.cst[3 * HP + cj[i][j]];
        }
    }
    auto rec_tile = [&](int i, const v4i (&b)[KS], v4i (&a)[3]) __attribute__((always_inline)) {
        a[0] = a[1] = a[2] = v4i{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KS - 1; ++ks)
#pragma unroll
            for (int d = 0; d < 3; ++d) a[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wh[i][ks][d], b[ks], a[d], 0, 0, 0);
        const long bt = ((long)b[KS - 1].y << 32) | (unsigned)b[KS - 1].x;
#pragma unroll
        for (int d = 0; d < 3; ++d) a[d] = __builtin_amdgcn_mfma_i32_16x16x32_i8(Wht[i][d], bt, a[d], 0, 0, 0);
    };
    // input product of tile i for frames (f, f+1) -> zc[i][0..1][*] (re-dealt: 2 values per lane and frame)
    auto in_tile = [&](int i, int f) __attribute__((always_inline)) {
        const int8_t* s0 = sring + ((f & 3) * 16) * LDH;
        const int8_t* s1 = sring + (((f + 1) & 3) * 16) * LDH;
        const int8_t* src = (n < 8 ? s0 : s1) + (n & 7) * LDH + q * 16;
        v4i e[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
#pragma unroll
        for (int ks = 0; ks < KS - 1; ++ks) {
            const v4i bs = *reinterpret_cast<const v4i*>(src + ks * 64);
#pragma unroll
            for (int d = 0; d < 3; ++d) e[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wi[i][ks][d], bs, e[d], 0, 0, 0);
        }
        {
            const long bs = *reinterpret_cast<const long*>(src + (KS - 1) * 64);
#pragma unroll
            for (int d = 0; d < 3; ++d) e[d] = __builtin_amdgcn_mfma_i32_16x16x32_i8(Wit[i][d], bs, e[d], 0, 0, 0);
        }
        float z[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) z[r] = __builtin_fmaf(recomb(e[0][r], e[1][r], e[2][r]), dqi4[i][r], bf4[i][r]);
        // frame f: columns 0..7 -> lanes 8..15 take element 1 / 3 of lane-8;  frame f+1: columns 8..15 -> lanes 0..7 take element 0 / 2 of lane+8
        zc[i][0][0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, z[0]), __builtin_bit_cast(int, z[1]), 0x118, 0xf, 0xC, false));
        zc[i][0][1] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, z[2]), __builtin_bit_cast(int, z[3]), 0x118, 0xf, 0xC, false));
        zc[i][1][0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, z[1]), __builtin_bit_cast(int, z[0]), 0x108, 0xf, 0x3, false));
        zc[i][1][1] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, z[3]), __builtin_bit_cast(int, z[2]), 0x108, 0xf, 0x3, false));
    };
    auto epi_tile = [&](int i, const v4i (&a)[3], int8_t* hn, int par) __attribute__((always_inline)) {
        int v[3][NV];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            v[d][0] = __builtin_amdgcn_update_dpp(a[d][0], a[d][1], 0x118, 0xf, 0xC, false);
            v[d][1] = __builtin_amdgcn_update_dpp(a[d][2], a[d][3], 0x118, 0xf, 0xC, false);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float z = par ? zc[i][1][j] : zc[i][0][j];
            const float y = cell(recomb(v[0][j], v[1][j], v[2][j]), dqh[i][j], z, db[i][j], c[i][j], al[i][j], be[i][j]);
            hn[row * LDH + cj[i][j]] = (y >= 0.f) ? 1 : 0;
        }
    };
    const long long t0 = __builtin_readcyclecounter();
    auto loop = [&](auto ntag) __attribute__((always_inline)) {
        constexpr int NTW = decltype(ntag)::value;
#pragma unroll 1
        for (int t2 = 0; t2 < p.T; t2 += 2) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int t = t2 + par;
                const int8_t* hc = hbuf + (t & 1) * 16 * LDH;
                int8_t* hn = hbuf + ((t & 1) ^ 1) * 16 * LDH;
                v4i b[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) b[ks] = *reinterpret_cast<const v4i*>(hc + n * LDH + ks * 64 + q * 16);
                v4i acc[NTL][3];
                if constexpr (ORD == 0) {
#pragma unroll
                    for (int i = 0; i < NTW; ++i) rec_tile(i, b, acc[i]);
#pragma unroll
                    for (int i = 0; i < NTW; ++i) epi_tile(i, acc[i], hn, par);
                    if (BATCH) { if (par < NTW) in_tile(par, t + 2); }   // the frames this tile needs next: its zc is free now
                    else {
#pragma unroll
                        for (int i = 0; i < NTW; ++i) in_tile(i, t + 1);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < NTW; ++i) {
                        rec_tile(i, b, acc[i]);
                        __builtin_amdgcn_sched_barrier(0);
                        epi_tile(i, acc[i], hn, par);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (BATCH) { if (par < NTW) in_tile(par, t + 2); }
                    else {
#pragma unroll
                        for (int i = 0; i < NTW; ++i) in_tile(i, t + 1);
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
            }
        }
    };
    if (two) loop(std::integral_constant<int, 2>{}); else loop(std::integral_constant<int, 1>{});
    const long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (i == 0 || two) p.cout[(size_t)(((int)blockIdx.x * 8 + row) % p.R) * H + cj[i][j]] = c[i][j];
    if (tid == 0 && blockIdx.x == 0) p.clk[0] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------
// VALU issue rates: 64 independent-ish instructions per iteration in 8 chains
template <int OP>
__global__ __launch_bounds__(1024) void valu_rate_kernel(int iters, int* out, long long* clk) {
    float f[8];
    int k[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { f[c] = 1.0f + threadIdx.x * 1e-3f + c; k[c] = threadIdx.x + c; }
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f g[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) g[c] = v2f{f[c], f[c] + 0.5f};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[c]) : "v"(f[(c + 1) & 7]));
                else if constexpr (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(g[c]) : "v"(g[(c + 1) & 7]));
                else if constexpr (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(f[c]));
                else if constexpr (OP == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[c]));
                else if constexpr (OP == 4) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[c]) : "v"(k[c]));
                else if constexpr (OP == 5) asm volatile("v_mov_b32_dpp %0, %1 row_shr:8 row_mask:0xf bank_mask:0xc" : "+v"(k[c]) : "v"(k[(c + 1) & 7]));
                else if constexpr (OP == 6) asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(k[c]) : "v"(k[(c + 1) & 7]));
                else if constexpr (OP == 7) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(g[c]) : "v"(g[(c + 1) & 7]));
                else if constexpr (OP == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(g[c]) : "v"(g[(c + 1) & 7]));
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0; int si = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) { s += f[c] + g[c].x + g[c].y; si += k[c]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)s + si;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------
static float* dalloc_f(size_t n, float lo, float hi) {
    float* h = (float*)malloc(n * 4);
    for (size_t i = 0; i < n; ++i) h[i] = lo + (hi - lo) * (float)(rand() & 0xffff) / 65535.f;
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice)); free(h);
    return d;
}

template <class F>
static void timeit(const char* name, int grid, int steps, long long* dclk, F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(50);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    long long clk = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        launch(steps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) { best = ms; CK(hipMemcpy(&clk, dclk, 8, hipMemcpyDeviceToHost)); }
    }
    printf("%-58s grid %3d : %7.1f ns/step  %7.1f clk/step (wg 0, s_memtime)\n", name, grid, best * 1e6 / steps, (double)clk / steps);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int T = 1000, R = 832;
    int* dout; CK(hipMalloc(&dout, 256 * 1024 * 4));
    long long* dclk; CK(hipMalloc(&dclk, 64));
    int8_t* hw = (int8_t*)malloc(3 * NT * KS * 1024);
    for (int i = 0; i < 3 * NT * KS * 1024; ++i) hw[i] = (int8_t)((rand() & 0xff) - 128);
    // the k tail beyond neuron 224 is zero in the real packing
    int8_t* dw; CK(hipMalloc(&dw, 3 * NT * KS * 1024)); CK(hipMemcpy(dw, hw, 3 * NT * KS * 1024, hipMemcpyHostToDevice));
    StepArgs a{};
    a.w = dw;
    {
        float* h = (float*)malloc(4 * HP * 4);
        for (int j = 0; j < HP; ++j) { h[j] = 1.0f / 8388608.f; h[HP + j] = 0.1f; h[2 * HP + j] = 1.1f; h[3 * HP + j] = -0.05f; }
        float* d; CK(hipMalloc(&d, 4 * HP * 4)); CK(hipMemcpy(d, h, 4 * HP * 4, hipMemcpyHostToDevice)); a.cst = d; free(h);
    }
    a.zin = dalloc_f((size_t)R * H, -1.f, 1.f);
    CK(hipMalloc(&a.cout, (size_t)R * H * 4));
    a.clk = dclk; a.T = T; a.R = R;
    CK(hipMalloc(&a.stamps, 16 * 8 * 8));

    printf("== matrix pipe: 48 MFMAs per wave and iteration ==\n");
#define RATE(SHAPE, CH, WAVES) timeit("mfma i8 16x16x" #SHAPE " chains=" #CH " waves/CU=" #WAVES, 256, 2000, dclk, [&](int it) { \
        hipLaunchKernelGGL((mfma_rate_kernel<SHAPE, CH>), dim3(256), dim3(WAVES * 64), 0, 0, it, dout, dclk); })
    RATE(64, 3, 4); RATE(64, 6, 4); RATE(64, 12, 4); RATE(64, 3, 16);
    RATE(32, 3, 4); RATE(32, 6, 4); RATE(32, 12, 4); RATE(32, 3, 16);
    printf("   (ns per iteration / 48 = ns per MFMA per wave; clk likewise)\n");

    printf("== LDS round trip: write, lgkmcnt(0), s_barrier, 4 x ds_read_b128, lgkmcnt(0) ==\n");
#define LRT(NW) timeit("lds_roundtrip waves=" #NW, 208, 20000, dclk, [&](int it) { hipLaunchKernelGGL((lds_roundtrip_kernel<NW>), dim3(208), dim3(NW * 64), 0, 0, it, dout, dclk); })
    LRT(4); LRT(8); LRT(16);

    printf("== one scan step, no global traffic ==\n");
#define V0(RPW, GRID) timeit("v0 16 waves x 1 tile, rows/wg=" #RPW, GRID, T, dclk, [&](int it) { StepArgs b = a; b.T = it; hipLaunchKernelGGL((step_v0_kernel<RPW>), dim3(GRID), dim3(1024), 0, 0, b); })
    V0(4, 208); V0(16, 52);
#define V1(NW, RPW, KT, SCHED, VPM, GRID) timeit("v1 waves=" #NW " rows/wg=" #RPW " ktail=" #KT " sched=" #SCHED " valu/mfma=" #VPM, GRID, T, dclk, [&](int it) { \
        StepArgs b = a; b.T = it; hipLaunchKernelGGL((step_v1_kernel<NW, RPW, KT, SCHED, VPM>), dim3(GRID), dim3(NW * 64), 0, 0, b); })
    V1(4, 4, 0, 0, 3, 208); V1(4, 4, 0, 1, 2, 208); V1(4, 4, 0, 1, 3, 208); V1(4, 4, 1, 1, 3, 208);
    V1(4, 8, 0, 0, 3, 104); V1(4, 8, 0, 1, 3, 104); V1(4, 8, 0, 1, 4, 104); V1(4, 8, 1, 1, 4, 104);
    V1(4, 16, 0, 0, 6, 52); V1(4, 16, 0, 1, 6, 52); V1(4, 16, 1, 1, 6, 52);
    V1(8, 4, 0, 0, 3, 208); V1(8, 4, 0, 1, 3, 208); V1(8, 8, 0, 0, 3, 104); V1(8, 8, 0, 1, 4, 104); V1(8, 8, 1, 1, 4, 104);
    V1(8, 16, 0, 1, 6, 52);
    V1(16, 4, 0, 0, 3, 208); V1(16, 8, 0, 0, 3, 104); V1(16, 8, 1, 0, 3, 104);
    printf("== VALU issue: 64 instructions per wave and iteration ==\n");
#define VR(OP, NAME, WAVES) timeit("valu " NAME " waves/CU=" #WAVES, 256, 4000, dclk, [&](int it) { hipLaunchKernelGGL((valu_rate_kernel<OP>), dim3(256), dim3(WAVES * 64), 0, 0, it, dout, dclk); })
    VR(0, "v_fma_f32", 4); VR(0, "v_fma_f32", 16); VR(1, "v_pk_fma_f32", 4); VR(1, "v_pk_fma_f32", 16); VR(2, "v_exp_f32", 4); VR(2, "v_exp_f32", 16);
    VR(3, "v_rcp_f32", 16); VR(4, "v_cvt_f32_i32", 16); VR(5, "v_mov_dpp", 4); VR(5, "v_mov_dpp", 16); VR(6, "v_lshl_add_u32", 16); VR(7, "v_pk_add_f32", 16); VR(8, "v_pk_mul_f32", 16);
    printf("== priorities / two groups half a step apart ==\n");
#define V1P(NW, RPW, SCHED, VPM, PRIO, GRID) timeit("v1 waves=" #NW " rows/wg=" #RPW " sched=" #SCHED " prio=" #PRIO, GRID, T, dclk, [&](int it) { \
        StepArgs b = a; b.T = it; hipLaunchKernelGGL((step_v1_kernel<NW, RPW, 0, SCHED, VPM, PRIO>), dim3(GRID), dim3(NW * 64), 0, 0, b); })
    V1P(16, 4, 0, 3, 1, 208); V1P(16, 4, 0, 3, 2, 208); V1P(16, 8, 0, 3, 1, 104); V1P(16, 8, 0, 3, 2, 104); V1P(16, 16, 0, 3, 0, 52); V1P(16, 16, 0, 3, 1, 52);
    V1P(8, 4, 0, 3, 1, 208); V1P(8, 8, 0, 3, 1, 104); V1P(8, 8, 1, 4, 1, 104); V1P(8, 16, 1, 6, 1, 52);
#define V3(RPW, PRIO, YO, GRID) timeit("v3 two groups rows/wg=" #RPW " prio=" #PRIO " yorder=" #YO, GRID, T, dclk, [&](int it) { \
        StepArgs b = a; b.T = it; hipLaunchKernelGGL((step_v3_kernel<RPW, PRIO, YO>), dim3(GRID), dim3(1024), 0, 0, b); })
    V3(4, 0, 0, 208); V3(4, 1, 0, 208); V3(4, 2, 0, 208); V3(4, 0, 1, 208);
    V3(8, 0, 0, 104); V3(8, 1, 0, 104); V3(8, 2, 0, 104); V3(8, 0, 1, 104); V3(8, 1, 1, 104);
    V3(16, 0, 0, 52); V3(16, 1, 0, 52); V3(16, 0, 1, 52);
    printf("== per-wave timeline of a step (s_memtime): top of step -> last MFMA issued -> epilogue issued -> past the barrier ==\n");
#define V1S(NW, RPW, ORDER, GRID) { timeit("v1 STAMPED waves=" #NW " rows/wg=" #RPW " order=" #ORDER, GRID, T, dclk, [&](int it) { \
        StepArgs b = a; b.T = it; hipLaunchKernelGGL((step_v1_kernel<NW, RPW, 0, 0, 3, 0, 1, ORDER>), dim3(GRID), dim3(NW * 64), 0, 0, b); }); \
        long long hs[16 * 8]; CK(hipMemcpy(hs, a.stamps, sizeof(hs), hipMemcpyDeviceToHost)); long long m0 = hs[0]; for (int w = 0; w < NW; ++w) if (hs[w * 8] < m0) m0 = hs[w * 8]; \
        for (int w = 0; w < NW; ++w) printf("   wave %2d (simd slot %d): last step: top +%4lld  mfma-issued +%4lld  epi-issued +%4lld  early-done +%4lld  past-barrier +%4lld | mean: mfma %6.1f epi %6.1f barrier %6.1f\n", w, w >> 2, \
            hs[w * 8] - m0, hs[w * 8 + 1] - m0, hs[w * 8 + 2] - m0, hs[w * 8 + 7] ? hs[w * 8 + 7] - m0 : 0ll, hs[w * 8 + 3] - m0, (double)hs[w * 8 + 4] / T, (double)hs[w * 8 + 5] / T, (double)hs[w * 8 + 6] / T); }
    if (argc > 1 && !strcmp(argv[1], "early")) { V1S(16, 4, 0, 208); V1S(16, 4, 2, 208); V1S(16, 8, 0, 104); V1S(16, 8, 2, 104); return 0; }
    V1S(16, 8, 0, 104); V1S(16, 4, 0, 208); V1S(8, 8, 0, 104); V1S(8, 8, 1, 104); V1S(4, 8, 1, 104);
#define V1O(NW, RPW, ORDER, PRIO, GRID) timeit("v1 waves=" #NW " rows/wg=" #RPW " order=" #ORDER " prio=" #PRIO, GRID, T, dclk, [&](int it) { \
        StepArgs b = a; b.T = it; hipLaunchKernelGGL((step_v1_kernel<NW, RPW, 0, 0, 3, PRIO, 0, ORDER>), dim3(GRID), dim3(NW * 64), 0, 0, b); })
    V1O(8, 8, 1, 0, 104); V1O(8, 8, 1, 1, 104); V1O(8, 4, 1, 0, 208); V1O(8, 16, 1, 0, 52); V1O(4, 8, 1, 0, 104);
    printf("== fused layer (recurrent + own input product), 8 waves x 2 tiles, all weights in registers, rows/wg = 8 ==\n");
#define F8(ORD, BATCH) timeit("f8 order=" #ORD " batch2=" #BATCH, 104, T, dclk, [&](int it) { StepArgs b = a; b.T = it; hipLaunchKernelGGL((step_f8_kernel<ORD, BATCH>), dim3(104), dim3(512), 0, 0, b); })
    F8(0, 1); F8(1, 1); F8(0, 0); F8(1, 0);
    return 0;
}
