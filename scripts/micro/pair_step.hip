// pair_step.hip -- round 4: can a layer >= 1 form its own input product S1.W_ih^T INSIDE the 8-row IO-wave scan?
//
// Stand-alone (no torch, no library).  H = 224 (14 tiles), K padded to 256, 8 rows per workgroup, 16 waves: 14 compute waves
// (one 16-neuron tile each, W_hh register resident: 48 VGPRs) + 2 idle stand-ins for the loader / storer.  The input product of
// a tile is batched over TWO frames in the MFMA columns the 8-row geometry leaves idle (columns 0..7 = the rows at frame f,
// 8..15 = the same rows at frame f+1): 12 matrix instructions per tile and TWO steps.  W_ih does not fit the registers beside
// W_hh at 16 waves (128 VGPRs): its digit planes 0 and 1 sit in LDS (2 x 56 KB, read as A fragments: 1 KiB contiguous per
// wave-instruction, conflict-free), plane 2 in registers (16 VGPRs).  The previous layer's int8 spikes come from an LDS ring
// (static here; in the real role the loader wave's LDS-DMA fills it), chunk (c + 2 r) mod 16 of row r -- the same bank spread as
// the state buffer's 288-byte row stride.
//
// What is measured: clk per step against WHERE the 12 extra matrix instructions are issued --
//   PLACE 0: none (the plain 8-row scan3 step: the floor)
//   PLACE 1: behind the epilogue (the matrix pipe is idle while the last wave of a SIMD finishes its epilogue)
//   PLACE 2: at the head of the step, before the wave's state fragments have arrived from LDS
//   PLACE 3: head for the LAST compute wave of each SIMD (its epilogue is the exposed one), tail for the others
//   SPLIT 0: all 12 in the even step; 1: k-steps 0-1 in the even step, 2-3 in the odd one (accumulators live across the barrier)
//   PRIO 1 : s_setprio 0 around the input product, 2 around the recurrent product + epilogue
//   EPI 0/1/2: no epilogue / the real one / twice (attribution of the VALU share)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o pair_step.bin pair_step.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdint.h>
#include <type_traits>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int H = 224, NT = 14, KS = 4, HP = 256, LDH = HP + 32;
constexpr int RD = 8, SLOT = 8 * HP;                       // input-spike ring: RD frames of 8 rows
constexpr int HBUF_OFF = RD * SLOT;                        // 16 KiB
constexpr int WIH_OFF = HBUF_OFF + 2 * 16 * LDH;           // + 9 KiB
constexpr int PLANE = NT * KS * 1024;                      // 56 KiB per digit plane
constexpr int CSTI_OFF = WIH_OFF + 2 * PLANE;              // {dq_ih, dq_ih, b_f, b_f} per neuron pair: 2 KiB
constexpr int LDS_BYTES = CSTI_OFF + (HP / 2) * 16;        // 139 KiB

struct PArgs {
    const int8_t* w_hh;   // packed digits [3][NT][KS][64][16]
    const int8_t* w_ih;
    const float* cst;     // [6][HP]: dq_hh, db, alpha, beta, dq_ih, bias_f
    float* cout;          // [grid * 8][H]
    int* hsum;            // [grid]: a checksum of the spikes (keeps everything live; compared between variants)
    long long* clk;
    long long* stamps;    // [16][8]
    int T;
};

__device__ __forceinline__ float recomb(int a0, int a1, int a2) { return (float)((a2 << 16) + (a1 << 8) + a0); }

template <int PLACE, int SPLIT, int PRIO, int EPI, int STAMP>
__global__ __launch_bounds__(1024) void pair_l2_kernel(const PArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int8_t* hbuf = reinterpret_cast<int8_t*>(smem + HBUF_OFF);
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- set-up: ring = pseudo-random spikes, state buffers zero, W_ih planes 0 / 1 -> LDS
    for (int i = tid; i < RD * SLOT / 4; i += 1024) reinterpret_cast<unsigned*>(smem)[i] = ((unsigned)(i + 977 * blockIdx.x) * 2654435761u >> 9) & 0x01010101u;
    for (int i = tid; i < 2 * 16 * LDH / 4; i += 1024) reinterpret_cast<int*>(hbuf)[i] = 0;
    for (int i = tid; i < 2 * PLANE / 16; i += 1024)
        reinterpret_cast<v4i*>(smem + WIH_OFF)[i] = reinterpret_cast<const v4i*>(p.w_ih)[i];
    for (int i = tid; i < HP / 2; i += 1024)
        reinterpret_cast<v4f*>(smem + CSTI_OFF)[i] = v4f{p.cst[4 * HP + 2 * i], p.cst[4 * HP + 2 * i + 1], p.cst[5 * HP + 2 * i], p.cst[5 * HP + 2 * i + 1]};
    __syncthreads();
    // the k tail beyond neuron 224 of the ring must be zero (as the real int8 rows are)
    for (int i = tid; i < RD * 8 * 32; i += 1024) {
        const int s = i / (8 * 32), r = (i / 32) & 7, kk = 224 + (i & 31);
        const int c = kk >> 4, pos = (c + 2 * r) & 15;
        smem[s * SLOT + r * HP + pos * 16 + (kk & 15)] = 0;
    }
    __syncthreads();

    long long st[6] = {0, 0, 0, 0, 0, 0};
    if (wave < NT) {
        const int ct = wave;
        const int row = n & 7, sub = 2 * (n >> 3);
        const int cj = ct * 16 + q * 4 + sub;
        // k-steps 0..2 as 16-byte A fragments; the k tail (neurons 192..223) as ONE 16x16x32 step: lane (n, q) holds k = 192 + 8 q + j,
        // which is bytes [(q & 1) * 8, +8) of lane (n, q >> 1) of the 16x16x64 fragment of k-step 3
        v4i Whh[KS - 1][3], Wi2[KS - 1];
        long Wht[3], Wi2t;
        const unsigned toff = (unsigned)((((q >> 1) * 16 + n) * 16) + (q & 1) * 8);
#pragma unroll
        for (int ks = 0; ks < KS - 1; ++ks) {
#pragma unroll
            for (int d = 0; d < 3; ++d) Whh[ks][d] = *reinterpret_cast<const v4i*>(p.w_hh + ((((size_t)d * NT + ct) * KS + ks) * 64 + lane) * 16);
            Wi2[ks] = *reinterpret_cast<const v4i*>(p.w_ih + ((((size_t)2 * NT + ct) * KS + ks) * 64 + lane) * 16);
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) Wht[d] = *reinterpret_cast<const long*>(p.w_hh + (((size_t)d * NT + ct) * KS + KS - 1) * 1024 + toff);
        Wi2t = *reinterpret_cast<const long*>(p.w_ih + (((size_t)2 * NT + ct) * KS + KS - 1) * 1024 + toff);
        float c[2] = {0.f, 0.f}, dq[2], db[2], al[2], be[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            dq[j] = p.cst[cj + j]; db[j] = p.cst[HP + cj + j]; al[j] = p.cst[2 * HP + cj + j]; be[j] = p.cst[3 * HP + cj + j];
        }
        const unsigned boff = (unsigned)(n * LDH + q * 16);
        const unsigned boft = (unsigned)(n * LDH + 192 + q * 8);
        const unsigned hoff = (unsigned)(row * LDH + cj);
        // my B fragment of the input product: column n = (frame f0 + (n >> 3), row n & 7), k chunk c = 4 ks + q at position (c + 2 r) & 15
        unsigned soff[KS - 1];
#pragma unroll
        for (int ks = 0; ks < KS - 1; ++ks) soff[ks] = (unsigned)((n >> 3) * SLOT + row * HP + ((ks * 4 + q + 2 * row) & 15) * 16);
        const unsigned soft = (unsigned)((n >> 3) * SLOT + row * HP + ((12 + (q >> 1) + 2 * row) & 15) * 16 + (q & 1) * 8);
        const unsigned woff = (unsigned)(WIH_OFF + (ct * KS) * 1024 + lane * 16);
        const unsigned wofft = (unsigned)(WIH_OFF + (ct * KS + KS - 1) * 1024) + toff;
        // is this wave the last compute wave of its SIMD?  waves w, w+4, w+8, w+12 share a SIMD; 14, 15 are the IO stand-ins
        const bool last_of_simd = wave + 4 >= NT;
        float zc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        v4i e[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
        int hs = 0;

        // k-steps [k0, k1) of the input product of frames (f0, f0 + 1), f0 even: ring slots f0 % RD and (f0 + 1) % RD are adjacent
        auto in_mfma = [&](int f0, int k0, int k1) __attribute__((always_inline)) {
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
            const char* ring = smem + (f0 % RD) * SLOT;
#pragma unroll
            for (int ks = 0; ks < KS - 1; ++ks) {
                if (ks < k0 || ks >= k1) continue;
                const v4i bs = *reinterpret_cast<const v4i*>(ring + soff[ks]);
                const v4i w0 = *reinterpret_cast<const v4i*>(smem + woff + ks * 1024);
                const v4i w1 = *reinterpret_cast<const v4i*>(smem + woff + PLANE + ks * 1024);
                e[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, bs, e[0], 0, 0, 0);
                e[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1, bs, e[1], 0, 0, 0);
                e[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wi2[ks], bs, e[2], 0, 0, 0);
            }
            if (k1 == KS) {
                const long bs = *reinterpret_cast<const long*>(ring + soft);
                const long w0 = *reinterpret_cast<const long*>(smem + wofft);
                const long w1 = *reinterpret_cast<const long*>(smem + wofft + PLANE);
                e[0] = __builtin_amdgcn_mfma_i32_16x16x32_i8(w0, bs, e[0], 0, 0, 0);
                e[1] = __builtin_amdgcn_mfma_i32_16x16x32_i8(w1, bs, e[1], 0, 0, 0);
                e[2] = __builtin_amdgcn_mfma_i32_16x16x32_i8(Wi2t, bs, e[2], 0, 0, 0);
            }
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(2);
        };
        // accumulators -> the input term of both frames, re-dealt to the epilogue's layout (two adjacent neurons per lane):
        // frame f0 = columns 0..7: lanes 8..15 of a row of 16 take elements 2, 3 of the lane 8 below; frame f0+1 = columns 8..15:
        // lanes 0..7 take elements 0, 1 of the lane 8 above
        auto in_finish = [&]() __attribute__((always_inline)) {
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = recomb(e[0][k], e[1][k], e[2][k]);
            const int f00 = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, r[0]), __builtin_bit_cast(int, r[2]), 0x118, 0xf, 0xC, false);
            const int f01 = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, r[1]), __builtin_bit_cast(int, r[3]), 0x118, 0xf, 0xC, false);
            const int f10 = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, r[2]), __builtin_bit_cast(int, r[0]), 0x108, 0xf, 0x3, false);
            const int f11 = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, r[3]), __builtin_bit_cast(int, r[1]), 0x108, 0xf, 0x3, false);
            // (called behind the epilogue of the odd step: both halves of zc are dead; dq_ih and b_f come from LDS, twice per 2 steps)
            const v4f cq = *reinterpret_cast<const v4f*>(smem + CSTI_OFF + (cj >> 1) * 16);
            zc[0][0] = __builtin_fmaf(__builtin_bit_cast(float, f00), cq.x, cq.z);
            zc[0][1] = __builtin_fmaf(__builtin_bit_cast(float, f01), cq.y, cq.w);
            zc[1][0] = __builtin_fmaf(__builtin_bit_cast(float, f10), cq.x, cq.z);
            zc[1][1] = __builtin_fmaf(__builtin_bit_cast(float, f11), cq.y, cq.w);
            e[0] = e[1] = e[2] = v4i{0, 0, 0, 0};
        };

        // prefetched operands of the NEXT step's head product: held across the barrier in registers that the state fragments and the
        // recurrent accumulators leave dead between a wave's epilogue and the top of the next step
        v4i pfb[2] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}}, pfw0[2] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}}, pfw1[2] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
        auto pf_load = [&](int f0, int kp) __attribute__((always_inline)) {  // k-steps 2 kp, 2 kp + 1 (k-step 3 = the 8-byte tail)
            const char* ring = smem + (f0 % RD) * SLOT;
            if (kp == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    pfb[i] = *reinterpret_cast<const v4i*>(ring + soff[i]);
                    pfw0[i] = *reinterpret_cast<const v4i*>(smem + woff + i * 1024);
                    pfw1[i] = *reinterpret_cast<const v4i*>(smem + woff + PLANE + i * 1024);
                }
            } else {
                pfb[0] = *reinterpret_cast<const v4i*>(ring + soff[2]);
                pfw0[0] = *reinterpret_cast<const v4i*>(smem + woff + 2 * 1024);
                pfw1[0] = *reinterpret_cast<const v4i*>(smem + woff + PLANE + 2 * 1024);
                const long b8 = *reinterpret_cast<const long*>(ring + soft);
                const long w08 = *reinterpret_cast<const long*>(smem + wofft);
                const long w18 = *reinterpret_cast<const long*>(smem + wofft + PLANE);
                pfb[1].x = (int)b8; pfb[1].y = (int)(b8 >> 32);
                pfw0[1].x = (int)w08; pfw0[1].y = (int)(w08 >> 32);
                pfw1[1].x = (int)w18; pfw1[1].y = (int)(w18 >> 32);
            }
        };
        auto pf_mfma = [&](int kp) __attribute__((always_inline)) {
            if (kp == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    e[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(pfw0[i], pfb[i], e[0], 0, 0, 0);
                    e[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(pfw1[i], pfb[i], e[1], 0, 0, 0);
                    e[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wi2[i], pfb[i], e[2], 0, 0, 0);
                }
            } else {
                e[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(pfw0[0], pfb[0], e[0], 0, 0, 0);
                e[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(pfw1[0], pfb[0], e[1], 0, 0, 0);
                e[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wi2[2], pfb[0], e[2], 0, 0, 0);
                const long b8 = ((long)pfb[1].y << 32) | (unsigned)pfb[1].x;
                const long w08 = ((long)pfw0[1].y << 32) | (unsigned)pfw0[1].x;
                const long w18 = ((long)pfw1[1].y << 32) | (unsigned)pfw1[1].x;
                e[0] = __builtin_amdgcn_mfma_i32_16x16x32_i8(w08, b8, e[0], 0, 0, 0);
                e[1] = __builtin_amdgcn_mfma_i32_16x16x32_i8(w18, b8, e[1], 0, 0, 0);
                e[2] = __builtin_amdgcn_mfma_i32_16x16x32_i8(Wi2t, b8, e[2], 0, 0, 0);
            }
        };

        // MODE: 0 no input product; 1 behind the epilogue; 2 at the head (operands fetched after the barrier); 3 at the head with
        // operands prefetched before the barrier; 4 behind the epilogue at priority 0 (the rest of the wave runs at priority 1)
        auto loop = [&](auto mode_tag) __attribute__((always_inline)) {
            constexpr int MODE = decltype(mode_tag)::value;
            if constexpr (MODE == 3) { pf_load(2, 0); __builtin_amdgcn_s_waitcnt(0xc07f); }
            if constexpr (MODE == 4 || (MODE == 3 && PRIO)) __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
            for (int t2 = 0; t2 < p.T; t2 += 2) {
#pragma unroll
                for (int par = 0; par < 2; ++par) {
                    const int t = t2 + par;
                    const int8_t* hc = hbuf + (t & 1) * 16 * LDH;
                    int8_t* hn = hbuf + ((t & 1) ^ 1) * 16 * LDH;
                    if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(st[0])); __builtin_amdgcn_sched_barrier(0); }
                    if constexpr (MODE == 2) {
                        if constexpr (SPLIT) in_mfma(t2 + 2, 2 * par, 2 * par + 2);
                        else if (par == 0) in_mfma(t2 + 2, 0, KS);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (MODE == 3) { pf_mfma(par); __builtin_amdgcn_sched_barrier(0); }
                    if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(st[1])); __builtin_amdgcn_sched_barrier(0); }
                    v4i b[KS - 1];
#pragma unroll
                    for (int ks = 0; ks < KS - 1; ++ks) b[ks] = *reinterpret_cast<const v4i*>(hc + boff + ks * 64);
                    const long bt = *reinterpret_cast<const long*>(hc + boft);
                    v4i a[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
#pragma unroll
                    for (int ks = 0; ks < KS - 1; ++ks)
#pragma unroll
                        for (int d = 0; d < 3; ++d) a[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Whh[ks][d], b[ks], a[d], 0, 0, 0);
#pragma unroll
                    for (int d = 0; d < 3; ++d) a[d] = __builtin_amdgcn_mfma_i32_16x16x32_i8(Wht[d], bt, a[d], 0, 0, 0);
                    if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(st[2])); __builtin_amdgcn_sched_barrier(0); }
                    if constexpr (EPI > 0) {
#pragma unroll
                        for (int rep = 0; rep < EPI; ++rep) {
                            int v[3][2];
#pragma unroll
                            for (int d = 0; d < 3; ++d) {
                                v[d][0] = __builtin_amdgcn_update_dpp(a[d][0], a[d][2], 0x118, 0xf, 0xC, false);
                                v[d][1] = __builtin_amdgcn_update_dpp(a[d][1], a[d][3], 0x118, 0xf, 0xC, false);
                            }
                            int ri[2];
#pragma unroll
                            for (int j = 0; j < 2; ++j) ri[j] = (v[2][j] << 16) + (v[1][j] << 8) + v[0][j];
                            if (rep == 0) {
                                // the state fragments and the accumulators are dead from here on: their registers take the operands of the
                                // next input product, which arrive from LDS under the cell's dependent chain
                                if constexpr (MODE == 4 || MODE == 5) { __builtin_amdgcn_sched_barrier(0); pf_load(t2 + 2, par); __builtin_amdgcn_sched_barrier(0); }
                                if constexpr (MODE == 3) { __builtin_amdgcn_sched_barrier(0); pf_load(par == 0 ? t2 + 2 : t2 + 4, par ^ 1); __builtin_amdgcn_sched_barrier(0); }
                            }
                            unsigned pk = 0;
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const float rec = (float)ri[j];
                                const float pre_f = __builtin_fmaf(rec, dq[j], zc[par][j]);
                                const float pre_g = pre_f + db[j];
                                const float f = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre_f * -1.44269504088896341f));
                                const float m = __builtin_fmaf(f, c[j] - pre_g, pre_g);
                                const float y = __builtin_fmaf(m, al[j], be[j]);
                                c[j] = y;
                                pk |= (y >= 0.0f) ? (1u << (8 * j)) : 0u;
                            }
                            *reinterpret_cast<unsigned short*>(hn + hoff) = (unsigned short)pk;
                            hs += (int)pk;
                            if (rep + 1 < EPI) { a[0][0] += (int)pk; asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2])); }
                        }
                    } else {
                        hs += a[0][0] + a[1][1] + a[2][2];
                    }
                    if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(st[3])); __builtin_amdgcn_sched_barrier(0); }
                    if constexpr (MODE == 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (SPLIT == 2) in_mfma(t2 + 2, 0, KS);  // (model of a 16-row role: the whole input product EVERY step)
                        else if constexpr (SPLIT) in_mfma(t2 + 2, 2 * par, 2 * par + 2);
                        else if (par == 0) in_mfma(t2 + 2, 0, KS);
                    }
                    if constexpr (MODE == 4 || MODE == 5) {
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (MODE == 4) __builtin_amdgcn_s_setprio(0);
                        pf_mfma(par);
                        if constexpr (MODE == 4) __builtin_amdgcn_s_setprio(1);
                    }
                    if (MODE != 0 && par == 1) in_finish();
                    if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(st[4])); __builtin_amdgcn_sched_barrier(0); }
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_s_barrier();
                    if constexpr (STAMP) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st[5])); }
                }
            }
        };
        __builtin_amdgcn_s_barrier();
        const long long t0 = __builtin_readcyclecounter();
        using std::integral_constant;
        // PLACE: 0 none | 1 tail | 2 head | 3 last wave of a SIMD head, others tail | 4 last head-prefetched, others tail
        //        | 5 last head-prefetched, others tail at low priority | 6 all head-prefetched | 7 last head-prefetched, others none
        //        (what the last waves' products cost alone) | 8 last none, others tail (what the others' cost alone)
        if constexpr (PLACE == 0) loop(integral_constant<int, 0>{});
        else if constexpr (PLACE == 1) loop(integral_constant<int, 1>{});
        else if constexpr (PLACE == 2) loop(integral_constant<int, 2>{});
        else if constexpr (PLACE == 3) { if (last_of_simd) loop(integral_constant<int, 2>{}); else loop(integral_constant<int, 1>{}); }
        else if constexpr (PLACE == 4) { if (last_of_simd) loop(integral_constant<int, 3>{}); else loop(integral_constant<int, 5>{}); }
        else if constexpr (PLACE == 5) { if (last_of_simd) loop(integral_constant<int, 3>{}); else loop(integral_constant<int, 4>{}); }
        else if constexpr (PLACE == 6) loop(integral_constant<int, 3>{});
        else if constexpr (PLACE == 7) { if (last_of_simd) loop(integral_constant<int, 3>{}); else loop(integral_constant<int, 0>{}); }
        else if constexpr (PLACE == 8) { if (last_of_simd) loop(integral_constant<int, 0>{}); else loop(integral_constant<int, 5>{}); }
        else if constexpr (PLACE == 9) { if (last_of_simd) loop(integral_constant<int, 0>{}); else loop(integral_constant<int, 4>{}); }
        else if constexpr (PLACE == 10) loop(integral_constant<int, 5>{});
        else loop(integral_constant<int, 4>{});
        const long long t1 = __builtin_readcyclecounter();
        p.cout[(size_t)(blockIdx.x * 8 + row) * H + cj] = c[0];
        p.cout[(size_t)(blockIdx.x * 8 + row) * H + cj + 1] = c[1];
        atomicAdd(p.hsum + blockIdx.x, hs);
        if (tid == 0 && blockIdx.x == 0) p.clk[0] = t1 - t0;
    } else {
        __builtin_amdgcn_s_barrier();
#pragma unroll 1
        for (int t = 0; t < ((p.T + 1) & ~1); ++t) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
    }
    if constexpr (STAMP) if (blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 6; ++k) p.stamps[wave * 8 + k] = st[k];
}

// ---------------------------------------------------------------------------------------------------------------------
template <class K>
static void run(const char* name, K kern, PArgs a, int grid, int T, bool stamps) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    PArgs w = a; w.T = 50;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), LDS_BYTES, 0, w);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    long long clk = 0;
    int hsum = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(a.hsum, 0, grid * 4));
        a.T = T;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), LDS_BYTES, 0, a);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) { best = ms; CK(hipMemcpy(&clk, a.clk, 8, hipMemcpyDeviceToHost)); }
        int hh[256]; CK(hipMemcpy(hh, a.hsum, grid * 4, hipMemcpyDeviceToHost));
        hsum = 0; for (int i = 0; i < grid; ++i) hsum ^= hh[i] * (i + 1);
    }
    printf("%-64s grid %3d : %7.1f ns/step  %7.1f clk/step  (launch %.3f ms, checksum %08x)\n", name, grid, best * 1e6 / T, (double)clk / T, best, (unsigned)hsum);
    if (stamps) {
        long long hs[16 * 8]; CK(hipMemcpy(hs, a.stamps, sizeof(hs), hipMemcpyDeviceToHost));
        long long m0 = hs[0]; for (int w2 = 0; w2 < NT; ++w2) if (hs[w2 * 8] < m0) m0 = hs[w2 * 8];
        for (int w2 = 0; w2 < NT; ++w2)
            printf("   wave %2d (simd %d slot %d): top +%4lld  head-in +%4lld  rec-issued +%4lld  epi +%4lld  tail-in +%4lld  past-barrier +%4lld\n", w2, w2 & 3, w2 >> 2,
                   hs[w2 * 8] - m0, hs[w2 * 8 + 1] - m0, hs[w2 * 8 + 2] - m0, hs[w2 * 8 + 3] - m0, hs[w2 * 8 + 4] - m0, hs[w2 * 8 + 5] - m0);
    }
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int T = 1000, GRID = 104;
    srand(1);
    const size_t wbytes = (size_t)3 * NT * KS * 1024;
    int8_t* hw = (int8_t*)malloc(wbytes);
    PArgs a{};
    for (int m = 0; m < 2; ++m) {
        for (size_t i = 0; i < wbytes; ++i) {
            const int ks = (int)((i / 1024) % KS), kk = ks * 64 + (int)(((i % 1024) / 16) / 16) * 16 + (int)(i % 16);  // A fragment: lane = (q, n), 16 k per lane
            hw[i] = kk < H ? (int8_t)((rand() & 0xff) - 128) : 0;
        }
        int8_t* d; CK(hipMalloc(&d, wbytes)); CK(hipMemcpy(d, hw, wbytes, hipMemcpyHostToDevice));
        if (m == 0) a.w_hh = d; else a.w_ih = d;
    }
    {
        float* h = (float*)malloc(6 * HP * 4);
        for (int j = 0; j < HP; ++j) {
            h[j] = 1.0f / 8388608.f; h[HP + j] = 0.1f; h[2 * HP + j] = 1.1f; h[3 * HP + j] = -0.05f;
            h[4 * HP + j] = 1.0f / 8388608.f; h[5 * HP + j] = 0.02f * ((j % 7) - 3);
        }
        float* d; CK(hipMalloc(&d, 6 * HP * 4)); CK(hipMemcpy(d, h, 6 * HP * 4, hipMemcpyHostToDevice)); a.cst = d; free(h);
    }
    CK(hipMalloc(&a.cout, (size_t)GRID * 8 * H * 4));
    CK(hipMalloc(&a.hsum, 256 * 4));
    CK(hipMalloc(&a.clk, 64));
    CK(hipMalloc(&a.stamps, 16 * 8 * 8));
    CK(hipMemset(a.stamps, 0, 16 * 8 * 8));

#define RUN(PLACE, SPLIT, PRIO, EPI) run("l2 place=" #PLACE " split=" #SPLIT " prio=" #PRIO " epi=" #EPI, pair_l2_kernel<PLACE, SPLIT, PRIO, EPI, 0>, a, GRID, T, false)
#define RUNS(PLACE, SPLIT, PRIO, EPI) run("l2 STAMPED place=" #PLACE " split=" #SPLIT " prio=" #PRIO " epi=" #EPI, pair_l2_kernel<PLACE, SPLIT, PRIO, EPI, 1>, a, GRID, T, true)
    printf("== 8-row IO-wave scan step; +12 input-product MFMAs per tile and two steps (W_ih planes 0-1 in LDS, plane 2 in registers) ==\n");
    RUN(0, 1, 0, 1);
    RUN(1, 1, 0, 1); RUN(4, 1, 0, 1); RUN(5, 1, 0, 1); RUN(5, 1, 1, 1); RUN(6, 1, 0, 1); RUN(7, 1, 0, 1); RUN(8, 1, 0, 1); RUN(9, 1, 0, 1); RUN(10, 1, 0, 1); RUN(11, 1, 0, 1);
    printf("== model of a 16-row fused role: all 12 input-product MFMAs behind the epilogue EVERY step; epilogue once / twice (4 values per lane) ==\n");
    RUN(1, 2, 0, 1); RUN(1, 2, 0, 2); RUN(0, 1, 0, 2);
    printf("== per-wave timeline of the last step ==\n");
    RUNS(7, 1, 0, 1); RUNS(8, 1, 0, 1); RUNS(9, 1, 0, 1); RUNS(5, 1, 1, 1); RUNS(10, 1, 0, 1);
    return 0;
}
