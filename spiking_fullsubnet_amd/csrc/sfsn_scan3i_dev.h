// sfsn_scan3i_dev.h -- the IO-wave scan of a layer >= 1 that forms its OWN input term (round 4), gfx950 only.
//
// Layer l+1 needs only frame t of layer l (efficient_spiking_neuron.py:56-61), and its input term S_l(t) . W_ih^T + b
// (efficient_spiking_neuron.py:140-145) is a product with a binary left operand like the recurrent one.  Round 3's scan3 role read
// that term as fp32 [T][R][H] (745 MB per sub-band layer at B = 64, T = 1000, written by sfsn_spike_proj just before); this role
// reads the previous layer's int8 spikes instead (a quarter of the bytes, no separate product launch) and runs the product on the
// matrix pipe beside the recurrent one:
//   * 8 rows per workgroup leave MFMA columns 8..15 idle, so the input product is batched over TWO frames: columns 0..7 = the rows
//     at frame f, columns 8..15 = the same rows at frame f + 1 -- 12 matrix instructions per tile and two steps instead of 12 per step
//     (the recurrent product cannot be batched: it needs h(t - 1));
//   * both weight matrices of a tile do not fit a wave's 128 registers (16 waves per CU): W_hh stays register resident (the k tail
//     of H = 224 / 160 as ONE 16x16x32 step with 8-byte fragments: 42 instead of 48 registers), digit planes 0 and 1 of W_ih sit in
//     LDS (2 x NT x KS KiB, read as A fragments: 1 KiB contiguous per wave instruction, conflict free), plane 2 in registers;
//   * the input product of frames (f, f + 1) runs during steps f - 2 and f - 1, half of its k-steps in each, BEHIND the wave's
//     epilogue; its operands are fetched from LDS under the epilogue's dependent chain into the registers the state fragments and
//     the recurrent accumulators have just left dead; the finished term is re-dealt with two DPP row shifts per frame into the
//     epilogue's layout and stays in registers (it never exists in LDS or HBM);
//   * loader wave: the previous layer's int8 rows -> an LDS ring (LDS-DMA, 2 KiB per frame, chunk (c + 2 r) mod 16 of row r: the
//     bank spread of the state buffer's 288-byte stride), gated on the producers' progress counters when they run in this launch;
//     storer wave and spare waves: scan3_role's.
// Same exact integer products and the same two roundings as sfsn_spike_proj + sfsn_gsn_layer_scan: fma(exact sum, dq_ih, b_f),
// then fma(exact sum, dq_hh, that) -- bit-identical results (tests/test_stack_scan.py).
// What it costs (scripts/micro/pair_step.hip, H = 224, no global traffic): 1860-1890 clk per step against 1400 for the plain 8-row
// step -- within a SIMD the matrix pipe (72 x 16 clk) and the four epilogues' VALU time largely serialise, wherever the extra 24
// matrix instructions are placed (head of the step, behind the epilogue, prefetched, by priority).
#ifndef SFSN_SCAN3I_DEV_H
#define SFSN_SCAN3I_DEV_H
#include "sfsn_scan3_dev.h"

template <int KS, int FLG>
struct Scan3iCfg {
    static constexpr int RPW = 8, HP = KS * 64, LDH = HP + 32, NCH = HP / 16;
    static constexpr bool GATED = (FLG & 1) != 0, PUB = (FLG & 2) != 0;
    // ring slot: 8 rows x 16 chunk positions of 16 bytes (positions whose chunk lies beyond HP are padding), two 1 KiB DMA pieces
    static constexpr int NP = 2, SLOT = NP * 1024;
    // frame t + A is requested during step t; frames up to t + 4 have landed when step t + 1 starts (its input product reads
    // frames t + 3, t + 4 at the latest).  An sc1 load of rows another workgroup has just written through takes ~3 us.
    static constexpr int A = GATED ? 10 : 6;
    static constexpr int D = A + 2;  // slots: the DMA of step t reuses the slot of frame t - 2 (dead for two barriers)
    static constexpr int HBUF_OFF = D * SLOT;
    static constexpr int WIH_OFF = HBUF_OFF + 2 * 16 * LDH;
    __host__ __device__ static constexpr int plane_bytes(int NT) { return NT * KS * 1024; }
    __host__ __device__ static constexpr int csti_off(int NT) { return WIH_OFF + 2 * plane_bytes(NT); }
    __host__ __device__ static constexpr int flag_off(int NT) { return csti_off(NT) + (HP / 2) * 16; }
    __host__ __device__ static constexpr int lds_bytes(int NT) { return flag_off(NT) + 16; }
};

struct Scan3iRole {
    const int8_t* spikes_in;  // the previous layer's int8 spikes [T][R][HP]
    const int8_t* w_ih;       // packed digits [3][NT][KS][64][16]
    const float* w_ih_dq;
    const int8_t* w_hh;
    const float* w_dq;
    const float* bias;        // [2 H]: b_f, b_g
    const float* bn_alpha;
    const float* bn_beta;
    float* h_state;
    float* c_state;
    float* spikes_f32;
    int8_t* spikes_i8;
    int R, row0;
    unsigned long long* count = nullptr;  // (as Scan3Role::count)
#ifdef SFSN_EXPERIMENTS
    unsigned long long* probe = nullptr;  // (as Scan3Role::probe)
#endif
};

// TL = 1: H mod 64 is in (0, 32]: the last k-step of the RECURRENT product is ONE 16x16x32 matrix instruction whose 8-byte fragments
// are cut out of the 16x16x64 fragment layout (lane (n, q) of the 32-wide step holds k = 8 q + j = bytes [(q & 1) * 8, +8) of lane
// (n, q >> 1)): six registers less of W_hh.
// FLG / OUT as in scan3_role.  8 rows per workgroup.
// D0 = 1 (round 6): the weights were packed with 16 bits (sfsn_w3_pack_bits): digit plane 0 of BOTH matrices is zero and its matrix
// instructions are skipped -- 12 instead of 18 per tile and step, the same sums (the 16-bit report mode, sfsn_gsn_stack_scan_x_w16).
template <int KS, int TL, int OUT, int FLG, int D0 = 0>
__device__ __forceinline__ void scan3i_role(const Scan3iRole& rl, const StackLink& lk, char* smem, int T, int H, int NT, int exp_flags = 0) {
    using C = Scan3iCfg<KS, FLG>;
    constexpr int RPW = 8, LDH = C::LDH, HP = C::HP, D = C::D, A = C::A, SLOT = C::SLOT, NP = C::NP, NCH = C::NCH;
    constexpr bool GATED = C::GATED, PUB = C::PUB;
    constexpr bool CWF = SFSN_S3_CWF && (OUT & 1);  // the compute waves write the fp32 spikes (see SFSN_S3_CWF)
    constexpr bool LSF = !CWF && SFSN_S3_LSF && PUB && (OUT & 1);
    constexpr int KSF = TL ? KS - 1 : KS;   // full 64-wide k-steps
    constexpr int NK = KS;                  // k-steps in all
    constexpr int NKA = NK < 2 ? NK : 2;    // k-steps [0, NKA) of an input product run in the even step, the rest (at most two) in the odd one
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int R = rl.R, row0 = rl.row0;
    const int PLANE = C::plane_bytes(NT);
    int8_t* hbuf = reinterpret_cast<int8_t*>(smem + C::HBUF_OFF);
    volatile int* flag = reinterpret_cast<volatile int*>(smem + C::flag_off(NT));

    // ---- set-up by all threads: state buffers zeroed, h_{-1} -> hbuf[0], W_ih planes 0 / 1 and the input product's constants -> LDS
    for (int i = tid; i < 2 * 16 * LDH / 4; i += 1024) reinterpret_cast<int*>(hbuf)[i] = 0;
    if (tid == 0) { flag[0] = 0; flag[1] = 0; }
    for (int i = tid; i < 2 * PLANE / 16; i += 1024) {
        const int d = i / (PLANE / 16), r = i - d * (PLANE / 16);
        reinterpret_cast<v4i*>(smem + C::WIH_OFF)[i] = *reinterpret_cast<const v4i*>(rl.w_ih + (size_t)d * PLANE + (size_t)r * 16);
    }
    for (int i = tid; i < HP / 2; i += 1024) {
        v4f cq = {0.f, 0.f, 0.f, 0.f};
        if (2 * i < H) cq = v4f{rl.w_ih_dq[2 * i], rl.w_ih_dq[2 * i + 1], rl.bias[2 * i], rl.bias[2 * i + 1]};
        reinterpret_cast<v4f*>(smem + C::csti_off(NT))[i] = cq;
    }
    __syncthreads();
    for (int idx = tid; idx < RPW * (H / 4); idx += 1024) {
        const int rr = idx / (H / 4), j4 = (idx - rr * (H / 4)) * 4;
        const int rsrc = row0 + rr < R ? row0 + rr : R - 1;  // (rows past R duplicate row R-1 in every value: see scan3_role)
        const v4f h = *reinterpret_cast<const v4f*>(rl.h_state + (size_t)rsrc * H + j4);
        const unsigned pk = (h.x > 0.5f ? 1u : 0u) | (h.y > 0.5f ? 0x100u : 0u) | (h.z > 0.5f ? 0x10000u : 0u) |
                            (h.w > 0.5f ? 0x1000000u : 0u);
        *reinterpret_cast<unsigned*>(hbuf + rr * LDH + j4) = pk;
    }

    if (wave < NT) {
        // ================================================= compute wave: output tile `wave` =================================================
        const int ct = wave;
        const int row = n & 7, sub = 2 * (n >> 3);
        const int cj = ct * 16 + q * 4 + sub;  // my first neuron (two adjacent ones per lane)
        const bool live = row0 + row < R;
        const int grow = live ? row0 + row : R - 1;
        const unsigned toff = (unsigned)((((q >> 1) * 16 + n) * 16) + (q & 1) * 8);  // my 8 bytes of a k-tail fragment
        v4i Whh[KSF > 0 ? KSF : 1][3], Wi2[KS];
        long Wht[3] = {0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KSF; ++ks)
#pragma unroll
            for (int d = 0; d < 3; ++d)
                Whh[ks][d] = *reinterpret_cast<const v4i*>(rl.w_hh + ((((size_t)d * NT + ct) * KS + ks) * 64 + lane) * 16);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) Wi2[ks] = *reinterpret_cast<const v4i*>(rl.w_ih + ((((size_t)2 * NT + ct) * KS + ks) * 64 + lane) * 16);
        if constexpr (TL) {
#pragma unroll
            for (int d = 0; d < 3; ++d) Wht[d] = *reinterpret_cast<const long*>(rl.w_hh + (((size_t)d * NT + ct) * KS + KS - 1) * 1024 + toff);
        }
        float c[2], dq[2], db[2], al[2], be[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            c[j] = rl.c_state[(size_t)grow * H + cj + j];
            dq[j] = rl.w_dq[cj + j];
            db[j] = rl.bias[H + cj + j] - rl.bias[cj + j];
            al[j] = rl.bn_alpha[cj + j];
            be[j] = rl.bn_beta[cj + j];
        }
        const unsigned boff = (unsigned)(n * LDH + q * 16);
        const unsigned boft = (unsigned)(n * LDH + (KS - 1) * 64 + q * 8);
        const unsigned hoff = (unsigned)(row * LDH + cj);
        // my B fragments of the input product: column n = (frame f0 + (n >> 3), row n & 7); k chunk c at position (c + 2 row) & 15
        unsigned soff[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) soff[ks] = (unsigned)((n >> 3) * SLOT + row * 256 + ((ks * 4 + q + 2 * row) & 15) * 16);
        const unsigned woff = (unsigned)(C::WIH_OFF + (ct * KS) * 1024 + lane * 16);
        const unsigned cqoff = (unsigned)(C::csti_off(NT) + (cj >> 1) * 16);
        const unsigned foff = (unsigned)(((size_t)grow * H + cj) * 4);  // my fp32 spikes within a frame of [T][R][H]
        const size_t fframe = (size_t)R * H;
        float zc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};  // the input term of my two neurons at the two frames of the current pair
        v4i e[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
        v4i pfb[2], pfw0[2], pfw1[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) pfb[i] = pfw0[i] = pfw1[i] = v4i{0, 0, 0, 0};

        // operands of k-steps [k0, k0 + 2) of the input product of frames (f0, f0 + 1), f0 even (their ring slots are adjacent).
        // PART 0: the spike fragments and digit plane 0 of W_ih (requested under the cell's dependent chain); PART 1: digit plane 1
        // (requested behind the cell, its latency covered by the matrix instructions of planes 0 and 2) -- all three at once would
        // need 24 registers at the point where the cell's temporaries are live: 128 VGPRs do not hold that without spills.
        // The input product uses full 16x16x64 steps throughout (zero padded k): a 16x16x32 step accumulating onto a 16x16x64
        // step's result ONE or TWO matrix instructions later returned wrong sums (scripts/micro/pair_role.hip: every case with a
        // full step and the tail step in the same half failed, H = 160 with the tail alone in its half did not; hipcc puts no
        // wait states between the two shapes) -- the recurrent product's tail step follows its accumulator by three instructions.
        auto pf_load = [&](int f0, int k0, int part) __attribute__((always_inline)) {
            const char* ring = smem + (f0 % D) * SLOT;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ks = k0 + i;
                if (ks >= NK) continue;
                if (part == 0) {
                    pfb[i] = *reinterpret_cast<const v4i*>(ring + soff[ks]);
                    if constexpr (!D0) pfw0[i] = *reinterpret_cast<const v4i*>(smem + woff + ks * 1024);
                } else {
                    pfw1[i] = *reinterpret_cast<const v4i*>(smem + woff + PLANE + ks * 1024);
                }
            }
        };
        auto pf_mfma = [&](int k0) __attribute__((always_inline)) {
            // planes 0 and 2 of both k-steps first, plane 1 (whose fragments were requested last) behind them
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ks = k0 + i;
                    if (ks >= NK) continue;
                    if (pass == 0) {
                        if constexpr (!D0) e[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(pfw0[i], pfb[i], e[0], 0, 0, 0);
                        e[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wi2[ks], pfb[i], e[2], 0, 0, 0);
                    } else {
                        e[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(pfw1[i], pfb[i], e[1], 0, 0, 0);
                    }
                }
            }
        };
        // accumulators -> the input term of both frames in the epilogue's layout: z = fma(exact sum, dq_ih, b_f) (= sfsn_spike_proj),
        // frame f0 = columns 0..7: lanes 8..15 of a row of 16 take elements 2, 3 of the lane 8 below; frame f0 + 1 = columns 8..15:
        // lanes 0..7 take elements 0, 1 of the lane 8 above.  (dq_ih and b_f of my two neurons: cq = {dq0, dq1, b0, b1}.)
        auto in_finish = [&](const v4f cq) __attribute__((always_inline)) {
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = (float)((e[2][k] << 16) + (e[1][k] << 8) + e[0][k]);
            const int f00 = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, r[0]), __builtin_bit_cast(int, r[2]), 0x118, 0xf, 0xC, false);
            const int f01 = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, r[1]), __builtin_bit_cast(int, r[3]), 0x118, 0xf, 0xC, false);
            const int f10 = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, r[2]), __builtin_bit_cast(int, r[0]), 0x108, 0xf, 0x3, false);
            const int f11 = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, r[3]), __builtin_bit_cast(int, r[1]), 0x108, 0xf, 0x3, false);
            zc[0][0] = __builtin_fmaf(__builtin_bit_cast(float, f00), cq.x, cq.z);
            zc[0][1] = __builtin_fmaf(__builtin_bit_cast(float, f01), cq.y, cq.w);
            zc[1][0] = __builtin_fmaf(__builtin_bit_cast(float, f10), cq.x, cq.z);
            zc[1][1] = __builtin_fmaf(__builtin_bit_cast(float, f11), cq.y, cq.w);
            e[0] = e[1] = e[2] = v4i{0, 0, 0, 0};
        };

        __syncthreads();                       // initial state in hbuf[0], W_ih planes and constants in LDS
        __builtin_amdgcn_s_barrier();          // the loader's prologue frames (0 .. A - 1) have landed
        // the input term of frames 0, 1 (all k-steps); the loop below forms frames t2 + 2, t2 + 3 during steps t2, t2 + 1
        pf_load(0, 0, 0); pf_load(0, 0, 1);
        pf_mfma(0);
        if constexpr (NK > 2) { pf_load(0, 2, 0); pf_load(0, 2, 1); pf_mfma(2); }
        in_finish(*reinterpret_cast<const v4f*>(smem + cqoff));
        int stop = 0;
        float sp[2] = {0.f, 0.f};
        S3_PB_DECL();
#pragma unroll 1
        for (int t2 = 0; t2 < T && !stop; t2 += 2) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int t = t2 + par;
                if (par == 1 && t >= T) break;
                const int8_t* hc = hbuf + (t & 1) * 16 * LDH;
                int8_t* hn = hbuf + ((t & 1) ^ 1) * 16 * LDH;
                if constexpr (GATED) stop = flag[t & 1];  // written by the loader during step t-1 (or before)
                v4i b[KSF > 0 ? KSF : 1];
#pragma unroll
                for (int ks = 0; ks < KSF; ++ks) b[ks] = *reinterpret_cast<const v4i*>(hc + boff + ks * 64);
                long bt = 0;
                if constexpr (TL) bt = *reinterpret_cast<const long*>(hc + boft);
                if constexpr (CWF) if (t > 0) s3_store_spikes<2>(rl.spikes_f32 + (size_t)(t - 1) * fframe, foff, sp);  // (under the LDS wait)
                v4i a[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
                if constexpr (TL) {
                    // The 32-wide tail step FIRST, from zero accumulators, as one block the compiler cannot reorder, with the wait
                    // states a 16x16x64 step needs before it may accumulate onto a 16x16x32 step's result: hipcc treats the two
                    // shapes as back-to-back compatible (no wait states), the hardware does not forward between them -- sums came
                    // out wrong whenever the two met within two matrix instructions (scripts/micro/pair_role.hip; in a kernel
                    // where the scheduler happened to move them together: H = 96 in tests/test_stack_scan.py).
                    if constexpr (D0) {
                        asm volatile(
                            "v_mfma_i32_16x16x32_i8 %0, %2, %4, 0\n\t"
                            "v_mfma_i32_16x16x32_i8 %1, %3, %4, 0\n\t"
                            "s_nop 7"
                            : "=&v"(a[1]), "=&v"(a[2])
                            : "v"(Wht[1]), "v"(Wht[2]), "v"(bt));
                    } else {
                        asm volatile(
                            "v_mfma_i32_16x16x32_i8 %0, %3, %6, 0\n\t"
                            "v_mfma_i32_16x16x32_i8 %1, %4, %6, 0\n\t"
                            "v_mfma_i32_16x16x32_i8 %2, %5, %6, 0\n\t"
                            "s_nop 5"
                            : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2])
                            : "v"(Wht[0]), "v"(Wht[1]), "v"(Wht[2]), "v"(bt));
                    }
                    if constexpr (KSF == 0) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // (the epilogue reads them next)
                }
#pragma unroll
                for (int ks = 0; ks < KSF; ++ks)
#pragma unroll
                    for (int d = D0; d < 3; ++d) a[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Whh[ks][d], b[ks], a[d], 0, 0, 0);
                // columns 0..7 are live: lanes 8..15 of a row of 16 take elements 2, 3 of the lane 8 below them; exact sum (= recombine3)
                int ri[2];
                {
                    int v[3][2];
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        v[d][0] = __builtin_amdgcn_update_dpp(a[d][0], a[d][2], 0x118, 0xf, 0xC, false);
                        v[d][1] = __builtin_amdgcn_update_dpp(a[d][1], a[d][3], 0x118, 0xf, 0xC, false);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) ri[j] = (v[2][j] << 16) + (v[1][j] << 8) + v[0][j];
                }
                // the state fragments and the accumulators are dead from here on: their registers take the operands of this step's
                // half of the input product, which arrive from LDS under the cell's dependent chain
                __builtin_amdgcn_sched_barrier(0);
                pf_load(t2 + 2, par == 0 ? 0 : NKA, 0);
                __builtin_amdgcn_sched_barrier(0);
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
                    sp[j] = (y >= 0.0f) ? 1.0f : 0.0f;
                }
                *reinterpret_cast<unsigned short*>(hn + hoff) = (unsigned short)pk;
                __builtin_amdgcn_sched_barrier(0);
                pf_load(t2 + 2, par == 0 ? 0 : NKA, 1);
                v4f cq = {0.f, 0.f, 0.f, 0.f};
                if (par == 1) cq = *reinterpret_cast<const v4f*>(smem + cqoff);  // (arrives under the matrix instructions)
                pf_mfma(par == 0 ? 0 : NKA);
                if (par == 1) in_finish(cq);
                __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
                S3_PB_TIC();
                __builtin_amdgcn_s_barrier();
                S3_PB_TOC(0);
                if constexpr (GATED) {
                    stop = __builtin_amdgcn_readfirstlane(stop);
                    if (stop) break;
                }
            }
        }
        S3_PB_OUT(rl, wave, lane);
        if constexpr (CWF) if (T > 0 && !stop) s3_store_spikes<2>(rl.spikes_f32 + (size_t)(T - 1) * fframe, foff, sp);
        // final state
        const int8_t* hl = hbuf + (T & 1) * 16 * LDH;
        if (live) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                rl.c_state[(size_t)grow * H + cj + j] = c[j];
                rl.h_state[(size_t)grow * H + cj + j] = (float)hl[hoff + j];
            }
        }
        return;
    }

    if (wave == NT) {
        // ================================================= loader wave =================================================
        // piece p, lane: chunk position e = 64 p + lane of the slot = (row e / 16, position e % 16) <- global chunk (position - 2 row) mod 16
        unsigned goff[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int e = 64 * p + lane, r = e >> 4, pos = e & 15;
            int cch = (pos - 2 * r) & 15;
            if (cch >= NCH) cch = 0;  // padding position (HP < 256): any valid chunk, never read
            const int grow = (row0 + r < R) ? row0 + r : R - 1;
            goff[p] = (unsigned)(grow * HP + cch * 16);
        }
        const size_t frame = (size_t)R * HP;
        int avail = GATED ? 0 : T;
        int failed = 0;
        S3FlushF<RPW, LDH> ff;
        if constexpr (LSF) ff.init(lane, row0, R, H);
        // frames t + 5 .. t + A may stay in flight behind the wait of step t (plus the stores of the steps in between)
        int allow = (A - 4) * NP;
        if constexpr (LSF) allow = (A - 4) * (NP + ff.nsf) + ff.nsf;
        if (allow > 62) allow = 62;
        auto ensure = [&](int need) __attribute__((always_inline)) {
            if constexpr (GATED) s3_ensure(lk, need, T, avail, failed, lane);
        };
        auto issue = [&](int slot, int td) __attribute__((always_inline)) {
            const float* st = reinterpret_cast<const float*>(rl.spikes_in + (size_t)td * frame);
#pragma unroll
            for (int p = 0; p < NP; ++p)
                dma16_to_lds<GATED>(__builtin_amdgcn_readfirstlane((unsigned)(slot * SLOT + p * 1024)), st, goff[p]);
        };
        __syncthreads();
        ensure(A < T ? A : T);
        if (!failed)
            for (int s0 = 0; s0 < A; ++s0) issue(s0, s0 < T ? s0 : (T > 0 ? T - 1 : 0));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (GATED) if (failed && lane == 0) flag[0] = 1;  // read during step 0
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        int stop = 0;
        S3_PB_DECL();
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            if constexpr (GATED) stop = failed;  // what the other waves read from flag[t & 1] during this step
            const int td = (t + A < T) ? t + A : T - 1;
            S3_PB_TIC();
            ensure(td + 1);
            S3_PB_TOC(2);
            if (!failed) issue((t + A) % D, td);
            if constexpr (GATED) if (failed && lane == 0) flag[(t + 1) & 1] = 1;  // read during step t+1 (see scan3_role)
            if constexpr (LSF) if (t > 0) ff.run(hbuf + (t & 1) * 16 * LDH, rl.spikes_f32 + ((size_t)(t - 1) * R + row0) * H, lane);
            S3_PB_TIC();
            wait_vmcnt_n(allow);
            S3_PB_TOC(1);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            S3_PB_TIC();
            __builtin_amdgcn_s_barrier();
            S3_PB_TOC(0);
            if constexpr (GATED) if (stop) break;
        }
        S3_PB_OUT(rl, 14, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // DMAs past the end are invisible to the compiler
        if constexpr (LSF) if (T > 0 && !(GATED && stop)) ff.run(hbuf + (T & 1) * 16 * LDH, rl.spikes_f32 + ((size_t)(T - 1) * R + row0) * H, lane);
        return;
    }

    if (wave == NT + 1) {
        // ================================================= storer wave (scan3_role's) =================================================
        constexpr int MAX8 = (RPW * KS * 4 + 63) / 64;
        constexpr int nu8 = RPW * (HP / 16), ns8 = (nu8 + 63) / 64;
        constexpr bool F32 = (OUT & 1) && !LSF && !CWF;
        S3FlushF<RPW, LDH> ff;
        if constexpr (F32) ff.init(lane, row0, R, H);
        int l8[MAX8];
        unsigned ok8 = 0;
        unsigned cnt = 0;  // (as scan3_role's storer)
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
        const int pf = spf > 0 ? (62 / spf < SFSN_S3_PFMAX ? 62 / spf : SFSN_S3_PFMAX) : 8;  // frames of my stores that may be in flight
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
                if constexpr (PUB) {
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
        S3_PB_OUT(rl, 15, lane);
        if (T > 0 && !(GATED && __builtin_amdgcn_readfirstlane(stop))) {
            const int8_t* hl = hbuf + (T & 1) * 16 * LDH;
            flush8(hl, T - 1);
            flushf(hl, T - 1);
        }
        if constexpr (PUB) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) stack_publish(lk, T);  // (also after an expired spin: consumers must not wait for us)
        }
        if constexpr (!(OUT & 1)) wave_count_add(rl.count, cnt);
        return;
    }

    // ================================================= spare waves (NT < 14): keep the barrier count =================================================
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
