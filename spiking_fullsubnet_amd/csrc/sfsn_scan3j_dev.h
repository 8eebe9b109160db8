// sfsn_scan3j_dev.h -- the 16-row fused-input scan with IO-specialised waves (round 6), gfx950 only.
//
// sfsn_gsn_layer_scan_fused is the scan bench.py's timed region runs for every layer >= 1 of the sub-band models (16 rows per
// workgroup: per CU-time the geometry that carries most rows).  Until round 6 it ran round 2's body (gsn_scan_fused_kernel: 8 waves x
// 2 tiles, every wave fetching its own share of the input spikes, flushing its share of the outputs, per-neuron constants read from
// LDS inside a 230-instruction epilogue): 1.55 us per step, of which the matrix pipe holds 0.56.  What rounds 3-5 measured: within a
// SIMD the matrix pipe's time and the VALU's time ADD (scripts/micro/pingpong_step.hip: not even two independent row blocks per
// workgroup overlap them), so a step costs (matrix instructions x 16 clk) + (VALU instructions x ~4.5 clk) per SIMD, and the way to a
// shorter step is fewer VALU instructions.  This role is scan3i_role's structure at 16 rows:
//   compute waves (one 16-neuron tile each, wave < NT <= 14): h(t-1) fragments from LDS, 12 recurrent matrix instructions, the cell on
//       the wave's four values per lane (no re-deal: all 16 MFMA columns are live), new spikes to LDS; then -- off the step's
//       dependency chain -- the input product of frame t + 1 (12 matrix instructions, W_ih planes 0 / 1 from LDS, plane 2 in registers,
//       spike fragments from the loader's ring), finished to fma(exact sum, dq_ih, b_f) in four registers for the next step.
//       No global memory instruction, no address arithmetic in the loop.
//   loader wave: the previous layer's int8 rows -> an LDS ring by LDS-DMA (4 KiB per frame, chunk (c + r) mod 16 of row r).
//   storer wave: the spikes of frame t - 1, LDS -> global as whole contiguous blocks (fp32 + int8), counting them when no fp32 tensor is
//       written.  (The loader wave can take a share of the fp32 store instructions: Scan3jRole::lsplit.)
// Arithmetic: sfsn_spike_proj + sfsn_gsn_layer_scan value for value (exact integer products, fma(exact sum, dq_ih, b_f), then
// fma(exact sum, dq_hh, that), the cell's fma / exp2 / rcp sequence): bit-identical to gsn_scan_fused_kernel (tests).
// Shared gate weights, 128 < H <= 224, 16 rows per workgroup, no links (a per-layer launch).
#ifndef SFSN_SCAN3J_DEV_H
#define SFSN_SCAN3J_DEV_H
#include "sfsn_scan3_dev.h"

#ifndef SFSN_S3J_LSPLIT
// fp32 store instructions per frame issued by the loader wave (of 14 at H = 224; the storer also carries the four int8 stores); the
// environment variable of the same name overrides it.  Measured (B = 64, T = 1000, 52 workgroups): 0 / 3 / 5 / 7 -> 1.51 / 1.40 / 1.36 / 1.36 ms
// per launch (round 2's body: 1.56); without fp32 spikes 1.35 (1.53).
#define SFSN_S3J_LSPLIT 6
#endif

template <int KS>
struct Scan3jCfg {
    static constexpr int RPW = 16, HP = KS * 64, LDH = HP + 32, NCH = HP / 16;
    static constexpr int NP = 4, SLOT = NP * 1024;     // a frame of 16 rows x 16 chunk positions
    // frame t + A is requested during step t into the slot of frame t - 1 (D = A + 1).  Not frame t's own slot: in the steady state it is
    // dead by then (read during step t - 1), but frame 0 is read AFTER the prologue's barrier, beside the loader's step 0 (found as a
    // timing-dependent mismatch in a module test, not in the kernel test)
    static constexpr int A = 6, D = 7;
    static constexpr int HBUF_OFF = D * SLOT;
    static constexpr int CST_OFF = HBUF_OFF + 2 * 16 * LDH;   // [6][HP] floats: dq_hh, b_g - b_f, alpha, beta, dq_ih, b_f
    static constexpr int WIH_OFF = CST_OFF + 6 * HP * 4;
    __host__ __device__ static constexpr int plane_bytes(int NT) { return NT * KS * 1024; }
    __host__ __device__ static constexpr int lds_bytes(int NT) { return WIH_OFF + 2 * plane_bytes(NT); }
    // OFF form (14 tiles): the input terms the two IO waves compute for tiles 12 and 13, [frame parity][tile - 12][lane] x 16 bytes
    __host__ __device__ static constexpr int mbox_off(int NT) { return lds_bytes(NT); }
    __host__ __device__ static constexpr int lds_bytes_off(int NT) { return lds_bytes(NT) + 4096; }
};

// The input term of frame f (ring slot f % D) for output tile `ct`, as the wave's four values per lane: 12 matrix instructions (full
// 16x16x64 steps, zero padded k), planes 0 / 1 of W_ih from LDS two k-steps at a time, plane 2 from the caller's registers;
// z = fma(exact sum, dq_ih, b_f) (= sfsn_spike_proj).  Used by the compute waves for their own tile and, in the OFF form, by the IO waves
// for tiles 12 / 13.
template <int KS>
__device__ __forceinline__ v4f s3j_in_product(const char* smem, int f, const unsigned (&soff)[KS], unsigned woff, int PLANE, const v4i (&Wi2)[KS],
                                              const char* cq) {
    using C = Scan3jCfg<KS>;
    constexpr int HP = C::HP;
    const char* ring = smem + (f % C::D) * C::SLOT;
    v4i e[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
#pragma unroll
    for (int k0 = 0; k0 < KS; k0 += 2) {
        v4i sb[2], w0[2], w1[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (k0 + i >= KS) continue;
            sb[i] = *reinterpret_cast<const v4i*>(ring + soff[k0 + i]);
            w0[i] = *reinterpret_cast<const v4i*>(smem + woff + (k0 + i) * 1024);
            w1[i] = *reinterpret_cast<const v4i*>(smem + woff + PLANE + (k0 + i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (k0 + i >= KS) continue;
            e[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0[i], sb[i], e[0], 0, 0, 0);
            e[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wi2[k0 + i], sb[i], e[2], 0, 0, 0);
            e[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1[i], sb[i], e[1], 0, 0, 0);
        }
    }
    const v4f dqi = *reinterpret_cast<const v4f*>(cq + 4 * HP * 4);
    const v4f bf = *reinterpret_cast<const v4f*>(cq + 5 * HP * 4);
    v4f z;
#pragma unroll
    for (int r = 0; r < 4; ++r) z[r] = __builtin_fmaf((float)((e[2][r] << 16) + (e[1][r] << 8) + e[0][r]), dqi[r], bf[r]);
    return z;
}

struct Scan3jRole;
// What an IO wave of the OFF form needs to compute tile `ct`'s input terms: plane 2 of its W_ih rows in registers, its fragment offsets
template <int KS>
struct S3jHelper {
    v4i Wi2[KS];
    unsigned soff[KS], woff;
    const char* cq;
    char* mbox;  // my lane's 16 bytes of parity 0; parity 1 at + 2048
    __device__ __forceinline__ void init(const Scan3jRole& rl, char* smem, int NT, int ct, int lane);
    __device__ __forceinline__ void run(const char* smem, int f, int PLANE) const {
        const v4f z = s3j_in_product<KS>(smem, f, soff, woff, PLANE, Wi2, cq);
        *reinterpret_cast<v4f*>(mbox + (f & 1) * 2048) = z;
    }
};

struct Scan3jRole {
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
    unsigned long long* count;  // nullable: a launch without fp32 spikes adds the number of spikes it wrote
    int lsplit;                 // fp32 store instructions per frame issued by the loader wave (the storer takes the rest)
};

template <int KS>
__device__ __forceinline__ void S3jHelper<KS>::init(const Scan3jRole& rl, char* smem, int NT, int ct, int lane) {
    using C = Scan3jCfg<KS>;
    const int n = lane & 15, q = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        Wi2[ks] = *reinterpret_cast<const v4i*>(rl.w_ih + ((((size_t)2 * NT + ct) * KS + ks) * 64 + lane) * 16);
        soff[ks] = (unsigned)(n * 256 + ((ks * 4 + q + n) & 15) * 16);
    }
    woff = (unsigned)(C::WIH_OFF + (ct * KS) * 1024 + lane * 16);
    cq = smem + C::CST_OFF + (ct * 16 + q * 4) * 4;
    mbox = smem + C::mbox_off(NT) + (ct - 12) * 1024 + lane * 16;
}

// TL = 1: H mod 64 in (0, 32]: the last k-step of the RECURRENT product is one 16x16x32 instruction (scan3i_role's form).
// OUT bit 0: fp32 spikes, bit 1: int8 spikes (always).
// OFF = 1 (14 tiles only): with one tile per compute wave SIMDs 0 and 1 carry four tiles and SIMDs 2 and 3 three tiles and an IO wave,
// and a SIMD's matrix instructions and VALU instructions share its time (file head) -- the step is the four-tile SIMDs'.  The input
// term of the next frame is off the step's dependency chain, so the IO waves (SIMD 2: loader, SIMD 3: storer) compute it for tiles 12
// (SIMD 0) and 13 (SIMD 1) and hand the four values per lane over through LDS at the step barrier (double-buffered by frame parity):
// ~3.6 tiles' worth of work on every SIMD instead of 4 / 4 / 3 / 3.  The same instructions on the same operands: bit-identical.
template <int KS, int TL, int OUT, int OFF = 0>
__device__ __forceinline__ void scan3j_role(const Scan3jRole& rl, char* smem, int T, int H, int NT) {
    using C = Scan3jCfg<KS>;
    constexpr int RPW = 16, LDH = C::LDH, HP = C::HP, D = C::D, A = C::A, SLOT = C::SLOT, NP = C::NP, NCH = C::NCH;
    constexpr int KSF = TL ? KS - 1 : KS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int R = rl.R, row0 = rl.row0;
    const int PLANE = C::plane_bytes(NT);
    int8_t* hbuf = reinterpret_cast<int8_t*>(smem + C::HBUF_OFF);
    float(*cst)[HP] = reinterpret_cast<float(*)[HP]>(smem + C::CST_OFF);

    // ---- set-up by all threads
    for (int i = tid; i < 2 * 16 * LDH / 4; i += 1024) reinterpret_cast<int*>(hbuf)[i] = 0;
    for (int i = tid; i < 2 * PLANE / 16; i += 1024) {
        const int d = i / (PLANE / 16), r = i - d * (PLANE / 16);
        reinterpret_cast<v4i*>(smem + C::WIH_OFF)[i] = *reinterpret_cast<const v4i*>(rl.w_ih + (size_t)d * PLANE + (size_t)r * 16);
    }
    for (int j = tid; j < HP; j += 1024) {
        const bool in = j < H;
        cst[0][j] = in ? rl.w_dq[j] : 0.0f;
        cst[1][j] = in ? rl.bias[H + j] - rl.bias[j] : 0.0f;
        cst[2][j] = in ? rl.bn_alpha[j] : 0.0f;
        cst[3][j] = in ? rl.bn_beta[j] : 0.0f;
        cst[4][j] = in ? rl.w_ih_dq[j] : 0.0f;
        cst[5][j] = in ? rl.bias[j] : 0.0f;
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
        const int cj = ct * 16 + q * 4;  // my four neurons (row n)
        const bool live = row0 + n < R;
        const int grow = live ? row0 + n : R - 1;
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
        v4f c = *reinterpret_cast<const v4f*>(rl.c_state + (size_t)grow * H + cj);
        const unsigned boff = (unsigned)(n * LDH + q * 16);
        const unsigned boft = (unsigned)(n * LDH + (KS - 1) * 64 + q * 8);
        const unsigned hoff = (unsigned)(n * LDH + cj);
        unsigned soff[KS];  // my B fragments of the input product: row n, k chunk c = 4 ks + q at position (c + n) & 15
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) soff[ks] = (unsigned)(n * 256 + ((ks * 4 + q + n) & 15) * 16);
        const unsigned woff = (unsigned)(C::WIH_OFF + (ct * KS) * 1024 + lane * 16);
        const char* cq = smem + C::CST_OFF + cj * 4;  // my four neurons' constants: vector k at cq + k * HP * 4
        v4f z = {0.f, 0.f, 0.f, 0.f};                 // the input term of my four values at the NEXT frame to be finished

        const bool served = OFF && ct >= 12;  // (wave-uniform) my input terms come from an IO wave's mailbox
        const char* mbox = smem + C::mbox_off(NT) + (ct >= 12 ? ct - 12 : 0) * 1024 + lane * 16;
        auto in_product = [&](int f) __attribute__((always_inline)) { z = s3j_in_product<KS>(smem, f, soff, woff, PLANE, Wi2, cq); };

        __syncthreads();                       // initial state in hbuf[0], W_ih planes and constants in LDS
        __builtin_amdgcn_s_barrier();          // the loader's prologue frames (0 .. A - 1) have landed
        if (!served) in_product(0);
        if constexpr (OFF) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();      // the IO waves' input terms of frame 0 are in the mailbox
        }
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            if constexpr (OFF) if (served) z = *reinterpret_cast<const v4f*>(mbox + (t & 1) * 2048);
            const int8_t* hc = hbuf + (t & 1) * 16 * LDH;
            int8_t* hn = hbuf + ((t & 1) ^ 1) * 16 * LDH;
            v4i b[KSF > 0 ? KSF : 1];
#pragma unroll
            for (int ks = 0; ks < KSF; ++ks) b[ks] = *reinterpret_cast<const v4i*>(hc + boff + ks * 64);
            long bt = 0;
            if constexpr (TL) bt = *reinterpret_cast<const long*>(hc + boft);
            // the cell's per-neuron constants arrive under the matrix instructions
            const v4f dq = *reinterpret_cast<const v4f*>(cq);
            const v4f db = *reinterpret_cast<const v4f*>(cq + 1 * HP * 4);
            const v4f al = *reinterpret_cast<const v4f*>(cq + 2 * HP * 4);
            const v4f be = *reinterpret_cast<const v4f*>(cq + 3 * HP * 4);
            v4i a[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
            if constexpr (TL) {
                // the 32-wide tail step FIRST, from zero accumulators, with the wait states a 16x16x64 step needs before it may accumulate
                // onto a 16x16x32 step's result (see scan3i_role)
                asm volatile(
                    "v_mfma_i32_16x16x32_i8 %0, %3, %6, 0\n\t"
                    "v_mfma_i32_16x16x32_i8 %1, %4, %6, 0\n\t"
                    "v_mfma_i32_16x16x32_i8 %2, %5, %6, 0\n\t"
                    "s_nop 5"
                    : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2])
                    : "v"(Wht[0]), "v"(Wht[1]), "v"(Wht[2]), "v"(bt));
                if constexpr (KSF == 0) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            }
#pragma unroll
            for (int ks = 0; ks < KSF; ++ks)
#pragma unroll
                for (int d = 0; d < 3; ++d) a[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Whh[ks][d], b[ks], a[d], 0, 0, 0);
            unsigned pk = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rec = (float)((a[2][r] << 16) + (a[1][r] << 8) + a[0][r]);  // exact sum, rounded once (= recombine3)
                const float pre_f = __builtin_fmaf(rec, dq[r], z[r]);
                const float pre_g = pre_f + db[r];
                const float f = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre_f * -1.44269504088896341f));
                const float m = __builtin_fmaf(f, c[r] - pre_g, pre_g);
                const float y = __builtin_fmaf(m, al[r], be[r]);
                c[r] = y;
                pk |= (y >= 0.0f) ? (1u << (8 * r)) : 0u;
            }
            *reinterpret_cast<unsigned*>(hn + hoff) = pk;
            __builtin_amdgcn_sched_barrier(0);
            // off the chain: the input term of frame t + 1 (the loader clamps frames past the end to the last one: harmless work)
            if (!served) in_product(t + 1);
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
        }
        // final state
        const int8_t* hl = hbuf + (T & 1) * 16 * LDH;
        if (live) {
            *reinterpret_cast<v4f*>(rl.c_state + (size_t)grow * H + cj) = c;
            const unsigned pk = *reinterpret_cast<const unsigned*>(hl + hoff);
            const v4f h = {(float)(pk & 1u), (float)((pk >> 8) & 1u), (float)((pk >> 16) & 1u), (float)((pk >> 24) & 1u)};
            *reinterpret_cast<v4f*>(rl.h_state + (size_t)grow * H + cj) = h;
        }
        return;
    }

    if (wave == NT) {
        // ================================================= loader wave =================================================
        // piece p, lane: slot unit e = 64 p + lane = (row e >> 4, position e & 15) <- global chunk (position - row) mod 16
        unsigned goff[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int e = 64 * p + lane, r = e >> 4, pos = e & 15;
            int cch = (pos - r) & 15;
            if (cch >= NCH) cch = 0;  // padding position (HP < 256): any valid chunk, never read
            const int grow = (row0 + r < R) ? row0 + r : R - 1;
            goff[p] = (unsigned)(grow * HP + cch * 16);
        }
        const size_t frame = (size_t)R * HP;
        constexpr bool LSF = (OUT & 1) != 0;
        S3FlushF<RPW, LDH> ff;
        if constexpr (LSF) ff.init(lane, row0, R, H, 0, rl.lsplit);
        // frames t + 3 .. t + A may stay in flight behind the wait of step t (plus my stores of the steps in between)
        int allow = (A - 2) * NP;
        if constexpr (LSF) allow = (A - 2) * (NP + ff.nsf) + ff.nsf;
        if (allow > 62) allow = 62;
        auto issue = [&](int slot, int td) __attribute__((always_inline)) {
            const float* st = reinterpret_cast<const float*>(rl.spikes_in + (size_t)td * frame);
#pragma unroll
            for (int p = 0; p < NP; ++p) dma16_to_lds<false>(__builtin_amdgcn_readfirstlane((unsigned)(slot * SLOT + p * 1024)), st, goff[p]);
        };
        S3jHelper<KS> hp;
        if constexpr (OFF) hp.init(rl, smem, NT, 12, lane);
        __syncthreads();
        for (int s0 = 0; s0 < A; ++s0) issue(s0, s0 < T ? s0 : (T > 0 ? T - 1 : 0));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        if constexpr (OFF) {
            hp.run(smem, 0, PLANE);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            const int td = (t + A < T) ? t + A : T - 1;
            issue((t + A) % D, td);  // the slot of frame t - 1: read during step t - 2
            if constexpr (LSF) if (t > 0 && ff.nsf > 0) ff.run(hbuf + (t & 1) * 16 * LDH, rl.spikes_f32 + ((size_t)(t - 1) * R + row0) * H, lane);
            if constexpr (OFF) hp.run(smem, t + 1, PLANE);  // tile 12's input term of frame t + 1 (landed before the barrier of step t - 1)
            wait_vmcnt_n(allow);     // frame t + 2 has landed: the compute waves read it during step t + 1
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // DMAs past the end are invisible to the compiler
        if constexpr (LSF) if (T > 0 && ff.nsf > 0) ff.run(hbuf + (T & 1) * 16 * LDH, rl.spikes_f32 + ((size_t)(T - 1) * R + row0) * H, lane);
        return;
    }

    if (wave == NT + 1) {
        // ================================================= storer wave =================================================
        constexpr int MAX8 = (RPW * KS * 4 + 63) / 64;
        constexpr int nu8 = RPW * (HP / 16), ns8 = (nu8 + 63) / 64;
        constexpr bool F32 = (OUT & 1) != 0;
        S3FlushF<RPW, LDH> ff;
        if constexpr (F32) ff.init(lane, row0, R, H, rl.lsplit);
        int l8[MAX8];
        unsigned ok8 = 0;
        unsigned cnt = 0;
#pragma unroll
        for (int k = 0; k < MAX8; ++k) {
            const int u = 64 * k + lane, rr = u / (HP / 16), c16 = u - rr * (HP / 16);
            l8[k] = rr * LDH + c16 * 16;
            if (k < ns8 && u < nu8 && row0 + rr < R) ok8 |= 1u << k;
        }
        auto flush = [&](const int8_t* hsrc, int ts) __attribute__((always_inline)) {
            int8_t* p8 = rl.spikes_i8 + ((size_t)ts * R + row0) * HP;
#pragma unroll
            for (int k = 0; k < MAX8; ++k) {
                if ((ok8 >> k) & 1u) {
                    const v4i d = *reinterpret_cast<const v4i*>(hsrc + l8[k]);
                    *reinterpret_cast<v4i*>(p8 + (size_t)(64 * k + lane) * 16) = d;
                    if constexpr (!(OUT & 1)) cnt += popc16(d);  // (live rows only; the pad columns of the state buffer hold zeros)
                }
            }
            if constexpr (F32) ff.run(hsrc, rl.spikes_f32 + ((size_t)ts * R + row0) * H, lane);
        };
        S3jHelper<KS> hp;
        if constexpr (OFF) hp.init(rl, smem, NT, 13, lane);
        __syncthreads();
        __builtin_amdgcn_s_barrier();
        if constexpr (OFF) {
            hp.run(smem, 0, PLANE);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            if (t > 0) flush(hbuf + (t & 1) * 16 * LDH, t - 1);  // = h_{t-1}
            if constexpr (OFF) hp.run(smem, t + 1, PLANE);       // tile 13's input term of frame t + 1
            __builtin_amdgcn_s_waitcnt(0xc07f);                  // my LDS reads are done before the buffer is rewritten (step t + 1)
            __builtin_amdgcn_s_barrier();
        }
        if (T > 0) flush(hbuf + (T & 1) * 16 * LDH, T - 1);
        if constexpr (!(OUT & 1)) wave_count_add(rl.count, cnt);
        return;
    }

    // ================================================= spare waves (NT < 14): keep the barrier count =================================================
    __syncthreads();
    __builtin_amdgcn_s_barrier();
    if constexpr (OFF) __builtin_amdgcn_s_barrier();
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    }
}

// =====================================================================================================================
// scan3y_role -- the layer-0 twin at 16 rows: the real-valued input product x . W_ih^T + b of a group with narrow feature rows (even
// I <= 64, rows a multiple of 16) inside the 16-row IO-wave scan, on the bf16 matrix cores with input_proj_bf3_kernel's 3-way split
// (the six products per 32 k of fusedx_body / scan3x_role in their order into the same two accumulators, the same (hi + lo) + b):
// what sfsn_gsn_layer_scan_fused_x launches for H <= 224 since round 6 (gsn_scan_fusedx3_kernel), bit-identical to round 2's body.
//   compute waves: as scan3j_role; W_ih piece 1 in registers, pieces 2 / 3 in LDS as A fragments (written by the wave at set-up); the
//       product of frame t + 1 behind the epilogue of frame t, finished to (hi + lo) + b_f in four registers.
//   loader wave: a frame's 16 feature rows are one contiguous block (16 x I floats <= 4 KiB): LDS-DMA into a ring and, three frames
//       ahead of their use, split by the same wave into three bf16 planes (row stride 72 elements).
//   storer wave / spare waves: scan3j_role's.
// =====================================================================================================================
#ifndef SFSN_S3Y_LSPLIT
// fp32 store instructions per frame issued by the loader wave (it also converts the features).  Measured (B = 64, T = 1000, 32 workgroups,
// ms per launch): 0 / 1-3 / 4 / 7 -> 1.49 / 1.47 / 1.52 / 1.66 (round 2's body 1.64); without fp32 spikes 1.37 (1.64)
#define SFSN_S3Y_LSPLIT 2
#endif

template <int KS, int KSB>
struct Scan3yCfg {
    static constexpr int RPW = 16, HP = KS * 64, LDH = HP + 32;
    static constexpr int NPX = 4, XSLOT = NPX * 1024;      // a frame's features: 16 rows x I floats <= 4 KiB
    static constexpr int A = 6, DX = 7;                    // frame t + A is requested during step t into the slot of frame t - 1 (converted at step t - 4)
    static constexpr int LDX = 72, PLANE = 16 * LDX * 2;   // one bf16 plane of a frame's 16 rows
    static constexpr int DP = 5;                           // plane slots: step t converts frame t + 3 into the slot of frame t - 2 (read during step t - 3)
    static constexpr int PL_OFF = DX * XSLOT;
    static constexpr int HBUF_OFF = PL_OFF + DP * 3 * PLANE;
    static constexpr int CST_OFF = HBUF_OFF + 2 * 16 * LDH;  // [5][HP] floats: dq_hh, b_g - b_f, alpha, beta, b_f
    static constexpr int WX_OFF = CST_OFF + 5 * HP * 4;
    __host__ __device__ static constexpr int wx_bytes(int NT) { return 2 * NT * KSB * 1024; }  // pieces 2, 3 x tiles x k-chunks x 1 KiB
    __host__ __device__ static constexpr int lds_bytes(int NT) { return WX_OFF + wx_bytes(NT); }
};

struct Scan3yRole {
    const float* x;       // [T][R][I] the layer input (sfsn_features' output), R a multiple of 16
    const float* w_ih;    // [H][I] fp32, row-major
    int I;
    const int8_t* w_hh;
    const float* w_dq;
    const float* bias;    // [2 H]: b_f, b_g
    const float* bn_alpha;
    const float* bn_beta;
    float* h_state;
    float* c_state;
    float* spikes_f32;
    int8_t* spikes_i8;
    int R, row0;
    unsigned long long* count;
    int lsplit;
};

template <int KS, int TL, int OUT, int KSB>
__device__ __forceinline__ void scan3y_role(const Scan3yRole& rl, char* smem, int T, int H, int NT) {
    using C = Scan3yCfg<KS, KSB>;
    constexpr int RPW = 16, LDH = C::LDH, HP = C::HP, A = C::A, DX = C::DX, XSLOT = C::XSLOT, NPX = C::NPX, LDX = C::LDX, PLANE = C::PLANE, DP = C::DP;
    constexpr int KSF = TL ? KS - 1 : KS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int R = rl.R, row0 = rl.row0, I = rl.I;
    int8_t* hbuf = reinterpret_cast<int8_t*>(smem + C::HBUF_OFF);
    float(*cst)[HP] = reinterpret_cast<float(*)[HP]>(smem + C::CST_OFF);

    // ---- set-up by all threads: state buffers and bf16 planes zeroed (k >= I must read as 0), constants, h_{-1} -> hbuf[0]
    for (int i = tid; i < 2 * 16 * LDH / 4; i += 1024) reinterpret_cast<int*>(hbuf)[i] = 0;
    for (int i = tid; i < DP * 3 * PLANE / 4; i += 1024) reinterpret_cast<int*>(smem + C::PL_OFF)[i] = 0;
    for (int j = tid; j < HP; j += 1024) {
        const bool in = j < H;
        cst[0][j] = in ? rl.w_dq[j] : 0.0f;
        cst[1][j] = in ? rl.bias[H + j] - rl.bias[j] : 0.0f;
        cst[2][j] = in ? rl.bn_alpha[j] : 0.0f;
        cst[3][j] = in ? rl.bn_beta[j] : 0.0f;
        cst[4][j] = in ? rl.bias[j] : 0.0f;
    }
    __syncthreads();
    for (int idx = tid; idx < RPW * (H / 4); idx += 1024) {
        const int rr = idx / (H / 4), j4 = (idx - rr * (H / 4)) * 4;
        const int rsrc = row0 + rr < R ? row0 + rr : R - 1;
        const v4f h = *reinterpret_cast<const v4f*>(rl.h_state + (size_t)rsrc * H + j4);
        const unsigned pk = (h.x > 0.5f ? 1u : 0u) | (h.y > 0.5f ? 0x100u : 0u) | (h.z > 0.5f ? 0x10000u : 0u) |
                            (h.w > 0.5f ? 0x1000000u : 0u);
        *reinterpret_cast<unsigned*>(hbuf + rr * LDH + j4) = pk;
    }

    if (wave < NT) {
        // ================================================= compute wave: output tile `wave` =================================================
        const int ct = wave;
        const int cj = ct * 16 + q * 4;
        const bool live = row0 + n < R;
        const int grow = live ? row0 + n : R - 1;
        const unsigned toff = (unsigned)((((q >> 1) * 16 + n) * 16) + (q & 1) * 8);
        v4i Whh[KSF > 0 ? KSF : 1][3];
        long Wht[3] = {0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KSF; ++ks)
#pragma unroll
            for (int d = 0; d < 3; ++d)
                Whh[ks][d] = *reinterpret_cast<const v4i*>(rl.w_hh + ((((size_t)d * NT + ct) * KS + ks) * 64 + lane) * 16);
        if constexpr (TL) {
#pragma unroll
            for (int d = 0; d < 3; ++d) Wht[d] = *reinterpret_cast<const long*>(rl.w_hh + (((size_t)d * NT + ct) * KS + KS - 1) * 1024 + toff);
        }
        // W_ih: A fragment = 8 consecutive k of weight row ct * 16 + n (as input_proj_bf3_kernel); piece 1 in registers, 2 / 3 -> LDS
        bf8 Wx1[KSB];
        const unsigned wxoff = (unsigned)(C::WX_OFF + (ct * KSB) * 1024 + lane * 16);
        const int wxplane = NT * KSB * 1024;
        {
            const int wr = ct * 16 + n;
#pragma unroll
            for (int ks = 0; ks < KSB; ++ks) {
                unsigned pw[3][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = ks * 32 + q * 8 + 2 * e;
                    const float a = (wr < H && k < I) ? rl.w_ih[(size_t)wr * I + k] : 0.0f;
                    const float b = (wr < H && k + 1 < I) ? rl.w_ih[(size_t)wr * I + k + 1] : 0.0f;
                    split3(a, b, pw[0][e], pw[1][e], pw[2][e]);
                }
                Wx1[ks] = *reinterpret_cast<const bf8*>(pw[0]);
                *reinterpret_cast<v4i*>(smem + wxoff + ks * 1024) = *reinterpret_cast<const v4i*>(pw[1]);
                *reinterpret_cast<v4i*>(smem + wxoff + wxplane + ks * 1024) = *reinterpret_cast<const v4i*>(pw[2]);
            }
        }
        v4f c = *reinterpret_cast<const v4f*>(rl.c_state + (size_t)grow * H + cj);
        const unsigned boff = (unsigned)(n * LDH + q * 16);
        const unsigned boft = (unsigned)(n * LDH + (KS - 1) * 64 + q * 8);
        const unsigned hoff = (unsigned)(n * LDH + cj);
        const unsigned xoff = (unsigned)(C::PL_OFF + (n * LDX + q * 8) * 2);  // my B fragment: row n, k = 32 ks + 8 q .. + 7 of a plane
        const char* cq = smem + C::CST_OFF + cj * 4;
        v4f z = {0.f, 0.f, 0.f, 0.f};
        // the input term of frame f (plane slot f % DP) -> z: the product sequence of input_proj_bf3_kernel, instruction for instruction
        auto in_product = [&](int f) __attribute__((always_inline)) {
            const char* pl = smem + xoff + (f % DP) * 3 * PLANE;
            v4f hi = {0.f, 0.f, 0.f, 0.f}, lo = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSB; ++ks) {
                const bf8 x1 = *reinterpret_cast<const bf8*>(pl + ks * 64);
                const bf8 x2 = *reinterpret_cast<const bf8*>(pl + PLANE + ks * 64);
                const bf8 x3 = *reinterpret_cast<const bf8*>(pl + 2 * PLANE + ks * 64);
                const bf8 w2 = *reinterpret_cast<const bf8*>(smem + wxoff + ks * 1024);
                const bf8 w3 = *reinterpret_cast<const bf8*>(smem + wxoff + wxplane + ks * 1024);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3, x1, lo, 0, 0, 0);
                hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wx1[ks], x1, hi, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, x2, lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wx1[ks], x3, lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, x1, lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wx1[ks], x2, lo, 0, 0, 0);
            }
            const v4f bf = *reinterpret_cast<const v4f*>(cq + 4 * HP * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = (hi[r] + lo[r]) + bf[r];  // = input_proj_bf3_kernel's epilogue
        };

        __syncthreads();                       // initial state, W_ih pieces 2 / 3 and constants in LDS
        __builtin_amdgcn_s_barrier();          // the loader's planes of frames 0 .. 2
        in_product(0);
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            const int8_t* hc = hbuf + (t & 1) * 16 * LDH;
            int8_t* hn = hbuf + ((t & 1) ^ 1) * 16 * LDH;
            v4i b[KSF > 0 ? KSF : 1];
#pragma unroll
            for (int ks = 0; ks < KSF; ++ks) b[ks] = *reinterpret_cast<const v4i*>(hc + boff + ks * 64);
            long bt = 0;
            if constexpr (TL) bt = *reinterpret_cast<const long*>(hc + boft);
            const v4f dq = *reinterpret_cast<const v4f*>(cq);
            const v4f db = *reinterpret_cast<const v4f*>(cq + 1 * HP * 4);
            const v4f al = *reinterpret_cast<const v4f*>(cq + 2 * HP * 4);
            const v4f be = *reinterpret_cast<const v4f*>(cq + 3 * HP * 4);
            v4i a[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
            if constexpr (TL) {  // (the 32-wide tail step first, with its wait states: see scan3i_role)
                asm volatile(
                    "v_mfma_i32_16x16x32_i8 %0, %3, %6, 0\n\t"
                    "v_mfma_i32_16x16x32_i8 %1, %4, %6, 0\n\t"
                    "v_mfma_i32_16x16x32_i8 %2, %5, %6, 0\n\t"
                    "s_nop 5"
                    : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2])
                    : "v"(Wht[0]), "v"(Wht[1]), "v"(Wht[2]), "v"(bt));
                if constexpr (KSF == 0) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            }
#pragma unroll
            for (int ks = 0; ks < KSF; ++ks)
#pragma unroll
                for (int d = 0; d < 3; ++d) a[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Whh[ks][d], b[ks], a[d], 0, 0, 0);
            unsigned pk = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rec = (float)((a[2][r] << 16) + (a[1][r] << 8) + a[0][r]);
                const float pre_f = __builtin_fmaf(rec, dq[r], z[r]);
                const float pre_g = pre_f + db[r];
                const float f = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre_f * -1.44269504088896341f));
                const float m = __builtin_fmaf(f, c[r] - pre_g, pre_g);
                const float y = __builtin_fmaf(m, al[r], be[r]);
                c[r] = y;
                pk |= (y >= 0.0f) ? (1u << (8 * r)) : 0u;
            }
            *reinterpret_cast<unsigned*>(hn + hoff) = pk;
            __builtin_amdgcn_sched_barrier(0);
            in_product(t + 1);                   // off the chain (frames past the end: the loader converts clamped copies of the last one)
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
        const int8_t* hl = hbuf + (T & 1) * 16 * LDH;
        if (live) {
            *reinterpret_cast<v4f*>(rl.c_state + (size_t)grow * H + cj) = c;
            const unsigned pk = *reinterpret_cast<const unsigned*>(hl + hoff);
            const v4f h = {(float)(pk & 1u), (float)((pk >> 8) & 1u), (float)((pk >> 16) & 1u), (float)((pk >> 24) & 1u)};
            *reinterpret_cast<v4f*>(rl.h_state + (size_t)grow * H + cj) = h;
        }
        return;
    }

    if (wave == NT) {
        // ================================================= loader wave: features -> ring -> bf16 planes =================================================
        const int nchunk = 4 * I;  // 16 * I floats / 4
        unsigned goff[NPX];
        int po[NPX][4];            // my four floats of piece p: element offset within a bf16 plane (clamped lanes repeat the last chunk)
#pragma unroll
        for (int p = 0; p < NPX; ++p) {
            int e = 64 * p + lane;
            if (e > nchunk - 1) e = nchunk - 1;
            goff[p] = (unsigned)e * 16u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = 4 * e + j, rr = f / I, k = f - rr * I;
                po[p][j] = rr * LDX + k;
            }
        }
        const size_t frame = (size_t)R * I;
        const float* xbase = rl.x + (size_t)row0 * I;
        S3FlushF<RPW, LDH> ffl;
        if constexpr (OUT & 1) ffl.init(lane, row0, R, H, 0, rl.lsplit);
        const int lst = (OUT & 1) ? ffl.nsf : 0;
        int allow = (A - 3) * (NPX + lst) + lst;  // behind the DMA of frame t + 3: the DMAs and stores of the steps since, and this step's stores
        if (allow > 62) allow = 62;
        auto issue = [&](int slot, int td) __attribute__((always_inline)) {
#pragma unroll
            for (int p = 0; p < NPX; ++p)
                dma16_to_lds(__builtin_amdgcn_readfirstlane((unsigned)(slot * XSLOT + p * 1024)), xbase + (size_t)td * frame, goff[p]);
        };
        auto convert = [&](int fr) __attribute__((always_inline)) {
            const char* src = smem + (fr % DX) * XSLOT;
            unsigned short* pl = reinterpret_cast<unsigned short*>(smem + C::PL_OFF + (fr % DP) * 3 * PLANE);
#pragma unroll
            for (int p = 0; p < NPX; ++p) {
                const v4f v = *reinterpret_cast<const v4f*>(src + p * 1024 + lane * 16);  // (a clamped lane's DMA landed at its OWN lds position)
                unsigned p1[2], p2[2], p3[2];
                split3(v[0], v[1], p1[0], p2[0], p3[0]);
                split3(v[2], v[3], p1[1], p2[1], p3[1]);
                // (I is even and a chunk starts at an even float: the two values of a pair sit side by side in one row -> one 32-bit store
                //  per pair and plane instead of two 16-bit ones)
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2) {
                    *reinterpret_cast<unsigned*>(pl + po[p][2 * j2]) = p1[j2];
                    *reinterpret_cast<unsigned*>(pl + PLANE / 2 + po[p][2 * j2]) = p2[j2];
                    *reinterpret_cast<unsigned*>(pl + PLANE + po[p][2 * j2]) = p3[j2];
                }
            }
        };
        __syncthreads();
        if (T > 0)
            for (int s0 = 0; s0 < A; ++s0) issue(s0, s0 < T ? s0 : T - 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int fr = 0; fr < 3; ++fr) convert(fr);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            issue((t + A) % DX, (t + A < T) ? t + A : T - 1);
            if constexpr (OUT & 1) if (t > 0 && lst > 0) ffl.run(hbuf + (t & 1) * 16 * LDH, rl.spikes_f32 + ((size_t)(t - 1) * R + row0) * H, lane);
            wait_vmcnt_n(allow);    // frame t + 3 has landed
            convert(t + 3);         // its planes are read during step t + 2; the slot held frame t - 2 (read during step t - 3)
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (OUT & 1) if (T > 0 && lst > 0) ffl.run(hbuf + (T & 1) * 16 * LDH, rl.spikes_f32 + ((size_t)(T - 1) * R + row0) * H, lane);
        return;
    }

    if (wave == NT + 1) {
        // ================================================= storer wave (scan3j_role's) =================================================
        constexpr int MAX8 = (RPW * KS * 4 + 63) / 64;
        constexpr int nu8 = RPW * (HP / 16), ns8 = (nu8 + 63) / 64;
        constexpr bool F32 = (OUT & 1) != 0;
        S3FlushF<RPW, LDH> ff;
        if constexpr (F32) ff.init(lane, row0, R, H, rl.lsplit);
        int l8[MAX8];
        unsigned ok8 = 0;
        unsigned cnt = 0;
#pragma unroll
        for (int k = 0; k < MAX8; ++k) {
            const int u = 64 * k + lane, rr = u / (HP / 16), c16 = u - rr * (HP / 16);
            l8[k] = rr * LDH + c16 * 16;
            if (k < ns8 && u < nu8 && row0 + rr < R) ok8 |= 1u << k;
        }
        auto flush = [&](const int8_t* hsrc, int ts) __attribute__((always_inline)) {
            int8_t* p8 = rl.spikes_i8 + ((size_t)ts * R + row0) * HP;
#pragma unroll
            for (int k = 0; k < MAX8; ++k) {
                if ((ok8 >> k) & 1u) {
                    const v4i d = *reinterpret_cast<const v4i*>(hsrc + l8[k]);
                    *reinterpret_cast<v4i*>(p8 + (size_t)(64 * k + lane) * 16) = d;
                    if constexpr (!(OUT & 1)) cnt += popc16(d);
                }
            }
            if constexpr (F32) ff.run(hsrc, rl.spikes_f32 + ((size_t)ts * R + row0) * H, lane);
        };
        __syncthreads();
        __builtin_amdgcn_s_barrier();
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            if (t > 0) flush(hbuf + (t & 1) * 16 * LDH, t - 1);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
        if (T > 0) flush(hbuf + (T & 1) * 16 * LDH, T - 1);
        if constexpr (!(OUT & 1)) wave_count_add(rl.count, cnt);
        return;
    }

    // ================================================= spare waves =================================================
    __syncthreads();
    __builtin_amdgcn_s_barrier();
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    }
}

#endif
