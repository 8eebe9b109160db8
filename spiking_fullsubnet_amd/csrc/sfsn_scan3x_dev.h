// sfsn_scan3x_dev.h -- the IO-wave scan of a LAYER 0 that forms its real-valued input product itself (round 4), gfx950 only.
//
// The layer-0 twin of the FUSED3 role (sfsn_scan3i_dev.h): x . W_ih^T + b (efficient_spiking_neuron.py:141-142) for a group with
// narrow feature rows (even I <= 64: baseline_s / m / l group 0, 62 % of the sub-band rows) runs inside the 8-row scan, on the bf16
// matrix cores with the 3-way split of input_proj_bf3_kernel -- the same six products per 32 k in the same order into the same two
// accumulators, the same final (hi + lo) + b, hence bit-identical to sfsn_input_proj_f32 + sfsn_gsn_layer_scan -- and batched over
// TWO frames in the MFMA columns 8 rows leave idle (columns 0..7 = the rows at frame f, 8..15 = the same rows at frame f + 1): six
// matrix instructions per tile and step.  The fp32 input term of these rows (459 MB written + read per forward at B = 64, T = 1000)
// and their sfsn_input_proj_f32 launch disappear; the scan reads the feature rows instead (152 B per row and frame).
//   * loader wave: the 8 rows' features of a frame are one contiguous block (8 x I floats): LDS-DMA into a ring, and -- two frames
//     later, behind its own counted wait -- the same wave splits them into three bf16 planes (row stride 72 elements: B fragments
//     of the two frames of a pair read conflict free); the compute waves only ever see planes that a step barrier has published;
//   * compute waves: W_ih piece 1 register resident (8 VGPRs), pieces 2 and 3 in LDS as A fragments (written by the wave itself at
//     set-up); k-chunk 0 in the even step, k-chunk 1 in the odd one (accumulators live across the barrier), operands requested
//     under the cell's dependent chain, products behind the epilogue, the finished term re-dealt by DPP -- as in the FUSED3 role;
//   * storer wave: all stores of the role (fp32 + int8 spikes; publishing when layer 1 runs in the same launch); spare waves keep
//     the barrier count.  W_hh as in the FUSED3 role (the 32-wide k tail as one 16x16x32 step, issued first).
#ifndef SFSN_SCAN3X_DEV_H
#define SFSN_SCAN3X_DEV_H
#include "sfsn_scan3i_dev.h"

template <int KS, int FLG>
struct Scan3xCfg {
    static constexpr int RPW = 8, HP = KS * 64, LDH = HP + 32;
    static constexpr bool PUB = (FLG & 2) != 0;
    static constexpr int NPX = 2, XSLOT = NPX * 1024;      // a frame's features: 8 rows x I floats <= 2 KiB (I <= 64)
    static constexpr int A = 12, DX = A + 2;               // frame t + A is requested during step t (8 frames in flight: a load takes 2-4 us beside the write-through traffic of the launch), frame t + 4 converted
    static constexpr int LDX = 72;                         // bf16 elements per plane row (64 k + 8: 16-byte chunks at an odd stride)
    static constexpr int PLANE = 8 * LDX * 2;              // one bf16 plane of a frame's 8 rows
    static constexpr int DP = 6;                           // plane slots (frames): the conversion of step t reuses the slot of frame t - 2
    static constexpr int PL_OFF = DX * XSLOT;
    static constexpr int HBUF_OFF = PL_OFF + DP * 3 * PLANE;
    static constexpr int WX_OFF = HBUF_OFF + 2 * 16 * LDH;
    __host__ __device__ static constexpr int wx_bytes(int NT) { return 2 * NT * 2 * 1024; }  // pieces 2, 3 x tiles x 2 k-chunks x 1 KiB
    __host__ __device__ static constexpr int cstx_off(int NT) { return WX_OFF + wx_bytes(NT); }
    __host__ __device__ static constexpr int flag_off(int NT) { return cstx_off(NT) + (HP / 2) * 8; }
    __host__ __device__ static constexpr int lds_bytes(int NT) { return flag_off(NT) + 16; }
};

struct Scan3xRole {
    const float* x;       // [T][R][I] the layer input (sfsn_features' output), R a multiple of 8
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
    unsigned long long* count = nullptr;  // (as Scan3Role::count)
    int lsplit = SFSN_S3X_LSPLIT;         // fp32 store instructions per frame issued by the loader wave (see SFSN_S3_LSPLIT)
#ifdef SFSN_EXPERIMENTS
    unsigned long long* probe = nullptr;  // (as Scan3Role::probe)
#endif
};

// KSB: 32-wide k-chunks of the input product (2: 32 < I <= 64, 1: I <= 32) -- compile time, so that a step is straight-line code (a
// wave-uniform `if` around the operand loads made hipcc drain lgkmcnt at every merge: 2.0 us per step instead of 0.9)
template <int KS, int TL, int OUT, int FLG, int KSB, int D0 = 0>  // (D0 = 1: 16-bit recurrent weights, digit plane 0 skipped -- see scan3i_role)
__device__ __forceinline__ void scan3x_role(const Scan3xRole& rl, const StackLink& lk, char* smem, int T, int H, int NT) {
    using C = Scan3xCfg<KS, FLG>;
    constexpr int RPW = 8, LDH = C::LDH, HP = C::HP, A = C::A, DX = C::DX, XSLOT = C::XSLOT, NPX = C::NPX, LDX = C::LDX, PLANE = C::PLANE, DP = C::DP;
    constexpr bool PUB = C::PUB;
    constexpr int KSF = TL ? KS - 1 : KS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int R = rl.R, row0 = rl.row0, I = rl.I;
    int8_t* hbuf = reinterpret_cast<int8_t*>(smem + C::HBUF_OFF);

    // ---- set-up by all threads: state buffers and bf16 planes zeroed (k >= I must read as 0), h_{-1} -> hbuf[0], {b_f} -> LDS
    for (int i = tid; i < 2 * 16 * LDH / 4; i += 1024) reinterpret_cast<int*>(hbuf)[i] = 0;
    for (int i = tid; i < DP * 3 * PLANE / 4; i += 1024) reinterpret_cast<int*>(smem + C::PL_OFF)[i] = 0;
    for (int i = tid; i < HP / 2; i += 1024) {
        v2f b = {0.f, 0.f};
        if (2 * i < H) b = v2f{rl.bias[2 * i], rl.bias[2 * i + 1]};
        reinterpret_cast<v2f*>(smem + C::cstx_off(NT))[i] = b;
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
        const int row = n & 7, sub = 2 * (n >> 3);
        const int cj = ct * 16 + q * 4 + sub;
        const bool live = row0 + row < R;
        const int grow = live ? row0 + row : R - 1;
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
        bf8 Wx1[2];
        const unsigned wxoff = (unsigned)(C::WX_OFF + (ct * 2) * 1024 + lane * 16);
        const int wxplane = NT * 2 * 1024;
        {
            const int wr = ct * 16 + n;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
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
        // my B fragment of the input product: column n = (frame f0 + (n >> 3), row n & 7), k = 32 ks + 8 q .. + 7 of plane pl
        const unsigned xoff = (unsigned)(C::PL_OFF + (n >> 3) * 3 * PLANE + (row * LDX + q * 8) * 2);
        const unsigned bqoff = (unsigned)(C::cstx_off(NT) + (cj >> 1) * 8);
        float zc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        v4f hi = {0.f, 0.f, 0.f, 0.f}, lo = {0.f, 0.f, 0.f, 0.f};
        bf8 px[3], pw2, pw3;

        auto pf_load = [&](int f0, int ks, int part) __attribute__((always_inline)) {
            const char* pl = smem + xoff + (f0 % DP) * 3 * PLANE + ks * 64;
            if (part == 0) {
                px[0] = *reinterpret_cast<const bf8*>(pl);
                px[1] = *reinterpret_cast<const bf8*>(pl + PLANE);
                px[2] = *reinterpret_cast<const bf8*>(pl + 2 * PLANE);
            } else {
                pw2 = *reinterpret_cast<const bf8*>(smem + wxoff + ks * 1024);
                pw3 = *reinterpret_cast<const bf8*>(smem + wxoff + wxplane + ks * 1024);
            }
        };
        auto pf_mfma = [&](int ks) __attribute__((always_inline)) {  // the product sequence of input_proj_bf3_kernel, instruction for instruction
            lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw3, px[0], lo, 0, 0, 0);
            hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wx1[ks], px[0], hi, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw2, px[1], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wx1[ks], px[2], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw2, px[0], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wx1[ks], px[1], lo, 0, 0, 0);
        };
        // z = (hi + lo) + b_f (= input_proj_bf3_kernel's epilogue), re-dealt as in the FUSED3 role
        auto in_finish = [&](const v2f bq) __attribute__((always_inline)) {
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = hi[k] + lo[k];
            const int f00 = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, r[0]), __builtin_bit_cast(int, r[2]), 0x118, 0xf, 0xC, false);
            const int f01 = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, r[1]), __builtin_bit_cast(int, r[3]), 0x118, 0xf, 0xC, false);
            const int f10 = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, r[2]), __builtin_bit_cast(int, r[0]), 0x108, 0xf, 0x3, false);
            const int f11 = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, r[3]), __builtin_bit_cast(int, r[1]), 0x108, 0xf, 0x3, false);
            zc[0][0] = __builtin_bit_cast(float, f00) + bq.x;
            zc[0][1] = __builtin_bit_cast(float, f01) + bq.y;
            zc[1][0] = __builtin_bit_cast(float, f10) + bq.x;
            zc[1][1] = __builtin_bit_cast(float, f11) + bq.y;
            hi = lo = v4f{0.f, 0.f, 0.f, 0.f};
        };

        __syncthreads();                       // initial state, W_ih pieces 2 / 3 and constants in LDS
        __builtin_amdgcn_s_barrier();          // the loader's planes of frames 0 .. 3
        pf_load(0, 0, 0); pf_load(0, 0, 1);
        pf_mfma(0);
        if constexpr (KSB > 1) { pf_load(0, 1, 0); pf_load(0, 1, 1); pf_mfma(1); }
        in_finish(*reinterpret_cast<const v2f*>(smem + bqoff));
        S3_PB_DECL();
#pragma unroll 1
        for (int t2 = 0; t2 < T; t2 += 2) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int t = t2 + par;
                if (par == 1 && t >= T) break;
                const int8_t* hc = hbuf + (t & 1) * 16 * LDH;
                int8_t* hn = hbuf + ((t & 1) ^ 1) * 16 * LDH;
                v4i b[KSF > 0 ? KSF : 1];
#pragma unroll
                for (int ks = 0; ks < KSF; ++ks) b[ks] = *reinterpret_cast<const v4i*>(hc + boff + ks * 64);
                long bt = 0;
                if constexpr (TL) bt = *reinterpret_cast<const long*>(hc + boft);
                v4i a[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
                if constexpr (TL) {  // (the 32-wide tail step first, with its wait states: see scan3i_role)
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
                    if constexpr (KSF == 0) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                }
#pragma unroll
                for (int ks = 0; ks < KSF; ++ks)
#pragma unroll
                    for (int d = D0; d < 3; ++d) a[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Whh[ks][d], b[ks], a[d], 0, 0, 0);
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
                __builtin_amdgcn_sched_barrier(0);
                if (par < KSB) pf_load(t2 + 2, par, 0);
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
                }
                *reinterpret_cast<unsigned short*>(hn + hoff) = (unsigned short)pk;
                __builtin_amdgcn_sched_barrier(0);
#ifndef SFSN_X3_EXP
#define SFSN_X3_EXP 0  // timing experiments (wrong results): 1 no matrix instructions, 2 no operand loads behind the cell, 4 no finish
#endif
                if (par < KSB && !(SFSN_X3_EXP & 2)) { pf_load(t2 + 2, par, 1); }
                v2f bq = {0.f, 0.f};
                if (par == 1) bq = *reinterpret_cast<const v2f*>(smem + bqoff);
                if (par < KSB && !(SFSN_X3_EXP & 1)) pf_mfma(par);
                if (par == 1 && !(SFSN_X3_EXP & 4)) in_finish(bq);
                __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
                S3_PB_TIC();
                __builtin_amdgcn_s_barrier();
                S3_PB_TOC(0);
            }
        }
        S3_PB_OUT(rl, wave, lane);
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
        // ================================================= loader wave: features -> ring -> bf16 planes =================================================
        // a frame's block = rows row0 .. row0 + 7 x I floats, contiguous in [T][R][I] (R is a multiple of 8): chunk e = 64 p + lane
        const int nchunk = 2 * I;  // 8 * I floats / 4
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
        // my share of a frame's fp32 spike stores (round 5, SFSN_S3_LSPLIT: nine 1 KiB stores per frame through the storer wave alone
        // were 1260 clk of store issue per step)
        S3FlushF<RPW, LDH> ffl;
        if constexpr (OUT & 1) ffl.init(lane, row0, R, H, 0, rl.lsplit);
        const int lst = (OUT & 1) ? ffl.nsf : 0;
        int allow = (A - 4) * (NPX + lst) + lst;  // behind the DMA of frame t + 4: the DMAs and stores of the steps since, and this step's stores
        if (allow > 62) allow = 62;
        auto issue = [&](int slot, int td) __attribute__((always_inline)) {
#pragma unroll
            for (int p = 0; p < NPX; ++p)
                dma16_to_lds(__builtin_amdgcn_readfirstlane((unsigned)(slot * XSLOT + p * 1024)), xbase + (size_t)td * frame, goff[p]);
        };
        // frame fr: ring slot -> the three bf16 planes of plane slot fr % DP (k >= I stays zero: never written)
        auto convert = [&](int fr) __attribute__((always_inline)) {
            const char* src = smem + (fr % DX) * XSLOT;
            unsigned short* pl = reinterpret_cast<unsigned short*>(smem + C::PL_OFF + (fr % DP) * 3 * PLANE);
#pragma unroll
            for (int p = 0; p < NPX; ++p) {
                const v4f v = *reinterpret_cast<const v4f*>(src + p * 1024 + lane * 16);  // (a clamped lane's DMA landed at its OWN lds position)
                unsigned p1[2], p2[2], p3[2];
                split3(v[0], v[1], p1[0], p2[0], p3[0]);
                split3(v[2], v[3], p1[1], p2[1], p3[1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int sh = (j & 1) * 16;
                    pl[po[p][j]] = (unsigned short)(p1[j >> 1] >> sh);
                    pl[PLANE / 2 + po[p][j]] = (unsigned short)(p2[j >> 1] >> sh);
                    pl[PLANE + po[p][j]] = (unsigned short)(p3[j >> 1] >> sh);
                }
            }
        };
        __syncthreads();
        if (T > 0)
            for (int s0 = 0; s0 < A; ++s0) issue(s0, s0 < T ? s0 : T - 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int fr = 0; fr < 4; ++fr) convert(fr);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        S3_PB_DECL();
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            issue((t + A) % DX, (t + A < T) ? t + A : T - 1);
            if constexpr (OUT & 1) if (t > 0) ffl.run(hbuf + (t & 1) * 16 * LDH, rl.spikes_f32 + ((size_t)(t - 1) * R + row0) * H, lane);
            // frames t + 5, t + 6 may stay in flight: frame t + 4 has landed -> convert it (its planes are read from step t + 1 or
            // t + 2 on; the slot it overwrites held frame t - 2, dead for two barriers -- also at step 0 / 1, where a compute wave may
            // still be reading frames 0, 1 for the product it forms before the loop)
            S3_PB_TIC();
            wait_vmcnt_n(allow);
            S3_PB_TOC(1);
#ifndef SFSN_X3_NOCONVERT  // (timing experiment: wrong results)
            S3_PB_TIC();
            convert(t + 4);
            S3_PB_TOC(2);  // (this role has no hand-off polls: the third counter is the bf16 split of a frame)
#endif
            __builtin_amdgcn_s_waitcnt(0xc07f);
            S3_PB_TIC();
            __builtin_amdgcn_s_barrier();
            S3_PB_TOC(0);
        }
        S3_PB_OUT(rl, 14, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (OUT & 1) if (T > 0) ffl.run(hbuf + (T & 1) * 16 * LDH, rl.spikes_f32 + ((size_t)(T - 1) * R + row0) * H, lane);
        return;
    }

    if (wave == NT + 1) {
        // ================================================= storer wave: every store of the role but the loader's share =================================================
        constexpr int MAX8 = (RPW * KS * 4 + 63) / 64;
        constexpr int nu8 = RPW * (HP / 16), ns8 = (nu8 + 63) / 64;
        constexpr bool F32 = (OUT & 1) != 0;
        S3FlushF<RPW, LDH> ff;
        if constexpr (F32) ff.init(lane, row0, R, H, rl.lsplit);
        int l8[MAX8];
        unsigned ok8 = 0;
        unsigned cnt = 0;  // (as scan3_role's storer)
#pragma unroll
        for (int k = 0; k < MAX8; ++k) {
            const int u = 64 * k + lane, rr = u / (HP / 16), c16 = u - rr * (HP / 16);
            l8[k] = rr * LDH + c16 * 16;
            if (k < ns8 && u < nu8 && row0 + rr < R) ok8 |= 1u << k;
        }
        auto flush = [&](const int8_t* hsrc, int ts) __attribute__((always_inline)) {
            if constexpr (OUT & 2) {  // (the int8 rows first: in a publishing role they are what the counted wait is about)
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
            if constexpr (F32) ff.run(hsrc, rl.spikes_f32 + ((size_t)ts * R + row0) * H, lane);
        };
        const int rows_live = (R - row0 < RPW) ? R - row0 : RPW;
        const int spf = (F32 ? ff.nsf : 0) + ((OUT & 2) ? (rows_live * (HP / 16) + 63) / 64 : 0);
        const int pf = spf > 0 ? (62 / spf < SFSN_S3_PFMAX ? 62 / spf : SFSN_S3_PFMAX) : 8;
        __syncthreads();
        __builtin_amdgcn_s_barrier();
        S3_PB_DECL();
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            if (t > 0) {
                flush(hbuf + (t & 1) * 16 * LDH, t - 1);
                if constexpr (PUB) {
                    S3_PB_TIC();
                    wait_vmcnt_n(pf * spf);
                    S3_PB_TOC(1);
                    if (lane == 0 && t - pf > 0) stack_publish(lk, t - pf);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            S3_PB_TIC();
            __builtin_amdgcn_s_barrier();
            S3_PB_TOC(0);
        }
        S3_PB_OUT(rl, 15, lane);
        if (T > 0) flush(hbuf + (T & 1) * 16 * LDH, T - 1);
        if constexpr (PUB) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) stack_publish(lk, T);
        }
        if constexpr (!(OUT & 1)) wave_count_add(rl.count, cnt);
        return;
    }

    // ================================================= spare waves (NT < 14): keep the barrier count =================================================
    __syncthreads();
    __builtin_amdgcn_s_barrier();
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    }
}

#endif
