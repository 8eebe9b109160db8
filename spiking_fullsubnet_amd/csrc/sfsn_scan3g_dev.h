// sfsn_scan3g_dev.h -- the IO-wave scan for SEPARATE gate weights (shared_weights = false, NEURON:137-139), round 6, gfx950 only.
//
// baseline_xl's sub-band layers (H = 224, two gates: recipes/intel_ndns/spiking_fullsubnet_freeze_phase/baseline_xl.toml:61,64) ran
// round 2's body as 8 waves x 2 tiles x 2 gates: 192 registers of weights per wave, 116 of them spilled, 2.6 us per step.  This is
// scan3_role's structure with both gates' rows of a tile in ONE compute wave (the cell needs the forget gate and the candidate of a
// neuron in the same lane): digit planes 1 and 2 of both gates in registers (64 VGPRs), plane 0 of both in LDS as A fragments
// (2 x NT x KS KiB: 112 KiB at H = 224, read 1 KiB contiguous per wave instruction), the two gates multiplied ONE AFTER THE OTHER
// through the same accumulators (the forget gate's exact sum is re-dealt and kept as an integer while the candidate's product runs),
// 24 matrix instructions per tile and step.  Loader / storer / spare waves as in scan3_role (the input term is [T][R][2 H]: both
// gates' columns of a frame in one ring slot).  4 rows per workgroup (the ring and the LDS plane leave no room for 8), no links.
// Arithmetic is scan_body<G = 2>'s value for value: pre_f = fma(rec_f, dq_f, z_f), pre_g = fma(rec_g, dq_g, z_g) (the input term
// carries both gates' biases), the same cell -- bit-identical outputs (tests/test_hip_parity.py).
#ifndef SFSN_SCAN3G_DEV_H
#define SFSN_SCAN3G_DEV_H
#include "sfsn_scan3_dev.h"

template <int KS>
struct Scan3gCfg {
    static constexpr int RPW = 4, HP = KS * 64, LDH = HP + 32;
    __host__ __device__ static constexpr int chunks(int NT) { return RPW * 2 * NT * 4; }
    __host__ __device__ static constexpr int pieces(int NT) { return (chunks(NT) + 63) / 64; }
    __host__ __device__ static constexpr int slot_bytes(int NT) { return pieces(NT) * 1024; }
    static constexpr int MAXP = (RPW * 28 * 4 + 63) / 64;  // pieces at NT = 14
    static constexpr int D = 4;                            // input-term ring depth (frames): a plain role's (Scan3Cfg)
    __host__ __device__ static constexpr int hbuf_off(int NT) { return D * slot_bytes(NT); }
    __host__ __device__ static constexpr int plane_off(int NT) { return hbuf_off(NT) + 2 * 16 * LDH; }
    __host__ __device__ static constexpr int lds_bytes(int NT) { return plane_off(NT) + 2 * NT * KS * 1024; }
};

template <int KS, int OUT>
__device__ __forceinline__ void scan3g_role(const Scan3Role& rl, char* smem, int T, int H, int NT) {
    using C = Scan3gCfg<KS>;
    constexpr int RPW = 4, LDH = C::LDH, HP = C::HP, D = C::D;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int R = rl.R, row0 = rl.row0;
    const int SLOT = C::slot_bytes(NT);
    int8_t* hbuf = reinterpret_cast<int8_t*>(smem + C::hbuf_off(NT));
    char* wplane = smem + C::plane_off(NT);

    // ---- set-up by all threads: state buffers zeroed, digit plane 0 of both gates -> LDS (the first 2 NT KS KiB of the packed array),
    //      h_{-1} -> hbuf[0]
    for (int i = tid; i < 2 * 16 * LDH / 4; i += 1024) reinterpret_cast<int*>(hbuf)[i] = 0;
    for (int i = tid; i < 2 * NT * KS * 64; i += 1024) reinterpret_cast<v4i*>(wplane)[i] = reinterpret_cast<const v4i*>(rl.w_hh)[i];
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
        // ================================================= compute wave: output tile `wave`, both gates =================================================
        const int ct = wave;
        const int row = n & 3, sub = n >> 2;
        const int cj = ct * 16 + q * 4 + sub;  // my neuron
        const bool live = row0 + row < R;
        const int grow = live ? row0 + row : R - 1;
        v4i W[2][KS][2];  // [gate][k-step][digit plane 1, 2]
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int d = 0; d < 2; ++d)
                    W[g][ks][d] = *reinterpret_cast<const v4i*>(rl.w_hh + ((((size_t)(d + 1) * 2 * NT + (size_t)g * NT + ct) * KS + ks) * 64 + lane) * 16);
        float c = rl.c_state[(size_t)grow * H + cj];
        const float dqf = rl.w_dq[cj], dqg = rl.w_dq[H + cj], al = rl.bn_alpha[cj], be = rl.bn_beta[cj];
        // my input-term bytes within a ring slot: chunk ((g NT + tile) 4 + q) RPW + row, element `sub`
        const unsigned zoff_f = (unsigned)((((ct * 4 + q) * RPW + row) * 16) + sub * 4);
        const unsigned zoff_g = (unsigned)(((((NT + ct) * 4 + q) * RPW + row) * 16) + sub * 4);
        const unsigned boff = (unsigned)(n * LDH + q * 16);
        const unsigned hoff = (unsigned)(row * LDH + cj);
        const unsigned wl_f = (unsigned)((ct * KS) * 1024 + lane * 16), wl_g = (unsigned)(((NT + ct) * KS) * 1024 + lane * 16);
        __syncthreads();                       // initial state in hbuf[0], the LDS digit plane
        __builtin_amdgcn_s_barrier();          // the loader's prologue frames have landed
        auto pick = [&](const v4i& a) __attribute__((always_inline)) {  // 4 rows: element n / 4 of lane (row, q) -> one value per lane
            int x = a[0];
            x = __builtin_amdgcn_update_dpp(x, a[1], 0x114, 0xf, 0x2, false);  // row_shr:4  -> lanes 4..7
            x = __builtin_amdgcn_update_dpp(x, a[2], 0x118, 0xf, 0x4, false);  // row_shr:8  -> lanes 8..11
            x = __builtin_amdgcn_update_dpp(x, a[3], 0x11C, 0xf, 0x8, false);  // row_shr:12 -> lanes 12..15
            return x;
        };
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            const int8_t* hc = hbuf + (t & 1) * 16 * LDH;
            int8_t* hn = hbuf + ((t & 1) ^ 1) * 16 * LDH;
            v4i b[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) b[ks] = *reinterpret_cast<const v4i*>(hc + boff + ks * 64);
            const char* zp = smem + (t % D) * SLOT;
            const float zf = *reinterpret_cast<const float*>(zp + zoff_f), zg = *reinterpret_cast<const float*>(zp + zoff_g);
            float rec[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                v4i a[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const v4i w0 = *reinterpret_cast<const v4i*>(wplane + (g == 0 ? wl_f : wl_g) + ks * 1024);
                    a[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, b[ks], a[0], 0, 0, 0);
                    a[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[g][ks][0], b[ks], a[1], 0, 0, 0);
                    a[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[g][ks][1], b[ks], a[2], 0, 0, 0);
                }
                rec[g] = recombine3(pick(a[0]), pick(a[1]), pick(a[2]));  // (scan_body's form: exact, rounded once)
            }
            const float pre_f = __builtin_fmaf(rec[0], dqf, zf);
            const float pre_g = __builtin_fmaf(rec[1], dqg, zg);
            const float f = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre_f * -1.44269504088896341f));
            const float m = __builtin_fmaf(f, c - pre_g, pre_g);
            const float y = __builtin_fmaf(m, al, be);
            c = y;
            hn[hoff] = (y >= 0.0f) ? 1 : 0;
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
        }
        const int8_t* hl = hbuf + (T & 1) * 16 * LDH;
        if (live) {
            rl.c_state[(size_t)grow * H + cj] = c;
            rl.h_state[(size_t)grow * H + cj] = (float)hl[hoff];
        }
        return;
    }

    if (wave == NT) {
        // ================================================= loader wave =================================================
        const int np = C::pieces(NT), nch = C::chunks(NT);
        unsigned goff[C::MAXP];
#pragma unroll
        for (int p = 0; p < C::MAXP; ++p) {
            int e = 64 * p + lane;
            if (e > nch - 1) e = nch - 1;  // surplus lanes of the last piece re-fetch the last chunk
            const int cidx = e / RPW, r = e - cidx * RPW;
            const int grow = (row0 + r < R) ? row0 + r : R - 1;
            goff[p] = (unsigned)((grow * 2 * H + cidx * 4) * 4);  // (chunk (g NT + tile) 4 + q = floats [4 cidx, +4) of the row's 2 H)
        }
        const size_t frame = (size_t)R * 2 * H;
        int allow = (D - 2) * np;
        if (allow > 62) allow = 62;
        auto issue = [&](int slot, int td) __attribute__((always_inline)) {
            const float* zt = rl.zin + (size_t)td * frame;
#pragma unroll
            for (int p = 0; p < C::MAXP; ++p)
                if (p < np) dma16_to_lds<false>(__builtin_amdgcn_readfirstlane((unsigned)(slot * SLOT + p * 1024)), zt, goff[p]);  // wave-uniform
        };
        __syncthreads();
        for (int s0 = 0; s0 < D - 1; ++s0) issue(s0, s0 < T ? s0 : (T > 0 ? T - 1 : 0));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            const int td = (t + D - 1 < T) ? t + D - 1 : T - 1;
            issue((t + D - 1) % D, td);  // the slot of frame t - 1: read during step t - 1
            wait_vmcnt_n(allow);         // frames t + 2 .. t + D - 1 may stay in flight: frame t + 1 has landed
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // DMAs past the end are invisible to the compiler
        return;
    }

    if (wave == NT + 1) {
        // ================================================= storer wave (scan3_role's, no links) =================================================
        constexpr int MAX8 = (RPW * KS * 4 + 63) / 64;
        constexpr int nu8 = RPW * (HP / 16), ns8 = (nu8 + 63) / 64;
        constexpr bool F32 = (OUT & 1) != 0;
        S3FlushF<RPW, LDH> ff;
        if constexpr (F32) ff.init(lane, row0, R, H);
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

    // ================================================= spare waves (NT < 14) =================================================
    __syncthreads();
    __builtin_amdgcn_s_barrier();
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    }
}

#endif
