// sfsn_kernels.hip -- gfx950 (MI355X / CDNA4) kernels behind include/sfsn.h.
//
// Written for gfx950 only: 64-lane wavefronts, v_mfma_i32_16x16x64_i8 / v_mfma_f32_16x16x4_f32,
// 160 KiB LDS, 512-entry unified VGPR/AGPR file.  No other target is supported or guarded for.
//
// Design notes (DESIGN.md has the full discussion):
//  * Spikes are exactly {0,1} (efficient_spiking_neuron.py:89), so every product whose left operand is a
//    spike tensor runs on the int8 matrix cores with the fp32 weights split into three base-256 digits
//    (sfsn_w3_pack): integer accumulation is exact and order independent, the recombination rounds once.
//  * MFMA orientation: A = weights (16 output neurons x 64 k), B = spikes^T (64 k x 16 rows).  The
//    accumulator then holds, per lane, 4 CONSECUTIVE output neurons of ONE row (row = lane & 15,
//    neuron = 16*tile + 4*(lane >> 4) + r), so every global access of the epilogue is a 16-byte vector
//    and the new spikes go to LDS as one packed dword.
//  * The scan keeps W_hh register resident for all T steps (one workgroup = 16 rows, W tiles dealt
//    round-robin to the waves so the four SIMDs carry equal MFMA load), the hidden state in LDS as int8
//    (double buffered: one barrier per step), the membrane in registers in accumulator layout.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "sfsn.h"

#include "sfsn_scan_dev.h"
#include "sfsn_scan3_dev.h"
#include "sfsn_scan3j_dev.h"
#include "sfsn_scan3g_dev.h"
#include "sfsn_feat_dev.h"

// G = 1: shared gate weights (W [H][*] used for both gates); G = 2: separate forget / cell weights.
// KS = 64-wide k steps; NW = waves per workgroup; TPW = tiles owned by the first (NT - NW*(TPW-1)) waves, the
// rest own TPW-1 (tiles are dealt round-robin, so the four SIMDs carry equal MFMA load and no wave computes a
// tile that does not exist).  OUT bit 0: fp32 spikes, bit 1: int8 spikes, bit 2: membranes -- compile-time so that
// the stores are straight-line code.  LDS (dynamic, one allocation): [input-term ring][hidden state x2][constants].
template <int G, int KS, int NW, int TPW, int OUT, int LP>
__global__ __launch_bounds__(NW * 64) void gsn_scan_kernel(const ScanParams p) {
    using C = ScanCfg<G, KS, NW, TPW, OUT, LP>;
    constexpr int LDH = C::LDH, HP = C::HP;
    extern __shared__ __attribute__((aligned(16))) char scan_smem[];
    char* smem = scan_smem;
    int8_t* hbuf = reinterpret_cast<int8_t*>(smem + C::HBUF_OFF);
    float(*cst)[HP] = reinterpret_cast<float(*)[HP]>(smem + C::CST_OFF);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15;  // row within the tile (B / D column)
    const int q = lane >> 4;  // k group for A/B fragments; 4-neuron group for D
    SFSN_WG_STAMP(p.wg_times, 0);

    int s = 0;
    for (int i = 1; i < p.nseg; ++i)
        if ((int)blockIdx.x >= p.seg[i].tile0) s = i;
    const ScanSegDev sg = p.seg[s];
    const int H = p.H, NT = p.NT, T = p.T, R = sg.R;
    // A workgroup owns rpw rows; MFMA columns n >= rpw are clamped duplicates of column n % rpw (same data, same
    // results, same addresses).  Fewer rows per workgroup = more CUs busy and less HBM traffic per CU per step: one CU
    // sustains only ~10-13 B/clk of HBM streaming, which at 16 rows x 224 neurons (32 KB per step) IS the step time.
    const int rpw = p.rpw;
    const int row0 = ((int)blockIdx.x - sg.tile0) * rpw;
    const int rowc = (row0 + (n & (rpw - 1)) < R) ? row0 + (n & (rpw - 1)) : R - 1;

    scan_prologue<G, KS, NW, TPW, OUT, LP>(sg, smem, tid, H, NT, R, row0, rpw);

    const int n_hi = NT - NW * (TPW - 1);  // waves [0, n_hi) own TPW tiles, the others TPW-1
    if (wave < n_hi)
        scan_body<G, KS, NW, TPW, OUT, LP, TPW>(sg.zin, sg.w_hh, sg.spikes_f32, sg.spikes_i8, sg.membrane, sg.h_state, sg.c_state, smem, T,
                                            H, NT, R, row0, rowc, n, q, tid, wave, rpw, nullptr, nullptr, sg.count);
    else
        scan_body<G, KS, NW, TPW, OUT, LP, TPW - 1>(sg.zin, sg.w_hh, sg.spikes_f32, sg.spikes_i8, sg.membrane, sg.h_state, sg.c_state,
                                                smem, T, H, NT, R, row0, rowc, n, q, tid, wave, rpw, nullptr, nullptr, sg.count);
    SFSN_WG_STAMP(p.wg_times, 1);
}


// ---- the scan with IO-specialised waves (sfsn_scan3_dev.h): shared gates, H <= 224, one layer per launch, any segments ----------
template <int KS, int RPW, int OUT, int D0 = 0>
__global__ __launch_bounds__(1024) void gsn_scan3_kernel(const ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) char scan_smem[];
    int s = 0;
    for (int i = 1; i < p.nseg; ++i)
        if ((int)blockIdx.x >= p.seg[i].tile0) s = i;
    const ScanSegDev& sg = p.seg[s];
    Scan3Role rl;
    rl.zin = sg.zin; rl.w_hh = sg.w_hh; rl.w_dq = sg.w_dq; rl.bias = sg.bias; rl.bn_alpha = sg.bn_alpha; rl.bn_beta = sg.bn_beta;
    rl.h_state = sg.h_state; rl.c_state = sg.c_state; rl.spikes_f32 = sg.spikes_f32; rl.spikes_i8 = sg.spikes_i8;
    rl.R = sg.R; rl.row0 = ((int)blockIdx.x - sg.tile0) * RPW;
    rl.count = sg.count;
    rl.lsplit = p.lsplit;
    StackLink lk;
    lk.in = nullptr; lk.n_in = 0; lk.out = nullptr; lk.err = nullptr; lk.lag = 0; lk.dbg = nullptr;
    scan3_role<KS, RPW, OUT, 0, D0>(rl, lk, scan_smem, p.T, p.H, p.NT);
}

// Round 6: the same structure for SEPARATE gate weights (sfsn_scan3g_dev.h): both gates of a tile in one compute wave, 4 rows per workgroup
template <int KS, int OUT>
__global__ __launch_bounds__(1024) void gsn_scan3g_kernel(const ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) char scan_smem[];
    int s = 0;
    for (int i = 1; i < p.nseg; ++i)
        if ((int)blockIdx.x >= p.seg[i].tile0) s = i;
    const ScanSegDev& sg = p.seg[s];
    Scan3Role rl;
    rl.zin = sg.zin; rl.w_hh = sg.w_hh; rl.w_dq = sg.w_dq; rl.bias = sg.bias; rl.bn_alpha = sg.bn_alpha; rl.bn_beta = sg.bn_beta;
    rl.h_state = sg.h_state; rl.c_state = sg.c_state; rl.spikes_f32 = sg.spikes_f32; rl.spikes_i8 = sg.spikes_i8;
    rl.R = sg.R; rl.row0 = ((int)blockIdx.x - sg.tile0) * 4;
    rl.count = sg.count;
    scan3g_role<KS, OUT>(rl, scan_smem, p.T, p.H, p.NT);
}

// ---- fused-input scan (layers >= 1, shared gates, 128 < H <= 256, 16 rows per workgroup) -----------------------------------
// With the chip full the scan is HBM-bound (PMC + scripts/exp_dupmfma.py: doubling its matrix work does not change the launch
// time), and half of what it reads is the fp32 input term x.W_ih^T + b that sfsn_spike_proj wrote just before: 2 x 745 MB per
// sub-band layer at B=64, T=1000.  For a layer whose input is the previous layer's spikes that term is the same kind of
// product as the recurrent one, so this variant computes it in place: it reads the int8 spikes (4 KB per step instead of
// 14 KB) and keeps BOTH weight matrices in the CU -- digit plane 0 of each in LDS (2 x NT*KS KB), planes 1 and 2 in
// registers (8 waves x 2 tiles: 128 VGPRs).  24 instead of 12 MFMAs per tile and step; results bit-identical to
// sfsn_spike_proj + sfsn_gsn_layer_scan (same exact integer products, same two roundings: fma(rec_ih, dq, b) then
// fma(rec_hh, dq, .)).  Input spikes arrive through an LDS-DMA ring like the input term of the other variant (row chunks
// rotated by the row index so that B-fragment reads spread over the banks), spikes leave through ScanFlush.
template <int KS, int OUT, int NTL>
__device__ __forceinline__ void fused_body(const ScanSegDev& sg, char* smem, int T, int H, int NT, int R, int row0, int rowc, int n,
                                           int q, int tid, int wave) {
    using C = ScanCfg<1, KS, 8, 2, OUT, 0>;  // geometry constants of the flush only (LDH, HP, FL, NSTF)
    constexpr int LDH = C::LDH, HP = C::HP, D = 3, NW = 8, NCH = KS * 4;  // NCH = 16-byte chunks per input row
    constexpr int SR_BYTES = 16 * HP;                                      // one ring slot: 16 rows of int8 spikes
    constexpr int HBUF_OFF = D * SR_BYTES, CST_OFF = HBUF_OFF + 2 * 16 * LDH, WHH_OFF = CST_OFF + 6 * HP * 4;
    const int WIH_OFF = WHH_OFF + NT * KS * 1024;
    int8_t* hbuf = reinterpret_cast<int8_t*>(smem + HBUF_OFF);
    const float(*cst)[HP] = reinterpret_cast<const float(*)[HP]>(smem + CST_OFF);  // b_f, b_g - b_f, alpha, beta, dq_hh, dq_ih
    ScanFlush<C> fl;
    fl.init(tid, row0, R, H, NW * 64, 16);
    const int lane = tid & 63;
    constexpr int CBASE = (D - 2) * 1, CSTRIDE = (D - 1) * C::NSTF;  // outstanding operations allowed at the end-of-step wait

    v4i Whh[NTL][KS][2], Wih[NTL][KS][2];
    v4f c[NTL];
    int col[NTL];
    unsigned wl_off[NTL];
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        const int ct = wave + NW * i;
        col[i] = ct * 16 + q * 4;
        wl_off[i] = (unsigned)((ct * KS) * 64 + lane) * 16u;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const size_t tile = (size_t)(d + 1) * NT + ct;
                Whh[i][ks][d] = *reinterpret_cast<const v4i*>(sg.w_hh + ((tile * KS + ks) * 64 + lane) * 16);
                Wih[i][ks][d] = *reinterpret_cast<const v4i*>(sg.w_ih + ((tile * KS + ks) * 64 + lane) * 16);
            }
        c[i] = *reinterpret_cast<const v4f*>(sg.c_state + (size_t)rowc * H + col[i]);
    }
    // input-spike ring: DMA piece k = 64 chunks e = 64k + lane of the 16 x NCH chunks of a step; chunk e sits at LDS byte
    // 16 e and holds global chunk (e % NCH - e / NCH) mod NCH of row e / NCH (a rotation by the row index).
    const int piece = (wave & 3) < KS ? (wave & 3) : KS - 1;  // KS pieces; surplus waves repeat the last one (same bytes)
    const int e = piece * 64 + lane, er = e / NCH, esl = e - er * NCH;
    const int erow = (row0 + er < R) ? row0 + er : R - 1;
    const unsigned src_off = (unsigned)(erow * HP + ((esl - er % NCH + NCH) % NCH) * 16);
    const size_t frame = (size_t)R * HP;
    auto issue = [&](int slot, int td) __attribute__((always_inline)) {
        dma16_to_lds(__builtin_amdgcn_readfirstlane((unsigned)(slot * SR_BYTES + piece * 1024)), reinterpret_cast<const float*>(sg.spikes_in + (size_t)td * frame), src_off);
    };
    for (int s0 = 0; s0 < D - 1; ++s0) issue(s0, s0 < T ? s0 : (T > 0 ? T - 1 : 0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);

    auto step = [&](int t, auto first) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first)::value;
        const int8_t* hc = hbuf + (t & 1) * 16 * LDH;
        int8_t* hn = hbuf + ((t & 1) ^ 1) * 16 * LDH;
        {
            const int td = (t + D - 1 < T) ? t + D - 1 : T - 1;
            issue((t + D - 1) % D, td);
        }
        v4i bh[KS], bs[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bh[ks] = *reinterpret_cast<const v4i*>(hc + n * LDH + ks * 64 + q * 16);
        const char* sslot = smem + (t % D) * SR_BYTES + n * HP;
#pragma unroll
        for (int i = 0; i < NTL; ++i) {
            const int cc = col[i];
            v4i a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v4i w0 = *reinterpret_cast<const v4i*>(smem + WHH_OFF + wl_off[i] + (unsigned)(ks * 1024));
                a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, bh[ks], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Whh[i][ks][0], bh[ks], a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Whh[i][ks][1], bh[ks], a2, 0, 0, 0);
            }
            if (i == 0) {
                // under the first tile's MFMA latency: spikes of step t-1 -> global; then this step's input spikes (every
                // wave's piece of the slot landed before the barrier that ended the previous step, see below)
                if constexpr (!FIRST) fl.template run<OUT>(hc, sg.spikes_f32, sg.spikes_i8, t - 1, R, H);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) bs[ks] = *reinterpret_cast<const v4i*>(sslot + ((ks * 4 + q + n) % NCH) * 16);
            }
            v4i e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0}, e2 = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v4i w0 = *reinterpret_cast<const v4i*>(smem + WIH_OFF + wl_off[i] + (unsigned)(ks * 1024));
                e0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, bs[ks], e0, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wih[i][ks][0], bs[ks], e1, 0, 0, 0);
                e2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wih[i][ks][1], bs[ks], e2, 0, 0, 0);
            }
            const v4f bf = *reinterpret_cast<const v4f*>(&cst[0][cc]), db = *reinterpret_cast<const v4f*>(&cst[1][cc]);
            const v4f alpha = *reinterpret_cast<const v4f*>(&cst[2][cc]), beta = *reinterpret_cast<const v4f*>(&cst[3][cc]);
            const v4f dqh = *reinterpret_cast<const v4f*>(&cst[4][cc]), dqi = *reinterpret_cast<const v4f*>(&cst[5][cc]);
            v4f cy;
            unsigned pk = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = __builtin_fmaf(recombine3(e0[r], e1[r], e2[r]), dqi[r], bf[r]);   // = sfsn_spike_proj's rec*dq + bias
                const float pre_f = __builtin_fmaf(recombine3(a0[r], a1[r], a2[r]), dqh[r], z);
                const float pre_g = pre_f + db[r];
                const float f = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre_f * -1.44269504088896341f));
                const float m = __builtin_fmaf(f, c[i][r] - pre_g, pre_g);
                const float y = __builtin_fmaf(m, alpha[r], beta[r]);
                cy[r] = y;
                pk |= (y >= 0.0f) ? (1u << (8 * r)) : 0u;
            }
            c[i] = cy;
            *reinterpret_cast<unsigned*>(hn + n * LDH + cc) = pk;
        }
        // The input-spike slot of step t+1 is read by EVERY wave but filled by several: each wave waits here for its own
        // piece (issued at the top of step t-1; since then: one more DMA and two steps' flush stores), and the barrier
        // makes all pieces visible to all.  (The input-term ring of the other scan variant needs no such rendezvous: there
        // a wave reads only what it loaded itself.)
        if (t < D || CBASE + fl.nact * CSTRIDE > 63) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            wait_vmcnt_affine<CBASE, CSTRIDE, C::FL>(fl.nact);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    };
    __builtin_amdgcn_s_barrier();  // the prologue's pieces (drained above by every wave) are now visible to all
    if (T > 0) step(0, std::true_type{});
#pragma unroll 1
    for (int t = 1; t < T; ++t) step(t, std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (T > 0) fl.template run<OUT>(hbuf + (T & 1) * 16 * LDH, sg.spikes_f32, sg.spikes_i8, T - 1, R, H);
    fl.template finish<OUT>(sg.count);
    const int8_t* hl = hbuf + (T & 1) * 16 * LDH;
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        *reinterpret_cast<v4f*>(sg.c_state + (size_t)rowc * H + col[i]) = c[i];
        const unsigned pk = *reinterpret_cast<const unsigned*>(hl + n * LDH + col[i]);
        const v4f h = {(float)(pk & 1u), (float)((pk >> 8) & 1u), (float)((pk >> 16) & 1u), (float)((pk >> 24) & 1u)};
        *reinterpret_cast<v4f*>(sg.h_state + (size_t)rowc * H + col[i]) = h;
    }
}

template <int KS, int OUT>
__global__ __launch_bounds__(512) void gsn_scan_fused_kernel(const ScanParams p) {
    using C = ScanCfg<1, KS, 8, 2, OUT, 0>;
    constexpr int LDH = C::LDH, HP = C::HP, D = 3, NW = 8;
    constexpr int HBUF_OFF = D * 16 * HP, CST_OFF = HBUF_OFF + 2 * 16 * LDH, WHH_OFF = CST_OFF + 6 * HP * 4;
    extern __shared__ __attribute__((aligned(16))) char scan_smem[];
    char* smem = scan_smem;
    int8_t* hbuf = reinterpret_cast<int8_t*>(smem + HBUF_OFF);
    float(*cst)[HP] = reinterpret_cast<float(*)[HP]>(smem + CST_OFF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    int s = 0;
    for (int i = 1; i < p.nseg; ++i)
        if ((int)blockIdx.x >= p.seg[i].tile0) s = i;
    const ScanSegDev sg = p.seg[s];
    const int H = p.H, NT = p.NT, T = p.T, R = sg.R;
    const int row0 = ((int)blockIdx.x - sg.tile0) * 16;
    const int rowc = (row0 + n < R) ? row0 + n : R - 1;
    for (int j = tid; j < HP; j += NW * 64) {
        const bool in = j < H;
        cst[0][j] = in ? sg.bias[j] : 0.0f;
        cst[1][j] = in ? sg.bias[H + j] - sg.bias[j] : 0.0f;
        cst[2][j] = in ? sg.bn_alpha[j] : 0.0f;
        cst[3][j] = in ? sg.bn_beta[j] : 0.0f;
        cst[4][j] = in ? sg.w_dq[j] : 0.0f;
        cst[5][j] = in ? sg.w_ih_dq[j] : 0.0f;
    }
    for (int i = tid; i < 2 * 16 * LDH / 4; i += NW * 64) reinterpret_cast<int*>(hbuf)[i] = 0;
    {   // digit plane 0 of both matrices -> LDS, once
        v4i* d0 = reinterpret_cast<v4i*>(smem + WHH_OFF);
        v4i* d1 = reinterpret_cast<v4i*>(smem + WHH_OFF + NT * KS * 1024);
        const v4i* s0 = reinterpret_cast<const v4i*>(sg.w_hh);
        const v4i* s1 = reinterpret_cast<const v4i*>(sg.w_ih);
        for (int i = tid; i < NT * KS * 64; i += NW * 64) {
            d0[i] = s0[i];
            d1[i] = s1[i];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < 16 * (H / 4); idx += NW * 64) {
        const int rr = idx / (H / 4), j4 = (idx - rr * (H / 4)) * 4;
        const int rsrc = (row0 + rr < R) ? row0 + rr : R - 1;
        const v4f h = *reinterpret_cast<const v4f*>(sg.h_state + (size_t)rsrc * H + j4);
        const unsigned pk = (h.x > 0.5f ? 1u : 0u) | (h.y > 0.5f ? 0x100u : 0u) | (h.z > 0.5f ? 0x10000u : 0u) |
                            (h.w > 0.5f ? 0x1000000u : 0u);
        *reinterpret_cast<unsigned*>(hbuf + rr * LDH + j4) = pk;
    }
    __syncthreads();
    SFSN_WG_STAMP(p.wg_times, 0);
    if (wave < NT - NW)  // waves [0, NT - 8) own two tiles, the others one (8 < NT <= 16)
        fused_body<KS, OUT, 2>(sg, smem, T, H, NT, R, row0, rowc, n, q, tid, wave);
    else
        fused_body<KS, OUT, 1>(sg, smem, T, H, NT, R, row0, rowc, n, q, tid, wave);
    SFSN_WG_STAMP(p.wg_times, 1);
}

// Round 6: the fused-input scan at 16 rows per workgroup with IO-specialised waves (sfsn_scan3j_dev.h): what sfsn_gsn_layer_scan_fused
// launches for H <= 224 (at most 14 tiles: one per compute wave + loader + storer); bit-identical to gsn_scan_fused_kernel.
template <int KS, int TL, int OUT, int OFF = 0>
__global__ __launch_bounds__(1024) void gsn_scan_fused3_kernel(const ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) char scan_smem[];
    int s = 0;
    for (int i = 1; i < p.nseg; ++i)
        if ((int)blockIdx.x >= p.seg[i].tile0) s = i;
    const ScanSegDev& sg = p.seg[s];
    Scan3jRole rl;
    rl.spikes_in = sg.spikes_in; rl.w_ih = sg.w_ih; rl.w_ih_dq = sg.w_ih_dq; rl.w_hh = sg.w_hh; rl.w_dq = sg.w_dq; rl.bias = sg.bias;
    rl.bn_alpha = sg.bn_alpha; rl.bn_beta = sg.bn_beta; rl.h_state = sg.h_state; rl.c_state = sg.c_state;
    rl.spikes_f32 = sg.spikes_f32; rl.spikes_i8 = sg.spikes_i8; rl.R = sg.R; rl.row0 = ((int)blockIdx.x - sg.tile0) * 16;
    rl.count = sg.count; rl.lsplit = p.lsplit;
    SFSN_WG_STAMP(p.wg_times, 0);
    scan3j_role<KS, TL, OUT, OFF>(rl, scan_smem, p.T, p.H, p.NT);
    SFSN_WG_STAMP(p.wg_times, 1);
}

// ... and its layer-0 twin (scan3y_role): what sfsn_gsn_layer_scan_fused_x launches for H <= 224; bit-identical to gsn_scan_fusedx_kernel.
template <int KS, int TL, int OUT>
__global__ __launch_bounds__(1024) void gsn_scan_fusedx3_kernel(const ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) char scan_smem[];
    int s = 0;
    for (int i = 1; i < p.nseg; ++i)
        if ((int)blockIdx.x >= p.seg[i].tile0) s = i;
    const ScanSegDev& sg = p.seg[s];
    Scan3yRole rl;
    rl.x = sg.x_in; rl.w_ih = sg.w_ih_f32; rl.I = sg.I; rl.w_hh = sg.w_hh; rl.w_dq = sg.w_dq; rl.bias = sg.bias;
    rl.bn_alpha = sg.bn_alpha; rl.bn_beta = sg.bn_beta; rl.h_state = sg.h_state; rl.c_state = sg.c_state;
    rl.spikes_f32 = sg.spikes_f32; rl.spikes_i8 = sg.spikes_i8; rl.R = sg.R; rl.row0 = ((int)blockIdx.x - sg.tile0) * 16;
    rl.count = sg.count; rl.lsplit = p.lsplit;
    SFSN_WG_STAMP(p.wg_times, 0);
    // (the k-chunk count is compile time: a wave-uniform `if` around operand loads makes hipcc drain lgkmcnt at every merge)
    if (sg.I > 32) scan3y_role<KS, TL, OUT, 2>(rl, scan_smem, p.T, p.H, p.NT);
    else scan3y_role<KS, TL, OUT, 1>(rl, scan_smem, p.T, p.H, p.NT);
    SFSN_WG_STAMP(p.wg_times, 1);
}

// ---- streamed-weights scan: the shapes whose W_hh cannot live in one CU (unshared gates with H > 256: baseline_xl's
// full-band layers, 2 x 320 x 320 x 3 B = 614 KB against 512 KB of registers + 160 KB of LDS) -------------------------
// Same arithmetic as gsn_scan_kernel, bit for bit (same digit MFMAs, same epilogue), but every A fragment is fetched
// from L2 each step (the packed planes are coalesced 1 KB fragments; every workgroup reads the same 600 KB, so they stay
// in L2 / Infinity Cache).  A workgroup owns 16 rows; 8 waves share the output tiles; one barrier per step; inputs and
// outputs go straight between global memory and the accumulator layout.  ~5 us per step: a functional path for a config
// whose trained generator the reference does not even ship, not a tuned one.
template <int G>
__global__ __launch_bounds__(512) void gsn_scan_stream_kernel(const ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) char scan_smem[];
    constexpr int NW = 8, MAXT = (SFSN_MAX_HIDDEN / 16 + NW - 1) / NW;  // tiles per wave at most
    const int H = p.H, NT = p.NT, T = p.T, KS = (H + 63) / 64, HP = KS * 64, LDH = HP + 32;
    int8_t* hbuf = reinterpret_cast<int8_t*>(scan_smem);                       // [2][16][LDH]
    float* cst = reinterpret_cast<float*>(scan_smem + 2 * 16 * LDH);            // [3 + G][HP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    int s = 0;
    for (int i = 1; i < p.nseg; ++i)
        if ((int)blockIdx.x >= p.seg[i].tile0) s = i;
    const ScanSegDev sg = p.seg[s];
    const int R = sg.R, row0 = ((int)blockIdx.x - sg.tile0) * 16;
    const int rowc = (row0 + n < R) ? row0 + n : R - 1;
    const int ldz = G * H;
    for (int j = tid; j < HP; j += NW * 64) {
        const bool in = j < H;
        cst[0 * HP + j] = in ? sg.bias[H + j] - sg.bias[j] : 0.0f;
        cst[1 * HP + j] = in ? sg.bn_alpha[j] : 0.0f;
        cst[2 * HP + j] = in ? sg.bn_beta[j] : 0.0f;
#pragma unroll
        for (int g = 0; g < G; ++g) cst[(3 + g) * HP + j] = in ? sg.w_dq[g * H + j] : 0.0f;
    }
    for (int i = tid; i < 2 * 16 * LDH / 4; i += NW * 64) reinterpret_cast<int*>(hbuf)[i] = 0;
    __syncthreads();
    for (int idx = tid; idx < 16 * (H / 4); idx += NW * 64) {
        const int rr = idx / (H / 4), j4 = (idx - rr * (H / 4)) * 4;
        const int rsrc = (row0 + rr < R) ? row0 + rr : R - 1;
        const v4f h = *reinterpret_cast<const v4f*>(sg.h_state + (size_t)rsrc * H + j4);
        const unsigned pk = (h.x > 0.5f ? 1u : 0u) | (h.y > 0.5f ? 0x100u : 0u) | (h.z > 0.5f ? 0x10000u : 0u) |
                            (h.w > 0.5f ? 0x1000000u : 0u);
        *reinterpret_cast<unsigned*>(hbuf + rr * LDH + j4) = pk;
    }
    v4f c[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        const int ct = wave + NW * i;
        c[i] = ct < NT ? *reinterpret_cast<const v4f*>(sg.c_state + (size_t)rowc * H + ct * 16 + q * 4) : v4f{0, 0, 0, 0};
    }
    __syncthreads();
    const size_t plane = (size_t)G * NT * KS * 1024;
    unsigned cnt = 0;  // spikes stored by this lane (launches without an fp32 spike tensor: sg.count)
    for (int t = 0; t < T; ++t) {
        const int8_t* hc = hbuf + (t & 1) * 16 * LDH;
        int8_t* hn = hbuf + ((t & 1) ^ 1) * 16 * LDH;
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
            const int ct = wave + NW * i;
            if (ct >= NT) break;  // wave-uniform
            const int cc = ct * 16 + q * 4;
            v4f z[G];
#pragma unroll
            for (int g = 0; g < G; ++g) z[g] = *reinterpret_cast<const v4f*>(sg.zin + ((size_t)t * R + rowc) * ldz + g * H + cc);
            v4f pre[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                v4i a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
                const int8_t* wp = sg.w_hh + (((size_t)g * NT + ct) * KS * 64 + lane) * 16;
                for (int ks = 0; ks < KS; ++ks) {
                    const v4i b = *reinterpret_cast<const v4i*>(hc + n * LDH + ks * 64 + q * 16);
                    const v4i w0 = *reinterpret_cast<const v4i*>(wp + (size_t)ks * 1024);
                    const v4i w1 = *reinterpret_cast<const v4i*>(wp + (size_t)ks * 1024 + plane);
                    const v4i w2 = *reinterpret_cast<const v4i*>(wp + (size_t)ks * 1024 + 2 * plane);
                    a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, b, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1, b, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w2, b, a2, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) pre[g][r] = __builtin_fmaf(recombine3(a0[r], a1[r], a2[r]), cst[(3 + g) * HP + cc + r], z[g][r]);
            }
            v4f cy;
            unsigned pk = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pre_g = (G == 2) ? pre[G - 1][r] : pre[0][r] + cst[cc + r];
                const float f = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre[0][r] * -1.44269504088896341f));
                const float m = __builtin_fmaf(f, c[i][r] - pre_g, pre_g);
                const float y = __builtin_fmaf(m, cst[1 * HP + cc + r], cst[2 * HP + cc + r]);
                cy[r] = y;
                pk |= (y >= 0.0f) ? (1u << (8 * r)) : 0u;
            }
            c[i] = cy;
            *reinterpret_cast<unsigned*>(hn + n * LDH + cc) = pk;
            if (row0 + n < R) {  // rows past R are computed (clamped duplicates) but not stored
                *reinterpret_cast<unsigned*>(sg.spikes_i8 + ((size_t)t * R + rowc) * HP + cc) = pk;
                cnt += (unsigned)__builtin_popcount(pk);
                if (sg.spikes_f32) {
                    const v4f sp = {(float)(pk & 1u), (float)((pk >> 8) & 1u), (float)((pk >> 16) & 1u), (float)(pk >> 24)};
                    *reinterpret_cast<v4f*>(sg.spikes_f32 + ((size_t)t * R + rowc) * H + cc) = sp;
                }
                if (sg.membrane) *reinterpret_cast<v4f*>(sg.membrane + ((size_t)t * R + rowc) * H + cc) = cy;
            }
        }
        __syncthreads();
    }
    const int8_t* hl = hbuf + (T & 1) * 16 * LDH;
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        const int ct = wave + NW * i;
        if (ct >= NT || row0 + n >= R) continue;
        const int cc = ct * 16 + q * 4;
        *reinterpret_cast<v4f*>(sg.c_state + (size_t)rowc * H + cc) = c[i];
        const unsigned pk = *reinterpret_cast<const unsigned*>(hl + n * LDH + cc);
        const v4f h = {(float)(pk & 1u), (float)((pk >> 8) & 1u), (float)((pk >> 16) & 1u), (float)((pk >> 24) & 1u)};
        *reinterpret_cast<v4f*>(sg.h_state + (size_t)rowc * H + cc) = h;
    }
    if (!sg.spikes_f32) wave_count_add(sg.count, cnt);
}

// ---- separate gate weights that do not fit one CU, SPLIT over several (round 5: sfsn_gsn_layer_scan_split) ---------------------------
// gsn_scan_stream_kernel fetches all 614 KB of W_hh from the L2 every step: 10 us per step on the FOUR compute units the 64 rows of
// baseline_xl's full-band model occupy -- 20 of that model's 27 ms.  Here the neuron tiles of a 16-row block are dealt to NSPL
// workgroups; each keeps the rows of BOTH gates of its TPS tiles resident in LDS (a wave per tile: 2 gates x 3 digit planes x KS KB)
// and the workgroups of a row block exchange the new spikes through the L2 every step: one data-tagged 32-bit word per lane (four
// spikes + the step's epoch, ONE write-through store: data and "ready" arrive together, the scheme of the training kernels'
// tr_read_tagged), two slots by step parity; every workgroup polls the 16 x NT pieces of its rows (bounded: error word, all
// workgroups give up).  Same digit MFMAs in the same order and the same epilogue as the streamed kernel: bit-identical.
// The launch's workgroups must be co-resident (host: <= compute units).
__device__ __forceinline__ v4i split_load16_sc1(const void* ptr) {
    v4i v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(ptr) : "memory");
    return v;
}
__device__ __forceinline__ unsigned split_unpack4(unsigned w) { return (w & 1u) | ((w & 2u) << 7) | ((w & 4u) << 14) | ((w & 8u) << 21); }

template <int G>
__global__ __launch_bounds__(512) void gsn_scan_split_kernel(const ScanParams p, unsigned* __restrict__ scratch, const int TPS, const int NSPL) {
    extern __shared__ __attribute__((aligned(16))) char scan_smem[];
    const int H = p.H, NT = p.NT, T = p.T, KS = (H + 63) / 64, HP = KS * 64, LDH = HP + 32, H4 = H >> 2;
    const int tid = threadIdx.x, lane = tid & 63, nthr = TPS * 64;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const size_t wtile = (size_t)G * 3 * KS * 1024;                              // bytes of one tile's fragments (both gates, three planes)
    int8_t* wl = reinterpret_cast<int8_t*>(scan_smem) + (size_t)wave * wtile;    // this wave's tile: [G][3][KS][64][16]
    int8_t* hbuf = reinterpret_cast<int8_t*>(scan_smem) + (size_t)TPS * wtile;   // [2][16][LDH]
    float* cst = reinterpret_cast<float*>(hbuf + 2 * 16 * LDH);                  // [3 + G][HP]
    const ScanSegDev sg = p.seg[0];
    const int R = sg.R, rb = (int)blockIdx.x / NSPL, sp = (int)blockIdx.x - rb * NSPL, row0 = rb * 16;
    const int rowc = (row0 + n < R) ? row0 + n : R - 1;
    const int ldz = G * H;
    unsigned* err = scratch;                  // word 0: sticky error word (a wait expired)
    unsigned* xch = scratch + 16;             // [2][R][H / 4] tagged words
    const int ct = sp * TPS + wave;           // my neuron tile (wave-uniform)
    const bool have = ct < NT;
    for (int j = tid; j < HP; j += nthr) {
        const bool in = j < H;
        cst[0 * HP + j] = in ? sg.bias[H + j] - sg.bias[j] : 0.0f;
        cst[1 * HP + j] = in ? sg.bn_alpha[j] : 0.0f;
        cst[2 * HP + j] = in ? sg.bn_beta[j] : 0.0f;
#pragma unroll
        for (int g = 0; g < G; ++g) cst[(3 + g) * HP + j] = in ? sg.w_dq[g * H + j] : 0.0f;
    }
    for (int i = tid; i < 2 * 16 * LDH / 4; i += nthr) reinterpret_cast<int*>(hbuf)[i] = 0;
    const size_t plane = (size_t)G * NT * KS * 1024;
    if (have) {
#pragma unroll
        for (int g = 0; g < G; ++g)
            for (int pl = 0; pl < 3; ++pl)
                for (int ks = 0; ks < KS; ++ks)
                    *reinterpret_cast<v4i*>(wl + (((size_t)g * 3 + pl) * KS + ks) * 1024 + lane * 16) =
                        *reinterpret_cast<const v4i*>(sg.w_hh + (size_t)pl * plane + ((((size_t)g * NT + ct) * KS + ks) * 64 + lane) * 16);
    }
    __syncthreads();
    for (int idx = tid; idx < 16 * H4; idx += nthr) {
        const int rr = idx / H4, j4 = (idx - rr * H4) * 4;
        const int rsrc = (row0 + rr < R) ? row0 + rr : R - 1;
        const v4f h = *reinterpret_cast<const v4f*>(sg.h_state + (size_t)rsrc * H + j4);
        const unsigned pk = (h.x > 0.5f ? 1u : 0u) | (h.y > 0.5f ? 0x100u : 0u) | (h.z > 0.5f ? 0x10000u : 0u) |
                            (h.w > 0.5f ? 0x1000000u : 0u);
        *reinterpret_cast<unsigned*>(hbuf + rr * LDH + j4) = pk;
    }
    const int cc = ct * 16 + q * 4;
    v4f c = have ? *reinterpret_cast<const v4f*>(sg.c_state + (size_t)rowc * H + cc) : v4f{0, 0, 0, 0};
    __syncthreads();
    unsigned cnt = 0, last_pk = 0;
    for (int t = 0; t < T; ++t) {
        const int8_t* hc = hbuf + (t & 1) * 16 * LDH;
        int8_t* hn = hbuf + ((t & 1) ^ 1) * 16 * LDH;
        if (have) {
            v4f z[G];
#pragma unroll
            for (int g = 0; g < G; ++g) z[g] = *reinterpret_cast<const v4f*>(sg.zin + ((size_t)t * R + rowc) * ldz + g * H + cc);
            v4f pre[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                v4i a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
                const int8_t* wp = wl + (size_t)g * 3 * KS * 1024 + lane * 16;
                for (int ks = 0; ks < KS; ++ks) {
                    const v4i b = *reinterpret_cast<const v4i*>(hc + n * LDH + ks * 64 + q * 16);
                    const v4i w0 = *reinterpret_cast<const v4i*>(wp + (size_t)ks * 1024);
                    const v4i w1 = *reinterpret_cast<const v4i*>(wp + (size_t)(KS + ks) * 1024);
                    const v4i w2 = *reinterpret_cast<const v4i*>(wp + (size_t)(2 * KS + ks) * 1024);
                    a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, b, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1, b, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w2, b, a2, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) pre[g][r] = __builtin_fmaf(recombine3(a0[r], a1[r], a2[r]), cst[(3 + g) * HP + cc + r], z[g][r]);
            }
            v4f cy;
            unsigned pk = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pre_g = (G == 2) ? pre[G - 1][r] : pre[0][r] + cst[cc + r];
                const float f = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre[0][r] * -1.44269504088896341f));
                const float m = __builtin_fmaf(f, c[r] - pre_g, pre_g);
                const float y = __builtin_fmaf(m, cst[1 * HP + cc + r], cst[2 * HP + cc + r]);
                cy[r] = y;
                pk |= (y >= 0.0f) ? (1u << (8 * r)) : 0u;
            }
            c = cy;
            last_pk = pk;
            if (row0 + n < R) {  // rows past R are computed (clamped duplicates) but neither published nor stored
                if (t + 1 < T) {
                    const unsigned bits = (pk & 1u) | ((pk >> 7) & 2u) | ((pk >> 14) & 4u) | ((pk >> 21) & 8u);
                    __hip_atomic_store(xch + ((size_t)(t & 1) * R + rowc) * H4 + (cc >> 2), ((unsigned)(t + 1) << 4) | bits, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
                *reinterpret_cast<unsigned*>(sg.spikes_i8 + ((size_t)t * R + rowc) * HP + cc) = pk;
                cnt += (unsigned)__builtin_popcount(pk);
                if (sg.spikes_f32) {
                    const v4f sp4 = {(float)(pk & 1u), (float)((pk >> 8) & 1u), (float)((pk >> 16) & 1u), (float)(pk >> 24)};
                    *reinterpret_cast<v4f*>(sg.spikes_f32 + ((size_t)t * R + rowc) * H + cc) = sp4;
                }
                if (sg.membrane) *reinterpret_cast<v4f*>(sg.membrane + ((size_t)t * R + rowc) * H + cc) = cy;
            }
        }
        if (t + 1 < T) {
            // h_t of my 16 rows, every tile: a piece = the four tagged words of one (row, tile), written by one wave of one workgroup
            int good = 1;
            const unsigned epoch = (unsigned)(t + 1);
            for (int pc = tid; pc < 16 * NT; pc += nthr) {
                const int rr = pc / NT, tl = pc - rr * NT;
                const int rsrc = (row0 + rr < R) ? row0 + rr : R - 1;
                const unsigned* src = xch + ((size_t)(t & 1) * R + rsrc) * H4 + tl * 4;
                v4i v;
                for (unsigned spins = 0;; ++spins) {
                    v = split_load16_sc1(src);
                    if (((unsigned)v.x >> 4) == epoch && ((unsigned)v.y >> 4) == epoch && ((unsigned)v.z >> 4) == epoch && ((unsigned)v.w >> 4) == epoch) break;
                    if (spins > 400000u) {
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        good = 0;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                v4i o;
                o.x = (int)split_unpack4((unsigned)v.x); o.y = (int)split_unpack4((unsigned)v.y);
                o.z = (int)split_unpack4((unsigned)v.z); o.w = (int)split_unpack4((unsigned)v.w);
                *reinterpret_cast<v4i*>(hn + rr * LDH + tl * 16) = o;
            }
            if (!__syncthreads_and(good)) return;  // (the error word is set: the host raises; nobody waits for this workgroup's words for long)
        }
    }
    if (have && row0 + n < R) {
        *reinterpret_cast<v4f*>(sg.c_state + (size_t)rowc * H + cc) = c;
        const v4f h = {(float)(last_pk & 1u), (float)((last_pk >> 8) & 1u), (float)((last_pk >> 16) & 1u), (float)((last_pk >> 24) & 1u)};
        *reinterpret_cast<v4f*>(sg.h_state + (size_t)rowc * H + cc) = h;
    }
    if (!sg.spikes_f32) wave_count_add(sg.count, cnt);
}

// =====================================================================================================
// spike projection: y[m][n] = dq[n] * sum_k s[m][k] * Wq[n][k] (+ bias[n]);  s int8 0/1
// Waves are dealt (column-tile group cg, row-tile lane mw); each wave keeps its W tiles in registers and
// streams 16-row B fragments straight from global memory (16 B per lane, L1/L2 served for the sibling waves).
// =====================================================================================================
template <int TPW, int KS>
__global__ __launch_bounds__(512) void spike_proj_kernel(const int8_t* __restrict__ s, const int8_t* __restrict__ w,
                                                          const float* __restrict__ dq, const float* __restrict__ bias,
                                                          float* __restrict__ y, int M, int N, int ldy, int NT, int NWN) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int MW = 8 / NWN;
    const int cg = wave % NWN, mw = wave / NWN;
    if (mw >= MW) return;
    constexpr int KP = KS * 64;

    v4i W[TPW][KS][3];
    v4f dqv[TPW], bv[TPW];
    int col[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int ct = cg + NWN * i;
        const bool have = ct < NT;
        col[i] = have ? ct * 16 + q * 4 : -1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const size_t tile = (size_t)d * NT + (have ? ct : 0);
                W[i][ks][d] = *reinterpret_cast<const v4i*>(w + ((tile * KS + ks) * 64 + lane) * 16);
            }
        dqv[i] = *reinterpret_cast<const v4f*>(dq + (have ? col[i] : 0));  // dq is padded to NT*16
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[i][r] = (bias && have && col[i] + r < N) ? bias[col[i] + r] : 0.0f;
    }

    const int MT = (M + 15) >> 4;
    const bool vec = (ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15u) == 0;
    for (int mt = (int)blockIdx.x * MW + mw; mt < MT; mt += (int)gridDim.x * MW) {
        const int row = mt * 16 + n;
        const int rowc = row < M ? row : M - 1;
        v4i b[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b[ks] = *reinterpret_cast<const v4i*>(s + (size_t)rowc * KP + ks * 64 + q * 16);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            if (col[i] < 0) continue;
            v4i a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[i][ks][0], b[ks], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[i][ks][1], b[ks], a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[i][ks][2], b[ks], a2, 0, 0, 0);
            }
            v4f o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = recombine3(a0[r], a1[r], a2[r]) * dqv[i][r] + bv[i][r];
            if (row < M) {
                float* yp = y + (size_t)row * ldy + col[i];
                if (vec && col[i] + 3 < N) {
                    *reinterpret_cast<v4f*>(yp) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col[i] + r < N) yp[r] = o[r];
                }
            }
        }
    }
}

// =====================================================================================================
// input projection (real-valued x): z[m][n] = sum_k x[m][k] w[n][k], exact fp32 on v_mfma_f32_16x16x4_f32.
// k is consumed in chunks of 16: lane (n|m, q) holds k = 16c + 4q + e for MFMA e of chunk c -- the same
// map for A and B, so the pairing is correct and the B operand is four consecutive floats per lane.
// =====================================================================================================
template <int TPW, int KC>
__global__ __launch_bounds__(512) void input_proj_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ z, int M, int K, int N, int ldz, int NT,
                                                          int NWN) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int MW = 8 / NWN;
    const int cg = wave % NWN, mw = wave / NWN;
    if (mw >= MW) return;

    float W[TPW][KC][4];
    v4f bv[TPW];
    int col[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int ct = cg + NWN * i;
        const bool have = ct < NT;
        col[i] = have ? ct * 16 + q * 4 : -1;
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[i][r] = (bias && have && col[i] + r < N) ? bias[col[i] + r] : 0.0f;
        const int wr = ct * 16 + n;  // weight row this lane supplies as A
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = c * 16 + q * 4 + e;
                W[i][c][e] = (have && wr < N && k < K) ? w[(size_t)wr * K + k] : 0.0f;
            }
    }

    const int MT = (M + 15) >> 4;
    for (int mt = (int)blockIdx.x * MW + mw; mt < MT; mt += (int)gridDim.x * MW) {
        const int row = mt * 16 + n;
        const int rowc = row < M ? row : M - 1;
        const float* xp = x + (size_t)rowc * K;
        float b[KC][4];
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = c * 16 + q * 4 + e;
                b[c][e] = k < K ? xp[k] : 0.0f;
            }
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            if (col[i] < 0) continue;
            v4f acc = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(W[i][c][e], b[c][e], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += bv[i][r];
            if (row < M) {
                float* zp = z + (size_t)row * ldz + col[i];
                if ((ldz & 3) == 0 && col[i] + 3 < N) {
                    *reinterpret_cast<v4f*>(zp) = acc;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col[i] + r < N) zp[r] = acc[r];
                }
            }
        }
    }
}

// =====================================================================================================
// Fast paths of the two time-parallel products (N % 4 == 0, ldy % 4 == 0): a workgroup takes 64 rows at a time,
// every wave keeps its W tiles in registers, the accumulator fragments are staged through LDS and leave as whole,
// contiguous rows (full cache lines).  The direct fragment store of the generic kernels writes 64 B per row per tile
// -- partial-line writes, measured ~3x slower on the 745 MB sub-band input term.
// =====================================================================================================
#define GEMM_MB 4  // 16-row tiles per workgroup iteration

// Both kernels are software pipelined over 64-row super tiles: the next tile's left operand is requested from HBM
// (coalesced, one pass, into registers) before the current tile is computed, and parked in the second LDS buffer
// after it -- the HBM latency hides under the MFMA phase, and no wave re-reads what another already fetched.
// (blk, nblk): this workgroup's index and the number of workgroups that share the job -- blockIdx.x / gridDim.x for a launch of
// one product, a block range of the launch for several products side by side (proj_multi_kernel)
template <int TPW, int KS>
__device__ __forceinline__ void spike_proj_fast_body(const int8_t* __restrict__ s, const int8_t* __restrict__ w,
                                                     const float* __restrict__ dq, const float* __restrict__ bias,
                                                     float* __restrict__ y, int M, int N, int ldy, int NT, int NWN, int blk, int nblk,
                                                     char* gemm_smem_c) {
    constexpr int KP = KS * 64;
    constexpr int SROW = KP + 16;                     // +16 B: the 16 rows of a B fragment hit distinct banks
    constexpr int SBUF = 64 * SROW;                   // one spike tile
    constexpr int NV = (64 * KP / 16 + 511) / 512;    // 16-byte vectors per thread per tile
    int8_t* sbuf = reinterpret_cast<int8_t*>(gemm_smem_c);            // [2][64][SROW]
    float* obuf = reinterpret_cast<float*>(gemm_smem_c + 2 * SBUF);   // [64][N + 4]
    const int NP = N + 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int MW = 8 / NWN;
    const int cg = wave % NWN, mw = wave / NWN;
    const bool worker = mw < MW;

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
                W[i][ks][d] = *reinterpret_cast<const v4i*>(w + ((tile * KS + ks) * 64 + lane) * 16);
            }
        dqv[i] = *reinterpret_cast<const v4f*>(dq + (have ? col[i] : 0));
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[i][r] = (bias && have && col[i] + r < N) ? bias[col[i] + r] : 0.0f;
    }

    const int NS = (M + 63) >> 6;  // 64-row super tiles
    const int n4 = N >> 2;
    v4i pre[NV];
    auto fetch = [&](int st) __attribute__((always_inline)) {  // rows are contiguous: the tile is one 64*KP-byte block
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int v = tid + j * 512;
            size_t byte = (size_t)st * 64 * KP + (size_t)v * 16;
            const size_t last = (size_t)M * KP - 16;
            if (byte > last) byte = last;  // clamp the ragged last tile (those rows are never stored)
            if (v < 64 * KP / 16) pre[j] = *reinterpret_cast<const v4i*>(s + byte);
        }
    };
    auto park = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int v = tid + j * 512;
            if (v < 64 * KP / 16) {
                const int r = v / (KP / 16), c16 = v - r * (KP / 16);
                *reinterpret_cast<v4i*>(sbuf + buf * SBUF + r * SROW + c16 * 16) = pre[j];
            }
        }
    };
    int cur = 0;
    if (blk < NS) {
        fetch(blk);
        park(0);
    }
    __syncthreads();
    for (int st = blk; st < NS; st += nblk) {
        const int m0 = st * 64;
        const int nxt = st + nblk;
        if (nxt < NS) fetch(nxt);
        if (worker) {
            for (int mi = mw; mi < GEMM_MB; mi += MW) {
                const int8_t* sr = sbuf + cur * SBUF + (mi * 16 + n) * SROW + q * 16;
                v4i b[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) b[ks] = *reinterpret_cast<const v4i*>(sr + ks * 64);
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    if (col[i] < 0) continue;
                    v4i a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[i][ks][0], b[ks], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[i][ks][1], b[ks], a1, 0, 0, 0);
                        a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[i][ks][2], b[ks], a2, 0, 0, 0);
                    }
                    v4f o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = recombine3(a0[r], a1[r], a2[r]) * dqv[i][r] + bv[i][r];
                    if (col[i] + 3 < N) {
                        *reinterpret_cast<v4f*>(&obuf[(mi * 16 + n) * NP + col[i]]) = o;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (col[i] + r < N) obuf[(mi * 16 + n) * NP + col[i] + r] = o[r];
                    }
                }
            }
        }
        // Park the prefetched tile BEFORE issuing this tile's stores: the wait for the prefetch then covers only loads
        // (vmcnt retires in order -- parking after the store loop drained every store of the tile first, each iteration).
        if (nxt < NS) park(cur ^ 1);
        cur ^= 1;
        __syncthreads();
        const int rows = (M - m0 < 64) ? M - m0 : 64;
        for (int idx = tid; idx < rows * n4; idx += 512) {
            const int r = idx / n4, c4 = idx - r * n4;
            *reinterpret_cast<v4f*>(y + (size_t)(m0 + r) * ldy + c4 * 4) = *reinterpret_cast<const v4f*>(&obuf[r * NP + c4 * 4]);
        }
        // Only the LDS reads above have to be done before the next tile rewrites obuf: a raw barrier behind lgkmcnt(0).  (Round 5:
        // __syncthreads() is a workgroup-scope release -- hipcc puts s_waitcnt vmcnt(0) in front of it, so every tile's row stores had
        // to RETIRE in HBM before the next tile's prefetch was even issued: 7-8 us per 64-row tile against ~2.5 us of work.)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    }
}

template <int TPW, int KS>
__global__ __launch_bounds__(512) void spike_proj_fast_kernel(const int8_t* __restrict__ s, const int8_t* __restrict__ w,
                                                               const float* __restrict__ dq, const float* __restrict__ bias,
                                                               float* __restrict__ y, int M, int N, int ldy, int NT, int NWN) {
    extern __shared__ __attribute__((aligned(16))) char gemm_smem_c[];
    spike_proj_fast_body<TPW, KS>(s, w, dq, bias, y, M, N, ldy, NT, NWN, (int)blockIdx.x, (int)gridDim.x, gemm_smem_c);
}

// Several spike products of the same K in ONE launch (round 5): the projections of the sub-band groups of a chunk are independent
// (MODEL:118 per sequence model) and each alone leaves compute units idle or pays a launch boundary (three launches of 15-25 us with
// ~8 us between them per 380-frame chunk).  Every workgroup takes the job whose block range it falls into and runs that job's OWN
// tiling (the body of its single launch, instruction for instruction: same results).
struct ProjJobDev {
    const int8_t* s;
    const int8_t* w;
    const float* dq;
    const float* bias;
    float* y;
    int M, N, ldy, NT, NWN, tpw, block0, nblocks;
};
struct ProjMultiParams {
    ProjJobDev job[SFSN_MAX_SEGMENTS];
    int n;
};
template <int KS>
__global__ __launch_bounds__(512) void spike_proj_multi_kernel(const ProjMultiParams p) {
    extern __shared__ __attribute__((aligned(16))) char gemm_smem_c[];
    int j = 0;
    for (int i = 1; i < p.n; ++i)
        if ((int)blockIdx.x >= p.job[i].block0) j = i;
    const ProjJobDev& b = p.job[j];
    const int blk = (int)blockIdx.x - b.block0;
    if (b.tpw == 1) spike_proj_fast_body<1, KS>(b.s, b.w, b.dq, b.bias, b.y, b.M, b.N, b.ldy, b.NT, b.NWN, blk, b.nblocks, gemm_smem_c);
    else spike_proj_fast_body<2, KS>(b.s, b.w, b.dq, b.bias, b.y, b.M, b.N, b.ldy, b.NT, b.NWN, blk, b.nblocks, gemm_smem_c);
}

template <int TPW, int KC>
__global__ __launch_bounds__(512) void input_proj_fast_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ z, int M, int K, int N,
                                                               int ldz, int NT, int NWN) {
    extern __shared__ __attribute__((aligned(16))) float gemm_smem[];
    constexpr int KQ = KC * 16;        // padded K
    constexpr int KPAD = KQ + 4;       // +4 floats: the 16 rows of a B fragment (ds_read_b128) land on distinct banks
    constexpr int XBUF = 64 * KPAD;
    constexpr int NV = (64 * KQ + 511) / 512;  // floats per thread per tile
    float* xbuf = gemm_smem;               // [2][64][KPAD]
    float* obuf = gemm_smem + 2 * XBUF;    // [64][N + 4]
    const int NP = N + 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int MW = 8 / NWN;
    const int cg = wave % NWN, mw = wave / NWN;
    const bool worker = mw < MW;

    float W[TPW][KC][4];
    v4f bv[TPW];
    int col[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int ct = cg + NWN * i;
        const bool have = worker && ct < NT;
        col[i] = have ? ct * 16 + q * 4 : -1;
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[i][r] = (bias && have && col[i] + r < N) ? bias[col[i] + r] : 0.0f;
        const int wr = ct * 16 + n;
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = c * 16 + q * 4 + e;
                W[i][c][e] = (have && wr < N && k < K) ? w[(size_t)wr * K + k] : 0.0f;
            }
    }
    const int NS = (M + 63) >> 6;
    const int n4 = N >> 2;
    float pre[NV];
    int pr[NV], pk[NV];  // (row, k) of my elements within a tile: fixed
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int e = tid + j * 512;
        pr[j] = e / KQ;
        pk[j] = e - pr[j] * KQ;
    }
    auto fetch = [&](int st) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            int row = st * 64 + pr[j];
            if (row > M - 1) row = M - 1;
            pre[j] = (pr[j] < 64 && pk[j] < K) ? x[(size_t)row * K + pk[j]] : 0.0f;
        }
    };
    auto park = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (pr[j] < 64) xbuf[buf * XBUF + pr[j] * KPAD + pk[j]] = pre[j];
    };
    int cur = 0;
    if ((int)blockIdx.x < NS) {
        fetch(blockIdx.x);
        park(0);
    }
    __syncthreads();
    for (int st = blockIdx.x; st < NS; st += gridDim.x) {
        const int m0 = st * 64;
        const int nxt = st + gridDim.x;
        if (nxt < NS) fetch(nxt);
        if (worker) {
            for (int mi = mw; mi < GEMM_MB; mi += MW) {
                v4f b[KC];
#pragma unroll
                for (int c = 0; c < KC; ++c) b[c] = *reinterpret_cast<const v4f*>(&xbuf[cur * XBUF + (mi * 16 + n) * KPAD + c * 16 + q * 4]);
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    if (col[i] < 0) continue;
                    v4f acc = {0, 0, 0, 0};
#pragma unroll
                    for (int c = 0; c < KC; ++c)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(W[i][c][e], b[c][e], acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] += bv[i][r];
                    if (col[i] + 3 < N) {
                        *reinterpret_cast<v4f*>(&obuf[(mi * 16 + n) * NP + col[i]]) = acc;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (col[i] + r < N) obuf[(mi * 16 + n) * NP + col[i] + r] = acc[r];
                    }
                }
            }
        }
        // Park the prefetched tile BEFORE issuing this tile's stores: the wait for the prefetch then covers only loads
        // (vmcnt retires in order -- parking after the store loop drained every store of the tile first, each iteration).
        if (nxt < NS) park(cur ^ 1);
        cur ^= 1;
        __syncthreads();
        const int rows = (M - m0 < 64) ? M - m0 : 64;
        for (int idx = tid; idx < rows * n4; idx += 512) {
            const int r = idx / n4, c4 = idx - r * n4;
            *reinterpret_cast<v4f*>(z + (size_t)(m0 + r) * ldz + c4 * 4) = *reinterpret_cast<const v4f*>(&obuf[r * NP + c4 * 4]);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // (LDS reads of obuf done; the row stores stay in flight: see spike_proj_fast_body)
        __builtin_amdgcn_s_barrier();
    }
}

// ---- layer-0 input product on the bf16 matrix cores, fp32-accurate (3-way split) ------------------------------------
// The fp32 MFMA (v_mfma_f32_16x16x4_f32, 256 FLOP/clk/CU) bounds input_proj_fast_kernel: 65 GFLOP per forward at B=64,
// T=1000 is 0.41 ms of matrix-core time.  v_mfma_f32_16x16x32_bf16 is 16x faster per FLOP.  Every fp32 value splits
// (round-to-nearest, v_cvt_pk_bf16_f32) into three bf16 pieces v = v1 + v2 + v3 with |v2| <= 2^-9 |v|, |v3| <= 2^-18 |v|
// (the last residual has at most 8 significant bits and is exact).  Products of pieces are exact in fp32; the six products
// x1w1, x1w2, x2w1, x1w3, x2w2, x3w1 carry x.w to 2^-26 |x||w| per term -- below half an fp32 ulp of the term -- and the
// matrix core accumulates them in fp32.  The five small products go to their own accumulator (not swamped by the large
// one; two independent MFMA chains) and are added once at the end.  6 MFMAs of 16 clk per 32 k instead of 8 of 32 clk.
// (bf8 / split3: sfsn_scan_dev.h -- shared with the FUSEDX3 role of the stack launch)

typedef float v4fu __attribute__((ext_vector_type(4), aligned(4)));  // a 16-byte load from a dword-aligned address

#ifdef IP_STAMPS  // scripts/micro/inproj_stamps.sh: shader-clock sums per phase as wave 0 of a launch's first workgroup sees them
__device__ unsigned long long ip_dbg[8];
#define IP_T(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); dbg_[k] += now_ - last_; last_ = now_; } while (0)
#else
#define IP_T(k) do {} while (0)
#endif

template <int TPW, int KS>
__device__ __forceinline__ void input_proj_bf3_body(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ z, int M, int K, int N,
                                                    int ldz, int NT, int NWN, int blk, int nblk, float* gemm_smem) {
#ifdef IP_STAMPS
    unsigned long long dbg_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
    constexpr int KQ = KS * 32;          // padded K
    constexpr int LDX = KQ + 8;          // bf16 elements per row: row stride = 16 B * odd -> conflict-free ds_read_b128
    constexpr int PLANE = 64 * LDX / 2;  // dwords per piece plane
    constexpr int NVP = (64 * KQ / 2 + 511) / 512;  // k-pairs per thread per tile
    unsigned* xb = reinterpret_cast<unsigned*>(gemm_smem);  // [3][64][LDX] bf16
    float* obuf = gemm_smem + 3 * PLANE;                    // [64][N + 4]
    const int NP = N + 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int MW = 8 / NWN;
    const int cg = wave % NWN, mw = wave / NWN;
    const bool worker = mw < MW;

    bf8 W[TPW][KS][3];
    v4f bv[TPW];
    int col[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int ct = cg + NWN * i;
        const bool have = worker && ct < NT;
        col[i] = have ? ct * 16 + q * 4 : -1;
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[i][r] = (bias && have && col[i] + r < N) ? bias[col[i] + r] : 0.0f;
        const int wr = ct * 16 + n;  // A fragment: lane holds 8 consecutive k of weight row wr
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            unsigned pw[3][4];
            // a k-step that lies wholly inside K (wave-uniform): the lane's eight weights as two 16-byte loads from a clamped row (K is
            // even and w 16-byte aligned: rows are 8-byte aligned, which a global dwordx4 load takes) instead of eight 4-byte loads --
            // the workgroup's prologue was 80 scalar loads per lane, 11.6 of a 43 us launch (scripts/micro/inproj_stamps.sh)
            float wv[8];
            if (ks * 32 + 32 <= K) {
                const int wrc = wr < N ? wr : N - 1;
                const v4fu* src = reinterpret_cast<const v4fu*>(w + (size_t)wrc * K + ks * 32 + q * 8);
                const v4fu lo4 = src[0], hi4 = src[1];
                const bool live = have && wr < N;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    wv[e] = live ? lo4[e] : 0.0f;
                    wv[4 + e] = live ? hi4[e] : 0.0f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = ks * 32 + q * 8 + e;
                    wv[e] = (have && wr < N && k < K) ? w[(size_t)wr * K + k] : 0.0f;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) split3(wv[2 * e], wv[2 * e + 1], pw[0][e], pw[1][e], pw[2][e]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) W[i][ks][pl] = *reinterpret_cast<const bf8*>(pw[pl]);
        }
    }
    const int NS = (M + 63) >> 6;
    const int n4 = N >> 2;
    v2f pre[NVP];
    int pr[NVP], pk[NVP];  // (row, k) of my k-pairs within a tile: fixed
#pragma unroll
    for (int j = 0; j < NVP; ++j) {
        const int e = tid + j * 512;
        pr[j] = e / (KQ / 2);
        pk[j] = (e - pr[j] * (KQ / 2)) * 2;
    }
    auto fetch = [&](int st) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NVP; ++j) {
            int row = st * 64 + pr[j];
            if (row > M - 1) row = M - 1;
            const v2f zero = {0.0f, 0.0f};
            pre[j] = (pr[j] < 64 && pk[j] < K) ? *reinterpret_cast<const v2f*>(x + (size_t)row * K + pk[j]) : zero;  // K is even
        }
    };
    auto park = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NVP; ++j)
            if (pr[j] < 64) {
                unsigned p1, p2, p3;
                split3(pre[j][0], pre[j][1], p1, p2, p3);
                const int o = (pr[j] * LDX + pk[j]) >> 1;
                xb[o] = p1;
                xb[PLANE + o] = p2;
                xb[2 * PLANE + o] = p3;
            }
    };
    if (blk < NS) {
        fetch(blk);
        park();
    }
    __syncthreads();
    IP_T(0);
    for (int st = blk; st < NS; st += nblk) {
        const int m0 = st * 64;
        const int nxt = st + nblk;
        if (nxt < NS) fetch(nxt);
        if (worker) {
            // Two row blocks at a time and NO branch around an absent column tile (its W pieces are zero, its result is not stored:
            // the waves that lack one would wait at the barrier anyway): 2 x TPW independent accumulator chains side by side.  Every
            // accumulator still sees its six products per k-step in the same order -- the same bits.  (Round 5: with the `continue`
            // the chains of the two column tiles sat in separate exec-masked blocks, and one tile's `lo` chain is five dependent
            // matrix instructions per k-step: the product phase ran at 1.8 x its issue time, scripts/micro/inproj_stamps.sh.)
            constexpr int MP = (TPW <= 2 && TPW * KS <= 6) ? 2 : 1;  // row blocks side by side where the registers allow it (W pieces: 12 TPW KS)
            for (int mi0 = mw; mi0 < GEMM_MB; mi0 += MP * MW) {
                const bool two = MP == 2 && mi0 + MW < GEMM_MB;  // wave-uniform
                const int mis[2] = {mi0, two ? mi0 + MW : mi0};
                v4f hi[MP][TPW], lo[MP][TPW];
#pragma unroll
                for (int m = 0; m < MP; ++m)
#pragma unroll
                    for (int i = 0; i < TPW; ++i) hi[m][i] = lo[m][i] = v4f{0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    bf8 b1[MP], b2[MP], b3[MP];
#pragma unroll
                    for (int m = 0; m < MP; ++m) {
                        const unsigned* src = xb + (((mis[m] * 16 + n) * LDX + ks * 32 + q * 8) >> 1);
                        b1[m] = *reinterpret_cast<const bf8*>(src);
                        b2[m] = *reinterpret_cast<const bf8*>(src + PLANE);
                        b3[m] = *reinterpret_cast<const bf8*>(src + 2 * PLANE);
                    }
#define IPB_STEP(ACC, PL, B)                                                                                          \
    _Pragma("unroll") for (int m = 0; m < MP; ++m) _Pragma("unroll") for (int i = 0; i < TPW; ++i)                    \
        ACC[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[i][ks][PL], B[m], ACC[m][i], 0, 0, 0);
                    IPB_STEP(lo, 2, b1)
                    IPB_STEP(hi, 0, b1)
                    IPB_STEP(lo, 1, b2)
                    IPB_STEP(lo, 0, b3)
                    IPB_STEP(lo, 1, b1)
                    IPB_STEP(lo, 0, b2)
#undef IPB_STEP
                }
#pragma unroll
                for (int m = 0; m < MP; ++m) {
                    if (m == 1 && !two) break;
                    const int mi = mis[m];
#pragma unroll
                    for (int i = 0; i < TPW; ++i) {
                        if (col[i] < 0) continue;
                        v4f acc;
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r] = (hi[m][i][r] + lo[m][i][r]) + bv[i][r];
                        if (col[i] + 3 < N) {
                            *reinterpret_cast<v4f*>(&obuf[(mi * 16 + n) * NP + col[i]]) = acc;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (col[i] + r < N) obuf[(mi * 16 + n) * NP + col[i] + r] = acc[r];
                        }
                    }
                }
            }
        }
        IP_T(1);
        __syncthreads();  // every wave is done with the x pieces of this tile; obuf is complete
        IP_T(2);
        // park the prefetched tile BEFORE issuing this tile's stores (vmcnt retires in order: the wait for the prefetch
        // would otherwise also wait for the stores)
        if (nxt < NS) park();
        IP_T(3);
        const int rows = (M - m0 < 64) ? M - m0 : 64;
        for (int idx = tid; idx < rows * n4; idx += 512) {
            const int r = idx / n4, c4 = idx - r * n4;
            *reinterpret_cast<v4f*>(z + (size_t)(m0 + r) * ldz + c4 * 4) = *reinterpret_cast<const v4f*>(&obuf[r * NP + c4 * 4]);
        }
        IP_T(4);
        __builtin_amdgcn_s_waitcnt(0xc07f);  // (LDS reads of obuf done; the row stores stay in flight: see spike_proj_fast_body)
        __builtin_amdgcn_s_barrier();
        IP_T(5);
    }
#ifdef IP_STAMPS
    if (blk == 0 && threadIdx.x == 0)
        for (int k = 0; k < 8; ++k) ip_dbg[k] = dbg_[k];
#endif
}

template <int TPW, int KS>
__global__ __launch_bounds__(512) void input_proj_bf3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ z, int M, int K, int N,
                                                              int ldz, int NT, int NWN) {
    extern __shared__ __attribute__((aligned(16))) float gemm_smem[];
    input_proj_bf3_body<TPW, KS>(x, w, bias, z, M, K, N, ldz, NT, NWN, (int)blockIdx.x, (int)gridDim.x, gemm_smem);
}

// Several real-valued input products in ONE launch (the layer-0 products of the sub-band groups whose feature rows are too wide for
// the in-scan form: two launches per chunk at baseline_m).  As spike_proj_multi_kernel: a block range per job, the job's own tiling.
struct InProjJobDev {
    const float* x;
    const float* w;
    const float* bias;
    float* z;
    int M, K, N, ldz, NT, NWN, tpw, ksb, block0, nblocks;
};
struct InProjMultiParams {
    InProjJobDev job[SFSN_MAX_SEGMENTS];
    int n;
};
__global__ __launch_bounds__(512) void input_proj_multi_kernel(const InProjMultiParams p) {
    extern __shared__ __attribute__((aligned(16))) float gemm_smem[];
    int j = 0;
    for (int i = 1; i < p.n; ++i)
        if ((int)blockIdx.x >= p.job[i].block0) j = i;
    const InProjJobDev& b = p.job[j];
    const int blk = (int)blockIdx.x - b.block0;
#define IPM_CASE(TPW_, KS_)                                                                                                               \
    if (b.tpw == TPW_ && b.ksb == KS_) {                                                                                                  \
        input_proj_bf3_body<TPW_, KS_>(b.x, b.w, b.bias, b.z, b.M, b.K, b.N, b.ldz, b.NT, b.NWN, blk, b.nblocks, gemm_smem);              \
        return;                                                                                                                           \
    }
    IPM_CASE(1, 2) IPM_CASE(1, 3) IPM_CASE(1, 5) IPM_CASE(1, 6) IPM_CASE(2, 2) IPM_CASE(2, 3) IPM_CASE(2, 5)
#undef IPM_CASE
}

// ---- fused real-valued input scan (layer 0 of a group with narrow feature rows: baseline_m group 0, 8 units x 38) ----------------
// The layer-0 twin of gsn_scan_fused_kernel: the input term x.W_ih^T + b is computed inside the scan, here from the fp32
// feature rows with the bf16 3-way split of input_proj_bf3_kernel (same six products per 32 k, same two accumulators, same
// final (hi + lo) + b: bit-identical to sfsn_input_proj_f32 + sfsn_gsn_layer_scan).  W_ih pieces live in registers (I <= 64:
// 2 tiles x 2 k-steps x 3 pieces x 4 VGPRs), W_hh as in the fused-input scan (plane 0 in LDS, 1-2 in registers).  The 16
// feature rows of a step are one contiguous block (16 x I floats): it arrives by LDS-DMA, and the wave that fetched a piece
// converts that same piece -- after its own wait, before the step barrier -- into the three bf16 planes the B fragments
// are read from (double-buffered), so no extra rendezvous is needed.  Needs R % 16 == 0 (a block is copied flat).
template <int KS, int OUT, int NTL>
__device__ __forceinline__ void fusedx_body(const ScanSegDev& sg, char* smem, int T, int H, int NT, int R, int row0, int rowc, int n,
                                            int q, int tid, int wave) {
    using C = ScanCfg<1, KS, 8, 2, OUT, 0>;
    constexpr int LDH = C::LDH, HP = C::HP, D = 3, NW = 8, KSB = 2, LDX = KSB * 32 + 8;
    const int I = sg.I;
    const int xbytes = 16 * I * 4;                              // one step's feature block
    const int XSLOT = (xbytes + 1023) & ~1023;                  // ring slot (whole 1 KB DMA pieces)
    constexpr int PL_BYTES = 16 * LDX * 2;                      // one bf16 plane of 16 rows
    const int PLANES_OFF = D * XSLOT, HBUF_OFF = PLANES_OFF + 2 * 3 * PL_BYTES;
    const int CST_OFF = HBUF_OFF + 2 * 16 * LDH, WHH_OFF = CST_OFF + 5 * HP * 4;
    int8_t* hbuf = reinterpret_cast<int8_t*>(smem + HBUF_OFF);
    const float(*cst)[HP] = reinterpret_cast<const float(*)[HP]>(smem + CST_OFF);  // b_f, b_g - b_f, alpha, beta, dq_hh
    ScanFlush<C> fl;
    fl.init(tid, row0, R, H, NW * 64, 16);
    const int lane = tid & 63;
    constexpr int CBASE = (D - 2) * 1, CSTRIDE = (D - 1) * C::NSTF;

    v4i Whh[NTL][KS][2];
    bf8 Wx[NTL][KSB][3];
    v4f c[NTL];
    int col[NTL];
    unsigned wl_off[NTL];
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        const int ct = wave + NW * i;
        col[i] = ct * 16 + q * 4;
        wl_off[i] = (unsigned)((ct * KS) * 64 + lane) * 16u;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const size_t tile = (size_t)(d + 1) * NT + ct;
                Whh[i][ks][d] = *reinterpret_cast<const v4i*>(sg.w_hh + ((tile * KS + ks) * 64 + lane) * 16);
            }
        const int wr = ct * 16 + n;  // A fragment: lane holds 8 consecutive k of weight row wr (as input_proj_bf3_kernel)
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks) {
            unsigned pw[3][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = ks * 32 + q * 8 + 2 * e;
                const float a = (wr < H && k < I) ? sg.w_ih_f32[(size_t)wr * I + k] : 0.0f;
                const float b = (wr < H && k + 1 < I) ? sg.w_ih_f32[(size_t)wr * I + k + 1] : 0.0f;
                split3(a, b, pw[0][e], pw[1][e], pw[2][e]);
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) Wx[i][ks][pl] = *reinterpret_cast<const bf8*>(pw[pl]);
        }
        c[i] = *reinterpret_cast<const v4f*>(sg.c_state + (size_t)rowc * H + col[i]);
    }
    // feature ring: DMA piece p = 64 16-byte chunks of the step's flat block; the wave that fetches piece p also converts it
    const int nchunk = 4 * I;                                  // 16 * I floats / 4
    const int npiece = (nchunk + 63) >> 6;
    const int piece = (wave < npiece) ? wave : npiece - 1;     // surplus waves repeat the last piece (same bytes, same writes)
    int chunk = piece * 64 + lane;
    if (chunk > nchunk - 1) chunk = nchunk - 1;                // surplus lanes repeat the last chunk
    const unsigned src_off = (unsigned)chunk * 16u;
    const size_t frame = (size_t)R * I;                        // floats per step of this segment
    const float* xbase = sg.x_in + (size_t)row0 * I;
    // my four floats of the block: flat index 4 chunk + j -> (row, k) -> bf16 index in a plane
    int poff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = 4 * chunk + j, rr = f / I, k = f - rr * I;
        poff[j] = rr * LDX + k;
    }
    // the DMA of a surplus lane lands at its OWN lds slot (piece * 1024 + lane * 16) but carries the clamped chunk: read it there
    const int my_lds = piece * 1024 + lane * 16;
    auto issue = [&](int slot, int td) __attribute__((always_inline)) {
        dma16_to_lds(__builtin_amdgcn_readfirstlane((unsigned)(slot * XSLOT + piece * 1024)), xbase + (size_t)td * frame, src_off);
    };
    auto convert = [&](int slot, int buf) __attribute__((always_inline)) {  // my piece of ring slot -> bf16 planes[buf]
        const v4f v = *reinterpret_cast<const v4f*>(smem + slot * XSLOT + my_lds);
        unsigned short* pl = reinterpret_cast<unsigned short*>(smem + PLANES_OFF + buf * 3 * PL_BYTES);
        unsigned p1[2], p2[2], p3[2];
        split3(v[0], v[1], p1[0], p2[0], p3[0]);
        split3(v[2], v[3], p1[1], p2[1], p3[1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sh = (j & 1) * 16;
            pl[poff[j]] = (unsigned short)(p1[j >> 1] >> sh);
            pl[PL_BYTES / 2 + poff[j]] = (unsigned short)(p2[j >> 1] >> sh);
            pl[2 * (PL_BYTES / 2) + poff[j]] = (unsigned short)(p3[j >> 1] >> sh);
        }
    };
    for (int s0 = 0; s0 < D - 1; ++s0) issue(s0, s0 < T ? s0 : (T > 0 ? T - 1 : 0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);
    convert(0, 0);  // step 0's features (my piece; the barrier below publishes everybody's)

    auto step = [&](int t, auto first) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first)::value;
        const int8_t* hc = hbuf + (t & 1) * 16 * LDH;
        int8_t* hn = hbuf + ((t & 1) ^ 1) * 16 * LDH;
        {
            const int td = (t + D - 1 < T) ? t + D - 1 : T - 1;
            issue((t + D - 1) % D, td);
        }
        v4i bh[KS];
        bf8 bx[KSB][3];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bh[ks] = *reinterpret_cast<const v4i*>(hc + n * LDH + ks * 64 + q * 16);
        const char* pbase = smem + PLANES_OFF + (t & 1) * 3 * PL_BYTES + (n * LDX + q * 8) * 2;
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bx[ks][pl] = *reinterpret_cast<const bf8*>(pbase + pl * PL_BYTES + ks * 64);
#pragma unroll
        for (int i = 0; i < NTL; ++i) {
            const int cc = col[i];
            v4i a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v4i w0 = *reinterpret_cast<const v4i*>(smem + WHH_OFF + wl_off[i] + (unsigned)(ks * 1024));
                a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, bh[ks], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Whh[i][ks][0], bh[ks], a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Whh[i][ks][1], bh[ks], a2, 0, 0, 0);
            }
            if (i == 0) {
                if constexpr (!FIRST) fl.template run<OUT>(hc, sg.spikes_f32, sg.spikes_i8, t - 1, R, H);
            }
            v4f hi = {0, 0, 0, 0}, lo = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KSB; ++ks) {  // the product sequence of input_proj_bf3_kernel, instruction for instruction
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wx[i][ks][2], bx[ks][0], lo, 0, 0, 0);
                hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wx[i][ks][0], bx[ks][0], hi, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wx[i][ks][1], bx[ks][1], lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wx[i][ks][0], bx[ks][2], lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wx[i][ks][1], bx[ks][0], lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wx[i][ks][0], bx[ks][1], lo, 0, 0, 0);
            }
            const v4f bf = *reinterpret_cast<const v4f*>(&cst[0][cc]), db = *reinterpret_cast<const v4f*>(&cst[1][cc]);
            const v4f alpha = *reinterpret_cast<const v4f*>(&cst[2][cc]), beta = *reinterpret_cast<const v4f*>(&cst[3][cc]);
            const v4f dqh = *reinterpret_cast<const v4f*>(&cst[4][cc]);
            v4f cy;
            unsigned pk = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = (hi[r] + lo[r]) + bf[r];  // = input_proj_bf3_kernel's epilogue
                const float pre_f = __builtin_fmaf(recombine3(a0[r], a1[r], a2[r]), dqh[r], z);
                const float pre_g = pre_f + db[r];
                const float f = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre_f * -1.44269504088896341f));
                const float m = __builtin_fmaf(f, c[i][r] - pre_g, pre_g);
                const float y = __builtin_fmaf(m, alpha[r], beta[r]);
                cy[r] = y;
                pk |= (y >= 0.0f) ? (1u << (8 * r)) : 0u;
            }
            c[i] = cy;
            *reinterpret_cast<unsigned*>(hn + n * LDH + cc) = pk;
        }
        // my piece of step t+1's features has landed after this wait; convert it into the other plane buffer (nobody reads
        // that one during step t); the barrier publishes the planes and the new hidden state together
        if (t < D || CBASE + fl.nact * CSTRIDE > 63) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            wait_vmcnt_affine<CBASE, CSTRIDE, C::FL>(fl.nact);
        }
        convert((t + 1) % D, (t + 1) & 1);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    };
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    if (T > 0) step(0, std::true_type{});
#pragma unroll 1
    for (int t = 1; t < T; ++t) step(t, std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (T > 0) fl.template run<OUT>(hbuf + (T & 1) * 16 * LDH, sg.spikes_f32, sg.spikes_i8, T - 1, R, H);
    fl.template finish<OUT>(sg.count);
    const int8_t* hl = hbuf + (T & 1) * 16 * LDH;
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        *reinterpret_cast<v4f*>(sg.c_state + (size_t)rowc * H + col[i]) = c[i];
        const unsigned pk = *reinterpret_cast<const unsigned*>(hl + n * LDH + col[i]);
        const v4f h = {(float)(pk & 1u), (float)((pk >> 8) & 1u), (float)((pk >> 16) & 1u), (float)((pk >> 24) & 1u)};
        *reinterpret_cast<v4f*>(sg.h_state + (size_t)rowc * H + col[i]) = h;
    }
}

template <int KS, int OUT>
__global__ __launch_bounds__(512) void gsn_scan_fusedx_kernel(const ScanParams p) {
    using C = ScanCfg<1, KS, 8, 2, OUT, 0>;
    constexpr int LDH = C::LDH, HP = C::HP, D = 3, NW = 8, LDX = 2 * 32 + 8, PL_BYTES = 16 * LDX * 2;
    extern __shared__ __attribute__((aligned(16))) char scan_smem[];
    char* smem = scan_smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    int s = 0;
    for (int i = 1; i < p.nseg; ++i)
        if ((int)blockIdx.x >= p.seg[i].tile0) s = i;
    const ScanSegDev sg = p.seg[s];
    const int H = p.H, NT = p.NT, T = p.T, R = sg.R;
    const int XSLOT = (16 * sg.I * 4 + 1023) & ~1023;
    const int PLANES_OFF = D * XSLOT, HBUF_OFF = PLANES_OFF + 2 * 3 * PL_BYTES, CST_OFF = HBUF_OFF + 2 * 16 * LDH, WHH_OFF = CST_OFF + 5 * HP * 4;
    int8_t* hbuf = reinterpret_cast<int8_t*>(smem + HBUF_OFF);
    float(*cst)[HP] = reinterpret_cast<float(*)[HP]>(smem + CST_OFF);
    const int row0 = ((int)blockIdx.x - sg.tile0) * 16;
    const int rowc = row0 + n;  // R % 16 == 0: every row of the tile exists
    for (int j = tid; j < HP; j += NW * 64) {
        const bool in = j < H;
        cst[0][j] = in ? sg.bias[j] : 0.0f;
        cst[1][j] = in ? sg.bias[H + j] - sg.bias[j] : 0.0f;
        cst[2][j] = in ? sg.bn_alpha[j] : 0.0f;
        cst[3][j] = in ? sg.bn_beta[j] : 0.0f;
        cst[4][j] = in ? sg.w_dq[j] : 0.0f;
    }
    for (int i = tid; i < 2 * 16 * LDH / 4; i += NW * 64) reinterpret_cast<int*>(hbuf)[i] = 0;
    for (int i = tid; i < 2 * 3 * PL_BYTES / 4; i += NW * 64) reinterpret_cast<int*>(smem + PLANES_OFF)[i] = 0;  // k >= I stays zero
    {
        v4i* d0 = reinterpret_cast<v4i*>(smem + WHH_OFF);
        const v4i* s0 = reinterpret_cast<const v4i*>(sg.w_hh);
        for (int i = tid; i < NT * KS * 64; i += NW * 64) d0[i] = s0[i];
    }
    __syncthreads();
    for (int idx = tid; idx < 16 * (H / 4); idx += NW * 64) {
        const int rr = idx / (H / 4), j4 = (idx - rr * (H / 4)) * 4;
        const v4f h = *reinterpret_cast<const v4f*>(sg.h_state + (size_t)(row0 + rr) * H + j4);
        const unsigned pk = (h.x > 0.5f ? 1u : 0u) | (h.y > 0.5f ? 0x100u : 0u) | (h.z > 0.5f ? 0x10000u : 0u) |
                            (h.w > 0.5f ? 0x1000000u : 0u);
        *reinterpret_cast<unsigned*>(hbuf + rr * LDH + j4) = pk;
    }
    __syncthreads();
    SFSN_WG_STAMP(p.wg_times, 0);
    if (wave < NT - NW)
        fusedx_body<KS, OUT, 2>(sg, smem, T, H, NT, R, row0, rowc, n, q, tid, wave);
    else
        fusedx_body<KS, OUT, 1>(sg, smem, T, H, NT, R, row0, rowc, n, q, tid, wave);
    SFSN_WG_STAMP(p.wg_times, 1);
}

// =====================================================================================================
// feature prologue
// =====================================================================================================
struct FeatGroupDev {
    float* x;
    const float* ln_w;
    const float* ln_b;
    const float* mu;
    int lo, N, ctr, nbr, ctr_fb, nbr_fb, I1, I, norm;
    float eps;
};
struct FeatParams {
    FeatGroupDev g[SFSN_MAX_GROUPS];
    int ng, B, F, T, FB;
    int t0, t1;  // frames [t0, t1) are produced
    int f_lo, f_cnt;  // magnitude bins [f_lo, f_lo + f_cnt) are the ones some group reads
    float fdrc;
};

#define FEAT_TT 32  // frames per workgroup

// grid (ceil(T/32), B), 256 threads.  LDS: magnitude tile [f_cnt][33] + full-band tile [32][FB] + a gather table of the
// current unit chunk + one staging segment per wave.
//
// A wave owns a frame.  For every unit of the chunk it gathers the row from the tiles (per-lane LDS offsets from the
// table, built once per chunk by the whole workgroup), normalises it with two DPP wave reductions and parks it in its
// staging segment; the units of one (frame, clip) are contiguous in x, so the segment then leaves as one contiguous run
// (1.1-1.3 KB at baseline_m) with every lane storing -- rows of 38/94/158 floats written one by one were partial-line
// writes 78 KB apart.  NU = ceil(I/64) feature slots per lane is a template parameter: no work for absent slots.
#define FEAT_CHUNK 512  // floats per staging segment / gather-table entries: units per chunk = FEAT_CHUNK / I

template <int NU, int NORM>
__device__ __forceinline__ void feat_chunk_rows(const FeatGroupDev& g, const float* magT, const float* fbT, const int* offs,
                                                float* stage, int k0, int nk, int b, int B, int FB, int t0, int tend, int lane, int wave) {
    float lw[NU], lb[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int j = lane + 64 * u;
        const bool in = j < g.I && NORM == SFSN_NORM_LAYERNORM;
        lw[u] = in ? g.ln_w[j] : 0.0f;
        lb[u] = in ? g.ln_b[j] : 0.0f;
    }
    const float inv_I = 1.0f / (float)g.I;
    const float lap_den = NORM == SFSN_NORM_LAPLACE ? g.mu[b] + 2.220446049250313e-16f : 1.0f;
    // offline_gaussian_norm (FROZEN:205-218): (x - mean) / (std + EPSILON), one (mean, unbiased std) per clip (g.mu, g.ln_w = std [B])
    const float gau_mu = NORM == SFSN_NORM_GAUSSIAN ? g.mu[b] : 0.0f;
    const float gau_den = NORM == SFSN_NORM_GAUSSIAN ? g.ln_w[b] + 2.220446049250313e-16f : 1.0f;
    const int seg = nk * g.I;
    for (int tt = wave; tt < FEAT_TT; tt += 4) {
        const int t = t0 + tt;
        if (t >= tend) break;  // wave-uniform
        for (int k = 0; k < nk; ++k) {
            float v[NU];
            bool have[NU];
            float sum = 0.0f;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int j = lane + 64 * u;
                have[u] = j < g.I;
                const int o = have[u] ? offs[k * g.I + j] : 0;
                v[u] = !have[u] ? 0.0f : (o < 0 ? fbT[tt * FB + (-1 - o)] : magT[o + tt]);
                sum += v[u];
            }
            float y[NU];
            if constexpr (NORM == SFSN_NORM_LAYERNORM) {
                const float mean = wave_sum(sum) * inv_I;
                float ss = 0.0f;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const float d = v[u] - mean;
                    if (have[u]) ss += d * d;
                }
                const float rstd = __builtin_amdgcn_rsqf(wave_sum(ss) * inv_I + g.eps);
#pragma unroll
                for (int u = 0; u < NU; ++u) y[u] = ((v[u] - mean) * rstd) * lw[u] + lb[u];
            } else if constexpr (NORM == SFSN_NORM_LAPLACE) {
#pragma unroll
                for (int u = 0; u < NU; ++u) y[u] = v[u] / lap_den;
            } else if constexpr (NORM == SFSN_NORM_GAUSSIAN) {
#pragma unroll
                for (int u = 0; u < NU; ++u) y[u] = (v[u] - gau_mu) / gau_den;
            } else {
#pragma unroll
                for (int u = 0; u < NU; ++u) y[u] = v[u];
            }
#pragma unroll
            for (int u = 0; u < NU; ++u)
                if (have[u]) stage[k * g.I + lane + 64 * u] = y[u];
        }
        // LDS operations of one wave execute in order: the segment is complete when these reads are served
        float* out = g.x + ((size_t)t * B * g.N + (size_t)b * g.N + k0) * g.I;
        for (int i = lane; i < seg; i += 64) out[i] = stage[i];
    }
}

// (zero_ptr, zero_n16): rows blockIdx.y >= B of the grid carry no feature tile -- they zero `zero_n16` 16-byte pieces at `zero_ptr`:
// the zero initial state of a forward's scans (MODEL:100-106), written by the first launch of the forward's chain instead of by a
// fill kernel of its own (round 3: 217 fill launches averaging 149 us each in the timed region, queued behind scan workgroups).
__global__ __launch_bounds__(256) void features_kernel(const float* __restrict__ stft, const float* __restrict__ fb,
                                                        const FeatParams p, float* __restrict__ zero_ptr, const size_t zero_n16) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.y >= p.B) {
        const size_t blk = (size_t)(blockIdx.y - p.B) * gridDim.x + blockIdx.x;
        v4f* dst = reinterpret_cast<v4f*>(zero_ptr);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const size_t k = (blk * 8 + i) * 256 + threadIdx.x;
            if (k < zero_n16) dst[k] = v4f{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    const int nf = p.F - 1, T = p.T, B = p.B, FB = p.FB;
    float* magT = smem;                                         // [f_cnt][33]
    float* fbT = smem + (size_t)p.f_cnt * 33;                   // [32][FB]
    int* offs = reinterpret_cast<int*>(fbT + FEAT_TT * (FB > 0 ? FB : 1));  // [FEAT_CHUNK]
    const int b = blockIdx.y, t0 = p.t0 + blockIdx.x * FEAT_TT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* stage = reinterpret_cast<float*>(offs + FEAT_CHUNK) + wave * FEAT_CHUNK;
    const int tend = p.t1;

    // Tile loads in batches of eight independent requests per thread (a load -> wait -> store loop costs one HBM round
    // trip per iteration: 32 of them for 256 bins).  Frames past the end are clamped to a valid address and zeroed.
    {
        const int tt = tid & 31, t = t0 + tt;
        const bool live = t < tend;
        const int tc = live ? t : tend - 1;
        const float* src = stft + (((size_t)b * p.F + p.f_lo) * T + tc) * 2;
        for (int f0 = tid >> 5; f0 < p.f_cnt; f0 += 64) {  // this thread's bins: f0, f0+8, ..., f0+56
            float2 c[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int fr = f0 + 8 * i;
                if (fr > p.f_cnt - 1) fr = p.f_cnt - 1;
                c[i] = *reinterpret_cast<const float2*>(src + (size_t)fr * T * 2);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int fr = f0 + 8 * i;
                if (fr < p.f_cnt) magT[fr * 33 + tt] = live ? compress_mag(c[i].x, c[i].y, p.fdrc) : 0.0f;
            }
        }
    }
    if (fb) {
        const int nfb = FEAT_TT * FB;
        for (int i0 = tid; i0 < nfb; i0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int idx = i0 + 256 * i;
                if (idx > nfb - 1) idx = nfb - 1;
                const int tt = idx / FB, f = idx - tt * FB;
                int t = t0 + tt;
                if (t > tend - 1) t = tend - 1;
                v[i] = fb[((size_t)t * B + b) * FB + f];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = i0 + 256 * i;
                if (idx < nfb) fbT[idx] = (t0 + idx / FB < tend) ? v[i] : 0.0f;
            }
        }
    }

    for (int gi = 0; gi < p.ng; ++gi) {
        const FeatGroupDev g = p.g[gi];
        const int kch = FEAT_CHUNK / g.I;  // units per chunk (>= 1: I <= 256)
        for (int k0 = 0; k0 < g.N; k0 += kch) {
            const int nk = (g.N - k0 < kch) ? g.N - k0 : kch;
            // tiles loaded / previous chunk's table no longer read: LDS traffic only, so a raw barrier behind lgkmcnt(0) -- __syncthreads()
            // is a workgroup-scope release and made every chunk's row stores retire in HBM before the next chunk began (round 5)
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            // gather table of the chunk: entry >= 0 = offset of the bin's row in magT, < 0 = -1 - full-band column
            for (int idx = tid; idx < nk * g.I; idx += 256) {
                const int k = idx / g.I, j = idx - k * g.I, ku = k0 + k;
                offs[idx] = j < g.I1 ? (reflect_bin(g.lo + ku * g.ctr - g.nbr + j, nf) - p.f_lo) * 33
                                     : -1 - reflect_bin(g.lo + ku * g.ctr_fb - g.nbr_fb + (j - g.I1), nf) % FB;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
#define FEAT_ROWS(NU_)                                                                                                             \
    do {                                                                                                                           \
        if (g.norm == SFSN_NORM_LAYERNORM) feat_chunk_rows<NU_, SFSN_NORM_LAYERNORM>(g, magT, fbT, offs, stage, k0, nk, b, B, FB, t0, tend, lane, wave); \
        else if (g.norm == SFSN_NORM_LAPLACE) feat_chunk_rows<NU_, SFSN_NORM_LAPLACE>(g, magT, fbT, offs, stage, k0, nk, b, B, FB, t0, tend, lane, wave); \
        else if (g.norm == SFSN_NORM_GAUSSIAN) feat_chunk_rows<NU_, SFSN_NORM_GAUSSIAN>(g, magT, fbT, offs, stage, k0, nk, b, B, FB, t0, tend, lane, wave); \
        else feat_chunk_rows<NU_, SFSN_NORM_NONE>(g, magT, fbT, offs, stage, k0, nk, b, B, FB, t0, tend, lane, wave);               \
    } while (0)
            if (g.I <= 64) FEAT_ROWS(1);
            else if (g.I <= 128) FEAT_ROWS(2);
            else if (g.I <= 192) FEAT_ROWS(3);
            else FEAT_ROWS(4);
#undef FEAT_ROWS
        }
    }
}

// ---- cumulative_laplace_norm (model_low_freq_count_time.py:182-204): three small launches ------------------------------------
// (1) row sums of every (frame, row): a wave per row, the same lane assignment and wave reduction as the streaming hop's
//     feature rows (slot u of lane l = element l + 64 u), so that both paths add the same numbers in the same order;
// (2) one thread per row walks the T sums: running fp32 sum (torch.cumsum's arithmetic), mean, denominator;
// (3) every element divided by its (frame, row) denominator.
__global__ __launch_bounds__(256) void cumlap_rowsum_kernel(const float* __restrict__ x, float* __restrict__ s, int rows, int I) {
    const int lane = threadIdx.x & 63, row = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;  // wave-uniform
    const float* p = x + (size_t)row * I;
    float sum = 0.0f;
#pragma unroll
    for (int u = 0; u < 4; ++u) sum += (lane + 64 * u < I) ? p[lane + 64 * u] : 0.0f;
    sum = wave_sum(sum);
    if (lane == 0) s[row] = sum;
}

__global__ __launch_bounds__(64) void cumlap_scan_kernel(float* __restrict__ s, float* __restrict__ cum_state, int T, int R, int I,
                                                         int frames_before) {
    const int r = (int)blockIdx.x * 64 + threadIdx.x;
    if (r >= R) return;
    float cum = cum_state ? cum_state[r] : 0.0f;
    for (int t0 = 0; t0 < T; t0 += 8) {  // eight independent loads in flight, then the dependent chain
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = s[(size_t)(t0 + i < T ? t0 + i : T - 1) * R + r];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (t0 + i < T) {
                cum += v[i];
                const float mean = cum / (float)((double)I * (frames_before + t0 + i + 1));
                s[(size_t)(t0 + i) * R + r] = mean + 2.220446049250313e-16f;
            }
    }
    if (cum_state) cum_state[r] = cum;
}

__global__ __launch_bounds__(256) void cumlap_divide_kernel(float* __restrict__ x, const float* __restrict__ den, size_t n, int I) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = x[i] / den[i / (size_t)I];
}

// ---- spike counts (SynOPs, audiozen/metric.py:303-327) ---------------------------------------------------------
struct CountParams {
    const int8_t* src[SFSN_MAX_COUNT_TENSORS];
    unsigned long long* dst[SFSN_MAX_COUNT_TENSORS];
    unsigned long long nvec[SFSN_MAX_COUNT_TENSORS];  // 16-byte vectors per tensor
    int blk0[SFSN_MAX_COUNT_TENSORS + 1];             // first block of each tensor
    int n;
};

// Spikes are bytes 0/1: popcount of a dword = number of spikes in it.  Each block streams 64 KB (256 threads x 16 x 16 B).
__global__ __launch_bounds__(256) void spike_count_kernel(const CountParams p) {
    int ti = 0;
    while (ti + 1 < p.n && (int)blockIdx.x >= p.blk0[ti + 1]) ++ti;
    const v4i* src = reinterpret_cast<const v4i*>(p.src[ti]);
    const unsigned long long nvec = p.nvec[ti];
    unsigned long long i = (unsigned long long)(blockIdx.x - p.blk0[ti]) * 4096 + threadIdx.x;
    unsigned cnt = 0;
#pragma unroll 4
    for (int k = 0; k < 16; ++k, i += 256) {
        if (i < nvec) {
            const v4i v = src[i];
            cnt += __builtin_popcount((unsigned)v.x) + __builtin_popcount((unsigned)v.y) + __builtin_popcount((unsigned)v.z) +
                   __builtin_popcount((unsigned)v.w);
        }
    }
    __shared__ unsigned part[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(p.dst[ti], (unsigned long long)(part[0] + part[1] + part[2] + part[3]));
}

// ---- Laplace means -----------------------------------------------------------------------------------
// rs[b][f] = sum_t mag[b][f][t] (f < nf), then rs[b][nf + f'] = sum_t fb[t][b][f'].  One wave per row.
// (rs2 non-null: also the row sums of squares, as doubles -- offline_gaussian_norm's second moment)
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ stft, const float* __restrict__ fb,
                                                      float* __restrict__ rs, double* __restrict__ rs2, int B, int F, int T, int FB, float fdrc) {
    const int nf = F - 1, per_b = nf + FB;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wid >= B * per_b) return;
    const int b = wid / per_b, f = wid - b * per_b;
    double acc = 0.0, acc2 = 0.0;
    if (f < nf) {
        const float* src = stft + ((size_t)b * F + f) * T * 2;
        for (int t = lane; t < T; t += 64) {
            const float2 c = *reinterpret_cast<const float2*>(src + 2 * (size_t)t);
            const double m = (double)compress_mag(c.x, c.y, fdrc);
            acc += m;
            acc2 += m * m;
        }
    } else if (fb) {
        for (int t = lane; t < T; t += 64) {
            const double m = (double)fb[((size_t)t * B + b) * FB + (f - nf)];
            acc += m;
            acc2 += m * m;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { acc += __shfl_xor(acc, o); acc2 += __shfl_xor(acc2, o); }
    if (lane == 0) {
        rs[wid] = (float)acc;
        if (rs2) { rs2[wid] = acc2; reinterpret_cast<double*>(rs2 + (size_t)B * per_b)[wid] = acc; }  // (second half: the sums unrounded)
    }
}

// offline_gaussian_norm's statistics per (group, clip) over the gathered index multiset: mean and UNBIASED standard deviation
// (torch.std) from the sums and the sums of squares of the rows, combined in double.  One wave per (g, b).
__global__ __launch_bounds__(64) void gaussian_stats_kernel(const double* __restrict__ rs2, const FeatParams p, float* __restrict__ mu,
                                                             float* __restrict__ sd) {
    const int gi = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const FeatGroupDev g = p.g[gi];
    const int nf = p.F - 1, per_b = nf + p.FB;
    const double* r2 = rs2 + (size_t)b * per_b;
    const double* r1 = rs2 + (size_t)p.B * per_b + (size_t)b * per_b;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < g.N; ++k)
        for (int j = lane; j < g.I; j += 64) {
            const int idx = j < g.I1 ? reflect_bin(g.lo + k * g.ctr - g.nbr + j, nf)
                                     : nf + (reflect_bin(g.lo + k * g.ctr_fb - g.nbr_fb + (j - g.I1), nf) % p.FB);
            s1 += r1[idx];
            s2 += r2[idx];
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (lane == 0) {
        const double n = (double)p.T * g.N * g.I, m = s1 / n;
        double var = (s2 - n * m * m) / (n - 1.0);
        if (var < 0.0) var = 0.0;
        mu[(size_t)gi * p.B + b] = (float)m;
        sd[(size_t)gi * p.B + b] = (float)sqrt(var);
    }
}

// mu[g][b] = (sum over the gathered index multiset of row sums) / (T * N * I).  One wave per (g, b).
__global__ __launch_bounds__(64) void laplace_mu_kernel(const float* __restrict__ rs, const FeatParams p,
                                                         float* __restrict__ mu) {
    const int gi = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const FeatGroupDev g = p.g[gi];
    const int nf = p.F - 1, per_b = nf + p.FB;
    const float* r = rs + (size_t)b * per_b;
    double acc = 0.0;
    for (int k = 0; k < g.N; ++k)
        for (int j = lane; j < g.I; j += 64) {
            if (j < g.I1)
                acc += (double)r[reflect_bin(g.lo + k * g.ctr - g.nbr + j, nf)];
            else
                acc += (double)r[nf + (reflect_bin(g.lo + k * g.ctr_fb - g.nbr_fb + (j - g.I1), nf) % p.FB)];
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) mu[(size_t)gi * p.B + b] = (float)(acc / ((double)p.T * g.N * g.I));
}

// =====================================================================================================
// deep-filter epilogue
// =====================================================================================================
struct DfGroupDev {
    const float* proj;
    int N, fc, df, lo;
};
struct DfParams {
    DfGroupDev g[SFSN_MAX_GROUPS];
    int ng, B, F, T, S, fcov;  // fcov = first bin not covered by any group
    int t0, t1;                // frames [t0, t1) are produced
};

// grid (ceil(T/32), B), 256 threads; LDS: one unit's projection tile [32][P+1].
__global__ __launch_bounds__(256) void deepfilter_kernel(const float* __restrict__ stft, const DfParams p,
                                                          float* __restrict__ enh, float* __restrict__ mag) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int B = p.B, F = p.F, T = p.T, S = p.S;
    const int b = blockIdx.y, t0 = p.t0 + blockIdx.x * 32;
    const int tid = threadIdx.x, tt = tid & 31, fs = tid >> 5, t = t0 + tt;
    const int tend = p.t1;

    for (int gi = 0; gi < p.ng; ++gi) {
        const DfGroupDev g = p.g[gi];
        const int P = 2 * g.fc * g.df * S, LDP = P + 1;
        for (int k = 0; k < g.N; ++k) {
            __syncthreads();
            for (int idx = tid; idx < 32 * P; idx += 256) {
                const int r = idx / P, c = idx - r * P, tr = t0 + r;
                smem[r * LDP + c] = tr < tend ? g.proj[((size_t)tr * B * g.N + (size_t)b * g.N + k) * P + c] : 0.0f;
            }
            __syncthreads();
            if (t < tend) {
                const float* pr = smem + tt * LDP;
                for (int fci = fs; fci < g.fc; fci += 8) {
                    const int f = g.lo + k * g.fc + fci;
                    const float* xrow = stft + ((size_t)b * F + f) * T * 2;
                    for (int s = 0; s < S; ++s) {
                        float yr = 0.0f, yi = 0.0f;
                        for (int d = 0; d < g.df; ++d) {
                            const int ts = t - (g.df - 1) + d;
                            float xr = 0.0f, xi = 0.0f;
                            if (ts >= 0) {
                                const float2 xv = *reinterpret_cast<const float2*>(xrow + 2 * (size_t)ts);
                                xr = xv.x;
                                xi = xv.y;
                            }
                            const float cr = pr[((0 * g.fc + fci) * g.df + d) * S + s];
                            const float ci = pr[((1 * g.fc + fci) * g.df + d) * S + s];
                            yr += xr * cr - xi * ci;
                            yi += xr * ci + xi * cr;
                        }
                        const size_t o = (((size_t)b * S + s) * F + f) * T + t;
                        *reinterpret_cast<float2*>(enh + 2 * o) = make_float2(yr, yi);
                        if (mag) mag[o] = fast_abs2(yr, yi);
                    }
                }
            }
        }
    }
    // bins no group covers (at least the Nyquist bin) pass through untouched (MODEL:461-470)
    if (t < tend)
        for (int f = p.fcov + fs; f < F; f += 8) {
            const float2 xv = *reinterpret_cast<const float2*>(stft + (((size_t)b * F + f) * T + t) * 2);
            for (int s = 0; s < S; ++s) {
                const size_t o = (((size_t)b * S + s) * F + f) * T + t;
                *reinterpret_cast<float2*>(enh + 2 * o) = xv;
                if (mag) mag[o] = fast_abs2(xv.x, xv.y);
            }
        }
}

// ---- pass-structured variant (every P a multiple of 4) --------------------------------------------------------------
// A pass = a run of consecutive units of one group whose coefficient rows are contiguous in memory ([t][b*N + k][P]:
// units k0..k0+U-1 of one (t, b) are U*P consecutive floats).  Per pass: the [32][U*P] coefficient tile is loaded with
// 16-byte loads (a wave per row, eight rows in flight per wave), the noisy-spectrum tile [U*fc][32 + df - 1] once (the df
// taps of an output then come from LDS instead of df global loads), one barrier pair, and all 256 threads have work
// (U*fc bins x 32 frames).  baseline_m: 7 passes instead of 13 unit rounds, no runtime integer division per element.
#define DF_MAX_PASSES 48
struct DfPassParams {
    DfParams base;
    unsigned char pg[DF_MAX_PASSES], pk0[DF_MAX_PASSES], pnu[DF_MAX_PASSES];
    int npass, ctile_floats;  // floats of LDS reserved for the coefficient tile (the X tile follows)
};

#define DFP_THREADS 512
__global__ __launch_bounds__(DFP_THREADS) void deepfilter_pass_kernel(const float* __restrict__ stft, const DfPassParams pp,
                                                               float* __restrict__ enh, float* __restrict__ mag) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DfParams& p = pp.base;
    const int B = p.B, F = p.F, T = p.T, S = p.S;
    const int b = blockIdx.y, t0 = p.t0 + blockIdx.x * 32;
    const int tid = threadIdx.x, tt = tid & 31, fs = tid >> 5, t = t0 + tt;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int NWV = DFP_THREADS / 64, NFS = DFP_THREADS / 32;  // waves, bin slots
    const int tend = p.t1;
    float* ct = smem;
    float2* xt = reinterpret_cast<float2*>(smem + pp.ctile_floats);

    for (int ps = 0; ps < pp.npass; ++ps) {
        const DfGroupDev g = p.g[pp.pg[ps]];
        const int k0 = pp.pk0[ps], U = pp.pnu[ps];
        const int P = 2 * g.fc * g.df * S, UP = U * P, LD = UP + 1, Q = UP >> 2;
        const int nb = U * g.fc, XW = 32 + g.df - 1;
        // (LDS only: the previous pass's reads of the tiles are done.  A raw barrier -- __syncthreads() would wait for the previous pass's
        //  enh / mag stores to retire in HBM before this pass's loads are issued)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        // coefficient tile: row r = frame t0 + r, UP contiguous floats
        for (int c4 = lane; c4 < Q; c4 += 64) {
#pragma unroll
            for (int i = 0; i < 32 / NWV; ++i) {
                const int r = wave + NWV * i, tr = t0 + r;
                v4f v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (tr < tend) v = *reinterpret_cast<const v4f*>(g.proj + ((size_t)tr * B * g.N + (size_t)b * g.N + k0) * P + 4 * c4);
                float* d = ct + r * LD + 4 * c4;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            }
        }
        // noisy-spectrum tile: bins of the pass x frames [t0-(df-1), t0+32), eight independent requests per thread in flight
        // (a load -> LDS-store loop pays one memory round trip per iteration: that was 70 % of this kernel's time)
        {
            const int fbase = g.lo + k0 * g.fc, tb = t0 - (g.df - 1), nx = nb * XW;
            for (int e0 = tid; e0 < nx; e0 += DFP_THREADS * 8) {
                float2 v[8];
                bool ok[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    int e = e0 + DFP_THREADS * i;
                    if (e > nx - 1) e = nx - 1;
                    const int j = e / XW, c = e - j * XW, ts = tb + c;
                    ok[i] = ts >= 0 && ts < T;
                    const int tsc = ts < 0 ? 0 : (ts > T - 1 ? T - 1 : ts);
                    v[i] = *reinterpret_cast<const float2*>(stft + (((size_t)b * F + fbase + j) * T + tsc) * 2);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = e0 + DFP_THREADS * i;
                    if (e < nx) xt[e] = ok[i] ? v[i] : make_float2(0.0f, 0.0f);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        if (t < tend) {
            const float* pr = ct + tt * LD;
            int u = 0, fci = fs;
            while (fci >= g.fc) { fci -= g.fc; ++u; }
            for (int j = fs; j < nb; j += NFS) {
                const int f = g.lo + (k0 + u) * g.fc + fci;
                const float* pu = pr + u * P;
                for (int s_ = 0; s_ < S; ++s_) {
                    float yr = 0.0f, yi = 0.0f;
                    const float2* xr_ = xt + j * XW + tt;
                    for (int d = 0; d < g.df; ++d) {
                        const float2 xv = xr_[d];
                        const float cr = pu[((0 * g.fc + fci) * g.df + d) * S + s_];
                        const float ci = pu[((1 * g.fc + fci) * g.df + d) * S + s_];
                        yr += xv.x * cr - xv.y * ci;
                        yi += xv.x * ci + xv.y * cr;
                    }
                    const size_t o = (((size_t)b * S + s_) * F + f) * T + t;
                    *reinterpret_cast<float2*>(enh + 2 * o) = make_float2(yr, yi);
                    if (mag) mag[o] = fast_abs2(yr, yi);
                }
                fci += NFS;
                while (fci >= g.fc) { fci -= g.fc; ++u; }
            }
        }
    }
    if (t < tend)
        for (int f = p.fcov + fs; f < F; f += NFS) {
            const float2 xv = *reinterpret_cast<const float2*>(stft + (((size_t)b * F + f) * T + t) * 2);
            for (int s_ = 0; s_ < S; ++s_) {
                const size_t o = (((size_t)b * S + s_) * F + f) * T + t;
                *reinterpret_cast<float2*>(enh + 2 * o) = xv;
                if (mag) mag[o] = fast_abs2(xv.x, xv.y);
            }
        }
}

// =====================================================================================================
// host side: argument checks, dispatch on compile-time shapes, launches
// =====================================================================================================
static inline int hip_ok(hipError_t e) { return e == hipSuccess ? SFSN_OK : SFSN_EHIP; }
// Per-DEVICE caches: one process may drive several GPUs (the function attribute below and the CU count are per device).
#define SFSN_MAX_DEVICES 64
static int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SFSN_MAX_DEVICES) dev = 0;
    return dev;
}
static int cu_count() {  // compute units of the current device (256 on MI355X); 256 if the query fails
    static int n[SFSN_MAX_DEVICES] = {0};
    const int dev = current_device();
    if (n[dev] == 0) {
        int v = 0;
        n[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return n[dev];
}
// hipFuncAttributeMaxDynamicSharedMemorySize raised to `bytes` on the current device (idempotent; a benign race between host
// threads sets it twice at worst).  `seen` is the caller's per-kernel table of the largest size set per device.
static int raise_lds(const void* kern, int bytes, int* seen) {
    const int dev = current_device();
    if (bytes > seen[dev]) {
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return SFSN_EHIP;
        seen[dev] = bytes;
    }
    return SFSN_OK;
}
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

extern "C" int sfsn_abi_version(void) { return SFSN_ABI_VERSION; }

// Test hook (not part of include/sfsn.h): wave n of a 64-wave launch runs wait_vmcnt_n's computed jump with n in its probe form (every
// table entry records its own index instead of waiting) -> out[n] must be n for all 64 entries (tests/test_hip_parity.py).
__global__ __launch_bounds__(64) void vmcnt_table_probe_kernel(int* __restrict__ out) {
    const int landed = wait_vmcnt_n<true>((int)blockIdx.x);
    wait_vmcnt_n((int)blockIdx.x);  // (the product form on an empty queue: must fall through)
    if (threadIdx.x == 0) out[blockIdx.x] = landed;
}
extern "C" int sfsn_debug_vmcnt_table(int* out_host /* [64] */) {
    int* d = nullptr;
    if (!out_host) return SFSN_EINVAL;
    if (hipMalloc(&d, 64 * sizeof(int)) != hipSuccess) return SFSN_EHIP;
    int rc = SFSN_OK;
    if (hipMemset(d, 0xff, 64 * sizeof(int)) != hipSuccess) rc = SFSN_EHIP;
    if (rc == SFSN_OK) {
        hipLaunchKernelGGL(vmcnt_table_probe_kernel, dim3(64), dim3(64), 0, 0, d);
        if (hipGetLastError() != hipSuccess || hipMemcpy(out_host, d, 64 * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) rc = SFSN_EHIP;
    }
    hipFree(d);
    return rc;
}

#ifdef SFSN_EXPERIMENTS
// the workgroup-stamp probe (sfsn_scan_dev.h): a caller-owned device buffer of 2 x capacity stamps, handed out launch by launch;
// kind 1 = gsn_scan_kernel, 2 = the fused scan, 3 = the fused-x scan, 4 = the narrow stack launch, 5 = the IO-wave full-band launch
// (25 slots per workgroup: stamps, then 12 waves x 4 stall counters), 6 = the wide launch (33 slots: 16 waves).  Single-threaded use only.
static struct { unsigned long long* buf; int cap, used, nrec; int rec[16384][3]; } g_wgp;
unsigned long long* sfsn_wgprobe_take(int kind, int n) {
    if (!g_wgp.buf || g_wgp.used + n > g_wgp.cap || g_wgp.nrec >= 16384) return nullptr;
    unsigned long long* p = g_wgp.buf + 2 * (size_t)g_wgp.used;
    g_wgp.rec[g_wgp.nrec][0] = kind; g_wgp.rec[g_wgp.nrec][1] = g_wgp.used; g_wgp.rec[g_wgp.nrec][2] = n;
    ++g_wgp.nrec;
    g_wgp.used += n;
    return p;
}
extern "C" void sfsn_debug_wg_times(unsigned long long* buf, int capacity_wgs) { g_wgp.buf = buf; g_wgp.cap = capacity_wgs; g_wgp.used = 0; g_wgp.nrec = 0; }
extern "C" int sfsn_debug_wg_log(int* out /* [max][3]: kind, first workgroup, workgroups */, int max_rec) {
    const int n = g_wgp.nrec < max_rec ? g_wgp.nrec : max_rec;
    for (int i = 0; i < n; ++i) { out[3 * i] = g_wgp.rec[i][0]; out[3 * i + 1] = g_wgp.rec[i][1]; out[3 * i + 2] = g_wgp.rec[i][2]; }
    return n;
}
#endif

extern "C" const char* sfsn_strerror(int code) {
    switch (code) {
        case SFSN_OK: return "ok";
        case SFSN_EINVAL: return "invalid argument";
        case SFSN_EUNSUPPORTED: return "unsupported shape for the gfx950 kernels";
        case SFSN_EHIP: return "HIP runtime error (no gfx950 device, bad pointer or failed launch)";
        case SFSN_EDIVISIBLE: return "Number of frequency bins must be divisible by the center frequency";
        default: return "unknown error";
    }
}

extern "C" int sfsn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

template <int G, int KS, int NW, int TPW, int OUT, int LP>
static int launch_scan_variant(const ScanParams& p, int tiles, hipStream_t st) {
    using C = ScanCfg<G, KS, NW, TPW, OUT, LP>;
    auto kern = gsn_scan_kernel<G, KS, NW, TPW, OUT, LP>;
    if (C::LDS_BYTES > 64 * 1024) {
        static int seen[SFSN_MAX_DEVICES] = {0};
        if (raise_lds(reinterpret_cast<const void*>(kern), C::LDS_BYTES, seen) != SFSN_OK) return SFSN_EHIP;
    }
    ScanParams q = p;
    q.wg_times = sfsn_wgprobe_take(1, tiles);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(NW * 64), C::LDS_BYTES, st, q);
    return hip_ok(hipGetLastError());
}

template <int G, int KS, int NW, int TPW, int LP>
static int launch_scan(const ScanParams& p, int tiles, int out, hipStream_t st) {
    // output sets compiled: int8 only (downstream products need it), fp32 + int8 (module API), + membranes (tests)
    switch (out) {
        case 514: return launch_scan_variant<G, KS, NW, TPW, 514, LP>(p, tiles, st);
        case 515: return launch_scan_variant<G, KS, NW, TPW, 515, LP>(p, tiles, st);
        case 519: return launch_scan_variant<G, KS, NW, TPW, 519, LP>(p, tiles, st);
        case 2: return launch_scan_variant<G, KS, NW, TPW, 2, LP>(p, tiles, st);
        case 3: return launch_scan_variant<G, KS, NW, TPW, 3, LP>(p, tiles, st);
        case 7: return launch_scan_variant<G, KS, NW, TPW, 7, LP>(p, tiles, st);
#ifdef SFSN_TIMING_EXPERIMENTS  // wrong-result variants for bottleneck attribution only (scripts/exp_scan.sh)
        case 19: return launch_scan_variant<G, KS, NW, TPW, 19, LP>(p, tiles, st);
        case 35: return launch_scan_variant<G, KS, NW, TPW, 35, LP>(p, tiles, st);
        case 51: return launch_scan_variant<G, KS, NW, TPW, 51, LP>(p, tiles, st);
        case 67: return launch_scan_variant<G, KS, NW, TPW, 67, LP>(p, tiles, st);
        case 131: return launch_scan_variant<G, KS, NW, TPW, 131, LP>(p, tiles, st);
        case 259: return launch_scan_variant<G, KS, NW, TPW, 259, LP>(p, tiles, st);
        case 387: return launch_scan_variant<G, KS, NW, TPW, 387, LP>(p, tiles, st);
#endif
        default: return SFSN_EUNSUPPORTED;
    }
}


template <int KS, int RPW, int OUT>
static int launch_scan3_variant(const ScanParams& p, int tiles, hipStream_t st) {
    using C = Scan3Cfg<KS, RPW, 0>;
    const int lds = C::lds_bytes(p.NT);
    if (p.w16) {
        auto k16 = gsn_scan3_kernel<KS, RPW, OUT, 1>;
        if (lds > 64 * 1024) {
            static int seen16[SFSN_MAX_DEVICES] = {0};
            if (raise_lds(reinterpret_cast<const void*>(k16), lds, seen16) != SFSN_OK) return SFSN_EHIP;
        }
        hipLaunchKernelGGL(k16, dim3(tiles), dim3(1024), lds, st, p);
        return hip_ok(hipGetLastError());
    }
    auto kern = gsn_scan3_kernel<KS, RPW, OUT>;
    if (lds > 64 * 1024) {
        static int seen[SFSN_MAX_DEVICES] = {0};
        if (raise_lds(reinterpret_cast<const void*>(kern), lds, seen) != SFSN_OK) return SFSN_EHIP;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(1024), lds, st, p);
    return hip_ok(hipGetLastError());
}

static int launch_scan3(const ScanParams& p, int tiles, int out, int KS, hipStream_t st) {
#define SCAN3_CASE(KS_, RPW_, OUT_) \
    if (KS == KS_ && p.rpw == RPW_ && out == OUT_) return launch_scan3_variant<KS_, RPW_, OUT_>(p, tiles, st);
#define SCAN3_KS(KS_) \
    SCAN3_CASE(KS_, 4, 2) SCAN3_CASE(KS_, 4, 3) SCAN3_CASE(KS_, 8, 2) SCAN3_CASE(KS_, 8, 3) SCAN3_CASE(KS_, 16, 2) SCAN3_CASE(KS_, 16, 3)
    SCAN3_KS(1) SCAN3_KS(2) SCAN3_KS(3) SCAN3_KS(4)
#undef SCAN3_KS
#undef SCAN3_CASE
    return SFSN_EUNSUPPORTED;
}

static int layer_scan_impl(const sfsn_scan_segment* segs, int n_segs, int T, int H, int shared, int rows_per_wg, int w16, void* stream) {
    if (!segs || n_segs <= 0 || n_segs > SFSN_MAX_SEGMENTS || T < 0 || H <= 0) return SFSN_EINVAL;
    if (H % 16 != 0 || H > SFSN_MAX_HIDDEN) return SFSN_EUNSUPPORTED;
    ScanParams p;
    p.wg_times = nullptr;
    p.w16 = w16;
    p.lsplit = sfsn_s3_lsplit_host();
    // rows per workgroup: as few as it takes to spread the launch over ~all 256 CUs (see the kernel comment)
    int rows_total = 0;
    for (int i = 0; i < n_segs; ++i) rows_total += segs[i].R > 0 ? segs[i].R : 0;
    int rpw = 16;
    const bool streamed = !shared && H > 256;  // W_hh does not fit one CU: streamed-weights kernel, 16 rows per workgroup
    if (rows_per_wg != 0 && rows_per_wg != 16 && rows_per_wg != 8 && rows_per_wg != 4) return SFSN_EINVAL;
    if (streamed) {
        rpw = 16;
    } else if (rows_per_wg != 0) {
        rpw = rows_per_wg;
    } else {
        while (rpw > 4 && (rows_total + rpw - 1) / rpw < 200) rpw >>= 1;
    }
#ifdef SFSN_TIMING_EXPERIMENTS
    if (const char* e = getenv("SFSN_SCAN_RPW")) rpw = atoi(e);
#endif
    p.rpw = rpw;
    int tiles = 0;
    // the set of outputs must be the same for every segment of a launch (it selects the kernel variant);
    // the int8 spikes are always produced (every consumer of a scan in this library reads them)
    int out = 2 | (segs[0].spikes_f32 ? 1 : 0) | (segs[0].membrane ? 4 : 0);
    if (out == 6) return SFSN_EUNSUPPORTED;  // membranes are a test output: request them together with fp32 spikes
    for (int i = 0; i < n_segs; ++i) {
        const sfsn_scan_segment& s = segs[i];
        if (!s.spikes_i8 || (s.spikes_f32 != nullptr) != ((out & 1) != 0) || (s.membrane != nullptr) != ((out & 4) != 0))
            return SFSN_EINVAL;
        if (s.R <= 0 || !s.zin || !s.w_hh || !s.w_dq || !s.bias || !s.bn_alpha || !s.bn_beta || !s.h_state || !s.c_state)
            return SFSN_EINVAL;
        if (!aligned16(s.zin) || !aligned16(s.w_hh) || !aligned16(s.h_state) || !aligned16(s.c_state) ||
            !aligned16(s.spikes_f32) || !aligned16(s.spikes_i8) || !aligned16(s.membrane))
            return SFSN_EINVAL;
        ScanSegDev& d = p.seg[i];
        d.zin = s.zin; d.w_hh = s.w_hh; d.w_dq = s.w_dq; d.bias = s.bias; d.bn_alpha = s.bn_alpha; d.bn_beta = s.bn_beta;
        d.h_state = s.h_state; d.c_state = s.c_state; d.spikes_f32 = s.spikes_f32; d.spikes_i8 = s.spikes_i8; d.count = s.spike_count;
        d.membrane = s.membrane; d.R = s.R; d.tile0 = tiles;
        tiles += (s.R + rpw - 1) / rpw;
    }
    p.nseg = n_segs; p.T = T; p.H = H; p.NT = H / 16;
#ifdef SFSN_TIMING_EXPERIMENTS
    if (const char* e = getenv("SFSN_SCAN_DEBUG_OUT")) out = atoi(e);
#endif
    const int NT = p.NT, KS = (H + 63) / 64;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // Waves per workgroup: as many as the per-wave share of W allows registers for (16 waves -> 128 VGPRs, 8 -> 256):
    // more waves per SIMD overlap one wave's epilogue VALU with another's MFMAs and hide LDS / VMEM latency.
    if (streamed) {
        const int HPs = KS * 64;
        const size_t lds = (size_t)2 * 16 * (HPs + 32) + (size_t)5 * HPs * 4;
        hipLaunchKernelGGL(gsn_scan_stream_kernel<2>, dim3(tiles), dim3(512), lds, st, p);
        return hip_ok(hipGetLastError());
    }
    // shared gates, at most 14 output tiles (two of the 16 waves are free for the input ring and the spike stores), no membrane
    // output: the scan with IO-specialised waves (sfsn_scan3_dev.h).  SFSN_SCAN_V2=1 keeps round 2's body (A/B runs, tests).
    // (at 16 rows per workgroup one loader wave issuing a DMA per output tile and step is slower than round 2's body, where
    //  every wave fetches its own tile: 1.18 against 0.95 us per step at H = 224 -- that geometry keeps the old body)
    if (shared && NT <= 14 && rpw <= 8 && (out == 2 || out == 3) && !getenv("SFSN_SCAN_V2")) return launch_scan3(p, tiles, out, KS, st);
    if (w16) return SFSN_EUNSUPPORTED;  // only the scan3 kernels have the two-plane form: the caller uses sfsn_gsn_layer_scan
    // round 6: separate gate weights, at most 14 tiles, 4 rows per workgroup: the IO-wave scan with both gates of a tile in one wave
    // (sfsn_scan3g_dev.h; baseline_xl's sub-band layers: round 2's body spilled 116 registers there)
    if (!shared && NT <= 14 && rpw == 4 && (out == 2 || out == 3) && !getenv("SFSN_SCAN_V2")) {
#define SCAN3G_CASE(KS_, OUT_)                                                                                            \
    if (KS == KS_ && out == OUT_) {                                                                                       \
        const int lds = Scan3gCfg<KS_>::lds_bytes(NT);                                                                     \
        if (lds <= 160 * 1024 - 64) {                                                                                     \
            auto kern = gsn_scan3g_kernel<KS_, OUT_>;                                                                     \
            static int seen[SFSN_MAX_DEVICES] = {0};                                                                      \
            if (raise_lds(reinterpret_cast<const void*>(kern), lds, seen) != SFSN_OK) return SFSN_EHIP;                   \
            hipLaunchKernelGGL(kern, dim3(tiles), dim3(1024), lds, st, p);                                                \
            return hip_ok(hipGetLastError());                                                                             \
        }                                                                                                                 \
    }
        SCAN3G_CASE(1, 2) SCAN3G_CASE(1, 3) SCAN3G_CASE(2, 2) SCAN3G_CASE(2, 3) SCAN3G_CASE(3, 2) SCAN3G_CASE(3, 3) SCAN3G_CASE(4, 2) SCAN3G_CASE(4, 3)
#undef SCAN3G_CASE
    }
    int NW, TPW;
    if (rpw == 4 && out <= 7) out |= 512;  // repacked-epilogue variant (see scan_body)
    if (shared) {
        NW = H <= 256 ? 16 : 8;  // H = 320: one digit plane in LDS, two in registers (120 per wave), 2 waves per SIMD
    } else {
        NW = H <= 128 ? 16 : 8;
    }
    TPW = (NT + NW - 1) / NW;
#define SCAN_CASE(G_, KS_, NW_, TPW_, LP_) \
    if (KS == KS_ && NW == NW_ && TPW == TPW_) return launch_scan<G_, KS_, NW_, TPW_, LP_>(p, tiles, out, st);
    if (shared) {
        SCAN_CASE(1, 1, 16, 1, 0) SCAN_CASE(1, 2, 16, 1, 0) SCAN_CASE(1, 3, 16, 1, 0) SCAN_CASE(1, 4, 16, 1, 0) SCAN_CASE(1, 5, 8, 3, 1)
    } else {
        SCAN_CASE(2, 1, 16, 1, 0) SCAN_CASE(2, 2, 16, 1, 0) SCAN_CASE(2, 3, 8, 2, 0) SCAN_CASE(2, 4, 8, 2, 0)
    }
#undef SCAN_CASE
    return SFSN_EUNSUPPORTED;
}

extern "C" int sfsn_gsn_layer_scan(const sfsn_scan_segment* segs, int n_segs, int T, int H, int shared, int rows_per_wg,
                                   void* stream) {
    return layer_scan_impl(segs, n_segs, T, H, shared, rows_per_wg, 0, stream);
}

extern "C" int sfsn_gsn_layer_scan_w16(const sfsn_scan_segment* segs, int n_segs, int T, int H, int shared, int rows_per_wg,
                                       void* stream) {
    return layer_scan_impl(segs, n_segs, T, H, shared, rows_per_wg, 1, stream);
}

// Separate gate weights too large for one compute unit (H > 256): the tiles of a 16-row block split over several workgroups that keep
// their share of W_hh resident and exchange the spikes every step (gsn_scan_split_kernel).  One segment, co-resident workgroups.
extern "C" size_t sfsn_scan_split_scratch_bytes(int R, int H) {
    if (R <= 0 || H <= 0 || H % 16) return 0;
    return 64 + (size_t)2 * R * (H / 4) * sizeof(unsigned);
}
extern "C" int sfsn_gsn_layer_scan_split(const sfsn_scan_segment* segs, int n_segs, int T, int H, int shared, void* scratch, size_t scratch_bytes,
                                         void* stream) {
    if (!segs || n_segs <= 0 || T < 0 || H <= 0 || !scratch) return SFSN_EINVAL;
    if (H % 16 != 0 || H > SFSN_MAX_HIDDEN || n_segs != 1 || shared || H <= 256) return SFSN_EUNSUPPORTED;  // (what fits one CU has faster kernels)
    const sfsn_scan_segment& s = segs[0];
    const int out = 2 | (s.spikes_f32 ? 1 : 0) | (s.membrane ? 4 : 0);
    if (out == 6) return SFSN_EUNSUPPORTED;
    if (s.R <= 0 || !s.spikes_i8 || !s.zin || !s.w_hh || !s.w_dq || !s.bias || !s.bn_alpha || !s.bn_beta || !s.h_state || !s.c_state) return SFSN_EINVAL;
    if (!aligned16(s.zin) || !aligned16(s.w_hh) || !aligned16(s.h_state) || !aligned16(s.c_state) || !aligned16(s.spikes_f32) ||
        !aligned16(s.spikes_i8) || !aligned16(s.membrane) || !aligned16(scratch))
        return SFSN_EINVAL;
    if (scratch_bytes < sfsn_scan_split_scratch_bytes(s.R, H)) return SFSN_EINVAL;
    if (T == 0) return SFSN_OK;
    constexpr int G = 2;
    const int NT = H / 16, KS = (H + 63) / 64, HP = KS * 64;
    const size_t wtile = (size_t)G * 3 * KS * 1024, fixed = (size_t)2 * 16 * (HP + 32) + (size_t)(3 + G) * HP * 4;
    int TPS = (int)((150 * 1024 - fixed) / wtile);
    if (TPS > 8) TPS = 8;
    if (TPS < 1) return SFSN_EUNSUPPORTED;
    const int NSPL = (NT + TPS - 1) / TPS;
    TPS = (NT + NSPL - 1) / NSPL;
    const int blocks = ((s.R + 15) / 16) * NSPL;
    if (blocks > cu_count()) return SFSN_EUNSUPPORTED;  // every workgroup needs a compute unit of its own for the whole launch (LDS)
    const size_t lds = (size_t)TPS * wtile + fixed;
    ScanParams p;
    p.wg_times = nullptr; p.w16 = 0; p.lsplit = 0; p.rpw = 16;
    ScanSegDev& d = p.seg[0];
    d.zin = s.zin; d.w_hh = s.w_hh; d.w_dq = s.w_dq; d.bias = s.bias; d.bn_alpha = s.bn_alpha; d.bn_beta = s.bn_beta;
    d.h_state = s.h_state; d.c_state = s.c_state; d.spikes_f32 = s.spikes_f32; d.spikes_i8 = s.spikes_i8; d.count = s.spike_count;
    d.membrane = s.membrane; d.R = s.R; d.tile0 = 0;
    p.nseg = 1; p.T = T; p.H = H; p.NT = NT;
    auto kern = gsn_scan_split_kernel<G>;
    static int lds_seen[SFSN_MAX_DEVICES] = {0};
    if (raise_lds(reinterpret_cast<const void*>(kern), (int)lds, lds_seen) != SFSN_OK) return SFSN_EHIP;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(TPS * 64), lds, static_cast<hipStream_t>(stream), p, static_cast<unsigned*>(scratch), TPS, NSPL);
    return hip_ok(hipGetLastError());
}

extern "C" int sfsn_gsn_layer_scan_fused(const sfsn_scan_segment* segs, const sfsn_fused_input* fin, int n_segs, int T, int H,
                                         void* stream) {
    if (!segs || !fin || n_segs <= 0 || n_segs > SFSN_MAX_SEGMENTS || T < 0 || H <= 0) return SFSN_EINVAL;
    if (H % 16 != 0 || H <= 128 || H > 256) return SFSN_EUNSUPPORTED;  // two output tiles per wave: 8 < H/16 <= 16
    ScanParams p;
    p.wg_times = nullptr;
    p.rpw = 16;
    p.w16 = 0;
    p.lsplit = sfsn_s3_lsplit_host();
    int tiles = 0;
    const int out = 2 | (segs[0].spikes_f32 ? 1 : 0);
    for (int i = 0; i < n_segs; ++i) {
        const sfsn_scan_segment& s = segs[i];
        if (!s.spikes_i8 || (s.spikes_f32 != nullptr) != ((out & 1) != 0) || s.membrane) return SFSN_EINVAL;
        if (s.R <= 0 || !fin[i].spikes_in || !fin[i].w_ih || !fin[i].w_ih_dq || !s.w_hh || !s.w_dq || !s.bias || !s.bn_alpha ||
            !s.bn_beta || !s.h_state || !s.c_state)
            return SFSN_EINVAL;
        if (!aligned16(fin[i].spikes_in) || !aligned16(fin[i].w_ih) || !aligned16(s.w_hh) || !aligned16(s.h_state) ||
            !aligned16(s.c_state) || !aligned16(s.spikes_f32) || !aligned16(s.spikes_i8))
            return SFSN_EINVAL;
        ScanSegDev& d = p.seg[i];
        d.zin = nullptr; d.w_hh = s.w_hh; d.w_dq = s.w_dq; d.bias = s.bias; d.bn_alpha = s.bn_alpha; d.bn_beta = s.bn_beta;
        d.h_state = s.h_state; d.c_state = s.c_state; d.spikes_f32 = s.spikes_f32; d.spikes_i8 = s.spikes_i8; d.count = s.spike_count;
        d.membrane = nullptr; d.R = s.R; d.tile0 = tiles;
        d.spikes_in = fin[i].spikes_in; d.w_ih = fin[i].w_ih; d.w_ih_dq = fin[i].w_ih_dq;
        tiles += (s.R + 15) / 16;
    }
    p.nseg = n_segs; p.T = T; p.H = H; p.NT = H / 16;
    const int KS = (H + 63) / 64, HP = KS * 64;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // round 6: at most 14 tiles -> the IO-wave form (sfsn_scan3j_dev.h; SFSN_FUSED_V2=1 / SFSN_SCAN_V2=1 keep round 2's body: A/B runs, tests)
    if (p.NT <= 14 && !getenv("SFSN_FUSED_V2") && !getenv("SFSN_SCAN_V2")) {
        const bool tl = (H & 63) != 0 && (H & 63) <= 32;
        const int lds3 = KS == 3 ? Scan3jCfg<3>::lds_bytes(p.NT) : Scan3jCfg<4>::lds_bytes(p.NT);
        {
            const char* e = getenv("SFSN_S3J_LSPLIT");  // fp32 store instructions per frame the loader wave takes (A/B runs)
            const int x = e ? atoi(e) : SFSN_S3J_LSPLIT;
            p.lsplit = x < 0 ? 0 : (x > 14 ? 14 : x);
        }
#define FUSED3_CASE(KS_, TL_, OUT_)                                                                                        \
    if (KS == KS_ && (int)tl == TL_ && out == OUT_ && lds3 <= 160 * 1024 - 64) {                                           \
        auto kern = gsn_scan_fused3_kernel<KS_, TL_, OUT_>;                                                                \
        static int seen[SFSN_MAX_DEVICES] = {0};                                                                           \
        if (raise_lds(reinterpret_cast<const void*>(kern), lds3, seen) != SFSN_OK) return SFSN_EHIP;                       \
        p.wg_times = sfsn_wgprobe_take(2, tiles);                                                                          \
        hipLaunchKernelGGL(kern, dim3(tiles), dim3(1024), lds3, st, p);                                                    \
        return hip_ok(hipGetLastError());                                                                                  \
    }
        // 14 tiles (H = 224) and no fp32 spike tensor: the IO waves compute the input terms of tiles 12 / 13 (scan3j_role, OFF form).
        // Measured (B = 64, T = 1000, 52 workgroups, ms per launch, scripts/exp_s3joff_r06.py): without fp32 spikes 1.320 -> 1.238; WITH
        // them 1.333 -> 1.466 -- there the IO waves' fp32 stores (~1,400 clk per step between them) already fill SIMDs 2 / 3 up to the
        // four-tile SIMDs' level (what SFSN_S3J_LSPLIT = 6 balanced), so the form is taken for OUT = 2 only.  SFSN_S3J_OFF = 0 / 1 / 2:
        // never / OUT = 2 only (default) / always (A/B runs).
        const int offm = getenv("SFSN_S3J_OFF") ? atoi(getenv("SFSN_S3J_OFF")) : 1;
        if (p.NT == 14 && KS == 4 && tl && Scan3jCfg<4>::lds_bytes_off(14) <= 160 * 1024 - 64 && (offm == 2 || (offm == 1 && out == 2))) {
            const int ldso = Scan3jCfg<4>::lds_bytes_off(14);
#define FUSED3O_CASE(OUT_)                                                                                                 \
    if (out == OUT_) {                                                                                                     \
        auto kern = gsn_scan_fused3_kernel<4, 1, OUT_, 1>;                                                                 \
        static int seen[SFSN_MAX_DEVICES] = {0};                                                                           \
        if (raise_lds(reinterpret_cast<const void*>(kern), ldso, seen) != SFSN_OK) return SFSN_EHIP;                       \
        p.wg_times = sfsn_wgprobe_take(2, tiles);                                                                          \
        hipLaunchKernelGGL(kern, dim3(tiles), dim3(1024), ldso, st, p);                                                    \
        return hip_ok(hipGetLastError());                                                                                  \
    }
            FUSED3O_CASE(2) FUSED3O_CASE(3)
#undef FUSED3O_CASE
        }
        FUSED3_CASE(3, 0, 2) FUSED3_CASE(3, 0, 3) FUSED3_CASE(3, 1, 2) FUSED3_CASE(3, 1, 3) FUSED3_CASE(4, 1, 2) FUSED3_CASE(4, 1, 3)
#undef FUSED3_CASE
    }
    const int lds = 3 * 16 * HP + 2 * 16 * (HP + 32) + 6 * HP * 4 + 2 * p.NT * KS * 1024;
    if (lds > 160 * 1024) return SFSN_EUNSUPPORTED;
#define FUSED_CASE(KS_, OUT_)                                                                                              \
    if (KS == KS_ && out == OUT_) {                                                                                        \
        auto kern = gsn_scan_fused_kernel<KS_, OUT_>;                                                                      \
        static int seen[SFSN_MAX_DEVICES] = {0}; /* per device, raised to the largest size seen (not a stream operation) */ \
        if (raise_lds(reinterpret_cast<const void*>(kern), lds, seen) != SFSN_OK) return SFSN_EHIP;                        \
        p.wg_times = sfsn_wgprobe_take(2, tiles);                                                                          \
        hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, st, p);                                                      \
        return hip_ok(hipGetLastError());                                                                                  \
    }
    FUSED_CASE(3, 2) FUSED_CASE(3, 3) FUSED_CASE(4, 2) FUSED_CASE(4, 3)
#undef FUSED_CASE
    return SFSN_EUNSUPPORTED;
}

extern "C" int sfsn_gsn_layer_scan_fused_x(const sfsn_scan_segment* segs, const sfsn_fused_x* fin, int n_segs, int T, int H,
                                           void* stream) {
    if (!segs || !fin || n_segs <= 0 || n_segs > SFSN_MAX_SEGMENTS || T < 0 || H <= 0) return SFSN_EINVAL;
    if (H % 16 != 0 || H <= 128 || H > 256) return SFSN_EUNSUPPORTED;
    ScanParams p;
    p.wg_times = nullptr;
    p.rpw = 16;
    p.w16 = 0;
    p.lsplit = sfsn_s3_lsplit_host();
    int tiles = 0, imax = 0;
    const int out = 2 | (segs[0].spikes_f32 ? 1 : 0);
    for (int i = 0; i < n_segs; ++i) {
        const sfsn_scan_segment& s = segs[i];
        if (!s.spikes_i8 || (s.spikes_f32 != nullptr) != ((out & 1) != 0) || s.membrane) return SFSN_EINVAL;
        if (s.R <= 0 || !fin[i].x || !fin[i].w_ih || !s.w_hh || !s.w_dq || !s.bias || !s.bn_alpha || !s.bn_beta || !s.h_state ||
            !s.c_state)
            return SFSN_EINVAL;
        if (fin[i].I <= 0 || fin[i].I > 64 || fin[i].I % 2 != 0 || s.R % 16 != 0) return SFSN_EUNSUPPORTED;
        if (!aligned16(fin[i].x) || !aligned16(s.w_hh) || !aligned16(s.h_state) || !aligned16(s.c_state) || !aligned16(s.spikes_f32) ||
            !aligned16(s.spikes_i8))
            return SFSN_EINVAL;
        ScanSegDev& d = p.seg[i];
        d.zin = nullptr; d.w_hh = s.w_hh; d.w_dq = s.w_dq; d.bias = s.bias; d.bn_alpha = s.bn_alpha; d.bn_beta = s.bn_beta;
        d.h_state = s.h_state; d.c_state = s.c_state; d.spikes_f32 = s.spikes_f32; d.spikes_i8 = s.spikes_i8; d.count = s.spike_count;
        d.membrane = nullptr; d.R = s.R; d.tile0 = tiles;
        d.spikes_in = nullptr; d.w_ih = nullptr; d.w_ih_dq = nullptr;
        d.x_in = fin[i].x; d.w_ih_f32 = fin[i].w_ih; d.I = fin[i].I;
        if (fin[i].I > imax) imax = fin[i].I;
        tiles += s.R / 16;
    }
    p.nseg = n_segs; p.T = T; p.H = H; p.NT = H / 16;
    const int KS = (H + 63) / 64, HP = KS * 64;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // round 6: at most 14 tiles -> the IO-wave form (scan3y_role; SFSN_FUSED_V2=1 / SFSN_SCAN_V2=1 keep round 2's body: A/B runs, tests)
    if (p.NT <= 14 && !getenv("SFSN_FUSED_V2") && !getenv("SFSN_SCAN_V2")) {
        const bool tl = (H & 63) != 0 && (H & 63) <= 32;
        const int lds3 = KS == 3 ? Scan3yCfg<3, 2>::lds_bytes(p.NT) : Scan3yCfg<4, 2>::lds_bytes(p.NT);
        {
            const char* e = getenv("SFSN_S3Y_LSPLIT");
            const int x = e ? atoi(e) : SFSN_S3Y_LSPLIT;
            p.lsplit = x < 0 ? 0 : (x > 14 ? 14 : x);
        }
#define FUSEDX3_CASE(KS_, TL_, OUT_)                                                                                       \
    if (KS == KS_ && (int)tl == TL_ && out == OUT_ && lds3 <= 160 * 1024 - 64) {                                           \
        auto kern = gsn_scan_fusedx3_kernel<KS_, TL_, OUT_>;                                                               \
        static int seen[SFSN_MAX_DEVICES] = {0};                                                                           \
        if (raise_lds(reinterpret_cast<const void*>(kern), lds3, seen) != SFSN_OK) return SFSN_EHIP;                       \
        p.wg_times = sfsn_wgprobe_take(3, tiles);                                                                          \
        hipLaunchKernelGGL(kern, dim3(tiles), dim3(1024), lds3, st, p);                                                    \
        return hip_ok(hipGetLastError());                                                                                  \
    }
        FUSEDX3_CASE(3, 0, 2) FUSEDX3_CASE(3, 0, 3) FUSEDX3_CASE(3, 1, 2) FUSEDX3_CASE(3, 1, 3) FUSEDX3_CASE(4, 1, 2) FUSEDX3_CASE(4, 1, 3)
#undef FUSEDX3_CASE
    }
    const int xslot = (16 * imax * 4 + 1023) & ~1023;
    const int lds = 3 * xslot + 2 * 3 * 16 * 72 * 2 + 2 * 16 * (HP + 32) + 5 * HP * 4 + p.NT * KS * 1024;
    if (lds > 160 * 1024) return SFSN_EUNSUPPORTED;
#define FUSEDX_CASE(KS_, OUT_)                                                                                             \
    if (KS == KS_ && out == OUT_) {                                                                                        \
        auto kern = gsn_scan_fusedx_kernel<KS_, OUT_>;                                                                     \
        static int seen[SFSN_MAX_DEVICES] = {0}; /* per device, raised to the largest size seen (not a stream operation) */ \
        if (raise_lds(reinterpret_cast<const void*>(kern), lds, seen) != SFSN_OK) return SFSN_EHIP;                        \
        p.wg_times = sfsn_wgprobe_take(3, tiles);                                                                          \
        hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, st, p);                                                      \
        return hip_ok(hipGetLastError());                                                                                  \
    }
    FUSEDX_CASE(3, 2) FUSEDX_CASE(3, 3) FUSEDX_CASE(4, 2) FUSEDX_CASE(4, 3)
#undef FUSEDX_CASE
    return SFSN_EUNSUPPORTED;
}

static inline void pick_tiling(int NT, int& TPW, int& NWN) {
    TPW = (NT + 7) / 8;
    NWN = (NT + TPW - 1) / TPW;
}

extern "C" int sfsn_spike_proj(const int8_t* s, const int8_t* w_packed, const float* w_dq, const float* bias, float* y, int M,
                               int K, int N, int ldy, void* stream) {
    if (!s || !w_packed || !w_dq || !y || M <= 0 || K <= 0 || N <= 0 || ldy < N) return SFSN_EINVAL;
    if (!aligned16(s) || !aligned16(w_packed) || !aligned16(w_dq) || (reinterpret_cast<uintptr_t>(y) & 3u)) return SFSN_EINVAL;
    const int NT = (N + 15) / 16, KS = (K + 63) / 64;
    int TPW, NWN;
    pick_tiling(NT, TPW, NWN);
    if (TPW > 3 || KS > 5) return SFSN_EUNSUPPORTED;
    const int MW = 8 / NWN, MT = (M + 15) / 16;
    int grid = (MT + MW - 1) / MW;
    if (grid > 2048) grid = 2048;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)2 * 64 * (KS * 64 + 16) + (size_t)64 * (N + 4) * sizeof(float);
    // (a y that is only 4-byte aligned -- a chunk offset t0*R*P*4 with odd t0*R*P -- takes the generic kernel's scalar stores)
    const bool fast = (N % 4 == 0) && (ldy % 4 == 0) && M >= 64 && lds <= 150 * 1024 && aligned16(y);
    if (fast) {
        int fgrid = (M + 63) / 64;
        { const int cap = cu_count() * (TPW == 1 ? 2 : 1); if (fgrid > cap) fgrid = cap; }  // resident workgroups only: the W tiles are loaded once per workgroup
#define SPF_CASE(TPW_, KS_)                                                                                              \
    if (TPW == TPW_ && KS == KS_) {                                                                                      \
        auto kern = spike_proj_fast_kernel<TPW_, KS_>;                                                                   \
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                  \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) \
            return SFSN_EHIP;                                                                                            \
        hipLaunchKernelGGL(kern, dim3(fgrid), dim3(512), lds, st, s, w_packed, w_dq, bias, y, M, N, ldy, NT, NWN);       \
        return hip_ok(hipGetLastError());                                                                                \
    }
        SPF_CASE(1, 1) SPF_CASE(1, 2) SPF_CASE(1, 3) SPF_CASE(1, 4) SPF_CASE(1, 5)
        SPF_CASE(2, 1) SPF_CASE(2, 2) SPF_CASE(2, 3) SPF_CASE(2, 4) SPF_CASE(2, 5)
        SPF_CASE(3, 1) SPF_CASE(3, 2) SPF_CASE(3, 3) SPF_CASE(3, 4) SPF_CASE(3, 5)
#undef SPF_CASE
    }
#define SP_CASE(TPW_, KS_)                                                                                          \
    if (TPW == TPW_ && KS == KS_) {                                                                                 \
        hipLaunchKernelGGL((spike_proj_kernel<TPW_, KS_>), dim3(grid), dim3(512), 0, st, s, w_packed, w_dq, bias, y, M, N, \
                           ldy, NT, NWN);                                                                                \
        return hip_ok(hipGetLastError());                                                                           \
    }
    SP_CASE(1, 1) SP_CASE(1, 2) SP_CASE(1, 3) SP_CASE(1, 4) SP_CASE(1, 5)
    SP_CASE(2, 1) SP_CASE(2, 2) SP_CASE(2, 3) SP_CASE(2, 4) SP_CASE(2, 5)
    SP_CASE(3, 1) SP_CASE(3, 2) SP_CASE(3, 3) SP_CASE(3, 4) SP_CASE(3, 5)
#undef SP_CASE
    return SFSN_EUNSUPPORTED;
}

// block ranges of a multi-job launch: `total` workgroups dealt in proportion to the jobs' bytes, at least one and at most NS each
static void deal_blocks(const double* weight, const int* ns, int n, int total, int* nblocks) {
    double sum = 0;
    for (int i = 0; i < n; ++i) sum += weight[i];
    for (int i = 0; i < n; ++i) {
        int b = (int)(total * weight[i] / sum + 0.5);
        if (b < 1) b = 1;
        if (b > ns[i]) b = ns[i];
        nblocks[i] = b;
    }
}

extern "C" int sfsn_spike_proj_multi(const sfsn_proj_job* jobs, int n, void* stream) {
    if (!jobs || n <= 0 || n > SFSN_MAX_SEGMENTS) return SFSN_EINVAL;
    ProjMultiParams p;
    p.n = n;
    int KS0 = 0, ns[SFSN_MAX_SEGMENTS], nb[SFSN_MAX_SEGMENTS];
    double wt[SFSN_MAX_SEGMENTS];
    size_t lds = 0;
    for (int i = 0; i < n; ++i) {
        const sfsn_proj_job& j = jobs[i];
        if (!j.s || !j.w_packed || !j.w_dq || !j.y || j.M <= 0 || j.K <= 0 || j.N <= 0 || j.ldy < j.N) return SFSN_EINVAL;
        if (!aligned16(j.s) || !aligned16(j.w_packed) || !aligned16(j.w_dq) || (reinterpret_cast<uintptr_t>(j.y) & 3u)) return SFSN_EINVAL;
        const int NT = (j.N + 15) / 16, KS = (j.K + 63) / 64;
        int TPW, NWN;
        pick_tiling(NT, TPW, NWN);
        if (i == 0) KS0 = KS;
        const size_t l = (size_t)2 * 64 * (KS * 64 + 16) + (size_t)64 * (j.N + 4) * sizeof(float);
        // (only what the single entry would run on its fast kernel; anything else: the caller issues the products one by one)
        if (KS != KS0 || KS > 5 || TPW > 2 || (j.N % 4) || (j.ldy % 4) || j.M < 64 || l > 150 * 1024 || !aligned16(j.y)) return SFSN_EUNSUPPORTED;
        if (l > lds) lds = l;
        ProjJobDev& d = p.job[i];
        d.s = j.s; d.w = j.w_packed; d.dq = j.w_dq; d.bias = j.bias; d.y = j.y; d.M = j.M; d.N = j.N; d.ldy = j.ldy; d.NT = NT; d.NWN = NWN; d.tpw = TPW;
        ns[i] = (j.M + 63) / 64;
        wt[i] = (double)ns[i] * (64.0 * KS * 64 + 64.0 * j.N * 4);
    }
    deal_blocks(wt, ns, n, cu_count(), nb);
    int blocks = 0;
    for (int i = 0; i < n; ++i) { p.job[i].block0 = blocks; p.job[i].nblocks = nb[i]; blocks += nb[i]; }
    hipStream_t st = static_cast<hipStream_t>(stream);
#define SPM_CASE(KS_)                                                                                                    \
    if (KS0 == KS_) {                                                                                                    \
        auto kern = spike_proj_multi_kernel<KS_>;                                                                        \
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                  \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) \
            return SFSN_EHIP;                                                                                            \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, st, p);                                                   \
        return hip_ok(hipGetLastError());                                                                                \
    }
    SPM_CASE(1) SPM_CASE(2) SPM_CASE(3) SPM_CASE(4) SPM_CASE(5)
#undef SPM_CASE
    return SFSN_EUNSUPPORTED;
}

extern "C" int sfsn_input_proj_f32_multi(const sfsn_inproj_job* jobs, int n, void* stream) {
    if (!jobs || n <= 0 || n > SFSN_MAX_SEGMENTS) return SFSN_EINVAL;
    static const bool no_bf3 = getenv("SFSN_INPROJ_F32") != nullptr;
    if (no_bf3) return SFSN_EUNSUPPORTED;
    InProjMultiParams p;
    p.n = n;
    int ns[SFSN_MAX_SEGMENTS], nb[SFSN_MAX_SEGMENTS];
    double wt[SFSN_MAX_SEGMENTS];
    size_t lds = 0;
    for (int i = 0; i < n; ++i) {
        const sfsn_inproj_job& j = jobs[i];
        if (!j.x || !j.w || !j.z || j.M <= 0 || j.K <= 0 || j.N <= 0 || j.ldz < j.N) return SFSN_EINVAL;
        if (!aligned16(j.z)) return SFSN_EINVAL;
        const int NT = (j.N + 15) / 16;
        int TPW, NWN;
        pick_tiling(NT, TPW, NWN);
        const int KSB = j.K <= 64 ? 2 : (j.K <= 96 ? 3 : (j.K <= 160 ? 5 : 6));
        const size_t l = ((size_t)3 * 64 * (KSB * 32 + 8) * 2) + (size_t)64 * (j.N + 4) * sizeof(float);
        const bool inst = (TPW == 1 && (KSB == 2 || KSB == 3 || KSB == 5 || KSB == 6)) || (TPW == 2 && (KSB == 2 || KSB == 3 || KSB == 5));
        // (only what the single entry runs on input_proj_bf3_kernel)
        if (!inst || (j.K % 2) || j.K > 192 || (j.N % 4) || (j.ldz % 4) || j.M < 64 || TPW * KSB > 12 || l > 150 * 1024 ||
            (reinterpret_cast<uintptr_t>(j.x) & 7))
            return SFSN_EUNSUPPORTED;
        if (l > lds) lds = l;
        InProjJobDev& d = p.job[i];
        d.x = j.x; d.w = j.w; d.bias = j.bias; d.z = j.z; d.M = j.M; d.K = j.K; d.N = j.N; d.ldz = j.ldz; d.NT = NT; d.NWN = NWN; d.tpw = TPW; d.ksb = KSB;
        ns[i] = (j.M + 63) / 64;
        wt[i] = (double)ns[i] * (64.0 * j.K * 4 + 64.0 * j.N * 4);
    }
    deal_blocks(wt, ns, n, cu_count(), nb);
    int blocks = 0;
    for (int i = 0; i < n; ++i) { p.job[i].block0 = blocks; p.job[i].nblocks = nb[i]; blocks += nb[i]; }
    auto kern = input_proj_multi_kernel;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
        return SFSN_EHIP;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, static_cast<hipStream_t>(stream), p);
    return hip_ok(hipGetLastError());
}

extern "C" int sfsn_input_proj_f32(const float* x, const float* w, const float* bias, float* z, int M, int K, int N, int ldz,
                                   void* stream) {
    if (!x || !w || !z || M <= 0 || K <= 0 || N <= 0 || ldz < N) return SFSN_EINVAL;
    if (!aligned16(z)) return SFSN_EINVAL;
    const int NT = (N + 15) / 16, KC = (K + 15) / 16;
    int TPW, NWN;
    pick_tiling(NT, TPW, NWN);
    if (TPW > 3 || KC > 12) return SFSN_EUNSUPPORTED;
    const int MW = 8 / NWN, MT = (M + 15) / 16;
    int grid = (MT + MW - 1) / MW;
    if (grid > 2048) grid = 2048;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int KSB = K <= 64 ? 2 : (K <= 96 ? 3 : (K <= 160 ? 5 : 6));
    const size_t blds = ((size_t)3 * 64 * (KSB * 32 + 8) * 2) + (size_t)64 * (N + 4) * sizeof(float);
    static const bool no_bf3 = getenv("SFSN_INPROJ_F32") != nullptr;  // diagnostic: force the fp32-MFMA kernels
    if (!no_bf3 && (K % 2 == 0) && K <= 192 && (N % 4 == 0) && (ldz % 4 == 0) && M >= 64 && TPW * KSB <= 12 && blds <= 150 * 1024 &&
        (reinterpret_cast<uintptr_t>(x) & 7) == 0) {
        int fgrid = (M + 63) / 64;
        if (fgrid > cu_count()) fgrid = cu_count();  // one resident workgroup per CU: the W split is paid once per CU
#define IPB_CASE(TPW_, KS_)                                                                                               \
    if (TPW == TPW_ && KSB == KS_) {                                                                                      \
        auto kern = input_proj_bf3_kernel<TPW_, KS_>;                                                                     \
        if (blds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                  \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) \
            return SFSN_EHIP;                                                                                             \
        hipLaunchKernelGGL(kern, dim3(fgrid), dim3(512), blds, st, x, w, bias, z, M, K, N, ldz, NT, NWN);                 \
        return hip_ok(hipGetLastError());                                                                                 \
    }
        IPB_CASE(1, 2) IPB_CASE(1, 3) IPB_CASE(1, 5) IPB_CASE(1, 6) IPB_CASE(2, 2) IPB_CASE(2, 3) IPB_CASE(2, 5)
        IPB_CASE(3, 2) IPB_CASE(3, 3)
#undef IPB_CASE
    }
    const int KCB = KC <= 3 ? 3 : (KC <= 6 ? 6 : (KC <= 10 ? 10 : 12));
    const size_t flds = ((size_t)2 * 64 * (KCB * 16 + 4) + (size_t)64 * (N + 4)) * sizeof(float);
    if ((N % 4 == 0) && (ldz % 4 == 0) && M >= 64 && flds <= 150 * 1024) {
        int fgrid = (M + 63) / 64;
        if (fgrid > 512) fgrid = 512;
#define IPF_CASE(TPW_, KC_)                                                                                               \
    if (TPW == TPW_ && KCB == KC_) {                                                                                      \
        auto kern = input_proj_fast_kernel<TPW_, KC_>;                                                                    \
        if (flds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                  \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) \
            return SFSN_EHIP;                                                                                             \
        hipLaunchKernelGGL(kern, dim3(fgrid), dim3(512), flds, st, x, w, bias, z, M, K, N, ldz, NT, NWN);                 \
        return hip_ok(hipGetLastError());                                                                                 \
    }
        IPF_CASE(1, 3) IPF_CASE(1, 6) IPF_CASE(1, 10) IPF_CASE(1, 12) IPF_CASE(2, 3) IPF_CASE(2, 6) IPF_CASE(2, 10) IPF_CASE(2, 12)
        IPF_CASE(3, 3) IPF_CASE(3, 6) IPF_CASE(3, 10) IPF_CASE(3, 12)
#undef IPF_CASE
    }
#define IP_CASE(TPW_, KC_)                                                                                           \
    if (TPW == TPW_ && KCB == KC_) {                                                                                 \
        hipLaunchKernelGGL((input_proj_kernel<TPW_, KC_>), dim3(grid), dim3(512), 0, st, x, w, bias, z, M, K, N, ldz, NT, \
                           NWN);                                                                                     \
        return hip_ok(hipGetLastError());                                                                            \
    }
    IP_CASE(1, 3) IP_CASE(1, 6) IP_CASE(1, 10) IP_CASE(1, 12) IP_CASE(2, 3) IP_CASE(2, 6) IP_CASE(2, 10) IP_CASE(2, 12)
    IP_CASE(3, 3) IP_CASE(3, 6) IP_CASE(3, 10) IP_CASE(3, 12)
#undef IP_CASE
    return SFSN_EUNSUPPORTED;
}

static int fill_feat(FeatParams& p, const sfsn_feature_group* groups, int n_groups, int B, int F, int T, int FB, float fdrc,
                     bool need_x, int t0 = 0, int nt = -1) {
    if (nt < 0) nt = T;
    if (t0 < 0 || nt <= 0 || t0 + nt > T) return SFSN_EINVAL;
    p.t0 = t0; p.t1 = t0 + nt;
    if (!groups || n_groups <= 0 || n_groups > SFSN_MAX_GROUPS || B <= 0 || F < 2 || T <= 0 || FB < 0) return SFSN_EINVAL;
    const int nf = F - 1;
    p.ng = n_groups; p.B = B; p.F = F; p.T = T; p.FB = FB; p.fdrc = fdrc;
    for (int i = 0; i < n_groups; ++i) {
        const sfsn_feature_group& g = groups[i];
        if (g.n_units <= 0 || g.ctr <= 0 || g.nbr < 0 || g.ctr_fb < 0 || g.nbr_fb < 0 || g.lo < 0) return SFSN_EINVAL;
        const int I1 = g.ctr + 2 * g.nbr, I2 = g.ctr_fb > 0 ? g.ctr_fb + 2 * g.nbr_fb : 0;
        if (I1 + I2 > 256) return SFSN_EUNSUPPORTED;
        if (g.lo + g.n_units * g.ctr > nf || g.nbr >= nf || (I2 && (FB <= 0 || g.nbr_fb >= nf))) return SFSN_EINVAL;
        if (need_x && (!g.x || (g.norm == SFSN_NORM_LAYERNORM && (!g.ln_w || !g.ln_b)) || (g.norm == SFSN_NORM_LAPLACE && !g.mu) ||
                       (g.norm == SFSN_NORM_GAUSSIAN && (!g.mu || !g.ln_w))))
            return SFSN_EINVAL;
        FeatGroupDev& d = p.g[i];
        d.x = g.x; d.ln_w = g.ln_w; d.ln_b = g.ln_b; d.mu = g.mu; d.lo = g.lo; d.N = g.n_units; d.ctr = g.ctr; d.nbr = g.nbr;
        d.ctr_fb = g.ctr_fb; d.nbr_fb = g.nbr_fb; d.I1 = I1; d.I = I1 + I2; d.norm = g.norm; d.eps = g.ln_eps;
    }
    // magnitude bins some group reads (reflected at both ends exactly as the kernel does): only those are loaded
    int fmin = nf, fmax = -1;
    for (int i = 0; i < n_groups; ++i) {
        const sfsn_feature_group& g = groups[i];
        const int ends[2] = {g.lo - g.nbr, g.lo + g.n_units * g.ctr - 1 + g.nbr};
        for (int e = 0; e < 2; ++e) {
            const int f = ends[e], r = f < 0 ? -f : (f > nf - 1 ? 2 * (nf - 1) - f : f);
            const int c = f < 0 ? 0 : (f > nf - 1 ? nf - 1 : f);  // the unreflected part of the range reaches the edge
            const int lo_ = r < c ? r : c, hi_ = r > c ? r : c;
            if (lo_ < fmin) fmin = lo_;
            if (hi_ > fmax) fmax = hi_;
        }
    }
    p.f_lo = fmin; p.f_cnt = fmax - fmin + 1;
    return SFSN_OK;
}

extern "C" int sfsn_features_z(const float* stft_ri, const float* fb_tbf, int B, int F, int T, int FB, float fdrc,
                               const sfsn_feature_group* groups, int n_groups, int t0, int nt, float* zero_ptr, size_t zero_bytes, void* stream);
extern "C" int sfsn_features(const float* stft_ri, const float* fb_tbf, int B, int F, int T, int FB, float fdrc,
                             const sfsn_feature_group* groups, int n_groups, int t0, int nt, void* stream) {
    return sfsn_features_z(stft_ri, fb_tbf, B, F, T, FB, fdrc, groups, n_groups, t0, nt, nullptr, 0, stream);
}

extern "C" int sfsn_features_z(const float* stft_ri, const float* fb_tbf, int B, int F, int T, int FB, float fdrc,
                               const sfsn_feature_group* groups, int n_groups, int t0, int nt, float* zero_ptr, size_t zero_bytes, void* stream) {
    if (!stft_ri) return SFSN_EINVAL;
    if ((zero_bytes != 0 && !zero_ptr) || (zero_bytes & 15) || (reinterpret_cast<uintptr_t>(zero_ptr) & 15)) return SFSN_EINVAL;
    FeatParams p;
    int rc = fill_feat(p, groups, n_groups, B, F, T, FB, fdrc, true, t0, nt);
    if (rc != SFSN_OK) return rc;
    for (int i = 0; i < n_groups; ++i)
        if (groups[i].ctr_fb > 0 && !fb_tbf) return SFSN_EINVAL;
    const size_t lds = ((size_t)p.f_cnt * 33 + (size_t)FEAT_TT * (FB > 0 ? FB : 1) + 5 * FEAT_CHUNK) * sizeof(float);
    if (lds > 150 * 1024) return SFSN_EUNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(features_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return SFSN_EHIP;
    }
    const unsigned gx = (unsigned)((nt + FEAT_TT - 1) / FEAT_TT);
    const size_t n16 = zero_bytes / 16, zblocks = (n16 + 8 * 256 - 1) / (8 * 256);
    const size_t zrows = (zblocks + gx - 1) / gx;
    if ((size_t)B + zrows > 65535) return SFSN_EUNSUPPORTED;
    hipLaunchKernelGGL(features_kernel, dim3(gx, (unsigned)(B + zrows)), dim3(256), lds, st, stft_ri, fb_tbf, p, zero_ptr, n16);
    return hip_ok(hipGetLastError());
}

extern "C" int sfsn_laplace_means(const float* stft_ri, const float* fb_tbf, int B, int F, int T, int FB, float fdrc,
                                  const sfsn_feature_group* groups, int n_groups, float* mu_out, float* scratch, void* stream) {
    if (!stft_ri || !mu_out || !scratch) return SFSN_EINVAL;
    FeatParams p;
    int rc = fill_feat(p, groups, n_groups, B, F, T, FB, fdrc, false);
    if (rc != SFSN_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rows = B * (F - 1 + FB);
    hipLaunchKernelGGL(rowsum_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, stft_ri, fb_tbf, scratch, static_cast<double*>(nullptr), B, F, T, FB, fdrc);
    hipLaunchKernelGGL(laplace_mu_kernel, dim3(n_groups, B), dim3(64), 0, st, scratch, p, mu_out);
    return hip_ok(hipGetLastError());
}

// Per-clip mean and unbiased standard deviation for offline_gaussian_norm (FROZEN:205-218) of the gathered, un-normalised group
// tensor: mu_out, sd_out [n_groups][B].  `scratch`: 5 * B * (F - 1 + FB) floats (row sums as floats, then sums of squares and
// sums as doubles), 8-byte aligned.
extern "C" int sfsn_gaussian_stats(const float* stft_ri, const float* fb_tbf, int B, int F, int T, int FB, float fdrc,
                                   const sfsn_feature_group* groups, int n_groups, float* mu_out, float* sd_out, float* scratch, void* stream) {
    if (!stft_ri || !mu_out || !sd_out || !scratch || (reinterpret_cast<uintptr_t>(scratch) & 7u)) return SFSN_EINVAL;
    if (T < 2) return SFSN_EINVAL;  // (the unbiased estimate needs two values; every group tensor of a clip has >= T of them)
    FeatParams p;
    int rc = fill_feat(p, groups, n_groups, B, F, T, FB, fdrc, false);
    if (rc != SFSN_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rows = B * (F - 1 + FB);
    double* rs2 = reinterpret_cast<double*>(scratch + (size_t)((rows + 1) & ~1));
    hipLaunchKernelGGL(rowsum_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, stft_ri, fb_tbf, scratch, rs2, B, F, T, FB, fdrc);
    hipLaunchKernelGGL(gaussian_stats_kernel, dim3(n_groups, B), dim3(64), 0, st, rs2, p, mu_out, sd_out);
    return hip_ok(hipGetLastError());
}

extern "C" int sfsn_cum_laplace_norm(float* x, int T, int R, int I, float* cum_state, int frames_before, float* scratch, void* stream) {
    if (!x || !scratch || T <= 0 || R <= 0 || I <= 0 || frames_before < 0) return SFSN_EINVAL;
    if (I > 256) return SFSN_EUNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rows = T * R;
    hipLaunchKernelGGL(cumlap_rowsum_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, scratch, rows, I);
    hipLaunchKernelGGL(cumlap_scan_kernel, dim3((R + 63) / 64), dim3(64), 0, st, scratch, cum_state, T, R, I, frames_before);
    const size_t n = (size_t)rows * I;
    hipLaunchKernelGGL(cumlap_divide_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, scratch, n, I);
    return hip_ok(hipGetLastError());
}

extern "C" int sfsn_spike_count(const sfsn_count_tensor* tensors, int n_tensors, void* stream) {
    if (!tensors || n_tensors <= 0 || n_tensors > SFSN_MAX_COUNT_TENSORS) return SFSN_EINVAL;
    CountParams p;
    p.n = n_tensors;
    long long blocks = 0;
    for (int i = 0; i < n_tensors; ++i) {
        const sfsn_count_tensor& t = tensors[i];
        if (!t.spikes_i8 || !t.count || t.n_bytes == 0 || (t.n_bytes & 15) || (reinterpret_cast<uintptr_t>(t.spikes_i8) & 15))
            return SFSN_EINVAL;
        p.src[i] = t.spikes_i8; p.dst[i] = t.count; p.nvec[i] = t.n_bytes / 16;
        p.blk0[i] = (int)blocks;
        blocks += (long long)((p.nvec[i] + 4095) / 4096);
        if (blocks > 0x7fffffffLL) return SFSN_EUNSUPPORTED;
    }
    p.blk0[n_tensors] = (int)blocks;
    hipLaunchKernelGGL(spike_count_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return hip_ok(hipGetLastError());
}

// ---- streaming: input history of a session ---------------------------------------------------------------
// hist [rows][D + hop] complex64 (rows = B * F): drop the oldest `hop` frames, append the new ones -- one launch instead of three
// elementwise copies per hop.  A thread owns one row: it reads the D frames it keeps and its `hop` new frames into registers
// before it writes anything, so the shift is safe in place.
#define SFSN_HIST_MAX 16
__global__ __launch_bounds__(256) void hist_shift_kernel(float2* __restrict__ hist, const float2* __restrict__ inp, int rows, int D, int hop) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float2* h = hist + (size_t)r * (D + hop);
    const float2* x = inp + (size_t)r * hop;
    float2 keep[SFSN_HIST_MAX];
#pragma unroll
    for (int i = 0; i < SFSN_HIST_MAX; ++i)
        if (i < D + hop) keep[i] = i < D ? h[hop + i] : x[i - D];
#pragma unroll
    for (int i = 0; i < SFSN_HIST_MAX; ++i)
        if (i < D + hop) h[i] = keep[i];
}

extern "C" int sfsn_hist_shift(float* hist_ri, const float* inp_ri, int rows, int D, int hop, void* stream) {
    if (!hist_ri || !inp_ri || rows <= 0 || D < 0 || hop <= 0) return SFSN_EINVAL;
    if (D + hop > SFSN_HIST_MAX) return SFSN_EUNSUPPORTED;
    hipLaunchKernelGGL(hist_shift_kernel, dim3((rows + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<float2*>(hist_ri), reinterpret_cast<const float2*>(inp_ri), rows, D, hop);
    return hip_ok(hipGetLastError());
}

extern "C" int sfsn_deepfilter(const float* stft_ri, int B, int F, int T, int S, const sfsn_df_group* groups, int n_groups,
                               float* enh_ri, float* enh_mag, int t0, int nt, void* stream) {
    if (!stft_ri || !enh_ri || !groups || n_groups <= 0 || n_groups > SFSN_MAX_GROUPS || B <= 0 || F < 2 || T <= 0 || S <= 0)
        return SFSN_EINVAL;
    if (t0 < 0 || nt <= 0 || t0 + nt > T) return SFSN_EINVAL;
    DfParams p;
    p.ng = n_groups; p.B = B; p.F = F; p.T = T; p.S = S; p.t0 = t0; p.t1 = t0 + nt;
    int lo = 0, maxP = 1;
    for (int i = 0; i < n_groups; ++i) {
        const sfsn_df_group& g = groups[i];
        if (!g.proj || g.n_units <= 0 || g.fc <= 0 || g.df <= 0) return SFSN_EINVAL;
        p.g[i].proj = g.proj; p.g[i].N = g.n_units; p.g[i].fc = g.fc; p.g[i].df = g.df; p.g[i].lo = lo;
        lo += g.n_units * g.fc;
        const int P = 2 * g.fc * g.df * S;
        if (P > maxP) maxP = P;
    }
    if (lo > F) return SFSN_EINVAL;
    p.fcov = lo;
    hipStream_t st = static_cast<hipStream_t>(stream);
    {   // pass-structured kernel: every P a multiple of 4 (16-byte coefficient loads), tiles within 48 KB of LDS
        static const bool no_pass = getenv("SFSN_DF_GENERIC") != nullptr;  // diagnostic: force the unit-by-unit kernel
        DfPassParams pp;
        pp.base = p;
        pp.npass = 0;
        bool ok = !no_pass;
        int max_up = 0, max_x = 0;
        for (int i = 0; i < n_groups && ok; ++i) {
            const sfsn_df_group& g = groups[i];
            const int P = 2 * g.fc * g.df * S;
            if (P % 4 != 0 || P > 2048 || (reinterpret_cast<uintptr_t>(g.proj) & 15) != 0) { ok = false; break; }
            int umax = 192 / P;  // <= 24.6 KB of coefficients per pass
            if (umax < 1) umax = 1;
            if (umax > 255) umax = 255;
            const int np = (g.n_units + umax - 1) / umax, U = (g.n_units + np - 1) / np;
            for (int k0 = 0; k0 < g.n_units; k0 += U) {
                if (pp.npass >= DF_MAX_PASSES || k0 > 255 || i > 255) { ok = false; break; }
                const int nu = (g.n_units - k0 < U) ? g.n_units - k0 : U;
                pp.pg[pp.npass] = (unsigned char)i; pp.pk0[pp.npass] = (unsigned char)k0; pp.pnu[pp.npass] = (unsigned char)nu;
                ++pp.npass;
                if (nu * P > max_up) max_up = nu * P;
                const int xw = nu * g.fc * (32 + g.df - 1) * 2;
                if (xw > max_x) max_x = xw;
            }
        }
        if (ok) {
            pp.ctile_floats = (32 * (max_up + 1) + 3) & ~3;
            const size_t plds = ((size_t)pp.ctile_floats + max_x) * sizeof(float);
            if (plds <= 48 * 1024) {
                hipLaunchKernelGGL(deepfilter_pass_kernel, dim3((nt + 31) / 32, B), dim3(DFP_THREADS), plds, st, stft_ri, pp, enh_ri, enh_mag);
                return hip_ok(hipGetLastError());
            }
        }
    }
    const size_t lds = (size_t)32 * (maxP + 1) * sizeof(float);
    if (lds > 150 * 1024) return SFSN_EUNSUPPORTED;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(deepfilter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return SFSN_EHIP;
    }
    hipLaunchKernelGGL(deepfilter_kernel, dim3((nt + 31) / 32, B), dim3(256), lds, st, stft_ri, p, enh_ri, enh_mag);
    return hip_ok(hipGetLastError());
}

#ifdef IP_STAMPS
extern "C" int sfsn_ip_debug(unsigned long long* out /* [8] host */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ip_dbg), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#endif
