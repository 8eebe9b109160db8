// sfsn_scan3_dev.h -- the recurrent scan with IO-specialised waves (round 3), gfx950 only.
//
// What round 2's scan_body (sfsn_scan_dev.h) paid for, measured with scripts/micro/scan_step_floor.hip (B=64 geometry, H=224):
//   * the step WITHOUT any global traffic costs 1235 / 1539 / 2000 clk at 4 / 8 / 16 rows per workgroup; the real kernel ran
//     0.75 / 0.91 / 0.97 us (1600 / 1960 / 2080 clk): every wave mixed input-term DMAs and spike stores into ONE in-order vmcnt
//     queue, so the wait for a DMA was a wait for the stores issued before it -- and with write-through (sc1) stores for an
//     in-launch consumer the producer slowed from 0.97 to 1.45 us per step;
//   * at 8 rows per workgroup the epilogue ran on all 16 MFMA columns (4 values per lane, half of them duplicates).
// Here the 16 waves of a workgroup have three jobs:
//   compute waves (one 16-neuron output tile each, wave < NT <= 14): B fragments from LDS, 12 MFMAs, epilogue on exactly the live
//       values (re-dealt with DPP row shifts: 4 / 2 / 1 values per lane at 16 / 8 / 4 rows), new spikes to LDS, barrier.
//       No global memory instruction in the loop, per-neuron constants in registers, ~35 VALU per step at 8 rows.
//   the loader wave (wave NT): the input term of frame t+D-1 -> LDS ring by LDS-DMA (only loads in its vmcnt queue: one counted
//       wait per step); in a stack launch it is the only wave that polls the producers' progress counters.
//   the storer wave (wave NT+1): the spikes of frame t-1, LDS -> global as whole contiguous blocks (fp32 + int8); in a stack
//       launch it writes the int8 rows write-through and publishes progress from its OWN store queue (only stores in it).
// One s_barrier per step for all 16 waves, as before.  Arithmetic is scan_body's, value for value (same three exact integer
// accumulators, recombined exactly -- (a2 << 16) + (a1 << 8) + a0 fits int32 for K <= 256 and is rounded once by the
// conversion, which is what recombine3's fma does -- same fma / exp2 / rcp sequence): bit-identical outputs (tested).
// Shared gate weights (G = 1), H <= 224 (two spare waves), 4 / 8 / 16 rows per workgroup.
#ifndef SFSN_SCAN3_DEV_H
#define SFSN_SCAN3_DEV_H
#include "sfsn_scan_dev.h"

// Ring depth by role (FLG bit 0: input written by other workgroups of the launch, bit 1: output read by others):
//   * plain role: the loader's queue holds DMAs only, 2 frames in flight cover a settled read (D = 4);
//   * gated: an sc1 load of data another workgroup has just written through takes ~3 us -- 2 frames in flight made the
//     consumer run at 1.5 us per frame on an idle chip (scripts/exp_stack_r03.sh); as deep as the 64 KiB LDS-DMA window and
//     the 6-bit vmcnt allow;
//   * publishing: the write-through int8 stores take ~5 us to retire and vmcnt retires in order, so they get a queue of their
//     own (the storer wave: 2 stores per frame, 8 frames in flight) and the fp32 spike stores move to the loader wave, whose
//     ring is deep enough that the DMA it waits for is older than any store it has issued in the last 4 steps.
#ifndef SFSN_S3_LSF
#define SFSN_S3_LSF 1  // 1: a publishing role's fp32 spikes are written by the loader wave (its storer's queue holds the write-through int8 stores only)
#endif
#ifndef SFSN_S3_CWF
// 1: the COMPUTE waves write the fp32 spikes themselves, one small store per wave and step (the 64-byte row segments of a tile; the
// L2 merges them into lines) -- a round-4 experiment, OFF.  Why it was tried: ONE wave issues a 1 KiB store every 0.058 us
// (17 KB/us, scripts/micro/store_retire.hip: idle chip or 208 workgroups alike, whatever the queue depth), so an 8-row role whose
// nine stores per frame leave through one IO wave spends 0.52 us per step issuing them: the publishing layer-1 role of a pair launch runs at 0.9 us per step on an
// idle chip whichever IO wave carries the seven fp32 stores of a frame (scripts/exp_stack_direct.py, ROWS=64), and the plain
// 8-row scan (nine stores per frame through the storer) sits at 0.69.  With fourteen store queues the publishing role ran at
// 0.63 -- but the compute-bound roles lost far more than that: one VMEM store per compute wave and step in front of the step
// barrier cost them 0.1-0.2 us per step, at the top of the next step (under the LDS wait) 0.5-0.9 us (4-row scan 0.55 -> 1.4).
#define SFSN_S3_CWF 0
#endif
#ifndef SFSN_S3_LSPLIT
// Round 5: at 8 rows per workgroup a frame's fp32 spikes are SEVEN 1 KiB stores and its int8 rows two; one wave issues a VMEM store
// every ~140 clk and an LDS-DMA piece every ~60 (scripts/micro/store_retire.hip, MI355X_MICROARCH.md), so whichever IO wave carried all
// of them was the role's step: the plain role's storer 9 x 140 = 1260 clk (0.69 us per step measured against 0.61 without the fp32
// tensor), the publishing role's loader 7 x 60 + 7 x 140 = 1400 clk (0.9 us measured on an idle chip).  The loader wave takes the first
// SFSN_S3_LSPLIT store instructions of a frame's fp32 block and the storer wave the rest: 7 x 60 + 3 x 140 = 840 against (4 + 2) x 140 =
// 840 clk -- neither IO wave is the step any more.  (Scan3Role::lsplit overrides it at run time for A/B runs: SFSN_S3_LSPLIT in the
// environment of the host library.)  Measured (scripts/exp_lsplit_r05.sh, B = 64, T = 1000, the pair launch as one whole-sequence
// launch; run-to-run noise ~2 %): loader share 0 / 1 / 2 / 3 / 7 of the scan3 roles x 0 / 1 / 2 of the FUSEDX3 role: 1.08-1.09 ms at
// (0, 0), 1.03-1.05 at (2, 1), (2, 2), (1, 1), 1.08-1.12 at (3, 0), (3, 1), 1.05-1.09 at (7, *) -- four per cent, not the forty
// the issue arithmetic promised: with the IO waves relieved the pair runs at its FUSED3 roles' compute floor (0.9-0.93 us per step
// alone).  That held while every counted wait of an IO wave cost ~400 clk (the 64-case switch behind wait_vmcnt_n); with
// the wait as a computed jump (~80 clk) the storer wave has the time for all of a frame's stores and the split only
// lengthens the loader's path: measured again at the end of round 5 (three interleaved rounds, scripts/exp_lsplit_r05.sh): (0, 0)
// 0.971-0.983 ms = 0.237-0.241 of the HBM roofline, (1, 0) 0.988-1.000, (1, 1) 0.994-1.004, (2, 1) 1.005-1.015; strict forward 2.49-2.53
// against 2.53-2.58; the timed region and B = 16 / 32 unchanged (scripts/exp_lsplit_ab2_r05.sh).  Defaults: 0 / 0 (no split).
#define SFSN_S3_LSPLIT 0
#endif
#ifndef SFSN_S3X_LSPLIT
#define SFSN_S3X_LSPLIT 0
#endif
#ifndef SFSN_S3_NT
// 1: the fp32 spike tensors (the module API's all_layer_outputs: written once, never read again on the device) leave as NON-TEMPORAL
// stores: 1.5 GB of write-once lines per forward need not displace what the L2 holds for the full-band stack's hand-offs and the
// time-parallel kernels.  Measured (round 6, interleaved A/B, scripts/ab_lib_r06.sh): strict forward 2.44-2.45 against 2.49-2.56 ms, pair
// launch 0.956-0.965 against 0.973 ms, timed region unchanged.
#define SFSN_S3_NT 1
#endif
#ifndef SFSN_S3_PFMAX
#define SFSN_S3_PFMAX 8   // frames of a publishing storer's stores that may be in flight (24 measured the same: the limit is elsewhere)
#endif
// the host's value of the split (the environment variable of the same name overrides the built-in default: A/B runs)
inline int sfsn_s3_lsplit_host() {
    static const int v = [] {
        const char* e = getenv("SFSN_S3_LSPLIT");
        const int x = e ? atoi(e) : SFSN_S3_LSPLIT;
        return x < 0 ? 0 : (x > 16 ? 16 : x);
    }();
    return v;
}
inline int sfsn_s3x_lsplit_host() {  // the FUSEDX3 role's own value (its loader wave also converts the features: less room)
    static const int v = [] {
        const char* e = getenv("SFSN_S3X_LSPLIT");
        const int x = e ? atoi(e) : SFSN_S3X_LSPLIT;
        return x < 0 ? 0 : (x > 16 ? 16 : x);
    }();
    return v;
}
template <int KS, int RPW, int FLG = 0>
struct Scan3Cfg {
    static constexpr int HP = KS * 64, LDH = HP + 32;
    // a frame's input term for the workgroup's RPW rows as 16-byte chunks, chunk e = (tile * 4 + q) * RPW + row: the chunks a
    // compute wave reads are contiguous (RPW * 64 bytes per tile) and a DMA piece (64 chunks = 1 KiB) covers 64 / RPW quads
    __host__ __device__ static constexpr int chunks(int NT) { return RPW * NT * 4; }
    __host__ __device__ static constexpr int pieces(int NT) { return (chunks(NT) + 63) / 64; }
    __host__ __device__ static constexpr int slot_bytes(int NT) { return pieces(NT) * 1024; }
    static constexpr int MAXP = (RPW * 14 * 4 + 63) / 64;  // pieces at NT = 14
    static constexpr bool GATED = (FLG & 1) != 0, PUB = (FLG & 2) != 0;
    // (8 rows: the loader wave also carries a share of the fp32 spike stores -- SFSN_S3_LSPLIT below -- and needs the deeper ring for
    //  the same reason as the publishing role's)
    static constexpr int DWANT = (PUB && SFSN_S3_LSF) ? 6 : ((GATED || PUB) ? 9 : (RPW == 8 ? 6 : 4));
    static constexpr int DFIT = 65536 / (MAXP * 1024);       // LDS-DMA destinations stay below 64 KiB
    static constexpr int D = DWANT < DFIT ? DWANT : DFIT;    // input-term ring depth (frames)
    __host__ __device__ static constexpr int hbuf_off(int NT) { return D * slot_bytes(NT); }
    __host__ __device__ static constexpr int flag_off(int NT) { return hbuf_off(NT) + 2 * 16 * LDH; }
    __host__ __device__ static constexpr int lds_bytes(int NT) { return flag_off(NT) + 16; }
    static constexpr int NV = RPW == 16 ? 4 : RPW == 8 ? 2 : 1;  // live values per lane and tile
};

// fp32 spikes of a compute wave: NV consecutive floats per lane at byte offset `off` of the frame's block (saddr form: the frame base
// is wave-uniform).  Fire and forget: compute waves never wait on vmcnt.  Issued at the TOP of the next step, while the wave waits
// for its state fragments from LDS anyway: a store in front of the step barrier cost the compute-bound roles 0.1-0.2 us per step
// (the issue of a VMEM instruction, address + data registers through the texture path, sits on every wave's way to the barrier).
template <int NV>
__device__ __forceinline__ void s3_store_spikes(const float* base_uniform, unsigned off, const float (&sp)[NV]) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base_uniform);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned long long base = ((unsigned long long)hi << 32) | lo;
    if constexpr (NV == 1) {
        asm volatile("global_store_dword %0, %1, %2" ::"v"(off), "v"(sp[0]), "s"(base));
    } else if constexpr (NV == 2) {
        typedef float v2f_ __attribute__((ext_vector_type(2)));
        const v2f_ d = {sp[0], sp[1]};
        asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(off), "v"(d), "s"(base));
    } else {
        const v4f d = {sp[0], sp[1], sp[2], sp[3]};
        asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(off), "v"(d), "s"(base));  // (see store16_sc1)
    }
}

// The gate of a loader wave: frames [0, need) published by ALL producers in lk.in[0 .. n_in)?  Polled by lane 0 with relaxed
// agent-scope loads and s_sleep between polls; the other waves of the workgroup never see the counters.  `avail` caches the
// last answer (producers run `lag` frames ahead before the consumer resumes, so that one poll covers many frames); a bounded
// spin that expires sets the launch's error word and `failed`.
__device__ __forceinline__ void s3_ensure(const StackLink& lk, int need, int T, int& avail, int& failed, int lane) {
    if (need > avail && !failed) {
        int vv = 0;
        if (lane == 0) {
            const int want = (need + lk.lag < T) ? need + lk.lag : T;
            for (unsigned spins = 0;; ++spins) {
                vv = 0x7fffffff;
                for (int i = 0; i < lk.n_in; ++i) {
                    const int pi = (int)__hip_atomic_load(lk.in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    vv = pi < vv ? pi : vv;
                }
                if (vv >= want) break;
                if (spins > SFSN_STACK_SPIN_LIMIT) {
                    __hip_atomic_store(lk.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    vv = -1;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
                if (lk.dbg) lk.dbg[1] += 1;
            }
            if (lk.dbg) lk.dbg[0] += 1;
        }
        vv = __builtin_amdgcn_readfirstlane(vv);
        if (vv < 0) failed = 1;
        else avail = vv;
    }
}

// fp32 spikes of one frame, LDS (int8, RPW rows x H) -> global: the block of RPW rows x H floats is contiguous in [T][R][H], so
// store k of a wave writes bytes [1024 k + 16 lane, +16) of it.  Rows past R are skipped.
template <int RPW, int LDH>
struct S3FlushF {
    static constexpr int MAXF = (RPW * 14 * 4 + 63) / 64;
    int lf[MAXF];
    unsigned okf;
    int nsf;  // store instructions per frame with at least one live lane (wave-uniform)
    // store instructions [k_lo, k_hi) of the frame's block are mine (the loader and the storer wave share a frame's stores)
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
    __device__ __forceinline__ void run(const int8_t* hsrc, float* pf /* block base of the frame */, int lane) const {
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

struct Scan3Role {
    const float* zin;
    const int8_t* w_hh;
    const float* w_dq;
    const float* bias;
    const float* bn_alpha;
    const float* bn_beta;
    float* h_state;
    float* c_state;
    float* spikes_f32;
    int8_t* spikes_i8;
    int R, row0;
    unsigned long long* count = nullptr;  // nullable: a role without fp32 spikes adds the number of spikes it wrote (ScanSegDev::count)
    int lsplit = SFSN_S3_LSPLIT;          // 8-row roles: fp32 store instructions per frame issued by the loader wave (the storer takes the rest)
#ifdef SFSN_EXPERIMENTS
    unsigned long long* probe = nullptr;  // [wave][4] shader-clock cycles: at the step barrier / in counted vmcnt waits / in hand-off polls / in all
#endif
};

// ---- per-wave stall accounting (make EXTRA=-DSFSN_EXPERIMENTS; scripts/exp_beside_r05.py): which wave of a workgroup is the one the
// others wait for at the step barrier, and what that wave waits for itself.  Not compiled into the product.
#ifdef SFSN_EXPERIMENTS
#define S3_PB_DECL() unsigned long long pb_[3] = {0, 0, 0}, pb_a_ = 0; const unsigned long long pb_0_ = __builtin_amdgcn_s_memtime()
#define S3_PB_TIC() pb_a_ = __builtin_amdgcn_s_memtime()
#define S3_PB_TOC(k) pb_[k] += __builtin_amdgcn_s_memtime() - pb_a_
#define S3_PB_OUT(rl, wave, lane)                                                                                        \
    do {                                                                                                                 \
        if ((rl).probe && (lane) == 0) {                                                                                 \
            unsigned long long* q_ = (rl).probe + 4 * (wave);                                                            \
            q_[0] = pb_[0]; q_[1] = pb_[1]; q_[2] = pb_[2]; q_[3] = __builtin_amdgcn_s_memtime() - pb_0_;                 \
        }                                                                                                                \
    } while (0)
#else
#define S3_PB_DECL() do {} while (0)
#define S3_PB_TIC() do {} while (0)
#define S3_PB_TOC(k) do {} while (0)
#define S3_PB_OUT(rl, wave, lane) do {} while (0)
#endif

// FLG bit 0: the input term is written by other workgroups of this launch (gated on lk.in, sc1 loads); bit 1: the int8 spikes
// feed other workgroups of this launch (sc1 stores, progress in lk.out).  OUT bit 0: fp32 spikes, bit 1: int8 spikes.
// D0 = 1: the weights were packed with 16 bits (sfsn_w3_pack_bits): digit plane 0 is zero and its products are skipped -- 8 instead
// of 12 matrix instructions per tile and step, the same sums.
template <int KS, int RPW, int OUT, int FLG, int D0 = 0>
__device__ __forceinline__ void scan3_role(const Scan3Role& rl, const StackLink& lk, char* smem, int T, int H, int NT, int exp_flags = 0) {
    using C = Scan3Cfg<KS, RPW, FLG>;
    constexpr int LDH = C::LDH, HP = C::HP, D = C::D, NV = C::NV;
    constexpr bool GATED = (FLG & 1) != 0, PUB = (FLG & 2) != 0;
#ifndef SFSN_S3_LSF
#define SFSN_S3_LSF 0
#endif
    constexpr bool CWF = SFSN_S3_CWF && (OUT & 1);  // the compute waves write the fp32 spikes (see SFSN_S3_CWF)
    // the loader wave writes fp32 spikes too: ALL of them in a publishing role at 4 / 16 rows (round 3; see Scan3Cfg), its share of
    // every frame at 8 rows (SFSN_S3_LSPLIT)
    constexpr bool LSPLIT = RPW == 8 && !CWF && (OUT & 1);
    constexpr bool LSF = !CWF && (OUT & 1) && ((SFSN_S3_LSF && PUB) || LSPLIT);
    const int ltake = LSPLIT ? rl.lsplit : 64;  // store instructions per frame the loader takes (64 = all)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int R = rl.R, row0 = rl.row0;
    const int SLOT = C::slot_bytes(NT);
    int8_t* hbuf = reinterpret_cast<int8_t*>(smem + C::hbuf_off(NT));
    volatile int* flag = reinterpret_cast<volatile int*>(smem + C::flag_off(NT));

    // ---- set-up by all threads: hidden-state buffers zeroed (pad rows / columns must read as 0 spikes), h_{-1} -> hbuf[0]
    for (int i = tid; i < 2 * 16 * LDH / 4; i += 1024) reinterpret_cast<int*>(hbuf)[i] = 0;
    if (tid == 0) { flag[0] = 0; flag[1] = 0; }
    __syncthreads();
    for (int idx = tid; idx < RPW * (H / 4); idx += 1024) {
        const int rr = idx / (H / 4), j4 = (idx - rr * (H / 4)) * 4;
        // (rows past R are duplicates of row R-1 in every value -- input term, membrane, last spikes --, so that whatever a lane of
        //  such a row stores goes to row R-1's address with row R-1's value)
        const int rsrc = row0 + rr < R ? row0 + rr : R - 1;
        const v4f h = *reinterpret_cast<const v4f*>(rl.h_state + (size_t)rsrc * H + j4);
        const unsigned pk = (h.x > 0.5f ? 1u : 0u) | (h.y > 0.5f ? 0x100u : 0u) | (h.z > 0.5f ? 0x10000u : 0u) |
                            (h.w > 0.5f ? 0x1000000u : 0u);
        *reinterpret_cast<unsigned*>(hbuf + rr * LDH + j4) = pk;
    }

    if (wave < NT) {
        // ================================================= compute wave: output tile `wave` =================================================
        const int ct = wave;
        const int row = RPW == 16 ? n : RPW == 8 ? (n & 7) : (n & 3);
        const int sub = RPW == 16 ? 0 : RPW == 8 ? 2 * (n >> 3) : (n >> 2);   // first of my NV neurons within the 4q group
        const int cj = ct * 16 + q * 4 + sub;                                  // my first neuron
        const bool live = row0 + row < R;
        const int grow = live ? row0 + row : R - 1;
        v4i W[KS][3];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int d = 0; d < 3; ++d)
                W[ks][d] = *reinterpret_cast<const v4i*>(rl.w_hh + ((((size_t)d * NT + ct) * KS + ks) * 64 + lane) * 16);
        float c[NV], dq[NV], db[NV], al[NV], be[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            c[j] = rl.c_state[(size_t)grow * H + cj + j];
            dq[j] = rl.w_dq[cj + j];
            db[j] = rl.bias[H + cj + j] - rl.bias[cj + j];
            al[j] = rl.bn_alpha[cj + j];
            be[j] = rl.bn_beta[cj + j];
        }
        const unsigned zoff = (unsigned)(((ct * 4 + q) * RPW + row) * 16 + sub * 4);  // my input-term bytes within a ring slot
        const unsigned boff = (unsigned)(n * LDH + q * 16);
        const unsigned hoff = (unsigned)(row * LDH + cj);
        const unsigned foff = (unsigned)(((size_t)grow * H + cj) * 4);  // my fp32 spikes within a frame of [T][R][H] (R * H * 4 < 4 GiB)
        const size_t fframe = (size_t)R * H;
        __syncthreads();                       // initial state in hbuf[0]
        __builtin_amdgcn_s_barrier();          // the loader's prologue frames have landed
        int stop = 0;
        float sp[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) sp[j] = 0.f;
        S3_PB_DECL();
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            const int8_t* hc = hbuf + (t & 1) * 16 * LDH;
            int8_t* hn = hbuf + ((t & 1) ^ 1) * 16 * LDH;
            v4i b[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) b[ks] = *reinterpret_cast<const v4i*>(hc + boff + ks * 64);
            if constexpr (CWF) if (t > 0) s3_store_spikes<NV>(rl.spikes_f32 + (size_t)(t - 1) * fframe, foff, sp);  // (under the LDS wait)
            const char* zp = smem + (t % D) * SLOT + zoff;
            float z[NV];
            if constexpr (NV == 4) {
                const v4f zz = *reinterpret_cast<const v4f*>(zp);
                z[0] = zz.x; z[1] = zz.y; z[2] = zz.z; z[3] = zz.w;
            } else if constexpr (NV == 2) {
                typedef float v2f __attribute__((ext_vector_type(2)));
                const v2f zz = *reinterpret_cast<const v2f*>(zp);
                z[0] = zz.x; z[1] = zz.y;
            } else {
                z[0] = *reinterpret_cast<const float*>(zp);
            }
            if constexpr (GATED) stop = flag[t & 1];  // written by the loader during step t-1 (or before)
            v4i a[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int d = D0; d < 3; ++d) a[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[ks][d], b[ks], a[d], 0, 0, 0);
            // (tried in round 3: MFMA columns beyond the RPW rows as DUPLICATES of the live ones -- same LDS address, a broadcast -- so
            //  that the re-deal below becomes a per-lane v_cndmask select instead of DPP row shifts: 0.611 / 0.744 us per step at
            //  4 / 8 rows against 0.579 / 0.690 with zero columns and DPP; zero operands also cost the matrix pipe less power)
            int v[3][NV];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if constexpr (RPW == 16) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[d][j] = a[d][j];
                } else if constexpr (RPW == 8) {
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
                const float rec = (float)((v[2][j] << 16) + (v[1][j] << 8) + v[0][j]);  // exact sum, rounded once (= recombine3)
                const float pre_f = __builtin_fmaf(rec, dq[j], z[j]);
                const float pre_g = pre_f + db[j];
                const float f = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre_f * -1.44269504088896341f));
                const float m = __builtin_fmaf(f, c[j] - pre_g, pre_g);
                const float y = __builtin_fmaf(m, al[j], be[j]);
                c[j] = y;
                pk |= (y >= 0.0f) ? (1u << (8 * j)) : 0u;
                sp[j] = (y >= 0.0f) ? 1.0f : 0.0f;
            }
            if constexpr (NV == 4) *reinterpret_cast<unsigned*>(hn + hoff) = pk;
            else if constexpr (NV == 2) *reinterpret_cast<unsigned short*>(hn + hoff) = (unsigned short)pk;
            else hn[hoff] = (int8_t)pk;
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            S3_PB_TIC();
            __builtin_amdgcn_s_barrier();
            S3_PB_TOC(0);
            if constexpr (GATED) if (__builtin_amdgcn_readfirstlane(stop)) break;
        }
        S3_PB_OUT(rl, wave, lane);
        if constexpr (CWF) if (T > 0 && !(GATED && __builtin_amdgcn_readfirstlane(stop))) s3_store_spikes<NV>(rl.spikes_f32 + (size_t)(T - 1) * fframe, foff, sp);
        // final state
        const int8_t* hl = hbuf + (T & 1) * 16 * LDH;
        if (live) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                rl.c_state[(size_t)grow * H + cj + j] = c[j];
                rl.h_state[(size_t)grow * H + cj + j] = (float)hl[hoff + j];
            }
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
            goff[p] = (unsigned)((grow * H + cidx * 4) * 4);
        }
        const size_t frame = (size_t)R * H;
        int avail = GATED ? 0 : T;
        int failed = 0;
        S3FlushF<RPW, LDH> ff;
        if constexpr (LSF) ff.init(lane, row0, R, H, 0, ltake);
        // operations issued after the DMAs of a frame and before the wait D-2 steps later: D-2 steps of DMAs (and stores)
        // plus the stores of the issuing step itself; the counter has 6 bits (a smaller allowance is only stricter)
        int allow = (D - 2) * np;
        if constexpr (LSF) allow = (D - 2) * (np + ff.nsf) + ff.nsf;
        if (allow > 62) allow = 62;
        auto ensure = [&](int need) __attribute__((always_inline)) {  // frames [0, need) published by all my producers
            if constexpr (GATED) s3_ensure(lk, need, T, avail, failed, lane);
        };
        auto issue = [&](int slot, int td) __attribute__((always_inline)) {
            const float* zt = rl.zin + (size_t)td * frame;
#pragma unroll
            for (int p = 0; p < C::MAXP; ++p)
                if (p < np) dma16_to_lds<GATED>(__builtin_amdgcn_readfirstlane((unsigned)(slot * SLOT + p * 1024)), zt, goff[p]);  // wave-uniform
        };
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
            if constexpr (GATED) stop = failed;  // what the other waves read from flag[t & 1] during this step
            const int td = (t + D - 1 < T) ? t + D - 1 : T - 1;
            S3_PB_TIC();
            ensure(td + 1);
            S3_PB_TOC(2);
            if (!failed) issue((t + D - 1) % D, td);
            // a failure is published in the word the OTHER parity reads: written during step t, read during step t+1 (a word
            // read during the step it is written in would be seen by some waves and not by others)
            if constexpr (GATED) if (failed && lane == 0) flag[(t + 1) & 1] = 1;
            if constexpr (LSF) if (t > 0) ff.run(hbuf + (t & 1) * 16 * LDH, rl.spikes_f32 + ((size_t)(t - 1) * R + row0) * H, lane);
            // frames t+2 .. t+D-1 may stay in flight: frame t+1 has landed when the barrier releases step t+1
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
        // ================================================= storer wave =================================================
        constexpr int MAX8 = (RPW * KS * 4 + 63) / 64;
        constexpr int nu8 = RPW * (HP / 16), ns8 = (nu8 + 63) / 64;
        constexpr bool F32 = (OUT & 1) && (!LSF || LSPLIT) && !CWF;  // (what the loader wave / the compute waves do not write)
        S3FlushF<RPW, LDH> ff;
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
                        if (PUB && !(exp_flags & 8)) store16_sc1(p8, (unsigned)((64 * k + lane) * 16), d);  // (bit 3: timing experiment, plain)
                        else *reinterpret_cast<v4i*>(p8 + (size_t)(64 * k + lane) * 16) = d;
                        if constexpr (!(OUT & 1)) cnt += popc16(d);  // (live rows only; the pad columns of the state buffer hold zeros)
                    }
                }
            }
        };
        // store instructions per frame that have at least one live lane (rows past R are skipped): the publishing wait below
        // counts on at least this many per frame being in the queue (an instruction without live lanes may or may not be issued)
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
                    // my queue holds nothing but these stores: all but the youngest PF frames' worth have retired -> frames
                    // [0, t-PF) are complete in memory (the int8 rows were written through).  PF as deep as the 6-bit counter
                    // allows: a write-through store takes microseconds to retire under load, and this wave stalling at the
                    // step barrier would stall the compute waves with it (measured: producers 1.0-1.4 us per step with PF = 2)
                    S3_PB_TIC();
                    if (!(exp_flags & 16)) wait_vmcnt_n(pf * spf);  // (bit 4: timing experiment, publish without the wait)
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

// ---------------------------------------------------------------------------------------------------------------------
// PROJ role with a loader wave: z[t][r][:] = (S[t][r][:] . W_ih^T) * dq + bias_f for 32 rows per workgroup, frame by frame behind
// the workgroups that produce S (the layer below, same launch).  Same products as sfsn_spike_proj (three int8 digit MFMAs,
// exact recombination, one fma).  Compute wave w < NT: output tile w, W_ih register resident, both 16-row column tiles of a
// frame; its results leave as 16-byte write-through stores -- the ONLY entries of its vmcnt queue, so the one counted wait per
// step (PF frames of them may be in flight) never waits for a load.  The loader wave (wave NT) fetches the int8 spike rows
// into a ring (sc1 loads) and is the only one that polls.  Round 2's PROJ16 mixed DMAs and write-through stores in every wave's
// queue: 2.4 us per frame at B = 64 against 0.7 for the scans it feeds.
// ---------------------------------------------------------------------------------------------------------------------
template <int KS>
struct Proj3Layout {
    static constexpr int HP = KS * 64, NCH = KS * 4, ROWS = 32, PF = 8;
    static constexpr int SLOT = ROWS * HP, NP = ROWS * NCH / 64;
    // ring depth: an sc1 load of rows the layer below has just written through takes ~3 us; 6 frames in flight (D = 4 made this
    // role run at 1.5 us per frame on an idle chip), within the 64 KiB LDS-DMA window and the 6-bit vmcnt
    static constexpr int D = (65536 / SLOT < 8 ? 65536 / SLOT : 8) < 63 / NP + 2 ? (65536 / SLOT < 8 ? 65536 / SLOT : 8) : 63 / NP + 2;
    static constexpr int ZLD = HP + 4;                       // staging row stride in floats (fragment writes hit distinct banks)
    static constexpr int ZBUF_OFF = D * SLOT, ZBUF_BYTES = ROWS * ZLD * 4;
    static constexpr int FLAG_OFF = ZBUF_OFF + 2 * ZBUF_BYTES, BYTES = FLAG_OFF + 16;
};

struct Proj3Role {
    const int8_t* spikes_in;
    const int8_t* w_ih;
    const float* w_ih_dq;
    const float* bias;
    float* zin;
    int R, row0;
};

// Output path: the accumulator fragment of a wave covers 64 bytes of each of 16 rows -- as global stores those are partial-line
// write-through writes (round 2's note (iii) and this round's first attempt: 2.4 us per frame).  The fragments go to an LDS
// staging buffer instead (double buffered); one step later every compute wave writes its share (2 KiB) of the frame's
// contiguous 32-row block with two full 16-byte-per-lane stores.
template <int KS>
__device__ __forceinline__ void proj3_role(const Proj3Role& rl, const StackLink& lk, char* smem, int T, int H, int NT, int exp_flags = 0) {
    using L = Proj3Layout<KS>;
    constexpr int HP = L::HP, D = L::D, NCH = L::NCH, SLOT = L::SLOT, NP = L::NP, PF = L::PF, ZLD = L::ZLD;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int R = rl.R, row0 = rl.row0;
    volatile int* flag = reinterpret_cast<volatile int*>(smem + L::FLAG_OFF);
    if (tid == 0) { flag[0] = 0; flag[1] = 0; }
    __syncthreads();
    if (wave < NT) {
        const int ct = wave, cc = ct * 16 + q * 4;
        v4i W[KS][3];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int d = 0; d < 3; ++d)
                W[ks][d] = *reinterpret_cast<const v4i*>(rl.w_ih + ((((size_t)d * NT + ct) * KS + ks) * 64 + lane) * 16);
        v4f bf, dqi;  // (element loads: the caller's vectors need no 16-byte alignment)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) { bf[r4] = rl.bias[cc + r4]; dqi[r4] = rl.w_ih_dq[cc + r4]; }
        unsigned zl[2], soff[2][KS];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int r = c * 16 + n;
            zl[c] = (unsigned)(L::ZBUF_OFF + (r * ZLD + cc) * 4);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) soff[c][ks] = (unsigned)(r * HP + ((ks * 4 + q + r) % NCH) * 16);
        }
        // my two 16-byte units of a frame's block (32 rows x H floats, contiguous in memory): units 64 (2 wave + k) + lane
        const int q4 = H / 4;
        unsigned ul[2], ug[2];
        bool uok[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int u = 64 * (2 * wave + k) + lane, rr = u / q4, c4 = u - rr * q4;
            ul[k] = (unsigned)(L::ZBUF_OFF + (rr * ZLD + c4 * 4) * 4);
            ug[k] = (unsigned)u * 16u;
            uok[k] = row0 + rr < R;
        }
        auto flush = [&](int ts) __attribute__((always_inline)) {  // frame ts: staging buffer ts & 1 -> global, write-through
            float* zt = rl.zin + ((size_t)ts * R + row0) * H;
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (uok[k]) {
                    const v4i d = *reinterpret_cast<const v4i*>(smem + ul[k] + (ts & 1) * L::ZBUF_BYTES);
                    if (exp_flags & 1) continue;                                                    // (timing experiment: no stores)
                    if (exp_flags & 2) *reinterpret_cast<v4i*>(reinterpret_cast<char*>(zt) + ug[k]) = d;  // (timing experiment: plain stores)
                    else store16_sc1(zt, ug[k], d);
                }
        };
        __builtin_amdgcn_s_barrier();  // the loader's prologue frames have landed
        int stop = 0;
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            stop = flag[t & 1];
            // frame f leaves at step f+1; the wait that ended step t-1 left at most PF steps of my stores in flight: the stores of
            // steps <= t-1-PF, i.e. frames <= t-2-PF, are complete for every wave
            if (wave == 0 && lane == 0 && t - 1 - PF > 0) stack_publish(lk, t - 1 - PF);
            const char* sl = smem + (t % D) * SLOT;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                v4i e[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const v4i bs = *reinterpret_cast<const v4i*>(sl + soff[c][ks]);
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        if (exp_flags & 4) e[d] += W[ks][d] ^ bs;  // (timing experiment: no matrix instructions)
                        else e[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[ks][d], bs, e[d], 0, 0, 0);
                    }
                }
                if (c == 0 && t > 0) flush(t - 1);  // under the first tile's matrix instructions
                v4f z;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    z[r4] = __builtin_fmaf((float)((e[2][r4] << 16) + (e[1][r4] << 8) + e[0][r4]), dqi[r4], bf[r4]);  // = sfsn_spike_proj
                *reinterpret_cast<v4f*>(smem + zl[c] + (t & 1) * L::ZBUF_BYTES) = z;
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PF) : "memory");  // my queue: two stores per frame, nothing else
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            if (__builtin_amdgcn_readfirstlane(stop)) break;
        }
        if (T > 0 && !__builtin_amdgcn_readfirstlane(stop)) flush(T - 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wave == 0 && lane == 0) stack_publish(lk, T);  // (also after an expired spin: consumers must not wait for us)
        return;
    }
    if (wave == NT) {
        // loader: a slot = the 32 rows of a frame as 32 * NCH 16-byte chunks, chunk (row r, position p) holds global chunk
        // (p - r) mod NCH of that row (the 16 rows of a B fragment then hit distinct banks); piece k = chunks [64 k, 64 k + 64)
        unsigned src_off[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int e = p * 64 + lane, er = e / NCH, esl = e - er * NCH;
            const int erow = (row0 + er < R) ? row0 + er : R - 1;
            src_off[p] = (unsigned)(erow * HP + ((esl - er % NCH + NCH) % NCH) * 16);
        }
        const size_t frame = (size_t)R * HP;
        int avail = (exp_flags & 32) ? T : 0, failed = 0;  // (bit 5: timing experiment, no gating)
        auto issue = [&](int slot, int td) __attribute__((always_inline)) {
            if (exp_flags & 64) return;                  // (bit 6: timing experiment, no input DMAs)
            const float* st = reinterpret_cast<const float*>(rl.spikes_in + (size_t)td * frame);
#pragma unroll
            for (int p = 0; p < NP; ++p) dma16_to_lds<true>(__builtin_amdgcn_readfirstlane((unsigned)(slot * SLOT + p * 1024)), st, src_off[p]);
        };
        s3_ensure(lk, D - 1 < T ? D - 1 : T, T, avail, failed, lane);
        if (!failed)
            for (int s0 = 0; s0 < D - 1; ++s0) issue(s0, s0 < T ? s0 : (T > 0 ? T - 1 : 0));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (failed && lane == 0) flag[0] = 1;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        int stop = 0;
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            stop = failed;
            const int td = (t + D - 1 < T) ? t + D - 1 : T - 1;
            s3_ensure(lk, td + 1, T, avail, failed, lane);
            if (!failed) issue((t + D - 1) % D, td);
            if (failed && lane == 0) flag[(t + 1) & 1] = 1;  // read during step t+1 (see scan3_role)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * NP) : "memory");
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            if (stop) break;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        return;
    }
    // spare waves keep the barrier count
    __builtin_amdgcn_s_barrier();
    int stop = 0;
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        stop = flag[t & 1];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        if (__builtin_amdgcn_readfirstlane(stop)) break;
    }
    __builtin_amdgcn_s_barrier();
}

#endif
