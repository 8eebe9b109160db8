// sfsn_stack.hip -- the layer-pipelined GSN stack scan for gfx950 (MI355X / CDNA4): sfsn_gsn_stack_scan.
//
// The reference runs a stack layer by layer (StackedGSU.forward, efficient_spiking_neuron.py:56-61: layer l over ALL T frames,
// then layer l+1), but layer l+1 needs frame t of layer l only at frame t.  This kernel runs every layer of a stack in ONE
// launch: each layer's rows are owned by their own workgroups (one per CU, weights resident for all T frames), and a
// workgroup of layer l+1 trails its producers of layer l by a few frames.  The hand-off goes through L2 / Infinity Cache
// with write-through (sc1) stores, a per-workgroup progress counter and sc1 loads (sfsn_scan_dev.h: StackLink; rules from
// MI355X_MICROARCH.md "inter-workgroup visibility" and "hand-off price list").  The critical path of a stack is then one
// chain of T dependent steps instead of L chains back to back, and the fp32 input term of layers >= 1 never exists in HBM.
//
// Roles (a role = one layer of one row segment = a contiguous range of workgroups; producers have lower block indices):
//   ZIN    recurrent scan whose input term x.W_ih^T + b arrives as fp32 [T][R][H] -- layer 0 (written by the time-parallel
//          input product before the launch) or a layer >= 1 fed by a PROJ role (H > 256);  = scan_body of sfsn_scan_dev.h
//   FUSED  recurrent scan of a layer >= 1 that computes its input term itself from the previous layer's int8 spikes
//          (shared gates, H <= 256: both weight matrices fit one CU)
//   PROJ   the time-parallel input product S.W_ih^T + b of a layer >= 1 whose two matrices do not fit one CU together
//          (H = 320, the full-band model): 16 rows per workgroup, no recurrence, feeds a ZIN role
//   FUSED3 (round 4) the IO-wave scan of a layer >= 1 with its input product inside, batched over two frames in the MFMA columns
//          8 rows leave idle (sfsn_scan3i_dev.h): H <= 224, shared gates, 8 rows per workgroup; what H <= 224 stacks without
//          input-term buffers run (the 8-wave FUSED role keeps 224 < H <= 256 and other row counts)
// Arithmetic is that of the per-layer kernels, instruction for instruction: results are bit-identical to
// sfsn_spike_proj + sfsn_gsn_layer_scan (tested).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "sfsn.h"
#include "sfsn_scan_dev.h"
#include "sfsn_scan3_dev.h"
#include "sfsn_scan3i_dev.h"
#include "sfsn_scan3x_dev.h"
#include "sfsn_scan3w_dev.h"

#define STACK_MAX_ROLES 24
#define STACK_ZIN 0
#define STACK_FUSED 1
#define STACK_PROJ 2
#define STACK_FUSEDX3 4  // layer 0 with its real-valued input product inside the IO-wave scan (sfsn_scan3x_dev.h): wide kernel, 8 rows
#define STACK_FUSED3 3  // the IO-wave scan with its input product inside (sfsn_scan3i_dev.h): wide kernel, 8 rows per workgroup

struct StackRoleDev {
    const int8_t* w_hh;
    const float* w_dq;
    const float* bias;
    const float* bn_alpha;
    const float* bn_beta;
    float* h_state;
    float* c_state;
    float* spikes_f32;
    int8_t* spikes_i8;
    float* zin;               // ZIN: the input term (read); PROJ: the input term (written)
    const int8_t* spikes_in;  // FUSED / PROJ: the previous layer's int8 spikes
    const int8_t* w_ih;       // FUSED / PROJ: packed input weights
    const float* w_ih_dq;
    const float* x;           // FUSEDX3: the layer input [T][R][I]
    const float* w_ih_f32;    // FUSEDX3: [H][I]
    int I;
    int R, block0, nblocks, kind, rpw;
    int src;      // producer role (-1: the input is complete before the launch)
    int src_rpw;  // rows per workgroup of the producer role
    int split;    // PROJ role of the narrow kernels: workgroups per 16-row block, each owning NT / split column tiles (1 elsewhere)
    int pub;      // 1: another role of this launch consumes what this role writes
    unsigned long long* count;  // nullable: without fp32 spikes the role adds the number of spikes it wrote (ScanSegDev::count)
};

struct StackParams {
    StackRoleDev role[STACK_MAX_ROLES];
    unsigned* prog;  // [0] error word, [1] workgroups that have exited, [2 + block] frames published by that workgroup
    int nblocks;     // grid size (padding blocks included): the LAST workgroup to exit zeroes the counters for the next launch
    unsigned* dbg;   // optional [2 * block]: hand-off waits / poll iterations per workgroup (SFSN_STACK_DEBUG=1)
    int nroles, T, H, NT, lag;
    int gate_off;    // byte offset of the gate word in the dynamic LDS allocation (behind every role's layout)
    int v2;          // 1: round 2's scan body in the wide flavour (SFSN_SCAN_V2=1: A/B runs)
    int exp_flags;   // timing experiments (SFSN_STACK_EXP, wrong results): see proj3_role
    int lsplit;      // 8-row IO-wave roles: fp32 store instructions per frame issued by the loader wave (SFSN_S3_LSPLIT)
    int lsplit_x;    // ... of the FUSEDX3 role
    unsigned long long* wg_times;  // EXPERIMENTS builds: per-workgroup residency stamps (sfsn_scan_dev.h)
};


// Every workgroup (padding blocks too) calls this as its last action: the last one to arrive -- nobody polls any more -- zeroes the
// progress counters and the exit counter, so that the next launch on this scratch buffer starts clean without a memset in front
// of it (two fill launches of ~5 us each per stack launch: a third of a streaming hop's stack time).  `word`: the gate word of
// the dynamic LDS allocation (NO static __shared__ in these kernels: the LDS-DMA destinations are absolute addresses from 0).
__device__ __forceinline__ void stack_exit(const StackParams& p, int* word) {
    // my own counter stores (write-through, issued by this workgroup's publishing wave) must have reached memory before I count
    // myself out -- otherwise one of them could land after the last workgroup's zeroing.  No cache fence is needed for that:
    // draining the waves' own store queues is enough (a __threadfence() here cost ~6 us per launch).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        *reinterpret_cast<volatile int*>(word) = atomicAdd(p.prog + 1, 1u) == (unsigned)(p.nblocks - 1) ? 1 : 0;
    __syncthreads();
    if (*reinterpret_cast<volatile int*>(word)) {
        for (int i = threadIdx.x; i < p.nblocks; i += blockDim.x) __hip_atomic_store(p.prog + 2 + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) __hip_atomic_store(p.prog + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int KS>
struct StackGeom {
    static constexpr int NW = 8;
    static constexpr int TPW = KS <= 2 ? 1 : (KS <= 4 ? 2 : 3);
    static constexpr int LP = KS == 5 ? 1 : 0;
    static constexpr int HP = KS * 64, LDH = HP + 32;
};

// ---------------------------------------------------------------------------------------------------------------------
// FUSED role: the fused-input scan of sfsn_kernels.hip (gsn_scan_fused_kernel) for any rows-per-workgroup, gated and
// (optionally) publishing.  8 waves; wave w owns output tiles w and w + 8 (those that exist).  Digit plane 0 of W_hh and
// W_ih in LDS, planes 1-2 in registers.  LDS: [input-spike ring D x SLOT][hbuf 2 x 16 x LDH][6 constants x HP][W_hh p0][W_ih p0].
// ---------------------------------------------------------------------------------------------------------------------
template <int KS>
struct FusedLayout {
    #ifndef SFSN_FUSED_RING
#define SFSN_FUSED_RING 3
#endif
    // input-spike ring depth: the slot of step t+1 is waited for at the end of step t and was requested D-2 steps before
    // that -- a hand-off through L2 / Infinity Cache (write-through producer) needs more lead than a read of settled data
    static constexpr int HP = KS * 64, LDH = HP + 32, D = SFSN_FUSED_RING, NCH = KS * 4;
    __device__ __host__ static constexpr int slot_bytes(int rpw) { return ((rpw * HP + 1023) / 1024) * 1024; }
    __device__ __host__ static constexpr int hbuf_off(int rpw) { return D * slot_bytes(rpw); }
    __device__ __host__ static constexpr int cst_off(int rpw) { return hbuf_off(rpw) + 2 * 16 * LDH; }
    __device__ __host__ static constexpr int whh_off(int rpw) { return cst_off(rpw) + 6 * HP * 4; }
    __device__ __host__ static constexpr int bytes(int rpw, int NT) { return whh_off(rpw) + 2 * NT * KS * 1024; }
};

template <int KS, int OUT, int NTL, bool PUB>
__device__ __forceinline__ void stack_fused_body(const StackRoleDev& rl, const StackLink& lk, char* smem, int* gate_word, int T, int H,
                                                 int NT, int row0, int rowc, int n, int q, int tid, int wave) {
    using C = ScanCfg<1, KS, 8, 2, OUT, 0>;  // geometry constants of the flush only (LDH, HP, FL, NSTF)
    using L = FusedLayout<KS>;
    constexpr int LDH = C::LDH, HP = C::HP, D = L::D, NW = 8, NCH = L::NCH;
    const int rpw = rl.rpw, R = rl.R;
    const int SLOT = L::slot_bytes(rpw);
    const int WHH_OFF = L::whh_off(rpw), WIH_OFF = WHH_OFF + NT * KS * 1024;
    int8_t* hbuf = reinterpret_cast<int8_t*>(smem + L::hbuf_off(rpw));
    const float(*cst)[HP] = reinterpret_cast<const float(*)[HP]>(smem + L::cst_off(rpw));  // b_f, b_g - b_f, alpha, beta, dq_hh, dq_ih
    ScanFlush<C> fl;
    fl.init(tid, row0, R, H, NW * 64, rpw);
    const int lane = tid & 63;
    // operations allowed in flight at the end-of-step wait: one more DMA and the flush stores of two steps
    constexpr int CBASE = (D - 2) * 1, CSTRIDE = (D - 1);
    const int nst = fl.template stores_per_frame<OUT, PUB>();

    v4i Whh[NTL > 0 ? NTL : 1][KS][2], Wih[NTL > 0 ? NTL : 1][KS][2];
    v4f c[NTL > 0 ? NTL : 1];
    int col[NTL > 0 ? NTL : 1];
    unsigned wl_off[NTL > 0 ? NTL : 1];
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
                Whh[i][ks][d] = *reinterpret_cast<const v4i*>(rl.w_hh + ((tile * KS + ks) * 64 + lane) * 16);
                Wih[i][ks][d] = *reinterpret_cast<const v4i*>(rl.w_ih + ((tile * KS + ks) * 64 + lane) * 16);
            }
        c[i] = *reinterpret_cast<const v4f*>(rl.c_state + (size_t)rowc * H + col[i]);
    }
    // input-spike ring: a slot holds the rpw rows of a frame as rpw * NCH 16-byte chunks; chunk e = (row er, position esl)
    // sits at LDS byte 16 e and holds global chunk (esl - er) mod NCH of that row (a rotation by the row index: the 16 rows
    // of a B fragment then hit distinct banks).  DMA piece k = chunks [64 k, 64 k + 64); wave w fetches piece min(w, last).
    const int nchunk = rpw * NCH;
    const int npiece = (nchunk + 63) >> 6;
    const int piece = wave < npiece ? wave : npiece - 1;
    int e = piece * 64 + lane;
    if (e > nchunk - 1) e = nchunk - 1;  // surplus lanes re-fetch the last chunk (into the slot's padding)
    const int er = e / NCH, esl = e - er * NCH;
    const int erow = (row0 + er < R) ? row0 + er : R - 1;
    const unsigned src_off = (unsigned)(erow * HP + ((esl - er % NCH + NCH) % NCH) * 16);
    const size_t frame = (size_t)R * HP;
    auto issue = [&](int slot, int td) __attribute__((always_inline)) {
        dma16_to_lds<true>(__builtin_amdgcn_readfirstlane((unsigned)(slot * SLOT + piece * 1024)),
                           reinterpret_cast<const float*>(rl.spikes_in + (size_t)td * frame), src_off);
    };
    int avail = 0;
    {   // the prologue and step 0 read frames [0, D)
        const int need = D < T ? D : T;
        avail = stack_refresh(lk, need, T, gate_word, wave, lane);
    }
    for (int s0 = 0; s0 < D - 1; ++s0) issue(s0, s0 < T ? s0 : (T > 0 ? T - 1 : 0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);

    const int nr = n & (rpw - 1);  // MFMA columns n >= rpw are duplicates of column n % rpw (same data, same results)
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
        const char* sslot = smem + (t % D) * SLOT + nr * HP;
        if constexpr (NTL == 0) {
            if constexpr (!FIRST) fl.template run<OUT, PUB>(hc, rl.spikes_f32, rl.spikes_i8, t - 1, R, H);
        }
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
                // wave's piece of the slot landed before the barrier that ended the previous step)
                if constexpr (!FIRST) fl.template run<OUT, PUB>(hc, rl.spikes_f32, rl.spikes_i8, t - 1, R, H);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) bs[ks] = *reinterpret_cast<const v4i*>(sslot + ((ks * 4 + q + nr) % NCH) * 16);
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
        // each wave waits for its own piece of step t+1's slot (issued at the top of step t-1; since then: one more DMA and
        // two steps' flush stores); the barrier makes all pieces and the new hidden state visible to all
        if (t < D || CBASE + nst * CSTRIDE > 63) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            wait_vmcnt_affine<CBASE, CSTRIDE, 5>(nst);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    };
    __builtin_amdgcn_s_barrier();  // the prologue's pieces (drained above by every wave) are now visible to all
    if (T > 0 && avail >= 0) step(0, std::true_type{});
#pragma unroll 1
    for (int t = 1; t < T; ++t) {
        {   // step t issues the DMA of frame t+D-1
            const int need = (t + D < T) ? t + D : T;
            if (avail >= 0 && need > avail) avail = stack_refresh(lk, need, T, gate_word, wave, lane);
            if (avail < 0) break;
        }
        if constexpr (PUB) {
            // After the barrier that ended step t-1 every wave has passed the wait of step t-1, which covers the flush
            // stores issued at steps <= t-D, i.e. frames <= t-D-1: t-D frames are complete in memory.
            if (wave == 0 && lane == 0 && t - D > 0) stack_publish(lk, t - D);
        }
        step(t, std::false_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (T > 0 && avail >= 0) fl.template run<OUT, PUB>(hbuf + (T & 1) * 16 * LDH, rl.spikes_f32, rl.spikes_i8, T - 1, R, H);
    if constexpr (PUB) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wave == 0 && lane == 0) stack_publish(lk, T);
    }
    fl.template finish<OUT>(rl.count);
    const int8_t* hl = hbuf + (T & 1) * 16 * LDH;
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        *reinterpret_cast<v4f*>(rl.c_state + (size_t)rowc * H + col[i]) = c[i];
        const unsigned pk = *reinterpret_cast<const unsigned*>(hl + n * LDH + col[i]);
        const v4f h = {(float)(pk & 1u), (float)((pk >> 8) & 1u), (float)((pk >> 16) & 1u), (float)((pk >> 24) & 1u)};
        *reinterpret_cast<v4f*>(rl.h_state + (size_t)rowc * H + col[i]) = h;
    }
}

template <int KS, int OUT, bool PUB>
__device__ __forceinline__ void stack_fused_role(const StackRoleDev& rl, const StackLink& lk, char* smem, int* gate_word, int T, int H,
                                                 int NT, int blk) {
    using L = FusedLayout<KS>;
    constexpr int LDH = L::LDH, HP = L::HP, NW = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int rpw = rl.rpw, R = rl.R;
    const int row0 = blk * rpw;
    const int rowc = (row0 + (n & (rpw - 1)) < R) ? row0 + (n & (rpw - 1)) : R - 1;
    int8_t* hbuf = reinterpret_cast<int8_t*>(smem + L::hbuf_off(rpw));
    float(*cst)[HP] = reinterpret_cast<float(*)[HP]>(smem + L::cst_off(rpw));
    for (int j = tid; j < HP; j += NW * 64) {
        const bool in = j < H;
        cst[0][j] = in ? rl.bias[j] : 0.0f;
        cst[1][j] = in ? rl.bias[H + j] - rl.bias[j] : 0.0f;
        cst[2][j] = in ? rl.bn_alpha[j] : 0.0f;
        cst[3][j] = in ? rl.bn_beta[j] : 0.0f;
        cst[4][j] = in ? rl.w_dq[j] : 0.0f;
        cst[5][j] = in ? rl.w_ih_dq[j] : 0.0f;
    }
    for (int i = tid; i < 2 * 16 * LDH / 4; i += NW * 64) reinterpret_cast<int*>(hbuf)[i] = 0;
    {   // digit plane 0 of both matrices -> LDS, once
        v4i* d0 = reinterpret_cast<v4i*>(smem + L::whh_off(rpw));
        v4i* d1 = reinterpret_cast<v4i*>(smem + L::whh_off(rpw) + NT * KS * 1024);
        const v4i* s0 = reinterpret_cast<const v4i*>(rl.w_hh);
        const v4i* s1 = reinterpret_cast<const v4i*>(rl.w_ih);
        for (int i = tid; i < NT * KS * 64; i += NW * 64) {
            d0[i] = s0[i];
            d1[i] = s1[i];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < 16 * (H / 4); idx += NW * 64) {
        const int rr = idx / (H / 4), j4 = (idx - rr * (H / 4)) * 4;
        const int rsrc = (row0 + (rr & (rpw - 1)) < R) ? row0 + (rr & (rpw - 1)) : R - 1;
        const v4f h = *reinterpret_cast<const v4f*>(rl.h_state + (size_t)rsrc * H + j4);
        const unsigned pk = (h.x > 0.5f ? 1u : 0u) | (h.y > 0.5f ? 0x100u : 0u) | (h.z > 0.5f ? 0x10000u : 0u) |
                            (h.w > 0.5f ? 0x1000000u : 0u);
        *reinterpret_cast<unsigned*>(hbuf + rr * LDH + j4) = pk;
    }
    __syncthreads();
    if (wave + NW < NT)
        stack_fused_body<KS, OUT, 2, PUB>(rl, lk, smem, gate_word, T, H, NT, row0, rowc, n, q, tid, wave);
    else if (wave < NT)
        stack_fused_body<KS, OUT, 1, PUB>(rl, lk, smem, gate_word, T, H, NT, row0, rowc, n, q, tid, wave);
    else
        stack_fused_body<KS, OUT, 0, PUB>(rl, lk, smem, gate_word, T, H, NT, row0, rowc, n, q, tid, wave);
}

// ---------------------------------------------------------------------------------------------------------------------
// PROJ role: z[t][r][:] = (S[t][r][:] . W_ih^T) * dq + bias_f for 16 rows per workgroup, frame by frame behind its producers.
// Same products as sfsn_spike_proj (three int8 digit MFMAs, one rounding in the recombination, rec * dq exact, + bias).
// Digit plane 0 of W_ih in LDS, planes 1-2 in registers (8 waves x up to 3 tiles).  Output fragments leave as 16-byte
// write-through stores.  LDS: [input-spike ring D x 16 x HP][bias, dq: 2 x HP floats][W_ih p0].
// ---------------------------------------------------------------------------------------------------------------------
#ifndef SFSN_PROJ_D
#define SFSN_PROJ_D 3
#endif
template <int KS>
struct ProjLayout {
    static constexpr int HP = KS * 64, D = SFSN_PROJ_D, NCH = KS * 4;
    static constexpr int SLOT = 16 * HP;
    static constexpr int CST_OFF = D * SLOT, W_OFF = CST_OFF + 2 * HP * 4;
    __device__ __host__ static constexpr int bytes(int NT) { return W_OFF + NT * KS * 1024; }
};

template <int KS, int NTL, int NW = 8>
__device__ __forceinline__ void stack_proj_body(const StackRoleDev& rl, const StackLink& lk, char* smem, int* gate_word, int T, int H,
                                                int NT, int row0, int rowc, int n, int q, int tid, int wave, int t_lo) {
    using L = ProjLayout<KS>;
    constexpr int HP = L::HP, D = L::D, NCH = L::NCH, SLOT = L::SLOT;
    const int R = rl.R;
    const float(*cst)[HP] = reinterpret_cast<const float(*)[HP]>(smem + L::CST_OFF);  // bias_f, dq_ih
    const int lane = tid & 63;
    constexpr int CBASE = (D - 2) * 1 + (D - 1) * NTL;  // one more DMA and two steps' output stores may stay in flight

    v4i Wih[NTL > 0 ? NTL : 1][KS][2];
    int col[NTL > 0 ? NTL : 1];
    unsigned wl_off[NTL > 0 ? NTL : 1];
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        const int ct = t_lo + wave + NW * i;
        col[i] = ct * 16 + q * 4;
        wl_off[i] = (unsigned)((ct * KS) * 64 + lane) * 16u;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const size_t tile = (size_t)(d + 1) * NT + ct;
                Wih[i][ks][d] = *reinterpret_cast<const v4i*>(rl.w_ih + ((tile * KS + ks) * 64 + lane) * 16);
            }
    }
    constexpr int nchunk = 16 * NCH, npiece = (nchunk + 63) >> 6;  // = KS pieces of 1 KiB
    const int piece = wave < npiece ? wave : npiece - 1;
    const int e = piece * 64 + lane;
    const int er = e / NCH, esl = e - er * NCH;
    const int erow = (row0 + er < R) ? row0 + er : R - 1;
    const unsigned src_off = (unsigned)(erow * HP + ((esl - er % NCH + NCH) % NCH) * 16);
    const size_t frame = (size_t)R * HP;
    auto issue = [&](int slot, int td) __attribute__((always_inline)) {
        dma16_to_lds<true>(__builtin_amdgcn_readfirstlane((unsigned)(slot * SLOT + piece * 1024)),
                           reinterpret_cast<const float*>(rl.spikes_in + (size_t)td * frame), src_off);
    };
    int avail = 0;
    {
        const int need = D < T ? D : T;
        avail = stack_refresh(lk, need, T, gate_word, wave, lane);
    }
    for (int s0 = 0; s0 < D - 1; ++s0) issue(s0, s0 < T ? s0 : (T > 0 ? T - 1 : 0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();

    const unsigned zrow = (unsigned)(rowc * H) * 4u;
    S3_PB_DECL();
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        if (t > 0) {
            const int need = (t + D < T) ? t + D : T;
            S3_PB_TIC();
            if (avail >= 0 && need > avail) avail = stack_refresh(lk, need, T, gate_word, wave, lane);
            S3_PB_TOC(2);
            // stores issued at steps <= t-D are complete for every wave (see the wait below): frames [0, t-D+1)
            if (wave == 0 && lane == 0 && t - D + 1 > 0) stack_publish(lk, t - D + 1);
        }
        if (avail < 0) break;
        {
            const int td = (t + D - 1 < T) ? t + D - 1 : T - 1;
            issue((t + D - 1) % D, td);
        }
        v4i bs[KS];
        const char* sslot = smem + (t % D) * SLOT + n * HP;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bs[ks] = *reinterpret_cast<const v4i*>(sslot + ((ks * 4 + q + n) % NCH) * 16);
        float* zt = rl.zin + (size_t)t * R * H;
#pragma unroll
        for (int i = 0; i < NTL; ++i) {
            const int cc = col[i];
            v4i e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0}, e2 = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v4i w0 = *reinterpret_cast<const v4i*>(smem + L::W_OFF + wl_off[i] + (unsigned)(ks * 1024));
                e0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, bs[ks], e0, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wih[i][ks][0], bs[ks], e1, 0, 0, 0);
                e2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wih[i][ks][1], bs[ks], e2, 0, 0, 0);
            }
            const v4f bf = *reinterpret_cast<const v4f*>(&cst[0][cc]), dqi = *reinterpret_cast<const v4f*>(&cst[1][cc]);
            v4f z;
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = __builtin_fmaf(recombine3(e0[r], e1[r], e2[r]), dqi[r], bf[r]);
            v4i zi;
            __builtin_memcpy(&zi, &z, 16);
            store16_sc1(zt, zrow + (unsigned)cc * 4u, zi);  // rows past R are clamped duplicates (same value, same address)
        }
        // my piece of step t+1's slot was issued at the top of step t-D+2; since then: D-2 more DMAs and D-1 steps' NTL stores
        S3_PB_TIC();
        if (t < D) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CBASE) : "memory");
        }
        S3_PB_TOC(1);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        S3_PB_TIC();
        __builtin_amdgcn_s_barrier();
        S3_PB_TOC(0);
    }
    S3_PB_OUT(lk, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave == 0 && lane == 0) stack_publish(lk, T);
}

// NW waves (8 in the 512-thread kernel, 12 in gsn_stack_fb_kernel).  Workgroup blk of the role = 16-row block blk / split, column part
// blk % split: the tiles [part NT / split, (part + 1) NT / split), dealt round-robin over the waves.  Round 5: the role's step is
// bound by how fast ONE compute unit gets 16 rows x H floats of write-through stores out (scripts/exp_beside_r05.py: 2,720 clk per
// frame alone, 3,420 beside the sub-band pair launch, with < 100 clk of counted vmcnt waits -- the store instructions themselves do
// not issue), and the role's block range is padded to a multiple of eight anyway: the padding workgroups take column parts.
template <int KS, int NW = 8>
__device__ __forceinline__ void stack_proj_role(const StackRoleDev& rl, const StackLink& lk, char* smem, int* gate_word, int T, int H,
                                                int NT, int blk) {
    using L = ProjLayout<KS>;
    constexpr int HP = L::HP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int R = rl.R;
    const int part = blk % rl.split;
    const int row0 = (blk / rl.split) * 16;
    const int t_lo = part * NT / rl.split, t_hi = (part + 1) * NT / rl.split;
    const int rowc = (row0 + n < R) ? row0 + n : R - 1;
    float(*cst)[HP] = reinterpret_cast<float(*)[HP]>(smem + L::CST_OFF);
    for (int j = tid; j < HP; j += NW * 64) {
        const bool in = j < H;
        cst[0][j] = in ? rl.bias[j] : 0.0f;
        cst[1][j] = in ? rl.w_ih_dq[j] : 0.0f;
    }
    {
        v4i* d1 = reinterpret_cast<v4i*>(smem + L::W_OFF);
        const v4i* s1 = reinterpret_cast<const v4i*>(rl.w_ih);
        for (int i = t_lo * KS * 64 + tid; i < t_hi * KS * 64; i += NW * 64) d1[i] = s1[i];  // (my tiles of digit plane 0)
    }
    __syncthreads();
    const int ntl = wave < t_hi - t_lo ? (t_hi - t_lo - wave + NW - 1) / NW : 0;  // tiles t_lo + wave, t_lo + wave + NW, ... < t_hi
    if (ntl >= 3)
        stack_proj_body<KS, 3, NW>(rl, lk, smem, gate_word, T, H, NT, row0, rowc, n, q, tid, wave, t_lo);
    else if (ntl == 2)
        stack_proj_body<KS, 2, NW>(rl, lk, smem, gate_word, T, H, NT, row0, rowc, n, q, tid, wave, t_lo);
    else if (ntl == 1)
        stack_proj_body<KS, 1, NW>(rl, lk, smem, gate_word, T, H, NT, row0, rowc, n, q, tid, wave, t_lo);
    else
        stack_proj_body<KS, 0, NW>(rl, lk, smem, gate_word, T, H, NT, row0, rowc, n, q, tid, wave, t_lo);
}

// ---------------------------------------------------------------------------------------------------------------------
// ZIN role = scan_body (sfsn_scan_dev.h) with 8 waves, gated and / or publishing.
// ---------------------------------------------------------------------------------------------------------------------
template <int KS, int OUT, int FLG>
__device__ __forceinline__ void stack_zin_role(const StackRoleDev& rl, const StackLink& lk, char* smem, int* gate_word, int T, int H, int NT,
                                               int blk) {
    using G = StackGeom<KS>;
    constexpr int NW = G::NW, TPW = G::TPW, LP = G::LP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int rpw = rl.rpw, R = rl.R;
    const int row0 = blk * rpw;
    const int rowc = (row0 + (n & (rpw - 1)) < R) ? row0 + (n & (rpw - 1)) : R - 1;
    ScanSegDev sg;
    sg.zin = rl.zin; sg.w_hh = rl.w_hh; sg.w_dq = rl.w_dq; sg.bias = rl.bias; sg.bn_alpha = rl.bn_alpha; sg.bn_beta = rl.bn_beta;
    sg.h_state = rl.h_state; sg.c_state = rl.c_state; sg.spikes_f32 = rl.spikes_f32; sg.spikes_i8 = rl.spikes_i8; sg.membrane = nullptr;
    sg.R = R;
    scan_prologue<1, KS, NW, TPW, OUT, LP>(sg, smem, tid, H, NT, R, row0, rpw);
    const int n_hi = NT - NW * (TPW - 1);  // waves [0, n_hi) own TPW tiles, the others TPW-1
    if (wave < n_hi)
        scan_body<1, KS, NW, TPW, OUT, LP, TPW, FLG>(sg.zin, sg.w_hh, sg.spikes_f32, sg.spikes_i8, nullptr, sg.h_state, sg.c_state, smem, T, H,
                                                     NT, R, row0, rowc, n, q, tid, wave, rpw, &lk, gate_word, rl.count);
    else
        scan_body<1, KS, NW, TPW, OUT, LP, TPW - 1, FLG>(sg.zin, sg.w_hh, sg.spikes_f32, sg.spikes_i8, nullptr, sg.h_state, sg.c_state, smem, T,
                                                         H, NT, R, row0, rowc, n, q, tid, wave, rpw, &lk, gate_word, rl.count);
}

// =====================================================================================================================
// The WIDE flavour (H <= 256, 16 waves per workgroup): every recurrent layer runs the 16-wave scan body of the per-layer
// kernel (one output tile per wave, four waves per SIMD hide each other's LDS / MFMA latency: 0.75 / 0.93 us per step at 4 /
// 8 rows per workgroup, against 1.6-1.8 us for any body that computes the input product inside the recurrent workgroup),
// and the input term of a layer >= 1 is produced, frame by frame, by PROJ16 workgroups of the same launch: 32 rows each,
// W_ih register resident (one tile per wave), both column tiles of a frame back to back -- a full-efficiency product whose
// fp32 result goes to the consumer through L2 / Infinity Cache (write-through stores, sc1 loads).
// =====================================================================================================================
template <int KS>
struct Proj16Layout {
    static constexpr int HP = KS * 64, D = 8, NCH = KS * 4, ROWS = 32;  // D: write-through stores retire slowly -- seven steps of them may be in flight
    static constexpr int SLOT = ROWS * HP;
    static constexpr int CST_OFF = D * SLOT;
    static constexpr int BYTES = CST_OFF + 2 * HP * 4;
};

template <int KS>
__device__ __forceinline__ void stack_proj16_role(const StackRoleDev& rl, const StackLink& lk, char* smem, int* gate_word, int T, int H,
                                                  int NT, int blk) {
    using L = Proj16Layout<KS>;
    constexpr int HP = L::HP, D = L::D, NCH = L::NCH, SLOT = L::SLOT, NW = 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int R = rl.R;
    const int row0 = blk * L::ROWS;
    float(*cst)[HP] = reinterpret_cast<float(*)[HP]>(smem + L::CST_OFF);  // bias_f, dq_ih
    for (int j = tid; j < HP; j += NW * 64) {
        const bool in = j < H;
        cst[0][j] = in ? rl.bias[j] : 0.0f;
        cst[1][j] = in ? rl.w_ih_dq[j] : 0.0f;
    }
    const bool have = wave < NT;
    const int ct = have ? wave : 0;
    const int cc = ct * 16 + q * 4;
    v4i W[KS][3];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const size_t tile = (size_t)d * NT + ct;
            W[ks][d] = *reinterpret_cast<const v4i*>(rl.w_ih + ((tile * KS + ks) * 64 + lane) * 16);
        }
    // input ring: a slot = the 32 rows of a frame as 32 * NCH 16-byte chunks, chunk (row r, position p) holds global chunk
    // (p - r) mod NCH of that row; piece k = chunks [64 k, 64 k + 64), wave w fetches piece min(w, last)
    constexpr int npiece = (L::ROWS * NCH) >> 6;
    const int piece = wave < npiece ? wave : npiece - 1;
    const int e = piece * 64 + lane;
    const int er = e / NCH, esl = e - er * NCH;
    const int erow = (row0 + er < R) ? row0 + er : R - 1;
    const unsigned src_off = (unsigned)(erow * HP + ((esl - er % NCH + NCH) % NCH) * 16);
    const size_t frame = (size_t)R * HP;
    auto issue = [&](int slot, int td) __attribute__((always_inline)) {
        dma16_to_lds<true>(__builtin_amdgcn_readfirstlane((unsigned)(slot * SLOT + piece * 1024)),
                           reinterpret_cast<const float*>(rl.spikes_in + (size_t)td * frame), src_off);
    };
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int avail = 0;
    {
        const int need = D < T ? D : T;
        avail = stack_refresh(lk, need, T, gate_word, wave, lane);
    }
    for (int s0 = 0; s0 < D - 1; ++s0) issue(s0, s0 < T ? s0 : (T > 0 ? T - 1 : 0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();

    const v4f bf = *reinterpret_cast<const v4f*>(&cst[0][cc]), dqi = *reinterpret_cast<const v4f*>(&cst[1][cc]);
    unsigned zoff[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int r = row0 + c * 16 + n;
        zoff[c] = (unsigned)(((r < R) ? r : R - 1) * H + cc) * 4u;  // rows past R are clamped duplicates (same value, same address)
    }
    // operations this wave issues per step: one DMA and (with a tile) two 16-byte stores
    const int ns = have ? 2 : 0;
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        if (t > 0) {
            const int need = (t + D < T) ? t + D : T;
            if (avail >= 0 && need > avail) avail = stack_refresh(lk, need, T, gate_word, wave, lane);
            // stores issued at steps <= t-D are complete for every wave (the wait that ended step t-1): frames [0, t-D+1)
            if (wave == 0 && lane == 0 && t - D + 1 > 0) stack_publish(lk, t - D + 1);
        }
        if (avail < 0) break;
        {
            const int td = (t + D - 1 < T) ? t + D - 1 : T - 1;
            issue((t + D - 1) % D, td);
        }
        float* zt = rl.zin + (size_t)t * R * H;
        const char* sl = smem + (t % D) * SLOT;
        if (have) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int r = c * 16 + n;
                v4i e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0}, e2 = {0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const v4i bs = *reinterpret_cast<const v4i*>(sl + r * HP + ((ks * 4 + q + r) % NCH) * 16);
                    e0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[ks][0], bs, e0, 0, 0, 0);
                    e1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[ks][1], bs, e1, 0, 0, 0);
                    e2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W[ks][2], bs, e2, 0, 0, 0);
                }
                v4f z;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) z[r4] = __builtin_fmaf(recombine3(e0[r4], e1[r4], e2[r4]), dqi[r4], bf[r4]);  // = sfsn_spike_proj
                v4i zi;
                __builtin_memcpy(&zi, &z, 16);
                store16_sc1(zt, zoff[c], zi);
            }
        }
        // my piece of step t+1's slot was issued at the top of step t-D+2; since then D-2 more DMAs and D-1 steps' stores
        if (t < D) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            wait_vmcnt_n((D - 1) * ns + (D - 2));
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave == 0 && lane == 0) stack_publish(lk, T);
}

template <int KS, int OUT, int FLG>
__device__ __forceinline__ void stack_zin16_role(const StackRoleDev& rl, const StackLink& lk, char* smem, int* gate_word, int T, int H, int NT,
                                                 int blk) {
    constexpr int NW = 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int rpw = rl.rpw, R = rl.R;
    const int row0 = blk * rpw;
    const int rowc = (row0 + (n & (rpw - 1)) < R) ? row0 + (n & (rpw - 1)) : R - 1;
    ScanSegDev sg;
    sg.zin = rl.zin; sg.w_hh = rl.w_hh; sg.w_dq = rl.w_dq; sg.bias = rl.bias; sg.bn_alpha = rl.bn_alpha; sg.bn_beta = rl.bn_beta;
    sg.h_state = rl.h_state; sg.c_state = rl.c_state; sg.spikes_f32 = rl.spikes_f32; sg.spikes_i8 = rl.spikes_i8; sg.membrane = nullptr;
    sg.R = R;
    scan_prologue<1, KS, NW, 1, OUT, 0>(sg, smem, tid, H, NT, R, row0, rpw);
    if (wave < NT)
        scan_body<1, KS, NW, 1, OUT, 0, 1, FLG>(sg.zin, sg.w_hh, sg.spikes_f32, sg.spikes_i8, nullptr, sg.h_state, sg.c_state, smem, T, H, NT, R, row0,
                                                rowc, n, q, tid, wave, rpw, &lk, gate_word, rl.count);
    else
        scan_body<1, KS, NW, 1, OUT, 0, 0, FLG>(sg.zin, sg.w_hh, sg.spikes_f32, sg.spikes_i8, nullptr, sg.h_state, sg.c_state, smem, T, H, NT, R, row0,
                                                rowc, n, q, tid, wave, rpw, &lk, gate_word, rl.count);
}

// D0 = 1 (round 6, sfsn_gsn_stack_scan_x_w16): 16-bit weights, the zero digit plane's matrix instructions skipped in every scan role
template <int KS, int OUT, int D0 = 0>
__global__ __launch_bounds__(1024) void gsn_stack_wide_kernel(const StackParams p) {
    extern __shared__ __attribute__((aligned(16))) char scan_smem[];
    int* gate_word_p = reinterpret_cast<int*>(scan_smem + p.gate_off);
    int ri = -1;
    for (int i = 0; i < p.nroles; ++i)
        if ((int)blockIdx.x >= p.role[i].block0 && (int)blockIdx.x < p.role[i].block0 + p.role[i].nblocks) ri = i;
    if (ri < 0) {
        stack_exit(p, gate_word_p);
        return;
    }
    SFSN_WG_STAMP(p.wg_times, 0);
    const StackRoleDev& rl = p.role[ri];
    const int blk = (int)blockIdx.x - rl.block0;
    StackLink lk;
    lk.in = nullptr; lk.n_in = 0; lk.out = nullptr; lk.err = p.prog; lk.lag = p.lag;
    lk.dbg = p.dbg ? p.dbg + 4 * blockIdx.x : nullptr;
#ifdef SFSN_EXPERIMENTS
    lk.probe = p.wg_times ? p.wg_times + 2 * gridDim.x + 64 * blockIdx.x : nullptr;  // 16 waves x 4 stall counters (WIDE_CASE)
#endif
    if (lk.dbg && threadIdx.x == 0) lk.dbg[2] = (unsigned)wall_clock64();
    if (rl.pub) lk.out = p.prog + 2 + blockIdx.x;
    const int my_rpw = rl.kind == STACK_PROJ ? Proj16Layout<KS>::ROWS : rl.rpw;
    if (rl.src >= 0) {
        const StackRoleDev& sr = p.role[rl.src];
        const int r0 = blk * my_rpw;
        int r1 = r0 + my_rpw - 1;
        if (r1 > rl.R - 1) r1 = rl.R - 1;
        const int b0 = r0 / rl.src_rpw, b1 = r1 / rl.src_rpw;
        lk.in = p.prog + 2 + sr.block0 + b0;
        lk.n_in = b1 - b0 + 1;
    }
    const int T = p.T, H = p.H, NT = p.NT;
    if (rl.kind == STACK_FUSEDX3) {
        Scan3xRole rx;
        rx.x = rl.x; rx.w_ih = rl.w_ih_f32; rx.I = rl.I; rx.w_hh = rl.w_hh; rx.w_dq = rl.w_dq; rx.bias = rl.bias;
        rx.bn_alpha = rl.bn_alpha; rx.bn_beta = rl.bn_beta; rx.h_state = rl.h_state; rx.c_state = rl.c_state;
        rx.spikes_f32 = rl.spikes_f32; rx.spikes_i8 = rl.spikes_i8; rx.R = rl.R; rx.row0 = blk * 8; rx.count = rl.count; rx.lsplit = p.lsplit_x;
#ifdef SFSN_EXPERIMENTS
        rx.probe = lk.probe;
#endif
        const bool tl = (H & 63) != 0 && (H & 63) <= 32;
#define X3_CASE(TL_, F_)                                                                    \
    {                                                                                       \
        if (rl.I > 32) scan3x_role<KS, TL_, OUT, F_, 2, D0>(rx, lk, scan_smem, T, H, NT);   \
        else scan3x_role<KS, TL_, OUT, F_, 1, D0>(rx, lk, scan_smem, T, H, NT);             \
    }
        if (rl.pub) {
            if (tl) X3_CASE(1, 2)
            else if constexpr (KS < 4) X3_CASE(0, 2)
        } else {
            if (tl) X3_CASE(1, 0)
            else if constexpr (KS < 4) X3_CASE(0, 0)
        }
#undef X3_CASE
    } else if (rl.kind == STACK_FUSED3) {
        Scan3iRole ri;
        ri.spikes_in = rl.spikes_in; ri.w_ih = rl.w_ih; ri.w_ih_dq = rl.w_ih_dq; ri.w_hh = rl.w_hh; ri.w_dq = rl.w_dq; ri.bias = rl.bias;
        ri.bn_alpha = rl.bn_alpha; ri.bn_beta = rl.bn_beta; ri.h_state = rl.h_state; ri.c_state = rl.c_state;
        ri.spikes_f32 = rl.spikes_f32; ri.spikes_i8 = rl.spikes_i8; ri.R = rl.R; ri.row0 = blk * 8; ri.count = rl.count;
#ifdef SFSN_EXPERIMENTS
        ri.probe = lk.probe;
#endif
        const bool tl = (H & 63) != 0 && (H & 63) <= 32;  // the k tail as one 16x16x32 step
        // (KS = 4 with at most 14 tiles is H = 208 or 224: always the tail form -- four full k-steps of both matrices do not fit)
        if (rl.pub) {
            if (tl) scan3i_role<KS, 1, OUT, 3, D0>(ri, lk, scan_smem, T, H, NT, p.exp_flags);
            else if constexpr (KS < 4) scan3i_role<KS, 0, OUT, 3, D0>(ri, lk, scan_smem, T, H, NT, p.exp_flags);
        } else {
            if (tl) scan3i_role<KS, 1, OUT, 1, D0>(ri, lk, scan_smem, T, H, NT, p.exp_flags);
            else if constexpr (KS < 4) scan3i_role<KS, 0, OUT, 1, D0>(ri, lk, scan_smem, T, H, NT, p.exp_flags);
        }
    } else if (rl.kind == STACK_PROJ) {
        if (NT <= 14 && !p.v2) {
            Proj3Role r3;
            r3.spikes_in = rl.spikes_in; r3.w_ih = rl.w_ih; r3.w_ih_dq = rl.w_ih_dq; r3.bias = rl.bias; r3.zin = rl.zin;
            r3.R = rl.R; r3.row0 = blk * Proj3Layout<KS>::ROWS;
            proj3_role<KS>(r3, lk, scan_smem, T, H, NT, p.exp_flags);
        } else {
            stack_proj16_role<KS>(rl, lk, scan_smem, gate_word_p, T, H, NT, blk);
        }
    } else {
        const bool rp4 = rl.rpw == 4;
        const int flg = (rl.src >= 0 ? 1 : 0) | (rl.pub ? 2 : 0);
        if (NT <= 14 && !p.v2) {
            // the scan with IO-specialised waves (sfsn_scan3_dev.h): the loader wave alone polls the producers, the storer wave
            // alone writes through and publishes -- the compute waves' step does not change with the role's links
            Scan3Role r3;
            r3.zin = rl.zin; r3.w_hh = rl.w_hh; r3.w_dq = rl.w_dq; r3.bias = rl.bias; r3.bn_alpha = rl.bn_alpha; r3.bn_beta = rl.bn_beta;
            r3.h_state = rl.h_state; r3.c_state = rl.c_state; r3.spikes_f32 = rl.spikes_f32; r3.spikes_i8 = rl.spikes_i8;
            r3.R = rl.R; r3.row0 = blk * rl.rpw; r3.count = rl.count; r3.lsplit = p.lsplit;
#ifdef SFSN_EXPERIMENTS
            r3.probe = lk.probe;
#endif
#define S3_CASE(RPW_, F) \
    if (rl.rpw == RPW_ && flg == F) scan3_role<KS, RPW_, OUT, F, D0>(r3, lk, scan_smem, T, H, NT, p.exp_flags);
            S3_CASE(4, 0) S3_CASE(4, 1) S3_CASE(4, 2) S3_CASE(4, 3)
            S3_CASE(8, 0) S3_CASE(8, 1) S3_CASE(8, 2) S3_CASE(8, 3)
            S3_CASE(16, 0) S3_CASE(16, 1) S3_CASE(16, 2) S3_CASE(16, 3)
#undef S3_CASE
        } else {
#define ZIN16_CASE(F)                                                                                     \
    if (flg == F) {                                                                                       \
        if (rp4)                                                                                          \
            stack_zin16_role<KS, OUT | 512, F>(rl, lk, scan_smem, gate_word_p, T, H, NT, blk);            \
        else                                                                                              \
            stack_zin16_role<KS, OUT, F>(rl, lk, scan_smem, gate_word_p, T, H, NT, blk);                  \
    }
            ZIN16_CASE(0) ZIN16_CASE(1) ZIN16_CASE(2) ZIN16_CASE(3)
#undef ZIN16_CASE
        }
    }
    if (lk.dbg && threadIdx.x == 0) lk.dbg[3] = (unsigned)wall_clock64();
    SFSN_WG_STAMP(p.wg_times, 1);
    stack_exit(p, gate_word_p);
}

// OUT: bit 0 fp32 spikes, bit 1 int8 spikes (always).  The 4-row repacked epilogue (bit 9) is selected per role from rpw.
template <int KS, int OUT>
__global__ __launch_bounds__(512) void gsn_stack_kernel(const StackParams p) {
    // (no static __shared__ here: the LDS-DMA destinations of the scan bodies are absolute LDS addresses from offset 0)
    extern __shared__ __attribute__((aligned(16))) char scan_smem[];
    int* gate_word_p = reinterpret_cast<int*>(scan_smem + p.gate_off);
    int ri = -1;
    for (int i = 0; i < p.nroles; ++i)
        if ((int)blockIdx.x >= p.role[i].block0 && (int)blockIdx.x < p.role[i].block0 + p.role[i].nblocks) ri = i;
    if (ri < 0) {  // padding block (role ranges start at multiples of 8: producer and consumer share an XCD)
        stack_exit(p, gate_word_p);
        return;
    }
    SFSN_WG_STAMP(p.wg_times, 0);
    const StackRoleDev& rl = p.role[ri];
    const int blk = (int)blockIdx.x - rl.block0;
    StackLink lk;
    lk.in = nullptr; lk.n_in = 0; lk.out = nullptr; lk.err = p.prog; lk.lag = p.lag;
    lk.dbg = p.dbg ? p.dbg + 4 * blockIdx.x : nullptr;
    if (lk.dbg && threadIdx.x == 0) lk.dbg[2] = (unsigned)wall_clock64();
    if (rl.pub) lk.out = p.prog + 2 + blockIdx.x;
    const int my_rpw = rl.kind == STACK_PROJ ? 16 : rl.rpw;
    if (rl.src >= 0) {  // the producer workgroups that own my rows (a PROJ producer: `split` workgroups per 16-row block)
        const StackRoleDev& sr = p.role[rl.src];
        const int r0 = (rl.kind == STACK_PROJ ? blk / rl.split : blk) * my_rpw;
        int r1 = r0 + my_rpw - 1;
        if (r1 > rl.R - 1) r1 = rl.R - 1;
        const int b0 = r0 / rl.src_rpw, b1 = r1 / rl.src_rpw;
        lk.in = p.prog + 2 + sr.block0 + b0 * sr.split;
        lk.n_in = (b1 - b0 + 1) * sr.split;
    }
    const int T = p.T, H = p.H, NT = p.NT;
    if (rl.kind == STACK_FUSED) {
        if constexpr (KS <= 4) {
            if (rl.pub)
                stack_fused_role<KS, OUT, true>(rl, lk, scan_smem, gate_word_p, T, H, NT, blk);
            else
                stack_fused_role<KS, OUT, false>(rl, lk, scan_smem, gate_word_p, T, H, NT, blk);
        }
    } else if (rl.kind == STACK_PROJ) {
        if constexpr (KS == 5) stack_proj_role<KS>(rl, lk, scan_smem, gate_word_p, T, H, NT, blk);
    } else {
        const bool rp4 = rl.rpw == 4;
        const int flg = (rl.src >= 0 ? 1 : 0) | (rl.pub ? 2 : 0);
#define ZIN_CASE(F)                                                                                       \
    if (flg == F) {                                                                                       \
        if (rp4)                                                                                          \
            stack_zin_role<KS, OUT | 512, F>(rl, lk, scan_smem, gate_word_p, T, H, NT, blk);               \
        else                                                                                              \
            stack_zin_role<KS, OUT, F>(rl, lk, scan_smem, gate_word_p, T, H, NT, blk);                     \
    }
        if constexpr (KS == 5) {  // layer 0 publishes to a PROJ role; layers >= 1 are gated on one (the last one publishes nothing)
            ZIN_CASE(0) ZIN_CASE(1) ZIN_CASE(2) ZIN_CASE(3)
        } else {  // layers >= 1 are FUSED roles: a ZIN role is a layer 0
            ZIN_CASE(0) ZIN_CASE(2)
        }
#undef ZIN_CASE
    }
    SFSN_WG_STAMP(p.wg_times, 1);
    stack_exit(p, gate_word_p);
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the full-band stack (256 < H <= 320) with IO-specialised waves -- 768 threads per workgroup: scan roles = scan3w_role
// (sfsn_scan3w_dev.h: ten compute waves x two tiles + loader + storer), PROJ roles = stack_proj_role on twelve waves.  Same roles,
// links and arithmetic as gsn_stack_kernel<5, OUT>; bit-identical results.
// ---------------------------------------------------------------------------------------------------------------------
template <int OUT>
__global__ __launch_bounds__(768) void gsn_stack_fb_kernel(const StackParams p) {
    extern __shared__ __attribute__((aligned(16))) char scan_smem[];
    constexpr int KS = 5;
    int* gate_word_p = reinterpret_cast<int*>(scan_smem + p.gate_off);
    int ri = -1;
    StackLink lk;
    lk.in = nullptr; lk.n_in = 0; lk.out = nullptr; lk.err = p.prog; lk.lag = p.lag; lk.dbg = nullptr;
    for (int i = 0; i < p.nroles; ++i)
        if ((int)blockIdx.x >= p.role[i].block0 && (int)blockIdx.x < p.role[i].block0 + p.role[i].nblocks) ri = i;
    if (ri < 0) {  // padding block
        stack_exit(p, gate_word_p);
        return;
    }
    SFSN_WG_STAMP(p.wg_times, 0);
    const StackRoleDev& rl = p.role[ri];
    const int blk = (int)blockIdx.x - rl.block0;
#ifdef SFSN_EXPERIMENTS
    lk.probe = p.wg_times ? p.wg_times + 2 * gridDim.x + 48 * blockIdx.x : nullptr;  // (behind the launch's stamps: launch_stack_fb)
#endif
    if (rl.pub) lk.out = p.prog + 2 + blockIdx.x;
    if (rl.src >= 0) {  // the producer workgroups that own my rows (a PROJ producer: `split` workgroups per 16-row block)
        const int my_rpw = rl.kind == STACK_PROJ ? 16 : rl.rpw;
        const StackRoleDev& sr = p.role[rl.src];
        const int r0 = (rl.kind == STACK_PROJ ? blk / rl.split : blk) * my_rpw;
        int r1 = r0 + my_rpw - 1;
        if (r1 > rl.R - 1) r1 = rl.R - 1;
        const int b0 = r0 / rl.src_rpw, b1 = r1 / rl.src_rpw;
        lk.in = p.prog + 2 + sr.block0 + b0 * sr.split;
        lk.n_in = (b1 - b0 + 1) * sr.split;
    }
    const int T = p.T, H = p.H, NT = p.NT;
    if (rl.kind == STACK_PROJ) {
        stack_proj_role<KS, 12>(rl, lk, scan_smem, gate_word_p, T, H, NT, blk);
    } else {
        const int flg = (rl.src >= 0 ? 1 : 0) | (rl.pub ? 2 : 0);
        Scan3Role r3;
        r3.zin = rl.zin; r3.w_hh = rl.w_hh; r3.w_dq = rl.w_dq; r3.bias = rl.bias; r3.bn_alpha = rl.bn_alpha; r3.bn_beta = rl.bn_beta;
        r3.h_state = rl.h_state; r3.c_state = rl.c_state; r3.spikes_f32 = rl.spikes_f32; r3.spikes_i8 = rl.spikes_i8;
        r3.R = rl.R; r3.row0 = blk * rl.rpw; r3.count = rl.count; r3.lsplit = p.lsplit;
#ifdef SFSN_EXPERIMENTS
        r3.probe = lk.probe;
#endif
#define S3W_CASE(RPW_, F) \
    if (rl.rpw == RPW_ && flg == F) scan3w_role<KS, RPW_, OUT, F>(r3, lk, scan_smem, T, H, NT);
        S3W_CASE(4, 0) S3W_CASE(4, 1) S3W_CASE(4, 2) S3W_CASE(4, 3)
        S3W_CASE(8, 0) S3W_CASE(8, 1) S3W_CASE(8, 2) S3W_CASE(8, 3)
#undef S3W_CASE
    }
    SFSN_WG_STAMP(p.wg_times, 1);
    stack_exit(p, gate_word_p);
}

// =====================================================================================================
// host side
// =====================================================================================================
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

extern "C" size_t sfsn_stack_scratch_bytes(int n_layers, int n_segs, int rows_total) {
    // progress counters: one per workgroup (at most one per 4 rows per layer, plus role padding and PROJ roles) + error word
    if (n_layers <= 0 || n_segs <= 0 || rows_total <= 0) return 0;
    const size_t blocks = (size_t)n_layers * ((size_t)(rows_total + 3) / 4 + (size_t)(rows_total + 15) / 16 + 16 * (size_t)n_segs);
    return (blocks + 1 + 16) * sizeof(unsigned) * 5;  // counters + (optional) two debug words per workgroup
}

// column parts per 16-row block of a narrow PROJ role: the largest of 1 / 2 / 4 that fits the padding (SFSN_PROJ_SPLIT caps it: A/B runs)
static int proj_split_host(int room) {
    int cap = 4;
    if (const char* e = getenv("SFSN_PROJ_SPLIT")) cap = atoi(e);
    int sp = 1;
    while (sp * 2 <= room && sp * 2 <= cap) sp *= 2;
    return sp;
}

template <int OUT>
static int launch_stack_fb(const StackParams& p, int blocks, int lds, hipStream_t st) {
    auto kern = gsn_stack_fb_kernel<OUT>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return SFSN_EHIP;
    StackParams q = p;
    q.wg_times = sfsn_wgprobe_take(5, 25 * blocks);  // 2 stamps + 12 waves x 4 stall counters per workgroup (S3_PB_*)
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(768), lds, st, q);
    return hipGetLastError() == hipSuccess ? SFSN_OK : SFSN_EHIP;
}

template <int KS, int OUT>
static int launch_stack(const StackParams& p, int blocks, int lds, hipStream_t st) {
    auto kern = gsn_stack_kernel<KS, OUT>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return SFSN_EHIP;  // (per device, cheap: set on every launch -- a process may drive several GPUs)
    StackParams q = p;
    q.wg_times = sfsn_wgprobe_take(4, blocks);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, st, q);
    return hipGetLastError() == hipSuccess ? SFSN_OK : SFSN_EHIP;
}

extern "C" int sfsn_gsn_stack_scan(const sfsn_scan_segment* segs, const sfsn_fused_input* fin, int n_layers, int n_segs, int T, int H,
                                   const int* rows_per_wg, int lag, void* scratch, size_t scratch_bytes, void* stream) {
    return sfsn_gsn_stack_scan_x(segs, fin, nullptr, n_layers, n_segs, T, H, rows_per_wg, lag, scratch, scratch_bytes, stream);
}

static int stack_scan_impl(const sfsn_scan_segment* segs, const sfsn_fused_input* fin, const sfsn_fused_x* fx, int n_layers, int n_segs, int T,
                           int H, const int* rows_per_wg, int lag, void* scratch, size_t scratch_bytes, void* stream, int w16);

extern "C" int sfsn_gsn_stack_scan_x(const sfsn_scan_segment* segs, const sfsn_fused_input* fin, const sfsn_fused_x* fx, int n_layers,
                                     int n_segs, int T, int H, const int* rows_per_wg, int lag, void* scratch, size_t scratch_bytes,
                                     void* stream) {
    return stack_scan_impl(segs, fin, fx, n_layers, n_segs, T, H, rows_per_wg, lag, scratch, scratch_bytes, stream, 0);
}

// The same launch for weights packed with 16 bits (sfsn_w3_pack_bits(.., 16, ..): digit plane 0 of every recurrent / spike-input matrix
// is zero): the sub-band pair layout (IO-wave scan roles, FUSEDX3, FUSED3) with the zero plane's matrix instructions skipped -- the same
// sums, 12 instead of 18 per tile and frame in the FUSED3 role.  SFSN_EUNSUPPORTED for any other layout (call sfsn_gsn_stack_scan_x:
// same results).  The 16-bit report mode of BASELINE configs[2] (module.weight_bits = 16), not the parity mode.
extern "C" int sfsn_gsn_stack_scan_x_w16(const sfsn_scan_segment* segs, const sfsn_fused_input* fin, const sfsn_fused_x* fx, int n_layers,
                                         int n_segs, int T, int H, const int* rows_per_wg, int lag, void* scratch, size_t scratch_bytes,
                                         void* stream) {
    return stack_scan_impl(segs, fin, fx, n_layers, n_segs, T, H, rows_per_wg, lag, scratch, scratch_bytes, stream, 1);
}

static int stack_scan_impl(const sfsn_scan_segment* segs, const sfsn_fused_input* fin, const sfsn_fused_x* fx, int n_layers, int n_segs, int T,
                           int H, const int* rows_per_wg, int lag, void* scratch, size_t scratch_bytes, void* stream, int w16) {
    if (!segs || !fin || n_layers <= 0 || n_segs <= 0 || n_segs > SFSN_MAX_SEGMENTS || T < 0 || H <= 0 || !scratch) return SFSN_EINVAL;
    if (H % 16 != 0 || H > SFSN_MAX_HIDDEN) return SFSN_EUNSUPPORTED;
    const int KS = (H + 63) / 64, NT = H / 16, HP = KS * 64;
    bool fused = H <= 256;  // both matrices of a layer >= 1 fit one CU
    // H <= 256 with input-term buffers for the layers >= 1: the WIDE flavour (16-wave scans + PROJ16 roles); without them
    // the 8-wave fused-input roles
    bool wide = fused;
    for (int l = 1; l < n_layers && wide; ++l)
        for (int i = 0; i < n_segs; ++i)
            if (!segs[l * n_segs + i].zin) wide = false;
    if (n_layers == 1 && fused) wide = true;
    if (getenv("SFSN_STACK_NARROW")) wide = false;
    // round 4: no input-term buffers, H <= 224, 8 rows per workgroup in the layers >= 1: the wide kernel with FUSED3 roles (the input
    // product inside the IO-wave scan) instead of the 8-wave FUSED roles (SFSN_STACK_FUSED8=1 keeps those: A/B runs)
    bool inscan = fused && !wide && NT <= 14 && n_layers >= 2 && !getenv("SFSN_STACK_FUSED8") && !getenv("SFSN_SCAN_V2");
    for (int l = 1; l < n_layers && inscan; ++l)
        if ((rows_per_wg ? rows_per_wg[l] : 8) != 8) inscan = false;
    if (inscan) wide = true;
    if (wide) fused = false;
    if (w16 && !(inscan && (KS == 3 || KS == 4))) return SFSN_EUNSUPPORTED;  // the two-plane form exists for the pair layout only
    // round 5: 256 < H <= 320 at 4 / 8 rows per workgroup: the 768-thread kernel with IO-specialised scan roles (SFSN_SCAN_V2=1 keeps
    // round 2's bodies: A/B runs)
    // Where it is used -- measured (scripts/exp_fb3_r05.py, exp_fb3_chunks_r05.sh; B = 64): the stack ALONE on the chip 1.09 / 1.34 ms
    // per 1000 frames at 4 / 8 rows against 1.30 / 1.60 for round 2's bodies, the twelve-lane region (8 rows) +1.5 %.  As a 240-380 frame
    // chunk BESIDE the sub-band pair launch (the strict forward's overlapped schedule, 4 rows) it first LOST to round 2's bodies (2.82-2.97
    // against 2.74-2.75 ms per forward); the per-wave stall counters of an EXPERIMENTS build (S3_PB_*, scripts/exp_beside_r05.py) found
    // the 16-frame hand-off hysteresis and the store-bound PROJ role (see stack_proj_role) -- with lag 4 and the PROJ role split by
    // columns the IO-wave kernel wins or ties there too (B = 4 / 16 / 32 / 64: 1.86 / 1.99 / 2.22 / 2.58-2.63 against 1.95 / 2.08 / 2.30 /
    // 2.61-2.64; no fp32 spike tensors 2.42-2.45 against 2.52-2.53; T = 2000 +1.4 %).  Only very short launches keep round 2's bodies (the
    // new kernel's hand-off chain costs ~24 us more to fill).  SFSN_STACK_FB3=1 / 0 forces it on / off (A/B runs, tests).
    bool fb3 = !fused && !wide && KS == 5 && !getenv("SFSN_SCAN_V2") && !getenv("SFSN_STACK_FB_V2");
    bool any4 = false;
    for (int l = 0; l < n_layers && fb3; ++l) {
        const int rp = rows_per_wg ? rows_per_wg[l] : 8;
        if (rp != 4 && rp != 8) fb3 = false;
        if (rp == 4) any4 = true;
    }
    if (const char* e = getenv("SFSN_STACK_FB3")) fb3 = fb3 && atoi(e) != 0;
    else if (any4 && T < 128) fb3 = false;
    const int roles_per_layer = (fused || inscan) ? 1 : 2;
    if (n_segs * (1 + (n_layers - 1) * roles_per_layer) > STACK_MAX_ROLES) return SFSN_EUNSUPPORTED;
    if (lag < 0) return SFSN_EINVAL;
    StackParams p;
    p.wg_times = nullptr;
    const int out = 2 | (segs[0].spikes_f32 ? 1 : 0);
    int blocks = 0, nroles = 0, lds = 0, rows_total = 0;
    int prev_role[SFSN_MAX_SEGMENTS];
    for (int l = 0; l < n_layers; ++l) {
        int rpw = rows_per_wg ? rows_per_wg[l] : 8;
        if (rpw != 4 && rpw != 8 && rpw != 16) return SFSN_EINVAL;
        for (int i = 0; i < n_segs; ++i) {
            const sfsn_scan_segment& s = segs[l * n_segs + i];
            const sfsn_fused_input& f = fin[l * n_segs + i];
            if (!s.spikes_i8 || (s.spikes_f32 != nullptr) != ((out & 1) != 0) || s.membrane) return SFSN_EINVAL;
            if (s.R <= 0 || !s.w_hh || !s.w_dq || !s.bias || !s.bn_alpha || !s.bn_beta || !s.h_state || !s.c_state) return SFSN_EINVAL;
            if (s.R != segs[i].R) return SFSN_EINVAL;  // a segment has the same rows in every layer
            if (!aligned16(s.w_hh) || !aligned16(s.h_state) || !aligned16(s.c_state) || !aligned16(s.spikes_f32) || !aligned16(s.spikes_i8) ||
                !aligned16(s.zin))
                return SFSN_EINVAL;
            if (l == 0) rows_total += s.R;
            const bool last = l == n_layers - 1;
            if (l > 0) {
                if (!f.w_ih || !f.w_ih_dq || !aligned16(f.w_ih)) return SFSN_EINVAL;
                if (f.spikes_in != segs[(l - 1) * n_segs + i].spikes_i8) return SFSN_EINVAL;  // the layer below, same segment
            }
            // layer 0 of a segment with fx[i].x: the real-valued input product inside the scan (FUSEDX3) -- the wide kernel's IO-wave
            // roles only, 8 rows per workgroup, even I <= 64, whole 8-row blocks, 16-byte aligned rows
            const bool xrole = l == 0 && fx && fx[i].x;
            if (xrole) {
                const sfsn_fused_x& g = fx[i];
                const bool can = (inscan || (n_layers == 1 && wide)) && NT <= 14 && rpw == 8 && !getenv("SFSN_SCAN_V2");
                if (!g.w_ih || g.I <= 0) return SFSN_EINVAL;
                if (!can || g.I > 64 || (g.I & 1) || (s.R & 7) || !aligned16(g.x)) return SFSN_EUNSUPPORTED;
            }
            if (!xrole && (l == 0 || !(fused || inscan))) {
                if (!s.zin) return SFSN_EINVAL;  // layer 0: the precomputed input term; H > 256: the PROJ role's output buffer
            }
            if (l > 0 && !fused && !inscan) {  // PROJ role first (lower block indices than the scan it feeds)
                StackRoleDev& r = p.role[nroles];
                r = StackRoleDev{};
                r.bias = s.bias; r.zin = const_cast<float*>(s.zin); r.spikes_in = f.spikes_in; r.w_ih = f.w_ih; r.w_ih_dq = f.w_ih_dq;
                const int prows = wide ? 32 : 16;
                r.R = s.R; r.kind = STACK_PROJ; r.rpw = prows; r.block0 = blocks; r.nblocks = (s.R + prows - 1) / prows;
                r.split = 1;
                if (!wide) {  // the role's block range is padded to a multiple of eight: the padding takes column parts (stack_proj_role)
                    const int pad = (r.nblocks + 7) & ~7;
                    r.split = proj_split_host(pad / r.nblocks);
                    r.nblocks *= r.split;
                }
                r.src = prev_role[i]; r.src_rpw = p.role[prev_role[i]].rpw; r.pub = 1;
                blocks = (blocks + r.nblocks + 7) & ~7;
                prev_role[i] = nroles++;
                int need = ProjLayout<5>::bytes(NT);
                if (wide) {
                    need = KS == 1 ? Proj16Layout<1>::BYTES : KS == 2 ? Proj16Layout<2>::BYTES : KS == 3 ? Proj16Layout<3>::BYTES : Proj16Layout<4>::BYTES;
                    const int n3 = KS == 1 ? Proj3Layout<1>::BYTES : KS == 2 ? Proj3Layout<2>::BYTES : KS == 3 ? Proj3Layout<3>::BYTES : Proj3Layout<4>::BYTES;
                    if (n3 > need) need = n3;
                }
                if (need > lds) lds = need;
            }
            StackRoleDev& r = p.role[nroles];
            r = StackRoleDev{};
            r.w_hh = s.w_hh; r.w_dq = s.w_dq; r.bias = s.bias; r.bn_alpha = s.bn_alpha; r.bn_beta = s.bn_beta;
            r.h_state = s.h_state; r.c_state = s.c_state; r.spikes_f32 = s.spikes_f32; r.spikes_i8 = s.spikes_i8; r.count = s.spike_count;
            r.zin = const_cast<float*>(s.zin);
            r.R = s.R; r.rpw = rpw; r.block0 = blocks; r.nblocks = (s.R + rpw - 1) / rpw; r.split = 1;
            r.pub = last ? 0 : 1;
            if (xrole) {
                r.kind = STACK_FUSEDX3; r.src = -1; r.src_rpw = rpw; r.x = fx[i].x; r.w_ih_f32 = fx[i].w_ih; r.I = fx[i].I; r.zin = nullptr;
            } else if (l == 0) {
                r.kind = STACK_ZIN; r.src = -1; r.src_rpw = rpw;
            } else if (fused || inscan) {
                r.kind = inscan ? STACK_FUSED3 : STACK_FUSED; r.spikes_in = f.spikes_in; r.w_ih = f.w_ih; r.w_ih_dq = f.w_ih_dq;
                r.src = prev_role[i]; r.src_rpw = p.role[prev_role[i]].rpw;
            } else {
                r.kind = STACK_ZIN; r.src = prev_role[i]; r.src_rpw = wide ? 32 : 16;
            }
            blocks = (blocks + r.nblocks + 7) & ~7;
            prev_role[i] = nroles++;
            int need = 0;
            if (wide) {
                switch (KS) {  // (the repacked-epilogue variant has the same layout)
                    case 1: need = ScanCfg<1, 1, 16, 1, 3, 0>::LDS_BYTES; break;
                    case 2: need = ScanCfg<1, 2, 16, 1, 3, 0>::LDS_BYTES; break;
                    case 3: need = ScanCfg<1, 3, 16, 1, 3, 0>::LDS_BYTES; break;
                    default: need = ScanCfg<1, 4, 16, 1, 3, 0>::LDS_BYTES; break;
                }
                if (NT <= 14) {  // the IO-wave scan's layout (ring of up to 9 frames of rpw rows + the state buffers)
                    const int dfit = 65536 / (((rpw * 14 * 4 + 63) / 64) * 1024);
                    const int n3 = (dfit < 9 ? dfit : 9) * (((rpw * NT * 4 + 63) / 64) * 1024) + 2 * 16 * (HP + 32) + 16;
                    if (n3 > need) need = n3;
                }
                if (r.kind == STACK_FUSEDX3) {
                    const int n3x = KS == 1 ? Scan3xCfg<1, 2>::lds_bytes(NT) : KS == 2 ? Scan3xCfg<2, 2>::lds_bytes(NT)
                                  : KS == 3 ? Scan3xCfg<3, 2>::lds_bytes(NT) : Scan3xCfg<4, 2>::lds_bytes(NT);
                    if (n3x > need) need = n3x;
                }
                if (r.kind == STACK_FUSED3) {  // + the input-spike ring, two digit planes of W_ih and the product's constants
                    const int n3i = KS == 1 ? Scan3iCfg<1, 3>::lds_bytes(NT) : KS == 2 ? Scan3iCfg<2, 3>::lds_bytes(NT)
                                  : KS == 3 ? Scan3iCfg<3, 3>::lds_bytes(NT) : Scan3iCfg<4, 3>::lds_bytes(NT);
                    if (n3i > need) need = n3i;
                }
            } else if (r.kind == STACK_FUSED) {
                switch (KS) {
                    case 1: need = FusedLayout<1>::bytes(rpw, NT); break;
                    case 2: need = FusedLayout<2>::bytes(rpw, NT); break;
                    case 3: need = FusedLayout<3>::bytes(rpw, NT); break;
                    default: need = FusedLayout<4>::bytes(rpw, NT); break;
                }
            } else {
                switch (KS) {  // (the repacked-epilogue variant has the same layout)
                    case 1: need = ScanCfg<1, 1, 8, 1, 3, 0>::LDS_BYTES; break;
                    case 2: need = ScanCfg<1, 2, 8, 1, 3, 0>::LDS_BYTES; break;
                    case 3: need = ScanCfg<1, 3, 8, 2, 3, 0>::LDS_BYTES; break;
                    case 4: need = ScanCfg<1, 4, 8, 2, 3, 0>::LDS_BYTES; break;
                    default: need = ScanCfg<1, 5, 8, 3, 3, 1>::LDS_BYTES; break;
                }
                if (fb3) need = rpw == 4 ? Scan3wCfg<5, 4, 3>::lds_bytes(NT) : Scan3wCfg<5, 8, 3>::lds_bytes(NT);
            }
            if (need > lds) lds = need;
        }
    }
    if (lds > 160 * 1024 - 64) return SFSN_EUNSUPPORTED;
    if ((size_t)(blocks + 2) * sizeof(unsigned) > scratch_bytes) return SFSN_EINVAL;
    p.prog = static_cast<unsigned*>(scratch);
    p.dbg = nullptr;
    if (getenv("SFSN_STACK_DEBUG") && (size_t)(blocks + 2) * 5 * sizeof(unsigned) <= scratch_bytes) p.dbg = p.prog + blocks + 2;
    p.nroles = nroles; p.T = T; p.H = H; p.NT = NT; p.lag = lag; p.nblocks = blocks;
    p.v2 = getenv("SFSN_SCAN_V2") ? 1 : 0;
    p.lsplit = sfsn_s3_lsplit_host();
    p.lsplit_x = sfsn_s3x_lsplit_host();
    // timing switches of the hand-off roles (wrong results by design: stores dropped, waits skipped): compiled in only with
    // -DSFSN_EXPERIMENTS (make EXTRA=-DSFSN_EXPERIMENTS, scripts/exp_stack_r03.sh) -- a stray environment variable must not be
    // able to corrupt a production launch
#ifdef SFSN_EXPERIMENTS
    p.exp_flags = getenv("SFSN_STACK_EXP") ? atoi(getenv("SFSN_STACK_EXP")) : 0;
#else
    p.exp_flags = 0;
#endif
    lds = (lds + 15) & ~15;
    p.gate_off = lds;
    lds += 16;
    (void)HP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // (no memset here: the counters are zero on entry -- the caller zeroes the scratch buffer ONCE, and every launch's last
    //  workgroup leaves them zeroed again)
    if (p.dbg && hipMemsetAsync(p.dbg, 0, (size_t)(blocks + 1) * 4 * sizeof(unsigned), st) != hipSuccess) return SFSN_EHIP;
#define WIDE_CASE(KS_, OUT_)                                                                                                  \
    if (wide && KS == KS_ && out == OUT_) {                                                                                   \
        auto kern = gsn_stack_wide_kernel<KS_, OUT_>;                                                                         \
        if (lds > 64 * 1024 &&                                                                                                \
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) \
            return SFSN_EHIP;                                                                                                 \
        StackParams q = p;                                                                                                    \
        q.wg_times = sfsn_wgprobe_take(6, 33 * blocks); /* 2 stamps + 16 waves x 4 stall counters per workgroup (S3_PB_*) */  \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds, st, q);                                                       \
        return hipGetLastError() == hipSuccess ? SFSN_OK : SFSN_EHIP;                                                         \
    }
#define WIDE16_CASE(KS_, OUT_)                                                                                                \
    if (w16 && wide && !p.v2 && KS == KS_ && out == OUT_) {                                                                   \
        auto kern = gsn_stack_wide_kernel<KS_, OUT_, 1>;                                                                      \
        if (lds > 64 * 1024 &&                                                                                                \
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) \
            return SFSN_EHIP;                                                                                                 \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds, st, p);                                                       \
        return hipGetLastError() == hipSuccess ? SFSN_OK : SFSN_EHIP;                                                         \
    }
    WIDE16_CASE(3, 2) WIDE16_CASE(3, 3) WIDE16_CASE(4, 2) WIDE16_CASE(4, 3)
#undef WIDE16_CASE
    if (w16) return SFSN_EUNSUPPORTED;
    WIDE_CASE(1, 2) WIDE_CASE(1, 3) WIDE_CASE(2, 2) WIDE_CASE(2, 3) WIDE_CASE(3, 2) WIDE_CASE(3, 3) WIDE_CASE(4, 2) WIDE_CASE(4, 3)
#undef WIDE_CASE
    if (fb3 && out == 2) return launch_stack_fb<2>(p, blocks, lds, st);
    if (fb3 && out == 3) return launch_stack_fb<3>(p, blocks, lds, st);
#define STACK_CASE(KS_, OUT_) \
    if (KS == KS_ && out == OUT_) return launch_stack<KS_, OUT_>(p, blocks, lds, st);
    STACK_CASE(1, 2) STACK_CASE(1, 3) STACK_CASE(2, 2) STACK_CASE(2, 3) STACK_CASE(3, 2) STACK_CASE(3, 3) STACK_CASE(4, 2) STACK_CASE(4, 3)
    STACK_CASE(5, 2) STACK_CASE(5, 3)
#undef STACK_CASE
    return SFSN_EUNSUPPORTED;
}
