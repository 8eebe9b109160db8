// sfsn_hop.hip -- the streaming hop: `hop` new frames of B clips through the whole live model in ONE launch (gfx950 only).
//
// BASELINE configs[4] (B = 1 per GPU, hop = 1 frame, state resident between frames).  A one-frame hop through the offline
// kernels is ~15 launches of 4.5-25 us each, every one of them a launch boundary and nothing else: the work of a frame is a
// few hundred MFMAs.  Here the whole frame is one launch of a few dozen small workgroups:
//
//   * an AGENT is one wave: it owns one 16-neuron output tile of one layer for one 16-row tile, keeps that tile's weight
//     fragments (W_hh, and W_ih of its layer) in registers, and its slice of the membrane in registers for the launch;
//   * the stages full-band layer 0 -> ... -> full-band projection -> sub-band layer 0 (all groups side by side) -> ... ->
//     sub-band projection + deep filter hand each frame over through L2: int8 spike bytes (or fp32 projections) leave with
//     write-through (sc1) stores, the wave drains its store queue and publishes its own 32-bit frame counter; consumers poll
//     the counters of the waves they depend on with one lane each (wave-wide ballot) and read the payload with sc1 loads.
//     No workgroup barrier sits on a hand-off: waves are the unit of synchronisation;
//   * the recurrent product h(t-1).W_hh of every layer is issued BEFORE the wave starts to wait for its input: only the
//     input-dependent half of a layer is on the frame's critical path;
//   * every weight fragment a wave needs is requested at launch, i.e. while the stages upstream are still computing.
//
// State between launches lives in device memory (membranes per agent, last spikes double-buffered by launch parity so that a
// fast wave cannot overwrite what a late peer still has to read, deep-filter history shifted by the thread that owns the bin).
//
// Arithmetic: the expressions of features_kernel / input_proj_kernel (fp32 MFMA chain, same k order) / spike_proj_kernel /
// scan_body / deepfilter_kernel, so a session built on this launch is bit-identical to one built on those kernels.
// Deadlock freedom: producers have lower block indices than their consumers, workgroups are dispatched in index order and the
// launch is refused unless every workgroup can be resident at once; every spin is bounded all the same (error word).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "sfsn.h"

#include "sfsn_scan_dev.h"
#include "sfsn_feat_dev.h"

#define HOP_THREADS 256
#define HOP_WAVES 4
#define HOP_MAX_SEQS (1 + SFSN_HOP_MAX_GROUPS)
#define HOP_MAX_STAGES (HOP_MAX_SEQS * (SFSN_HOP_MAX_LAYERS + 1))
#define HOP_KS_MAX 5     // 64-wide k steps of the int8 products: H <= 320
#define HOP_KC_MAX 12    // 16-wide k chunks of the fp32 input product: I <= 192
#define HOP_NU_MAX 3     // feature slots per lane: I <= 192
#define HOP_SPIN_LIMIT 2000000u
#define HOP_CNT0 4       // counters: [0] error word, [1] exit counter, [2] launch counter, [3] reserved, [4 + agent] frames done

struct HopLayerDev {
    const float* w_ih_f32;
    const int8_t* w_ih;
    const float* w_ih_dq;
    const int8_t* w_hh;
    const float* w_hh_dq;
    const float* bias;
    const float* alpha;
    const float* beta;
    int8_t* h[2];
    float* c;
    int8_t* spikes;
};
struct HopSeqDev {
    HopLayerDev layer[SFSN_HOP_MAX_LAYERS];
    const int8_t* w_p;
    const float* w_p_dq;
    const float* b_p;
    const float* ln_w;
    const float* ln_b;
    int nl, H, P, R, KS, NT, PT, I, I1, KC;
    int lo, N, ctr, nbr, ctr_fb, nbr_fb, norm, df, fc;
    float eps;
};
struct HopStageDev {
    int seq, layer;  // layer = -1: the projection (+ deep filter for a sub-band group)
    int wg0, nwg;    // workgroups [wg0, wg0 + nwg)
    int agent0;      // first counter of the stage; agent (rt, tile) = agent0 + rt * ntpad + tile
    int ntile, ntpad, nrt;
    int prod;        // stage whose agents feed this one (-1: the input frames)
};
struct HopParams {
    HopSeqDev seq[HOP_MAX_SEQS];
    HopStageDev st[HOP_MAX_STAGES];
    int nseq, nstage, nblocks, nagents;
    int B, F, S, hop, D, FB, fcov;
    float fdrc;
    const float* inp;
    float* hist;
    float* fb_out;
    float* enh;
    float* mag;
    unsigned* cnt;
};

// ---- coherent accesses (agent scope: global_load / global_store ... sc1) ------------------------------------------------
__device__ __forceinline__ unsigned ld_agent(const void* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(void* p, unsigned v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ v4i ld16_agent(const void* p) {
    const unsigned* u = reinterpret_cast<const unsigned*>(p);
    v4i r;
    r[0] = (int)ld_agent(u);
    r[1] = (int)ld_agent(u + 1);
    r[2] = (int)ld_agent(u + 2);
    r[3] = (int)ld_agent(u + 3);
    return r;
}

// Wave-level wait: every lane watches one counter of [cnt + a0, cnt + a0 + n), the wave leaves when all are >= need.
// Returns false when the bounded spin expired (error word set); the caller then stops waiting for anything (garbage out,
// the host raises) but keeps executing its barriers.
__device__ __forceinline__ bool hop_wait(const unsigned* cnt, int a0, int n, unsigned need, int lane, bool ok) {
    if (!ok) return false;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        for (unsigned spins = 0;; ++spins) {
            const unsigned v = i < n ? ld_agent(cnt + HOP_CNT0 + a0 + i) : 0xffffffffu;
            if (__ballot(v < need) == 0) break;
            if (spins > HOP_SPIN_LIMIT) {
                if (lane == 0) st_agent(const_cast<unsigned*>(cnt), 1u);
                return false;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return true;
}

__device__ __forceinline__ void hop_publish(unsigned* cnt, int agent, unsigned frames, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's payload stores have been written through
    if (lane == 0) st_agent(cnt + HOP_CNT0 + agent, frames);
}

// The last workgroup to leave zeroes the progress counters and the exit counter and advances the launch counter: the next
// launch starts clean without a memset in front of it.
__device__ __forceinline__ void hop_exit(const HopParams& p, int* word) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) *reinterpret_cast<volatile int*>(word) = atomicAdd(p.cnt + 1, 1u) == (unsigned)(p.nblocks - 1) ? 1 : 0;
    __syncthreads();
    if (*reinterpret_cast<volatile int*>(word)) {
        for (int i = threadIdx.x; i < p.nagents; i += blockDim.x) st_agent(p.cnt + HOP_CNT0 + i, 0u);
        if (threadIdx.x == 0) {
            st_agent(p.cnt + 2, ld_agent(p.cnt + 2) + 1u);
            st_agent(p.cnt + 1, 0u);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// layer role: 4 agents per workgroup (tiles 4w .. 4w+3 of one row tile).  LDS (layer 0 only): the feature rows of the tile.
// ---------------------------------------------------------------------------------------------------------------------
template <bool L0>
__device__ __forceinline__ void hop_layer_role(const HopParams& p, const HopStageDev& sd, const HopSeqDev& sq, char* smem, unsigned epoch) {
    const int l = sd.layer;
    const HopLayerDev& L = sq.layer[l];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int wgl = (int)blockIdx.x - sd.wg0, wpr = sd.ntpad / HOP_WAVES;  // workgroups per row tile
    const int rt = wgl / wpr, tile_raw = (wgl - rt * wpr) * HOP_WAVES + wave;
    const bool active = tile_raw < sd.ntile;
    const int tile = active ? tile_raw : 0;
    const int agent = sd.agent0 + rt * sd.ntpad + tile_raw;
    const int H = sq.H, KS = sq.KS, NT = sq.NT, R = sq.R, HP = KS * 64, I = sq.I, KC = sq.KC;
    const int row = 16 * rt + n, rowc = row < R ? row : R - 1;
    const int cc = 16 * tile + 4 * q;
    const int hop = p.hop;
    float* xrow = reinterpret_cast<float*>(smem + 64);
    const int KPX = KC * 16 + 4;

    // ---- weights of my tile into registers (requested now: the stages upstream are still at work)
    v4i Whh[3][HOP_KS_MAX], Wih[3][HOP_KS_MAX];
    float W0[HOP_KC_MAX][4];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int ks = 0; ks < HOP_KS_MAX; ++ks) {
            Whh[d][ks] = v4i{0, 0, 0, 0};
            Wih[d][ks] = v4i{0, 0, 0, 0};
            if (ks < KS) {
                const size_t off = ((((size_t)d * NT + tile) * KS + ks) * 64 + lane) * 16;
                Whh[d][ks] = *reinterpret_cast<const v4i*>(L.w_hh + off);
                if (!L0) Wih[d][ks] = *reinterpret_cast<const v4i*>(L.w_ih + off);
            }
        }
#pragma unroll
    for (int c = 0; c < HOP_KC_MAX; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = c * 16 + q * 4 + e;
            W0[c][e] = (L0 && c < KC && k < I) ? L.w_ih_f32[(size_t)(16 * tile + n) * I + k] : 0.0f;
        }
    const v4f dq = *reinterpret_cast<const v4f*>(L.w_hh_dq + cc);
    v4f dqi = {0.0f, 0.0f, 0.0f, 0.0f};
    if (!L0) dqi = *reinterpret_cast<const v4f*>(L.w_ih_dq + cc);
    const v4f bf = *reinterpret_cast<const v4f*>(L.bias + cc);
    const v4f bg = *reinterpret_cast<const v4f*>(L.bias + H + cc);
    const v4f alpha = *reinterpret_cast<const v4f*>(L.alpha + cc);
    const v4f beta = *reinterpret_cast<const v4f*>(L.beta + cc);
    v4f db;
#pragma unroll
    for (int r = 0; r < 4; ++r) db[r] = bg[r] - bf[r];
    v4f c = *reinterpret_cast<const v4f*>(L.c + (size_t)rowc * H + cc);
    float lw[HOP_NU_MAX], lb[HOP_NU_MAX];
#pragma unroll
    for (int u = 0; u < HOP_NU_MAX; ++u) {
        const int j = lane + 64 * u;
        const bool in = L0 && j < I && sq.norm == SFSN_NORM_LAYERNORM;
        lw[u] = in ? sq.ln_w[j] : 0.0f;
        lb[u] = in ? sq.ln_b[j] : 0.0f;
    }
    if (L0) {  // zero the feature rows once: the k padding of the input product must read zeros
        for (int i = tid; i < 16 * KPX; i += HOP_THREADS) xrow[i] = 0.0f;
        __syncthreads();
    }

    const HopStageDev* ps = sd.prod >= 0 ? &p.st[sd.prod] : nullptr;
    const int8_t* hprev = L.h[epoch & 1u];
    int8_t* hnext = L.h[(epoch + 1u) & 1u];
    bool ok = true;
    unsigned pk = 0;

    for (int t = 0; t < hop; ++t) {
        // ---- recurrent half: needs frame t-1 of my own layer only
        const int8_t* hsrc = hprev;
        if (t > 0) {
            ok = hop_wait(p.cnt, sd.agent0 + rt * sd.ntpad, sd.ntile, (unsigned)t, lane, ok);
            hsrc = L.spikes + (size_t)(t - 1) * R * HP;
        }
        v4i a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
        {
            v4i b[HOP_KS_MAX];
#pragma unroll
            for (int ks = 0; ks < HOP_KS_MAX; ++ks)
                if (ks < KS) b[ks] = t > 0 ? ld16_agent(hsrc + (size_t)rowc * HP + ks * 64 + q * 16)
                                           : *reinterpret_cast<const v4i*>(hsrc + (size_t)rowc * HP + ks * 64 + q * 16);
#pragma unroll
            for (int ks = 0; ks < HOP_KS_MAX; ++ks)
                if (ks < KS) {
                    a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Whh[0][ks], b[ks], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Whh[1][ks], b[ks], a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Whh[2][ks], b[ks], a2, 0, 0, 0);
                }
        }
        // ---- input half
        v4f z;
        if constexpr (L0) {
            if (ps) ok = hop_wait(p.cnt, ps->agent0, ps->nrt * ps->ntpad, (unsigned)(t + 1), lane, ok);
            if (t > 0) __syncthreads();  // everyone has read the previous frame's rows
            const int nf = p.F - 1;
            for (int rl = wave; rl < 16; rl += HOP_WAVES) {
                const int frow = 16 * rt + rl;
                if (frow >= R) break;
                const int b = frow / sq.N, k = frow - b * sq.N;
                float v[HOP_NU_MAX];
                bool have[HOP_NU_MAX];
                float sum = 0.0f;
#pragma unroll
                for (int u = 0; u < HOP_NU_MAX; ++u) {
                    const int j = lane + 64 * u;
                    have[u] = j < I;
                    v[u] = 0.0f;
                    if (have[u]) {
                        if (j < sq.I1) {
                            const int bin = reflect_bin(sq.lo + k * sq.ctr - sq.nbr + j, nf);
                            const float2 xc = *reinterpret_cast<const float2*>(p.inp + (((size_t)b * p.F + bin) * hop + t) * 2);
                            v[u] = compress_mag(xc.x, xc.y, p.fdrc);
                        } else {
                            const int col = reflect_bin(sq.lo + k * sq.ctr_fb - sq.nbr_fb + (j - sq.I1), nf) % p.FB;
                            v[u] = __uint_as_float(ld_agent(p.fb_out + ((size_t)t * p.B + b) * p.FB + col));
                        }
                    }
                    sum += v[u];
                }
                float y[HOP_NU_MAX];
                if (sq.norm == SFSN_NORM_LAYERNORM) {
                    const float inv_I = 1.0f / (float)I;
                    const float mean = wave_sum(sum) * inv_I;
                    float ss = 0.0f;
#pragma unroll
                    for (int u = 0; u < HOP_NU_MAX; ++u) {
                        const float d = v[u] - mean;
                        if (have[u]) ss += d * d;
                    }
                    const float rstd = __builtin_amdgcn_rsqf(wave_sum(ss) * inv_I + sq.eps);
#pragma unroll
                    for (int u = 0; u < HOP_NU_MAX; ++u) y[u] = ((v[u] - mean) * rstd) * lw[u] + lb[u];
                } else {
#pragma unroll
                    for (int u = 0; u < HOP_NU_MAX; ++u) y[u] = v[u];
                }
#pragma unroll
                for (int u = 0; u < HOP_NU_MAX; ++u)
                    if (have[u]) xrow[rl * KPX + lane + 64 * u] = y[u];
            }
            __syncthreads();
            // fp32 MFMA chain in input_proj_kernel's k order: lane (n, q) holds k = 16c + 4q + e of row n
            v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
            const int xr = (16 * rt + n < R) ? n : (R - 1 - 16 * rt);
#pragma unroll
            for (int cch = 0; cch < HOP_KC_MAX; ++cch)
                if (cch < KC) {
                    const v4f bx = *reinterpret_cast<const v4f*>(xrow + xr * KPX + cch * 16 + q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(W0[cch][e], bx[e], acc, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = acc[r] + bf[r];
        } else {
            ok = hop_wait(p.cnt, ps->agent0 + rt * ps->ntpad, ps->ntile, (unsigned)(t + 1), lane, ok);
            const int8_t* ssrc = sq.layer[l - 1].spikes + (size_t)t * R * HP;
            v4i b[HOP_KS_MAX];
#pragma unroll
            for (int ks = 0; ks < HOP_KS_MAX; ++ks)
                if (ks < KS) b[ks] = ld16_agent(ssrc + (size_t)rowc * HP + ks * 64 + q * 16);
            v4i i0 = {0, 0, 0, 0}, i1 = {0, 0, 0, 0}, i2 = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < HOP_KS_MAX; ++ks)
                if (ks < KS) {
                    i0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wih[0][ks], b[ks], i0, 0, 0, 0);
                    i1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wih[1][ks], b[ks], i1, 0, 0, 0);
                    i2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wih[2][ks], b[ks], i2, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = recombine3(i0[r], i1[r], i2[r]) * dqi[r] + bf[r];  // sfsn_spike_proj's epilogue
        }
        // ---- cell (scan_body's epilogue, shared gates)
        pk = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pre_f = __builtin_fmaf(recombine3(a0[r], a1[r], a2[r]), dq[r], z[r]);
            const float pre_g = pre_f + db[r];
            const float f = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre_f * -1.44269504088896341f));
            const float m = __builtin_fmaf(f, c[r] - pre_g, pre_g);
            const float y = __builtin_fmaf(m, alpha[r], beta[r]);
            c[r] = y;
            pk |= (y >= 0.0f) ? (1u << (8 * r)) : 0u;
        }
        if (active && row < R) st_agent(L.spikes + ((size_t)t * R + row) * HP + cc, pk);
        if (active) hop_publish(p.cnt, agent, (unsigned)(t + 1), lane);
    }
    if (active && row < R) {
        *reinterpret_cast<v4f*>(L.c + (size_t)row * H + cc) = c;
        *reinterpret_cast<unsigned*>(hnext + (size_t)row * HP + cc) = pk;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// projection role: one workgroup per row tile, its 4 waves loop over the 16-column tiles of P with the weight fragments in
// LDS.  Full-band: fp32 rows to fb_out (read by the sub-band layer-0 agents).  Sub-band group: rows to LDS, then the deep
// filter of the group's bins for this frame, then (after the last frame) the history shift of those bins.
// LDS: [64 B control][W_p: 3 x PT x KS KB][16 x (P + 4) floats].
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void hop_proj_role(const HopParams& p, const HopStageDev& sd, const HopSeqDev& sq, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int rt = (int)blockIdx.x - sd.wg0;
    const int agent = sd.agent0 + rt * sd.ntpad + wave;
    const int H = sq.H, KS = sq.KS, R = sq.R, HP = KS * 64, P = sq.P, PT = sq.PT;
    const int row = 16 * rt + n, rowc = row < R ? row : R - 1;
    const int hop = p.hop, D = p.D, S = p.S, F = p.F;
    const bool is_fb = sq.df == 0;
    char* wp = smem + 64;
    const int LDP = P + 4;
    float* pbuf = reinterpret_cast<float*>(wp + (size_t)3 * PT * KS * 1024);

    {  // W_p -> LDS, 16 bytes per thread per request, eight requests in flight
        const int n16 = 3 * PT * KS * 64;
        const v4i* src = reinterpret_cast<const v4i*>(sq.w_p);
        v4i* dst = reinterpret_cast<v4i*>(wp);
        for (int i0 = tid; i0 < n16; i0 += HOP_THREADS * 8) {
            v4i v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int idx = i0 + HOP_THREADS * i;
                if (idx > n16 - 1) idx = n16 - 1;
                v[i] = src[idx];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = i0 + HOP_THREADS * i;
                if (idx < n16) dst[idx] = v[i];
            }
        }
        __syncthreads();
    }
    const HopStageDev& ps = p.st[sd.prod];
    const HopLayerDev& last = sq.layer[sq.nl - 1];
    bool ok = true;

    for (int t = 0; t < hop; ++t) {
        ok = hop_wait(p.cnt, ps.agent0 + rt * ps.ntpad, ps.ntile, (unsigned)(t + 1), lane, ok);
        const int8_t* ssrc = last.spikes + (size_t)t * R * HP;
        v4i b[HOP_KS_MAX];
#pragma unroll
        for (int ks = 0; ks < HOP_KS_MAX; ++ks)
            if (ks < KS) b[ks] = ld16_agent(ssrc + (size_t)rowc * HP + ks * 64 + q * 16);
        if (!is_fb && t > 0) __syncthreads();  // the previous frame's deep filter has read pbuf
        for (int pt = wave; pt < PT; pt += HOP_WAVES) {
            v4i a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < HOP_KS_MAX; ++ks)
                if (ks < KS) {
                    const v4i w0 = *reinterpret_cast<const v4i*>(wp + ((((size_t)0 * PT + pt) * KS + ks) * 64 + lane) * 16);
                    const v4i w1 = *reinterpret_cast<const v4i*>(wp + ((((size_t)1 * PT + pt) * KS + ks) * 64 + lane) * 16);
                    const v4i w2 = *reinterpret_cast<const v4i*>(wp + ((((size_t)2 * PT + pt) * KS + ks) * 64 + lane) * 16);
                    a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, b[ks], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1, b[ks], a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w2, b[ks], a2, 0, 0, 0);
                }
            const int col = pt * 16 + q * 4;
            const v4f dq = *reinterpret_cast<const v4f*>(sq.w_p_dq + col);  // padded to PT * 16
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float bias = col + r < P ? sq.b_p[col + r] : 0.0f;
                const float o = recombine3(a0[r], a1[r], a2[r]) * dq[r] + bias;
                if (col + r < P) {
                    if (is_fb) {
                        if (row < R) st_agent(p.fb_out + ((size_t)t * p.B + row) * p.FB + col + r, __float_as_uint(o));
                    } else {
                        pbuf[n * LDP + col + r] = o;
                    }
                }
            }
        }
        if (is_fb) {
            hop_publish(p.cnt, agent, (unsigned)(t + 1), lane);
            continue;
        }
        __syncthreads();
        // ---- deep filter of this row tile's bins for frame t (deepfilter_kernel's expressions and tap order)
        const int fc = sq.fc, df = sq.df, nrow = (R - 16 * rt) < 16 ? (R - 16 * rt) : 16;
        for (int idx = tid; idx < nrow * fc; idx += HOP_THREADS) {
            const int rl = idx / fc, fci = idx - rl * fc;
            const int frow = 16 * rt + rl, b = frow / sq.N, k = frow - b * sq.N;
            const int f = sq.lo + k * fc + fci;
            const float* pr = pbuf + rl * LDP;
            const float* hrow = p.hist + ((size_t)b * F + f) * D * 2;
            const float* irow = p.inp + ((size_t)b * F + f) * hop * 2;
            for (int s = 0; s < S; ++s) {
                float yr = 0.0f, yi = 0.0f;
                for (int d = 0; d < df; ++d) {
                    const int ti = D + t - (df - 1) + d;
                    const float2 xv = ti < D ? *reinterpret_cast<const float2*>(hrow + 2 * ti)
                                             : *reinterpret_cast<const float2*>(irow + 2 * (ti - D));
                    const float cr = pr[((0 * fc + fci) * df + d) * S + s];
                    const float ci = pr[((1 * fc + fci) * df + d) * S + s];
                    yr += xv.x * cr - xv.y * ci;
                    yi += xv.x * ci + xv.y * cr;
                }
                const size_t o = (((size_t)b * S + s) * F + f) * hop + t;
                *reinterpret_cast<float2*>(p.enh + 2 * o) = make_float2(yr, yi);
                if (p.mag) p.mag[o] = fast_abs2(yr, yi);
            }
        }
        // bins no group covers (at least the Nyquist bin) pass through (MODEL:461-470): the first group's first workgroup
        if (sd.seq == 1 && rt == 0)
            for (int idx = tid; idx < p.B * (F - p.fcov); idx += HOP_THREADS) {
                const int b = idx / (F - p.fcov), f = p.fcov + idx - b * (F - p.fcov);
                const float2 xv = *reinterpret_cast<const float2*>(p.inp + (((size_t)b * F + f) * hop + t) * 2);
                for (int s = 0; s < S; ++s) {
                    const size_t o = (((size_t)b * S + s) * F + f) * hop + t;
                    *reinterpret_cast<float2*>(p.enh + 2 * o) = xv;
                    if (p.mag) p.mag[o] = fast_abs2(xv.x, xv.y);
                }
            }
    }
    if (is_fb || D == 0) return;
    // ---- history of my bins: the last D of [old history | new frames]; one thread owns a bin, ascending order reads ahead
    __syncthreads();
    const int fc = sq.fc, nrow = (R - 16 * rt) < 16 ? (R - 16 * rt) : 16;
    for (int idx = tid; idx < nrow * fc; idx += HOP_THREADS) {
        const int rl = idx / fc, fci = idx - rl * fc;
        const int frow = 16 * rt + rl, b = frow / sq.N, k = frow - b * sq.N;
        const int f = sq.lo + k * fc + fci;
        float* hrow = p.hist + ((size_t)b * F + f) * D * 2;
        const float* irow = p.inp + ((size_t)b * F + f) * hop * 2;
        for (int i = 0; i < D; ++i) {
            const int src = i + hop;
            const float2 v = src < D ? *reinterpret_cast<const float2*>(hrow + 2 * src) : *reinterpret_cast<const float2*>(irow + 2 * (src - D));
            *reinterpret_cast<float2*>(hrow + 2 * i) = v;
        }
    }
}

__global__ __launch_bounds__(HOP_THREADS) void stream_hop_kernel(const HopParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int si = 0;
    for (int i = 0; i < p.nstage; ++i)
        if ((int)blockIdx.x >= p.st[i].wg0) si = i;
    const HopStageDev& sd = p.st[si];
    const HopSeqDev& sq = p.seq[sd.seq];
    const unsigned epoch = ld_agent(p.cnt + 2);
    if (sd.layer >= 0) {
        if (sd.layer == 0)
            hop_layer_role<true>(p, sd, sq, smem, epoch);
        else
            hop_layer_role<false>(p, sd, sq, smem, epoch);
    } else {
        hop_proj_role(p, sd, sq, smem);
    }
    hop_exit(p, reinterpret_cast<int*>(smem));
}

// =====================================================================================================================
// host side
// =====================================================================================================================
static int hop_fill_seq(HopSeqDev& d, const sfsn_hop_seq& s, int B, int F, int S, bool is_fb, int FB) {
    if (s.n_layers < 1 || s.n_layers > SFSN_HOP_MAX_LAYERS) return SFSN_EUNSUPPORTED;
    if (s.H <= 0 || s.H % 16 != 0 || s.H > 64 * HOP_KS_MAX) return SFSN_EUNSUPPORTED;
    if (s.P <= 0 || s.P > 256) return SFSN_EUNSUPPORTED;
    const sfsn_feature_group& g = s.feat;
    if (g.n_units <= 0 || g.ctr <= 0 || g.nbr < 0 || g.ctr_fb < 0 || g.nbr_fb < 0 || g.lo < 0) return SFSN_EINVAL;
    const int nf = F - 1;
    const int I1 = g.ctr + 2 * g.nbr, I2 = g.ctr_fb > 0 ? g.ctr_fb + 2 * g.nbr_fb : 0, I = I1 + I2;
    if (I > 16 * HOP_KC_MAX || I > 64 * HOP_NU_MAX) return SFSN_EUNSUPPORTED;
    if (g.lo + g.n_units * g.ctr > nf || g.nbr >= nf || (I2 && (FB <= 0 || g.nbr_fb >= nf))) return SFSN_EINVAL;
    if (g.norm == SFSN_NORM_LAPLACE) return SFSN_EUNSUPPORTED;  // utterance-level statistics: not causal
    if (g.norm == SFSN_NORM_LAYERNORM && (!g.ln_w || !g.ln_b)) return SFSN_EINVAL;
    if (!s.w_p || !s.w_p_dq || !s.b_p) return SFSN_EINVAL;
    memset(&d, 0, sizeof(d));
    d.nl = s.n_layers; d.H = s.H; d.P = s.P; d.R = B * g.n_units; d.KS = (s.H + 63) / 64; d.NT = s.H / 16; d.PT = (s.P + 15) / 16;
    d.I = I; d.I1 = I1; d.KC = (I + 15) / 16;
    d.lo = g.lo; d.N = g.n_units; d.ctr = g.ctr; d.nbr = g.nbr; d.ctr_fb = g.ctr_fb; d.nbr_fb = g.nbr_fb; d.norm = g.norm; d.eps = g.ln_eps;
    d.ln_w = g.ln_w; d.ln_b = g.ln_b;
    d.w_p = s.w_p; d.w_p_dq = s.w_p_dq; d.b_p = s.b_p;
    d.df = is_fb ? 0 : s.df; d.fc = s.fc;
    if (!is_fb) {
        if (s.df < 1 || s.fc != g.ctr || s.P != 2 * s.fc * s.df * S) return SFSN_EINVAL;
    }
    for (int l = 0; l < s.n_layers; ++l) {
        const sfsn_hop_layer& L = s.layer[l];
        if (!L.w_hh || !L.w_hh_dq || !L.bias || !L.bn_alpha || !L.bn_beta || !L.h[0] || !L.h[1] || !L.c || !L.spikes) return SFSN_EINVAL;
        if (l == 0 ? !L.w_ih_f32 : (!L.w_ih || !L.w_ih_dq)) return SFSN_EINVAL;
        HopLayerDev& o = d.layer[l];
        o.w_ih_f32 = L.w_ih_f32; o.w_ih = L.w_ih; o.w_ih_dq = L.w_ih_dq; o.w_hh = L.w_hh; o.w_hh_dq = L.w_hh_dq; o.bias = L.bias;
        o.alpha = L.bn_alpha; o.beta = L.bn_beta; o.h[0] = L.h[0]; o.h[1] = L.h[1]; o.c = L.c; o.spikes = L.spikes;
    }
    return SFSN_OK;
}

static int hop_plan(HopParams& p, size_t& lds, const sfsn_hop_desc* d) {
    if (!d || d->n_groups < 1 || d->n_groups > SFSN_HOP_MAX_GROUPS) return d ? SFSN_EUNSUPPORTED : SFSN_EINVAL;
    if (d->B <= 0 || d->F < 2 || d->S < 1 || d->hop < 1 || d->D < 0 || d->D + d->hop > 32) return SFSN_EUNSUPPORTED;
    if (!d->inp_ri || !d->fb_out || !d->enh_ri || (d->D > 0 && !d->hist_ri)) return SFSN_EINVAL;
    memset(&p, 0, sizeof(p));
    p.B = d->B; p.F = d->F; p.S = d->S; p.hop = d->hop; p.D = d->D; p.FB = d->fb.P; p.fdrc = d->fdrc;
    p.inp = d->inp_ri; p.hist = d->hist_ri; p.fb_out = d->fb_out; p.enh = d->enh_ri; p.mag = d->enh_mag;
    p.nseq = 1 + d->n_groups;
    int rc = hop_fill_seq(p.seq[0], d->fb, d->B, d->F, d->S, true, 0);
    if (rc != SFSN_OK) return rc;
    if (d->fb.feat.ctr_fb != 0) return SFSN_EINVAL;
    int fcov = 0, dmax = 1;
    for (int g = 0; g < d->n_groups; ++g) {
        rc = hop_fill_seq(p.seq[1 + g], d->sb[g], d->B, d->F, d->S, false, p.FB);
        if (rc != SFSN_OK) return rc;
        const int top = p.seq[1 + g].lo + p.seq[1 + g].N * p.seq[1 + g].fc;
        if (top > fcov) fcov = top;
        if (d->sb[g].df > dmax) dmax = d->sb[g].df;
    }
    if (dmax - 1 > d->D || fcov > d->F) return SFSN_EINVAL;
    p.fcov = fcov;
    // stages in dependency order: producers get the lower block indices
    int ns = 0, wg = 0, ag = 0;
    lds = 64;
    auto add_layers = [&](int si_seq, int prod0) {
        const HopSeqDev& q = p.seq[si_seq];
        const int nrt = (q.R + 15) / 16;
        int prev = prod0;
        for (int l = 0; l < q.nl; ++l) {
            HopStageDev& s = p.st[ns];
            s.seq = si_seq; s.layer = l; s.ntile = q.NT; s.ntpad = (q.NT + HOP_WAVES - 1) / HOP_WAVES * HOP_WAVES; s.nrt = nrt;
            s.wg0 = wg; s.nwg = nrt * s.ntpad / HOP_WAVES; s.agent0 = ag; s.prod = prev;
            wg += s.nwg; ag += nrt * s.ntpad;
            prev = ns++;
            if (l == 0) {
                const size_t need = 64 + (size_t)16 * (q.KC * 16 + 4) * sizeof(float);
                if (need > lds) lds = need;
            }
        }
        return prev;
    };
    auto add_proj = [&](int si_seq, int prod) {
        const HopSeqDev& q = p.seq[si_seq];
        const int nrt = (q.R + 15) / 16;
        HopStageDev& s = p.st[ns];
        s.seq = si_seq; s.layer = -1; s.ntile = HOP_WAVES; s.ntpad = HOP_WAVES; s.nrt = nrt;
        s.wg0 = wg; s.nwg = nrt; s.agent0 = ag; s.prod = prod;
        wg += s.nwg; ag += nrt * HOP_WAVES;
        const size_t need = 64 + (size_t)3 * q.PT * q.KS * 1024 + (size_t)16 * (q.P + 4) * sizeof(float);
        if (need > lds) lds = need;
        return ns++;
    };
    const int fb_last = add_layers(0, -1);
    const int fb_proj = add_proj(0, fb_last);
    // sub-band stages layer by layer over all groups (the groups' layer-0 agents all wait for the same full-band projection)
    int prev[SFSN_HOP_MAX_GROUPS];
    int maxl = 0;
    for (int g = 0; g < d->n_groups; ++g) {
        prev[g] = fb_proj;
        if (p.seq[1 + g].nl > maxl) maxl = p.seq[1 + g].nl;
    }
    for (int l = 0; l < maxl; ++l)
        for (int g = 0; g < d->n_groups; ++g) {
            const HopSeqDev& q = p.seq[1 + g];
            if (l >= q.nl) continue;
            const int nrt = (q.R + 15) / 16;
            HopStageDev& s = p.st[ns];
            s.seq = 1 + g; s.layer = l; s.ntile = q.NT; s.ntpad = (q.NT + HOP_WAVES - 1) / HOP_WAVES * HOP_WAVES; s.nrt = nrt;
            s.wg0 = wg; s.nwg = nrt * s.ntpad / HOP_WAVES; s.agent0 = ag; s.prod = prev[g];
            wg += s.nwg; ag += nrt * s.ntpad;
            prev[g] = ns++;
            if (l == 0) {
                const size_t need = 64 + (size_t)16 * (q.KC * 16 + 4) * sizeof(float);
                if (need > lds) lds = need;
            }
        }
    for (int g = 0; g < d->n_groups; ++g) add_proj(1 + g, prev[g]);
    p.nstage = ns; p.nblocks = wg; p.nagents = ag;
    if (lds > 160 * 1024) return SFSN_EUNSUPPORTED;  // all of a CU's LDS
    return SFSN_OK;
}

// every workgroup must be resident at once (peers of a stage wait for each other): one per compute unit at most
static int hop_fits_device(int nblocks) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SFSN_EHIP;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return SFSN_EHIP;
    return nblocks > cus ? SFSN_EUNSUPPORTED : SFSN_OK;
}

extern "C" size_t sfsn_hop_scratch_bytes(const sfsn_hop_desc* desc) {
    HopParams p;
    size_t lds;
    if (hop_plan(p, lds, desc) != SFSN_OK) return 0;
    if (hop_fits_device(p.nblocks) == SFSN_EUNSUPPORTED) return 0;  // (no device at all: the launch reports it)
    return (size_t)(HOP_CNT0 + p.nagents + 16) * sizeof(unsigned);
}

extern "C" int sfsn_stream_hop(const sfsn_hop_desc* desc, void* stream) {
    HopParams local;
    size_t lds;
    const int rc = hop_plan(local, lds, desc);
    if (rc != SFSN_OK) return rc;
    if (!desc->scratch || desc->scratch_bytes < (size_t)(HOP_CNT0 + local.nagents) * sizeof(unsigned)) return SFSN_EINVAL;
    local.cnt = static_cast<unsigned*>(desc->scratch);
    int dev = 0;
    const int fit = hop_fits_device(local.nblocks);
    if (fit != SFSN_OK) return fit;
    if (hipGetDevice(&dev) != hipSuccess) return SFSN_EHIP;
    static int lds_set[64];
    if (lds > 64 * 1024 && dev < 64 && lds_set[dev] < (int)lds) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(stream_hop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
            hipSuccess)
            return SFSN_EHIP;
        lds_set[dev] = 160 * 1024;
    }
    hipLaunchKernelGGL(stream_hop_kernel, dim3(local.nblocks), dim3(HOP_THREADS), lds, static_cast<hipStream_t>(stream), local);
    return hipGetLastError() == hipSuccess ? SFSN_OK : SFSN_EHIP;
}
