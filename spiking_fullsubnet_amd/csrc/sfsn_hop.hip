// sfsn_hop.hip -- the streaming hop: `hop` new frames of B clips through the whole live model in ONE launch (gfx950 only).
//
// BASELINE configs[4] (B = 1 per GPU, hop = 1 frame, state resident between frames).  A one-frame hop through the offline
// kernels is ~15 launches of 4.5-25 us each, every one of them a launch boundary and little else: the work of a frame is a
// few hundred MFMAs.  Here the whole frame is one launch of a few dozen workgroups:
//
//   * an AGENT is one wave: it owns one 16-neuron output tile of one layer for one 16-row tile, keeps that tile's weight
//     fragments (W_hh, and W_ih of its layer) in registers, and its slice of the membrane in registers for the launch;
//     eight agents share a workgroup;
//   * the stages  full-band layer 0 -> ... -> last full-band layer -> sub-band layer 0 (all groups side by side; these
//     workgroups compute the full-band projection they need themselves: 4 tiles, cheaper than one more hop through L2)
//     -> ... -> sub-band projection + deep filter  hand each frame over through L2 as DATA-TAGGED GRANULES: a spike word is
//     four bytes 0/1 whose upper seven bits carry the launch's tag, written by one write-through (sc1) store -- data and
//     "ready" are the same word, so the producer neither drains its store queue nor publishes a flag, and the consumer
//     polls the payload itself.  In a consumer workgroup every word is polled by exactly one lane (wave k takes the k-th
//     64-neuron slice), the masked fragments are parked in LDS, one barrier, and all eight waves read their B operand from
//     LDS: polling traffic does not grow with the number of waves that need the data;
//   * the recurrent product h(t-1).W_hh of every layer is issued BEFORE the workgroup starts to wait for its input: only
//     the input-dependent half of a layer is on the frame's critical path;
//   * every weight fragment a wave needs is requested at launch, i.e. while the stages upstream are still computing.
//
// State between launches lives in device memory (membranes per agent, last spikes double-buffered by launch parity so that a
// fast wave cannot overwrite what a late peer still has to read, deep-filter history shifted by the thread that owns the bin).
// The caller numbers the launches (tag and parity come from that number): no counter has to be read, reset or waited for.
//
// Arithmetic: the expressions of features_kernel / spike_proj_kernel / scan_body / deepfilter_kernel; the real-valued input
// product uses input_proj_kernel's fp32 MFMA operand layout with four accumulators (a quarter of the dependent chain).
// Deadlock freedom: producers have lower block indices than their consumers, workgroups are dispatched in index order and the
// launch is refused unless every workgroup can be resident at once; every spin is bounded all the same (error word).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "sfsn.h"

#include "sfsn_scan_dev.h"
#include "sfsn_feat_dev.h"
#include "sfsn_fft_dev.h"

#define HOP_THREADS 512
#define HOP_WAVES 8
#define HOP_MAX_SEQS (1 + SFSN_HOP_MAX_GROUPS)
#define HOP_MAX_STAGES (HOP_MAX_SEQS * (SFSN_HOP_MAX_LAYERS + 1) + 2)
#define HOP_KS_MAX 5     // 64-wide k steps of the int8 products: H <= 320
#define HOP_KC_MAX 12    // 16-wide k chunks of the fp32 input product: I <= 192
#define HOP_NU_MAX 3     // feature slots per lane: I <= 192
#define HOP_ROWS_PER_WAVE (16 / HOP_WAVES)
#define HOP_PT_PER_WAVE 2  // P <= 256
#define HOP_DF_MAX 6      // deep-filter taps held in registers by the one-frame fast path (D + 1 <= 6: df <= 6)
#define HOP_DF_ITEMS 2    // bins per thread on that path: 16 rows x 64 centre bins = 1024 = 2 x 512 threads
#define HOP_SPIN_LIMIT 2000000u
#define HOP_MAX_BLOCKS 256  // workgroups per launch (one per compute unit at most)

struct HopLayerDev {
    const float* w_ih_f32;  // (fragment order)
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
    float* cum[2];  // SFSN_NORM_CUMLAPLACE: running sum of every row, double-buffered by launch parity
    int nl, H, P, R, KS, NT, PT, I, I1, KC;
    int lo, N, ctr, nbr, ctr_fb, nbr_fb, norm, df, fc;
    float eps;
};
struct HopStageDev {
    int seq, layer;  // layer = -1: the projection + deep filter of a sub-band group; -2 / -3: waveform mode's STFT / inverse STFT
    int wg0, nwg;    // workgroups [wg0, wg0 + nwg)
    int ntile, ntpad, nrt;
};
struct HopParams {
    HopSeqDev seq[HOP_MAX_SEQS];
    HopStageDev st[HOP_MAX_STAGES];
    unsigned stage_of_block[HOP_MAX_BLOCKS / 4];  // one byte per workgroup: a single scalar load finds the role
    int nseq, nstage, nblocks;
    int B, F, S, hop, D, FB, fcov;
    int G;  // 1: the forget and the cell gate share their weights (one product serves both); 2: separate weights, 2H rows per image
    float fdrc;
    unsigned launch;  // launches made on this state since it was zeroed: tag and state parity
    int frames_before;  // frames the state has seen since it was zeroed (SFSN_NORM_CUMLAPLACE's denominator)
    const float* inp;
    // waveform mode (hop == 1): the new samples, the last 512 input samples, the output's overlap-add accumulator, the output,
    // the window, the noisy / enhanced frame as {re, tag, im, tag} granules, the index of the frame this launch computes
    const float* wave_in;
    float* wave_state;
    float* ola_state;
    float* wave_out;
    const float* window;
    float* spec_g;
    float* enh_g;
    int frame_index;
    unsigned* done;  // optional (host-visible): one word per (clip, speaker), set to launch + 1 when its samples are out
    float* hist;
    float* enh;
    float* mag;
    unsigned* cnt;    // [0] error word
    unsigned long long* dbg;  // -DSFSN_HOP_STAMPS builds + SFSN_HOP_DEBUG: 8 time stamps (100 MHz) per wave
};

// what changes from hop to hop (a launch takes them from its kernel arguments; the resident kernel counts them up itself)
struct HopStep {
    unsigned launch;
    int frame_index, frames_before;
};

#ifdef SFSN_HOP_STAMPS
// (every lane stores the same value to the same word: no branch, so the compiler's wait-count bookkeeping is not disturbed;
// a stamps build always has a valid dbg pointer)
#define HOP_STAMP(i) p.dbg[((size_t)blockIdx.x * HOP_WAVES + wave) * 8 + (i)] = wall_clock64()
#else
#define HOP_STAMP(i) \
    do {             \
    } while (0)
#endif

// ---- coherent accesses (agent scope: global_load / global_store ... sc1) ------------------------------------------------
__device__ __forceinline__ unsigned ld_agent(const void* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(void* p, unsigned v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ unsigned long long ld64_agent(const void* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st64_agent(void* p, unsigned long long v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- hand-off: data-tagged granules (MI355X_MICROARCH.md, price list row handoff-1to1) ------------------------------------
__device__ __forceinline__ unsigned hop_tag(unsigned launch) { return launch % 127u + 1u; }

// All waves of the workgroup call this (one barrier inside).  `blk` = the [R][HP] spike block of one frame, written by other
// workgroups of this launch.  Wave k < KS polls the k-th 64-column slice of row tile `rt16` until every word carries `tagw`
// (= tag * 0x02020202; columns >= H are padding, never written, read as zero), parks the masked fragment in LDS (`hb`,
// KS KB); after the barrier every wave reads all KS fragments.  Returns false when the bounded spin expired (error word
// set) -- the wave then stops polling for good (garbage out, the host raises) but keeps executing its barriers.
__device__ __forceinline__ bool hop_gather(const int8_t* blk, int rt16, int R, int KS, int H, unsigned tagw, char* hb, v4i (&b)[HOP_KS_MAX],
                                           bool ok, unsigned* err, int wave, int lane) {
    const int n = lane & 15, q = lane >> 4;
    if (wave < KS) {
        const int row = 16 * rt16 + n, rowc = row < R ? row : R - 1;
        v4i v = {0, 0, 0, 0};
        if (wave * 64 + q * 16 < H) {
            const unsigned* u = reinterpret_cast<const unsigned*>(blk + ((size_t)rowc * KS + wave) * 64 + q * 16);
            for (unsigned spins = 0;; ++spins) {
                bool bad = false;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned w = ld_agent(u + i);
                    bad |= (w & 0xfefefefeu) != tagw;
                    v[i] = (int)(w & 0x01010101u);
                }
                if (!ok || __ballot(bad) == 0) break;
                if (spins > HOP_SPIN_LIMIT) {
                    st_agent(err, 1u);
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        *reinterpret_cast<v4i*>(hb + (wave * 64 + lane) * 16) = v;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < HOP_KS_MAX; ++ks) {
        b[ks] = v4i{0, 0, 0, 0};
        if (ks < KS) b[ks] = *reinterpret_cast<const v4i*>(hb + (ks * 64 + lane) * 16);
    }
    return ok;
}

// A complex value as two 8-byte {value, tag} granules (waveform mode: the noisy frame comes from the STFT workgroups of this
// launch, the enhanced frame goes to the inverse-STFT workgroups).  Every lane polls its own granules.
__device__ __forceinline__ void hop_put_cplx(float* g, float2 v, unsigned tagw) {
    st64_agent(g, ((unsigned long long)tagw << 32) | __float_as_uint(v.x));
    st64_agent(g + 2, ((unsigned long long)tagw << 32) | __float_as_uint(v.y));
}
__device__ __forceinline__ float2 hop_take_cplx(const float* g, unsigned tagw, bool& ok, unsigned* err) {
    for (unsigned spins = 0;; ++spins) {
        const unsigned long long a = ld64_agent(g), c = ld64_agent(g + 2);
        if (((unsigned)(a >> 32) == tagw && (unsigned)(c >> 32) == tagw) || !ok)
            return make_float2(__uint_as_float((unsigned)a), __uint_as_float((unsigned)c));
        if (spins > HOP_SPIN_LIMIT) {
            st_agent(err, 1u);
            ok = false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// bin f of clip b's new frame t: the caller's spectrum, or (waveform mode) the STFT workgroups' granules
__device__ __forceinline__ float2 hop_in_bin(const HopParams& p, int b, int f, int t, int hop, unsigned tagw, bool& ok) {
    if (!p.spec_g) return *reinterpret_cast<const float2*>(p.inp + (((size_t)b * p.F + f) * hop + t) * 2);
    return hop_take_cplx(p.spec_g + (((size_t)b * p.F + f) * hop + t) * 4, tagw, ok, p.cnt);
}

// ---------------------------------------------------------------------------------------------------------------------
// layer role: 8 agents per workgroup (tiles 8w .. 8w+7 of one row tile).
// LDS: [64 B][hbA: KS_MAX KB (own layer, frame t-1)][hbB: KS_MAX KB (input spikes, frame t)][fbl: 16 x FB floats][xrow: 16 x KPX]
//      [fbw: the full-band projection's weight fragments, 3 x PT x KS KB (gated layer 0 only)].
// ---------------------------------------------------------------------------------------------------------------------
// ONE: hop == 1 (the configuration that matters): no frame loop, so nothing stays live across it.
// G = 2 (separate forget / cell gate weights, baseline_xl): a wave takes the two gates of its tile ONE AFTER THE OTHER through the same
// weight registers -- both gates' fragments at once would be 240 registers of weights beside everything else.  The second gate's
// recurrent weights are requested when the first gate's matrix instructions have been issued, its input weights when the
// recurrent half is done: both arrive while the wave waits for the stage upstream.  Same products, same two roundings per gate and
// the same cell as scan_body's G = 2 epilogue: bit-identical to the offline kernels.
template <bool L0, bool ONE, int G>
__device__ __forceinline__ void hop_layer_role(const HopParams& p, const HopStep& hs, const HopStageDev& sd, const HopSeqDev& sq, char* smem) {
    const int l = sd.layer;
    const HopLayerDev& L = sq.layer[l];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int wgl = (int)blockIdx.x - sd.wg0, wpr = sd.ntpad / HOP_WAVES;  // workgroups per row tile
    const int rt = wgl / wpr, tile_raw = (wgl - rt * wpr) * HOP_WAVES + wave;
    const bool active = tile_raw < sd.ntile;
    const int tile = active ? tile_raw : 0;
    const int H = sq.H, KS = sq.KS, NT = sq.NT, R = sq.R, HP = KS * 64, I = sq.I, KC = sq.KC;
    const int row = 16 * rt + n, rowc = row < R ? row : R - 1;
    const int cc = 16 * tile + 4 * q;
    const int hop = ONE ? 1 : p.hop;
    const unsigned tagw = hop_tag(hs.launch) * 0x02020202u;
    char* hbA = smem + 64;
    char* hbB = hbA + HOP_KS_MAX * 1024;
    float* fbl = reinterpret_cast<float*>(hbB + HOP_KS_MAX * 1024);
    float* xrow = fbl + 16 * p.FB;
    const int KPX = KC * 16 + 4;
    char* fbw = reinterpret_cast<char*>(xrow + 16 * KPX);
    const bool gated = L0 && sd.seq > 0;  // layer 0 of a sub-band group: its rows need the full-band projection
    const HopSeqDev& fbq = p.seq[0];

    if (gated) {  // 16 bytes per thread per request, four requests in flight (visible after the first gather's barrier)
        const int n16 = 3 * fbq.PT * fbq.KS * 64;
        const v4i* src = reinterpret_cast<const v4i*>(fbq.w_p);
        v4i* dst = reinterpret_cast<v4i*>(fbw);
        for (int i0 = tid; i0 < n16; i0 += HOP_THREADS * 4) {
            v4i v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int idx = i0 + HOP_THREADS * i;
                if (idx > n16 - 1) idx = n16 - 1;
                v[i] = src[idx];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = i0 + HOP_THREADS * i;
                if (idx < n16) dst[idx] = v[i];
            }
        }
    }
    // ---- everything this wave will need is requested now (the stages upstream are still at work); the recurrent half's
    // operands first
    const int8_t* hprev = L.h[hs.launch & 1u];
    int8_t* hnext = L.h[(hs.launch + 1u) & 1u];
    v4i h0[HOP_KS_MAX];  // h of the last frame of the previous launch (plain bytes 0/1)
    v4i Whh[3][HOP_KS_MAX], Wih[3][HOP_KS_MAX];
#pragma unroll
    for (int ks = 0; ks < HOP_KS_MAX; ++ks) {
        h0[ks] = v4i{0, 0, 0, 0};
        if (ks < KS) h0[ks] = *reinterpret_cast<const v4i*>(hprev + (size_t)rowc * HP + ks * 64 + q * 16);
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int ks = 0; ks < HOP_KS_MAX; ++ks) {
            Whh[d][ks] = v4i{0, 0, 0, 0};
            if (ks < KS) Whh[d][ks] = *reinterpret_cast<const v4i*>(L.w_hh + ((((size_t)d * (G * NT) + tile) * KS + ks) * 64 + lane) * 16);
        }
    v4f c = *reinterpret_cast<const v4f*>(L.c + (size_t)rowc * H + cc);
    const v4f dq = *reinterpret_cast<const v4f*>(L.w_hh_dq + cc);
    const v4f bf = *reinterpret_cast<const v4f*>(L.bias + cc);
    const v4f bg = *reinterpret_cast<const v4f*>(L.bias + H + cc);
    const v4f alpha = *reinterpret_cast<const v4f*>(L.alpha + cc);
    const v4f beta = *reinterpret_cast<const v4f*>(L.beta + cc);
    v4f dqi = {0.0f, 0.0f, 0.0f, 0.0f};
    if (!L0) dqi = *reinterpret_cast<const v4f*>(L.w_ih_dq + cc);
    // layer >= 1: W_ih of my tile.  Gated layer 0: the full-band projection's fragments go to LDS (waves < PT use one tile each).
    const bool fbp = gated && wave < fbq.PT;
    const int fcol = (fbp ? wave : 0) * 16 + q * 4;
    v4f fdq = {0.0f, 0.0f, 0.0f, 0.0f}, fbias = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int ks = 0; ks < HOP_KS_MAX; ++ks) {
            Wih[d][ks] = v4i{0, 0, 0, 0};
            if (!L0 && ks < KS) Wih[d][ks] = *reinterpret_cast<const v4i*>(L.w_ih + ((((size_t)d * (G * NT) + tile) * KS + ks) * 64 + lane) * 16);
        }
    if (L0 && fbp) {
        fdq = *reinterpret_cast<const v4f*>(fbq.w_p_dq + fcol);
#pragma unroll
        for (int r = 0; r < 4; ++r) fbias[r] = fcol + r < fbq.P ? fbq.b_p[fcol + r] : 0.0f;
    }
    // layer 0: W_ih of my tile in MFMA fragment order [tile][chunk][lane][4] (k = 16 chunk + 4 q + e of row 16 tile + n):
    // one coalesced 16-byte request per chunk
    v4f W0[HOP_KC_MAX];
#pragma unroll
    for (int ch = 0; ch < HOP_KC_MAX; ++ch) {
        W0[ch] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
        if (L0 && ch < KC) W0[ch] = *reinterpret_cast<const v4f*>(L.w_ih_f32 + ((((size_t)tile * KC + ch) * 64 + lane) * 4));
    }
    v4f db;
#pragma unroll
    for (int r = 0; r < 4; ++r) db[r] = bg[r] - bf[r];
    v4f dqg = {0.0f, 0.0f, 0.0f, 0.0f}, dqig = {0.0f, 0.0f, 0.0f, 0.0f};  // G = 2: the cell gate's dequantisation scales
    if constexpr (G == 2) {
        dqg = *reinterpret_cast<const v4f*>(L.w_hh_dq + H + cc);
        if (!L0) dqig = *reinterpret_cast<const v4f*>(L.w_ih_dq + H + cc);
    }
    float lw[HOP_NU_MAX], lb[HOP_NU_MAX];
#pragma unroll
    for (int u = 0; u < HOP_NU_MAX; ++u) {
        const int j = lane + 64 * u;
        const bool in = L0 && j < I && sq.norm == SFSN_NORM_LAYERNORM;
        lw[u] = in ? sq.ln_w[j] : 0.0f;
        lb[u] = in ? sq.ln_b[j] : 0.0f;
    }
    HOP_STAMP(1);

    const int nf = p.F - 1;
    bool ok = true;
    unsigned pk = 0;
    // cumulative_laplace_norm: the running sums of my rows (every workgroup of the row tile computes the same sequence; the first
    // one stores it for the next launch, into the other half of the double buffer)
    float cumr[HOP_ROWS_PER_WAVE];
#pragma unroll
    for (int ri = 0; ri < HOP_ROWS_PER_WAVE; ++ri) {
        const int frow = 16 * rt + wave + HOP_WAVES * ri;
        // (agent-scope load: in the resident form the previous hop's sum was written by ANOTHER workgroup of the same launch --
        //  no launch boundary has made it visible to this CU's L1 / this XCD's L2)
        cumr[ri] = (L0 && sq.norm == SFSN_NORM_CUMLAPLACE && frow < R) ? __uint_as_float(ld_agent(&sq.cum[hs.launch & 1u][frow])) : 0.0f;
    }

    for (int t = 0; t < hop; ++t) {
        // ---- recurrent half: needs frame t-1 of my own layer only
        v4i a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
        v4i g0 = {0, 0, 0, 0}, g1 = {0, 0, 0, 0}, g2 = {0, 0, 0, 0};  // G = 2: the cell gate's recurrent sums
        {
            v4i b[HOP_KS_MAX];
            if (t == 0) {
#pragma unroll
                for (int ks = 0; ks < HOP_KS_MAX; ++ks) b[ks] = h0[ks];
            } else {
                ok = hop_gather(L.spikes + (size_t)(t - 1) * R * HP, rt, R, KS, H, tagw, hbA, b, ok, p.cnt, wave, lane);
            }
#pragma unroll
            for (int ks = 0; ks < HOP_KS_MAX; ++ks)
                if (ks < KS) {
                    a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Whh[0][ks], b[ks], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Whh[1][ks], b[ks], a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Whh[2][ks], b[ks], a2, 0, 0, 0);
                }
            if constexpr (G == 2) {  // the cell gate's recurrent product through the same registers (tile NT + tile of the image)
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int ks = 0; ks < HOP_KS_MAX; ++ks)
                        if (ks < KS) Whh[d][ks] = *reinterpret_cast<const v4i*>(L.w_hh + ((((size_t)d * (G * NT) + NT + tile) * KS + ks) * 64 + lane) * 16);
#pragma unroll
                for (int ks = 0; ks < HOP_KS_MAX; ++ks)
                    if (ks < KS) {
                        g0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Whh[0][ks], b[ks], g0, 0, 0, 0);
                        g1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Whh[1][ks], b[ks], g1, 0, 0, 0);
                        g2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Whh[2][ks], b[ks], g2, 0, 0, 0);
                    }
                if (!ONE) {  // (more frames follow in this launch: the forget gate's weights back for the next one)
#pragma unroll
                    for (int d = 0; d < 3; ++d)
#pragma unroll
                        for (int ks = 0; ks < HOP_KS_MAX; ++ks)
                            if (ks < KS) Whh[d][ks] = *reinterpret_cast<const v4i*>(L.w_hh + ((((size_t)d * (G * NT) + tile) * KS + ks) * 64 + lane) * 16);
                }
            }
        }
        if (t == 0) HOP_STAMP(2);
        // ---- input half
        v4f z, zg = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (L0) {
            // features of my rows (wave w: rows w and w + 8): the magnitude part needs the new frame only -- requested before
            // the wait for the full-band model
            float v[HOP_ROWS_PER_WAVE][HOP_NU_MAX];
#pragma unroll
            for (int ri = 0; ri < HOP_ROWS_PER_WAVE; ++ri) {
                const int frow = 16 * rt + wave + HOP_WAVES * ri;
                const int frc = frow < R ? frow : R - 1;
                const int b_ = frc / sq.N, k = frc - b_ * sq.N;
#pragma unroll
                for (int u = 0; u < HOP_NU_MAX; ++u) {
                    const int j = lane + 64 * u;
                    v[ri][u] = 0.0f;
                    if (j < sq.I1) {
                        const int bin = reflect_bin(sq.lo + k * sq.ctr - sq.nbr + j, nf);
                        const float2 xc = hop_in_bin(p, b_, bin, t, hop, tagw, ok);
                        v[ri][u] = compress_mag(xc.x, xc.y, p.fdrc);
                    }
                }
            }
            const int b0 = (16 * rt) / sq.N;  // first clip of this row tile
            if (gated) {
                // the full-band projection of my clips, computed here from the last full-band layer's spikes (MODEL:118 for
                // the clips this row tile covers): gather, 4 waves x one 16-column tile, park in LDS
                int b1 = (16 * rt + 15) / sq.N;
                if (b1 > p.B - 1) b1 = p.B - 1;
                const HopLayerDev& fl = fbq.layer[fbq.nl - 1];
                for (int f = b0 / 16; f <= b1 / 16; ++f) {
                    v4i b[HOP_KS_MAX];
                    ok = hop_gather(fl.spikes + (size_t)t * fbq.R * fbq.KS * 64, f, fbq.R, fbq.KS, fbq.H, tagw, hbB, b, ok, p.cnt, wave, lane);
                    if (fbp) {
                        v4i i0 = {0, 0, 0, 0}, i1 = {0, 0, 0, 0}, i2 = {0, 0, 0, 0};
#pragma unroll
                        for (int ks = 0; ks < HOP_KS_MAX; ++ks)
                            if (ks < fbq.KS) {
                                const v4i w0 = *reinterpret_cast<const v4i*>(fbw + ((((size_t)0 * fbq.PT + wave) * fbq.KS + ks) * 64 + lane) * 16);
                                const v4i w1 = *reinterpret_cast<const v4i*>(fbw + ((((size_t)1 * fbq.PT + wave) * fbq.KS + ks) * 64 + lane) * 16);
                                const v4i w2 = *reinterpret_cast<const v4i*>(fbw + ((((size_t)2 * fbq.PT + wave) * fbq.KS + ks) * 64 + lane) * 16);
                                i0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, b[ks], i0, 0, 0, 0);
                                i1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1, b[ks], i1, 0, 0, 0);
                                i2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w2, b[ks], i2, 0, 0, 0);
                            }
                        const int clip = 16 * f + n;  // full-band row = clip
                        if (clip >= b0 && clip <= b1) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (fcol + r < fbq.P) fbl[(clip - b0) * p.FB + fcol + r] = recombine3(i0[r], i1[r], i2[r]) * fdq[r] + fbias[r];
                        }
                    }
                    __syncthreads();  // fbl rows of this full-band tile are in place / hbB may be reused
                }
#pragma unroll
                for (int ri = 0; ri < HOP_ROWS_PER_WAVE; ++ri) {
                    const int frow = 16 * rt + wave + HOP_WAVES * ri;
                    const int frc = frow < R ? frow : R - 1;
                    const int b_ = frc / sq.N, k = frc - b_ * sq.N;
#pragma unroll
                    for (int u = 0; u < HOP_NU_MAX; ++u) {
                        const int j = lane + 64 * u;
                        if (j >= sq.I1 && j < I) {
                            const int col = reflect_bin(sq.lo + k * sq.ctr_fb - sq.nbr_fb + (j - sq.I1), nf) % p.FB;
                            v[ri][u] = fbl[(b_ - b0) * p.FB + col];
                        }
                    }
                }
            }
            if (t == 0) HOP_STAMP(3);
#pragma unroll
            for (int ri = 0; ri < HOP_ROWS_PER_WAVE; ++ri) {
                const int rl = wave + HOP_WAVES * ri;
                if (16 * rt + rl >= R) break;
                float sum = 0.0f;
#pragma unroll
                for (int u = 0; u < HOP_NU_MAX; ++u) sum += v[ri][u];
                float y[HOP_NU_MAX];
                if (sq.norm == SFSN_NORM_LAYERNORM) {
                    const float inv_I = 1.0f / (float)I;
                    const float mean = wave_sum(sum) * inv_I;
                    float ss = 0.0f;
#pragma unroll
                    for (int u = 0; u < HOP_NU_MAX; ++u) {
                        const float d = v[ri][u] - mean;
                        if (lane + 64 * u < I) ss += d * d;
                    }
                    const float rstd = __builtin_amdgcn_rsqf(wave_sum(ss) * inv_I + sq.eps);
#pragma unroll
                    for (int u = 0; u < HOP_NU_MAX; ++u) y[u] = ((v[ri][u] - mean) * rstd) * lw[u] + lb[u];
                } else if (sq.norm == SFSN_NORM_CUMLAPLACE) {
                    // cumlap_rowsum_kernel / cumlap_scan_kernel's arithmetic: fp32 row sum (same lanes, same reduction), fp32
                    // running sum, mean over everything the row has seen, x / (mean + eps)
                    cumr[ri] += wave_sum(sum);
                    const float den = cumr[ri] / (float)((double)I * (hs.frames_before + t + 1)) + 2.220446049250313e-16f;
#pragma unroll
                    for (int u = 0; u < HOP_NU_MAX; ++u) y[u] = v[ri][u] / den;
                } else {
#pragma unroll
                    for (int u = 0; u < HOP_NU_MAX; ++u) y[u] = v[ri][u];
                }
#pragma unroll
                for (int u = 0; u < HOP_NU_MAX; ++u) {  // the k padding of the input product must read zeros
                    const int j = lane + 64 * u;
                    if (j < KC * 16) xrow[rl * KPX + j] = j < I ? y[u] : 0.0f;
                }
            }
            __syncthreads();
            if (t == 0) HOP_STAMP(4);
            // fp32 MFMA products in input_proj_kernel's operand layout (lane (n, q) holds k = 16c + 4q + e of row n); four
            // accumulators by chunk so that the dependent chain is a quarter as long
            v4f acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
            const int xr = (16 * rt + n < R) ? n : (R - 1 - 16 * rt);
#pragma unroll
            for (int cch = 0; cch < HOP_KC_MAX; ++cch)
                if (cch < KC) {
                    const v4f bx = *reinterpret_cast<const v4f*>(xrow + xr * KPX + cch * 16 + q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[cch & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(W0[cch][e], bx[e], acc[cch & 3], 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = ((acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r])) + bf[r];
            if constexpr (G == 2) {  // the cell gate's rows of W_ih (fragment tile NT + tile), same operand, same association
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int cch = 0; cch < HOP_KC_MAX; ++cch)
                    if (cch < KC) {
                        const v4f wg = *reinterpret_cast<const v4f*>(L.w_ih_f32 + ((((size_t)(NT + tile) * KC + cch) * 64 + lane) * 4));
                        const v4f bx = *reinterpret_cast<const v4f*>(xrow + xr * KPX + cch * 16 + q * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[cch & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wg[e], bx[e], acc[cch & 3], 0, 0, 0);
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) zg[r] = ((acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r])) + bg[r];
            }
        } else {
            v4i b[HOP_KS_MAX];
            ok = hop_gather(sq.layer[l - 1].spikes + (size_t)t * R * HP, rt, R, KS, H, tagw, hbB, b, ok, p.cnt, wave, lane);
            if (t == 0) HOP_STAMP(4);
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
            if constexpr (G == 2) {
                v4i j0 = {0, 0, 0, 0}, j1 = {0, 0, 0, 0}, j2 = {0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < HOP_KS_MAX; ++ks)
                    if (ks < KS) {
                        const v4i w0 = *reinterpret_cast<const v4i*>(L.w_ih + ((((size_t)0 * (G * NT) + NT + tile) * KS + ks) * 64 + lane) * 16);
                        const v4i w1 = *reinterpret_cast<const v4i*>(L.w_ih + ((((size_t)1 * (G * NT) + NT + tile) * KS + ks) * 64 + lane) * 16);
                        const v4i w2 = *reinterpret_cast<const v4i*>(L.w_ih + ((((size_t)2 * (G * NT) + NT + tile) * KS + ks) * 64 + lane) * 16);
                        j0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, b[ks], j0, 0, 0, 0);
                        j1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1, b[ks], j1, 0, 0, 0);
                        j2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(w2, b[ks], j2, 0, 0, 0);
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) zg[r] = recombine3(j0[r], j1[r], j2[r]) * dqig[r] + bg[r];
            }
        }
        // ---- cell (scan_body's epilogue; G = 2: the cell gate has its own products)
        pk = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pre_f = __builtin_fmaf(recombine3(a0[r], a1[r], a2[r]), dq[r], z[r]);
            const float pre_g = G == 2 ? __builtin_fmaf(recombine3(g0[r], g1[r], g2[r]), dqg[r], zg[r]) : pre_f + db[r];
            const float f = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre_f * -1.44269504088896341f));
            const float m = __builtin_fmaf(f, c[r] - pre_g, pre_g);
            const float y = __builtin_fmaf(m, alpha[r], beta[r]);
            c[r] = y;
            pk |= (y >= 0.0f) ? (1u << (8 * r)) : 0u;
        }
        if (active && row < R) st_agent(L.spikes + ((size_t)t * R + row) * HP + cc, pk | tagw);  // data + tag: published
        if (t == 0) HOP_STAMP(5);
    }
    if (active && row < R) {
        *reinterpret_cast<v4f*>(L.c + (size_t)row * H + cc) = c;
        st_agent(hnext + (size_t)row * HP + cc, pk);  // (write-through: the resident form has no launch boundary to write L2 back)
    }
    if (L0 && sq.norm == SFSN_NORM_CUMLAPLACE && wgl - rt * wpr == 0 && lane == 0) {
#pragma unroll
        for (int ri = 0; ri < HOP_ROWS_PER_WAVE; ++ri) {
            const int frow = 16 * rt + wave + HOP_WAVES * ri;
            if (frow < R) st_agent(&sq.cum[(hs.launch + 1u) & 1u][frow], __float_as_uint(cumr[ri]));  // (write-through, as the last spikes are)
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// projection + deep filter of a sub-band group: one workgroup per row tile; wave w owns the 16-column tiles w and w + 8 of
// P (weight fragments in registers).  Rows to LDS, then the deep filter of the tile's bins for this frame, then (after the
// last frame) the history shift of those bins.  LDS: [64 B][hb: KS_MAX KB][16 x (P + 4) floats].
// ---------------------------------------------------------------------------------------------------------------------
template <bool ONE>
__device__ __forceinline__ void hop_proj_role(const HopParams& p, const HopStep& hs, const HopStageDev& sd, const HopSeqDev& sq, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int rt = (int)blockIdx.x - sd.wg0;
    const int H = sq.H, KS = sq.KS, R = sq.R, HP = KS * 64, P = sq.P, PT = sq.PT;
    const int hop = ONE ? 1 : p.hop, D = p.D, S = p.S, F = p.F;
    const unsigned tagw = hop_tag(hs.launch) * 0x02020202u;
    const int LDP = P + 4;
    char* hb = smem + 64;
    float* pbuf = reinterpret_cast<float*>(hb + HOP_KS_MAX * 1024);

    v4i Wp[HOP_PT_PER_WAVE][3][HOP_KS_MAX];
    v4f dqv[HOP_PT_PER_WAVE], bv[HOP_PT_PER_WAVE];
#pragma unroll
    for (int i = 0; i < HOP_PT_PER_WAVE; ++i) {
        const int pt = wave + HOP_WAVES * i;
        const bool have = pt < PT;
        const int col = (have ? pt : 0) * 16 + q * 4;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int ks = 0; ks < HOP_KS_MAX; ++ks) {
                Wp[i][d][ks] = v4i{0, 0, 0, 0};
                if (have && ks < KS) Wp[i][d][ks] = *reinterpret_cast<const v4i*>(sq.w_p + ((((size_t)d * PT + pt) * KS + ks) * 64 + lane) * 16);
            }
        dqv[i] = *reinterpret_cast<const v4f*>(sq.w_p_dq + col);  // padded to PT * 16
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[i][r] = col + r < P ? sq.b_p[col + r] : 0.0f;
    }
    HOP_STAMP(1);
    const HopLayerDev& last = sq.layer[sq.nl - 1];
    bool ok = true;
    const int fc = sq.fc, df = sq.df, nrow = (R - 16 * rt) < 16 ? (R - 16 * rt) : 16;

    // ---- what does not depend on the network happens before the wait: the pass-through bins, and (hop == 1, the usual shape:
    // one bin per thread, df <= HOP_DF_MAX) this thread's deep-filter taps into registers
    if (sd.seq == 1 && rt == 0)  // bins no group covers (at least the Nyquist bin) pass through (MODEL:461-470)
        for (int idx = tid; idx < p.B * (F - p.fcov) * hop; idx += HOP_THREADS) {
            const int t = idx % hop, r_ = idx / hop;
            const int b_ = r_ / (F - p.fcov), f = p.fcov + r_ - b_ * (F - p.fcov);
            const float2 xv = hop_in_bin(p, b_, f, t, hop, tagw, ok);
            for (int s = 0; s < S; ++s) {
                const size_t o = (((size_t)b_ * S + s) * F + f) * hop + t;
                *reinterpret_cast<float2*>(p.enh + 2 * o) = xv;
                if (p.mag) p.mag[o] = fast_abs2(xv.x, xv.y);
                if (p.enh_g) hop_put_cplx(p.enh_g + 4 * o, xv, tagw);
            }
        }
    const bool fast = ONE && nrow * fc <= HOP_THREADS * HOP_DF_ITEMS && D <= HOP_DF_MAX - 1;
    float2 tap[HOP_DF_ITEMS][HOP_DF_MAX];  // [old history (D) | new frame] of this thread's bins tid, tid + 512
#pragma unroll
    for (int it = 0; it < HOP_DF_ITEMS; ++it) {
        const int idx = tid + HOP_THREADS * it;
        if (fast && idx < nrow * fc) {
            const int rl = idx / fc, fci = idx - rl * fc;
            const int frow = 16 * rt + rl, b_ = frow / sq.N, k = frow - b_ * sq.N;
            const int f = sq.lo + k * fc + fci;
            const float* hrow = p.hist + ((size_t)b_ * F + f) * D * 2;
#pragma unroll
            for (int i = 0; i < HOP_DF_MAX; ++i) {
                tap[it][i] = make_float2(0.0f, 0.0f);
                if (i < D) tap[it][i] = *reinterpret_cast<const float2*>(hrow + 2 * i);
                if (i == D) tap[it][i] = hop_in_bin(p, b_, f, 0, 1, tagw, ok);
            }
        }
    }

    for (int t = 0; t < hop; ++t) {
        v4i b[HOP_KS_MAX];
        // (for t > 0 the barrier inside also orders the previous frame's deep-filter reads of pbuf before the writes below)
        ok = hop_gather(last.spikes + (size_t)t * R * HP, rt, R, KS, H, tagw, hb, b, ok, p.cnt, wave, lane);
        if (t == 0) HOP_STAMP(4);
#pragma unroll
        for (int i = 0; i < HOP_PT_PER_WAVE; ++i) {
            const int pt = wave + HOP_WAVES * i;
            if (pt >= PT) break;
            v4i a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < HOP_KS_MAX; ++ks)
                if (ks < KS) {
                    a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wp[i][0][ks], b[ks], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wp[i][1][ks], b[ks], a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wp[i][2][ks], b[ks], a2, 0, 0, 0);
                }
            const int col = pt * 16 + q * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (col + r < P) pbuf[n * LDP + col + r] = recombine3(a0[r], a1[r], a2[r]) * dqv[i][r] + bv[i][r];
        }
        if (t == 0) HOP_STAMP(5);
        __syncthreads();
        // ---- deep filter of this row tile's bins for frame t (deepfilter_kernel's expressions and tap order)
        if (fast) {
#pragma unroll
            for (int it = 0; it < HOP_DF_ITEMS; ++it) {
                const int idx = tid + HOP_THREADS * it;
                if (idx >= nrow * fc) break;
                const int rl = idx / fc, fci = idx - rl * fc;
                const int frow = 16 * rt + rl, b_ = frow / sq.N, k = frow - b_ * sq.N;
                const int f = sq.lo + k * fc + fci;
                const float* pr = pbuf + rl * LDP;
                for (int s = 0; s < S; ++s) {
                    float yr = 0.0f, yi = 0.0f;
#pragma unroll
                    for (int i = 0; i < HOP_DF_MAX; ++i) {  // tap i of [history | frame] is filter tap d = i - (D - (df - 1))
                        const int d = i - (D - (df - 1));
                        if (d >= 0 && i <= D) {
                            const float cr = pr[((0 * fc + fci) * df + d) * S + s];
                            const float ci = pr[((1 * fc + fci) * df + d) * S + s];
                            yr += tap[it][i].x * cr - tap[it][i].y * ci;
                            yi += tap[it][i].x * ci + tap[it][i].y * cr;
                        }
                    }
                    const size_t o = ((size_t)b_ * S + s) * F + f;
                    *reinterpret_cast<float2*>(p.enh + 2 * o) = make_float2(yr, yi);
                    if (p.mag) p.mag[o] = fast_abs2(yr, yi);
                    if (p.enh_g) hop_put_cplx(p.enh_g + 4 * o, make_float2(yr, yi), tagw);
                }
                float* hrow = p.hist + ((size_t)b_ * F + f) * D * 2;  // history: drop the oldest frame, append the new one
#pragma unroll
                for (int i = 0; i < HOP_DF_MAX - 1; ++i)
                    if (i < D) *reinterpret_cast<float2*>(hrow + 2 * i) = tap[it][i + 1];
            }
            if (t == 0) HOP_STAMP(6);
            continue;
        }
        for (int idx = tid; idx < nrow * fc; idx += HOP_THREADS) {
            const int rl = idx / fc, fci = idx - rl * fc;
            const int frow = 16 * rt + rl, b_ = frow / sq.N, k = frow - b_ * sq.N;
            const int f = sq.lo + k * fc + fci;
            const float* pr = pbuf + rl * LDP;
            const float* hrow = p.hist + ((size_t)b_ * F + f) * D * 2;
            const float* irow = p.inp + ((size_t)b_ * F + f) * hop * 2;
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
                const size_t o = (((size_t)b_ * S + s) * F + f) * hop + t;
                *reinterpret_cast<float2*>(p.enh + 2 * o) = make_float2(yr, yi);
                if (p.mag) p.mag[o] = fast_abs2(yr, yi);
            }
        }
        if (t == 0) HOP_STAMP(6);
    }
    if (fast || D == 0) return;
    // ---- history of my bins: the last D of [old history | new frames]; one thread owns a bin, ascending order reads ahead
    __syncthreads();
    for (int idx = tid; idx < nrow * fc; idx += HOP_THREADS) {
        const int rl = idx / fc, fci = idx - rl * fc;
        const int frow = 16 * rt + rl, b_ = frow / sq.N, k = frow - b_ * sq.N;
        const int f = sq.lo + k * fc + fci;
        float* hrow = p.hist + ((size_t)b_ * F + f) * D * 2;
        const float* irow = p.inp + ((size_t)b_ * F + f) * hop * 2;
        for (int i = 0; i < D; ++i) {
            const int src = i + hop;
            const float2 v = src < D ? *reinterpret_cast<const float2*>(hrow + 2 * src) : *reinterpret_cast<const float2*>(irow + 2 * (src - D));
            *reinterpret_cast<float2*>(hrow + 2 * i) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// waveform mode (hop == 1), first stage: the new frame's spectrum.  One workgroup per 16 clips, a wave per clip: the frame is
// the last 384 samples of the state followed by the 128 new ones; sfsn_fft.hip's transform (same code, same bits); the bins
// leave as granules; then the state moves on by one hop.  LDS: [64 B][unit table 4 KB][8 x 2 KB exchange].
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void hop_stft_role(const HopParams& p, const HopStep& hs, const HopStageDev& sd, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = (int)blockIdx.x - sd.wg0;
    const int nclip = (p.B - 16 * rt) < 16 ? (p.B - 16 * rt) : 16;
    const unsigned tagw = hop_tag(hs.launch) * 0x02020202u;
    float2* unit = reinterpret_cast<float2*>(smem + 64);
    float2(*fbuf)[FFT_N] = reinterpret_cast<float2(*)[FFT_N]>(smem + 64 + FFT_NFFT * 8);
    fill_unit_table(unit, tid, HOP_THREADS);
    __syncthreads();
    const Twiddles tw = make_twiddles<false>(unit, lane);
    float2 win[4], wk[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = 2 * (lane + 64 * r);
        win[r] = make_float2(p.window[n], p.window[n + 1]);
        wk[r] = unit_at<false>(unit, lane + 64 * r);
    }
    for (int ci = wave; ci < nclip; ci += HOP_WAVES) {
        const int b = 16 * rt + ci;
        const float* ws = p.wave_state + (size_t)b * FFT_NFFT;
        const float* wn = p.wave_in + (size_t)b * 128;
        float2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 2 * (lane + 64 * r);  // sample j of the frame: state[128 + j] for j < 384, then the new samples
            const float2 x = j < 384 ? *reinterpret_cast<const float2*>(ws + 128 + j) : *reinterpret_cast<const float2*>(wn + j - 384);
            v[r] = make_float2(x.x * win[r].x, x.y * win[r].y);
        }
        fft256<false>(v, fbuf[wave], lane, tw);
        float2 X[4], nyq = make_float2(0.0f, 0.0f);
        rfft512_split(v, fbuf[wave], lane, wk, X, nyq);
#pragma unroll
        for (int r = 0; r < 4; ++r) hop_put_cplx(p.spec_g + ((size_t)b * p.F + lane + 64 * r) * 4, X[r], tagw);
        if (lane == 0) hop_put_cplx(p.spec_g + ((size_t)b * p.F + FFT_N) * 4, nyq, tagw);
    }
    // the state moves on by one hop (every sample is read before any is written)
    float keep[16];
#pragma unroll
    for (int ci = 0; ci < 16; ++ci) {
        keep[ci] = 0.0f;
        if (ci < nclip) {
            const int b = 16 * rt + ci;
            keep[ci] = tid < 384 ? p.wave_state[(size_t)b * FFT_NFFT + 128 + tid] : p.wave_in[(size_t)b * 128 + tid - 384];
        }
    }
    __syncthreads();
#pragma unroll
    for (int ci = 0; ci < 16; ++ci)
        if (ci < nclip) p.wave_state[(size_t)(16 * rt + ci) * FFT_NFFT + tid] = keep[ci];
}

// ---------------------------------------------------------------------------------------------------------------------
// waveform mode, last stage: the enhanced frame back to samples.  A wave per (clip, speaker): polls the 257 bins of the
// enhanced frame (granules written by the deep-filter workgroups), sfsn_fft.hip's inverse transform and window, overlap-add
// in registers against the carried accumulator (ascending frame order, as istft_kernel adds them), the hop that is now
// complete divided by the squared-window envelope of the frames that exist (t - q >= 0), accumulator moved on by one hop.
// LDS: [64 B][unit table 4 KB][8 x 2 KB exchange][8 x 264 float2 spectrum rows].
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void hop_istft_role(const HopParams& p, const HopStep& hs, const HopStageDev& sd, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = ((int)blockIdx.x - sd.wg0) * HOP_WAVES + wave;  // clip * S + speaker
    const unsigned tagw = hop_tag(hs.launch) * 0x02020202u;
    float2* unit = reinterpret_cast<float2*>(smem + 64);
    float2(*fbuf)[FFT_N] = reinterpret_cast<float2(*)[FFT_N]>(smem + 64 + FFT_NFFT * 8);
    float2(*xs)[264] = reinterpret_cast<float2(*)[264]>(smem + 64 + FFT_NFFT * 8 + HOP_WAVES * FFT_N * 8);
    fill_unit_table(unit, tid, HOP_THREADS);
    __syncthreads();
    if (pair >= p.B * p.S) return;
    const Twiddles tw = make_twiddles<true>(unit, lane);
    float2 win[4], wk[4], ola[4];
    float* os = p.ola_state + (size_t)pair * FFT_NFFT;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = 2 * (lane + 64 * r);
        win[r] = make_float2(p.window[n], p.window[n + 1]);
        wk[r] = unit_at<true>(unit, lane + 64 * r);
        ola[r] = *reinterpret_cast<const float2*>(os + n);
    }
    bool ok = true;
    const float* eg = p.enh_g + (size_t)pair * p.F * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) xs[wave][lane + 64 * r] = hop_take_cplx(eg + (size_t)(lane + 64 * r) * 4, tagw, ok, p.cnt);
    if (lane == 0) xs[wave][FFT_N] = hop_take_cplx(eg + (size_t)FFT_N * 4, tagw, ok, p.cnt);
    __builtin_amdgcn_wave_barrier();
    float2 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = lane + 64 * r;
        v[r] = irfft512_presplit(xs[wave][k], xs[wave][FFT_N - k], k, wk[r]);
    }
    fft256<true>(v, fbuf[wave], lane, tw);
    const float sc = 1.0f / (float)FFT_N;
    float2 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float2 res = make_float2(v[r].x * sc * win[r].x, v[r].y * sc * win[r].y);
        acc[r] = make_float2(ola[r].x + res.x, ola[r].y + res.y);
    }
    // the hop that is complete now: padded positions n = 128 t + 2 lane + e; envelope over the frames t - q that exist, oldest first
    float2 env = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int q = 3; q >= 0; --q)
        if (hs.frame_index - q >= 0) {
            env.x += win[q].x * win[q].x;
            env.y += win[q].y * win[q].y;
        }
    const float2 out = make_float2(env.x > 1e-11f ? acc[0].x / env.x : 0.0f, env.y > 1e-11f ? acc[0].y / env.y : 0.0f);
    *reinterpret_cast<float2*>(p.wave_out + (size_t)pair * 128 + 2 * lane) = out;
    if (p.done) {
        // wave_out (and this word) may be host memory the device can reach: a caller that keeps its samples on the host spins on
        // the word instead of synchronising the stream -- the samples are there when it changes (system-scope release)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(p.done + pair, hs.launch + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = 2 * (lane + 64 * r);
        *reinterpret_cast<float2*>(os + n) = r < 3 ? acc[r + 1] : make_float2(0.0f, 0.0f);
    }
}

// one hop of one workgroup: the role its block index selects
template <bool ONE, int G>
__device__ __forceinline__ void hop_dispatch(const HopParams& p, const HopStep& hs, char* smem) {
    if ((int)blockIdx.x < p.st[0].nwg) {
        // layer 0 of the full-band model is the head of the frame's critical path: its descriptors sit at fixed kernarg
        // offsets, so every scalar load is issued at once instead of table -> stage -> sequence
        hop_layer_role<true, ONE, G>(p, hs, p.st[0], p.seq[0], smem);
        return;
    }
    const int si = (int)((p.stage_of_block[blockIdx.x >> 2] >> (8 * (blockIdx.x & 3))) & 0xffu);
    const HopStageDev& sd = p.st[si];
    const HopSeqDev& sq = p.seq[sd.seq];
    if (sd.layer >= 0) {
        if (sd.layer == 0)
            hop_layer_role<true, ONE, G>(p, hs, sd, sq, smem);
        else
            hop_layer_role<false, ONE, G>(p, hs, sd, sq, smem);
    } else if (sd.layer == -1) {
        hop_proj_role<ONE>(p, hs, sd, sq, smem);
    } else if (ONE && sd.layer == -2) {
        hop_stft_role(p, hs, sd, smem);
    } else if (ONE) {
        hop_istft_role(p, hs, sd, smem);
    }
}

// (G = 2, separate gate weights, is a kernel of its own: the shared-weights kernels keep their register allocation -- 252 registers,
//  no spills for the one-frame hop)
template <bool ONE, int G>
__global__ __launch_bounds__(HOP_THREADS) void stream_hop_kernel(const HopParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    HOP_STAMP(0);
    HopStep hs;
    hs.launch = p.launch; hs.frame_index = p.frame_index; hs.frames_before = p.frames_before;
    hop_dispatch<ONE, G>(p, hs, smem);
    HOP_STAMP(7);
}

// ---- the RESIDENT form (BASELINE configs[4]: "persistent kernel", round 3): one launch serves hop after hop.  The host rings a
// doorbell word in pinned host memory (value k + 1 for hop k, 0xFFFFFFFF = stop) after it has put the hop's samples into the
// pinned input buffer; one wave of every workgroup polls it over PCIe (forwarding it through a device word polled by the
// others cost 2 us more per hop).  Within a hop the stages hand over exactly as in a launch
// (tagged granules, tag = launch index + k); the completion word per (clip, speaker) tells the host the enhanced samples are in
// its memory, and only then may it ring the next hop (every consumer of hop k has read its inputs by then: the last stage
// depends on all of them).  Bounded: a doorbell that stays silent for `idle_polls` polls ends the kernel (it must never outlive
// its host thread), as does a hand-off wait that expires inside a hop.  Waveform mode, one-frame hops.
template <int G>
__global__ __launch_bounds__(HOP_THREADS) void stream_hop_resident_kernel(const HopParams p, unsigned* doorbell, unsigned idle_ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* fin = p.cnt + 2;  // workgroups that have finished (and released) a hop, counted up over the hops
    for (unsigned k = 0;; ++k) {
        // one wave per workgroup waits, the other waves sleep at the barrier.  Relaxed polls (an acquire per poll would
        // invalidate the caches under the stages that are still computing) of two words at once: the host's doorbell, over
        // PCIe, and the count of workgroups that have released hop k - 1 -- the state a hop leaves for the next one (last
        // spikes, read by every workgroup of the layer) crosses compute units without a launch boundary here.
        if (threadIdx.x < 64) {
            unsigned v;
            const unsigned long long t0 = wall_clock64();  // 100 MHz
            for (unsigned spins = 0;; ++spins) {
                v = __hip_atomic_load(doorbell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned f = ld_agent(fin);
                // (wrap-safe compares: hop and completion counts are 32-bit and a stream may outlive them)
                const bool rung = (int)(v - (k + 1u)) >= 0;
                if (v == 0xFFFFFFFFu || (rung && (int)(f - k * gridDim.x) >= 0)) break;
                // the idle watchdog only fires while the doorbell is silent: once a hop has been rung this workgroup serves it (a peer
                // that is still finishing the previous hop is waited for; the hop's own bounded hand-off spins report a peer that left)
                if ((spins & 63u) == 63u && ((!rung && wall_clock64() - t0 > (unsigned long long)idle_ticks) || ld_agent(p.cnt) != 0u)) {
                    v = 0xFFFFFFFFu;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (threadIdx.x == 0) *reinterpret_cast<unsigned*>(smem) = v;  // (the hop's LDS is free between hops)
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: the hop's samples sit in host memory
        unsigned v = *reinterpret_cast<const unsigned*>(smem);
        __syncthreads();
        v = __builtin_amdgcn_readfirstlane(v);
        if (v == 0xFFFFFFFFu) break;
#ifdef SFSN_HOP_STAMPS
        const int wave = threadIdx.x >> 6;
        HOP_STAMP(0);
#endif
        HopStep hs;
        hs.launch = p.launch + k; hs.frame_index = p.frame_index + (int)k; hs.frames_before = p.frames_before + (int)k * p.hop;
        hop_dispatch<true, G>(p, hs, smem);
#ifdef SFSN_HOP_STAMPS
        HOP_STAMP(7);
#endif
        // the state that crosses compute units between hops (the last spikes) left as write-through stores: once they have
        // completed they are visible to the agent -- no L2 write-back (a release fence here cost the hops 4 us: buffer_wbl2
        // stalls the L2 under the stages that are still computing)
        __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0)
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(fin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the host learns that the kernel has left (watchdog, error word or its own stop) without a runtime call
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(doorbell + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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
    if (g.norm == SFSN_NORM_CUMLAPLACE && (!s.cum[0] || !s.cum[1])) return SFSN_EINVAL;
    if (g.norm == SFSN_NORM_LAYERNORM && (!g.ln_w || !g.ln_b)) return SFSN_EINVAL;
    if (!s.w_p || !s.w_p_dq || !s.b_p) return SFSN_EINVAL;
    memset(&d, 0, sizeof(d));
    d.nl = s.n_layers; d.H = s.H; d.P = s.P; d.R = B * g.n_units; d.KS = (s.H + 63) / 64; d.NT = s.H / 16; d.PT = (s.P + 15) / 16;
    d.I = I; d.I1 = I1; d.KC = (I + 15) / 16;
    d.lo = g.lo; d.N = g.n_units; d.ctr = g.ctr; d.nbr = g.nbr; d.ctr_fb = g.ctr_fb; d.nbr_fb = g.nbr_fb; d.norm = g.norm; d.eps = g.ln_eps;
    d.ln_w = g.ln_w; d.ln_b = g.ln_b; d.cum[0] = s.cum[0]; d.cum[1] = s.cum[1];
    d.w_p = s.w_p; d.w_p_dq = s.w_p_dq; d.b_p = s.b_p;
    d.df = is_fb ? 0 : s.df; d.fc = s.fc;
    if (!is_fb) {
        if (s.df < 1 || s.fc != g.ctr || s.P != 2 * s.fc * s.df * S) return SFSN_EINVAL;
    }
    for (int l = 0; l < s.n_layers; ++l) {
        const sfsn_hop_layer& L = s.layer[l];
        if (!L.w_hh || !L.w_hh_dq || !L.bias || !L.bn_alpha || !L.bn_beta || !L.h[0] || !L.h[1] || !L.c || !L.spikes) return SFSN_EINVAL;
        if (l == 0 ? !L.w_ih_frag : (!L.w_ih || !L.w_ih_dq)) return SFSN_EINVAL;
        HopLayerDev& o = d.layer[l];
        o.w_ih_f32 = L.w_ih_frag; o.w_ih = L.w_ih; o.w_ih_dq = L.w_ih_dq; o.w_hh = L.w_hh; o.w_hh_dq = L.w_hh_dq; o.bias = L.bias;
        o.alpha = L.bn_alpha; o.beta = L.bn_beta; o.h[0] = L.h[0]; o.h[1] = L.h[1]; o.c = L.c; o.spikes = L.spikes;
    }
    return SFSN_OK;
}

static int hop_plan(HopParams& p, size_t& lds, const sfsn_hop_desc* d) {
    if (!d || d->n_groups < 1 || d->n_groups > SFSN_HOP_MAX_GROUPS) return d ? SFSN_EUNSUPPORTED : SFSN_EINVAL;
    if (d->B <= 0 || d->F < 2 || d->S < 1 || d->hop < 1 || d->D < 0 || d->D + d->hop > 32) return SFSN_EUNSUPPORTED;
    const bool wave = d->wave_in != nullptr;
    if ((!wave && !d->inp_ri) || !d->enh_ri || (d->D > 0 && !d->hist_ri)) return SFSN_EINVAL;
    if (wave) {
        if (!d->wave_state || !d->ola_state || !d->wave_out || !d->window || !d->spec_g || !d->enh_g) return SFSN_EINVAL;
        if (d->hop != 1 || d->F != FFT_F) return SFSN_EUNSUPPORTED;  // one 128-sample hop per launch, 512-point frames
    }
    memset(&p, 0, sizeof(p));
    p.B = d->B; p.F = d->F; p.S = d->S; p.hop = d->hop; p.D = d->D; p.FB = d->fb.P; p.fdrc = d->fdrc;
    p.G = d->unshared ? 2 : 1;
    p.inp = d->inp_ri; p.hist = d->hist_ri; p.enh = d->enh_ri; p.mag = d->enh_mag;
    if (wave) {
        p.wave_in = d->wave_in; p.wave_state = d->wave_state; p.ola_state = d->ola_state; p.wave_out = d->wave_out;
        p.window = d->window; p.spec_g = d->spec_g; p.enh_g = d->enh_g; p.frame_index = d->frame_index;
        p.done = d->done;
    }
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
    if (p.seq[0].PT > HOP_WAVES) return SFSN_EUNSUPPORTED;  // the sub-band layer-0 workgroups compute it with one tile per wave
    // stages in dependency order: producers get the lower block indices.  Sub-band stages go layer by layer over all groups
    // (the groups' layer-0 workgroups all wait for the same full-band layer).
    int ns = 0, wg = 0;
    auto add = [&](int si_seq, int layer) {
        const HopSeqDev& q = p.seq[si_seq];
        HopStageDev& s = p.st[ns++];
        s.seq = si_seq; s.layer = layer; s.nrt = (q.R + 15) / 16;
        s.ntile = layer >= 0 ? q.NT : HOP_WAVES;
        s.ntpad = (s.ntile + HOP_WAVES - 1) / HOP_WAVES * HOP_WAVES;
        s.wg0 = wg; s.nwg = s.nrt * s.ntpad / HOP_WAVES;
        wg += s.nwg;
    };
    int maxl = 0, kcmax = 0, pmax = 0;
    for (int i = 0; i < p.nseq; ++i) {
        if (i > 0 && p.seq[i].nl > maxl) maxl = p.seq[i].nl;
        if (p.seq[i].KC > kcmax) kcmax = p.seq[i].KC;
        if (i > 0 && p.seq[i].P > pmax) pmax = p.seq[i].P;
    }
    auto add_wave = [&](int layer, int n) {
        HopStageDev& s = p.st[ns++];
        s.seq = 0; s.layer = layer; s.nrt = n; s.ntile = HOP_WAVES; s.ntpad = HOP_WAVES;
        s.wg0 = wg; s.nwg = n;
        wg += n;
    };
    for (int l = 0; l < p.seq[0].nl; ++l) {
        add(0, l);
        if (l == 0 && wave) add_wave(-2, (d->B + 15) / 16);  // (behind the full-band layer 0, whose descriptors sit at stage 0)
    }
    for (int l = 0; l < maxl; ++l)
        for (int g = 0; g < d->n_groups; ++g)
            if (l < p.seq[1 + g].nl) add(1 + g, l);
    for (int g = 0; g < d->n_groups; ++g) add(1 + g, -1);
    if (wave) {
        add_wave(-3, (d->B * d->S + HOP_WAVES - 1) / HOP_WAVES);
        // the deep filter's one-bin-per-thread path is the one that writes the enhanced granules
        for (int g = 0; g < d->n_groups; ++g) {
            const HopSeqDev& q = p.seq[1 + g];
            const int rows = q.R < 16 ? q.R : 16;
            if (rows * q.fc > HOP_THREADS * HOP_DF_ITEMS || d->D > HOP_DF_MAX - 1) return SFSN_EUNSUPPORTED;
        }
    }
    p.nstage = ns; p.nblocks = wg;
    if (wg > HOP_MAX_BLOCKS) return SFSN_EUNSUPPORTED;
    for (int i = 0; i < ns; ++i)
        for (int b = p.st[i].wg0; b < p.st[i].wg0 + p.st[i].nwg; ++b) p.stage_of_block[b >> 2] |= (unsigned)i << (8 * (b & 3));
    const size_t lds_layer = 64 + (size_t)2 * HOP_KS_MAX * 1024 + ((size_t)16 * p.FB + (size_t)16 * (kcmax * 16 + 4)) * sizeof(float) +
                             (size_t)3 * p.seq[0].PT * p.seq[0].KS * 1024;
    const size_t lds_proj = 64 + (size_t)HOP_KS_MAX * 1024 + (size_t)16 * (pmax + 4) * sizeof(float);
    lds = lds_layer > lds_proj ? lds_layer : lds_proj;
    const size_t lds_wave = 64 + (size_t)FFT_NFFT * 8 + (size_t)HOP_WAVES * FFT_N * 8 + (size_t)HOP_WAVES * 264 * 8;
    if (wave && lds_wave > lds) lds = lds_wave;
    return SFSN_OK;
}

static size_t hop_counter_bytes(const HopParams&) { return 64; }

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
    return hop_counter_bytes(p) + (size_t)p.nblocks * HOP_WAVES * 8 * sizeof(unsigned long long);
}

extern "C" int sfsn_hop_stages(const sfsn_hop_desc* desc, int* out, int cap) {
    HopParams p;
    size_t lds;
    const int rc = hop_plan(p, lds, desc);
    if (rc != SFSN_OK) return rc;
    for (int i = 0; i < p.nstage && i < cap; ++i) {
        out[4 * i + 0] = p.st[i].seq; out[4 * i + 1] = p.st[i].layer; out[4 * i + 2] = p.st[i].wg0; out[4 * i + 3] = p.st[i].nwg;
    }
    return p.nstage;
}

extern "C" int sfsn_stream_hop(const sfsn_hop_desc* desc, void* stream) {
    HopParams local;
    size_t lds;
    const int rc = hop_plan(local, lds, desc);
    if (rc != SFSN_OK) return rc;
    if (!desc->scratch || desc->scratch_bytes < hop_counter_bytes(local) + (size_t)local.nblocks * HOP_WAVES * 64) return SFSN_EINVAL;
    local.cnt = static_cast<unsigned*>(desc->scratch);
    local.launch = desc->launch_index;
    local.frames_before = desc->frames_before;
    // per-wave time stamps behind the control words (written by -DSFSN_HOP_STAMPS builds only; scripts/exp_hop.py reads them)
    local.dbg = reinterpret_cast<unsigned long long*>(static_cast<char*>(desc->scratch) + hop_counter_bytes(local));
    int dev = 0;
    const int fit = hop_fits_device(local.nblocks);
    if (fit != SFSN_OK) return fit;
    if (hipGetDevice(&dev) != hipSuccess) return SFSN_EHIP;
    static int lds_set[2][2][64];  // per kernel instantiation (hop == 1 or not, one or two gate matrices) and device (round-4 advisor finding)
    const int one = local.hop == 1 ? 1 : 0;
    const bool g2 = local.G == 2;
    const void* kern = one ? (g2 ? reinterpret_cast<const void*>(stream_hop_kernel<true, 2>) : reinterpret_cast<const void*>(stream_hop_kernel<true, 1>))
                           : (g2 ? reinterpret_cast<const void*>(stream_hop_kernel<false, 2>) : reinterpret_cast<const void*>(stream_hop_kernel<false, 1>));
    if (lds > 64 * 1024 && (dev < 0 || dev >= 64 || !lds_set[one][g2 ? 1 : 0][dev])) {  // (devices beyond the cache: set on every launch)
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return SFSN_EHIP;
        if (dev >= 0 && dev < 64) lds_set[one][g2 ? 1 : 0][dev] = 1;
    }
    if (lds > 160 * 1024) return SFSN_EUNSUPPORTED;
    if (one)
        if (g2) hipLaunchKernelGGL((stream_hop_kernel<true, 2>), dim3(local.nblocks), dim3(HOP_THREADS), lds, static_cast<hipStream_t>(stream), local);
        else hipLaunchKernelGGL((stream_hop_kernel<true, 1>), dim3(local.nblocks), dim3(HOP_THREADS), lds, static_cast<hipStream_t>(stream), local);
    else
        if (g2) hipLaunchKernelGGL((stream_hop_kernel<false, 2>), dim3(local.nblocks), dim3(HOP_THREADS), lds, static_cast<hipStream_t>(stream), local);
        else hipLaunchKernelGGL((stream_hop_kernel<false, 1>), dim3(local.nblocks), dim3(HOP_THREADS), lds, static_cast<hipStream_t>(stream), local);
    return hipGetLastError() == hipSuccess ? SFSN_OK : SFSN_EHIP;
}

extern "C" int sfsn_stream_hop_resident(const sfsn_hop_desc* desc, void* doorbell, unsigned idle_ms, void* stream) {
    HopParams local;
    size_t lds;
    const int rc = hop_plan(local, lds, desc);
    if (rc != SFSN_OK) return rc;
    if (!doorbell || !desc->wave_in || !desc->done || desc->hop != 1) return SFSN_EINVAL;  // waveform mode with host completion words
    if (!desc->scratch || desc->scratch_bytes < hop_counter_bytes(local) + (size_t)local.nblocks * HOP_WAVES * 64) return SFSN_EINVAL;
    local.cnt = static_cast<unsigned*>(desc->scratch);
    local.launch = desc->launch_index;
    local.frames_before = desc->frames_before;
    local.dbg = reinterpret_cast<unsigned long long*>(static_cast<char*>(desc->scratch) + hop_counter_bytes(local));
    const int fit = hop_fits_device(local.nblocks);
    if (fit != SFSN_OK) return fit;
    if (lds > 160 * 1024) return SFSN_EUNSUPPORTED;
    const bool g2 = local.G == 2;
    const void* kern = g2 ? reinterpret_cast<const void*>(stream_hop_resident_kernel<2>) : reinterpret_cast<const void*>(stream_hop_resident_kernel<1>);
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return SFSN_EHIP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(local.cnt + 1, 0, 2 * sizeof(unsigned), st) != hipSuccess) return SFSN_EHIP;  // (word 2: the hop-finished count)
    const unsigned polls = (idle_ms > 30000u ? 30000u : idle_ms) * 100000u;  // ticks of the 100 MHz wall clock
    if (g2) hipLaunchKernelGGL(stream_hop_resident_kernel<2>, dim3(local.nblocks), dim3(HOP_THREADS), lds, st, local, static_cast<unsigned*>(doorbell), polls);
    else hipLaunchKernelGGL(stream_hop_resident_kernel<1>, dim3(local.nblocks), dim3(HOP_THREADS), lds, st, local, static_cast<unsigned*>(doorbell), polls);
    return hipGetLastError() == hipSuccess ? SFSN_OK : SFSN_EHIP;
}
