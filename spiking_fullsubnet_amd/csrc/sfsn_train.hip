// sfsn_train.hip -- training-mode GSN cell steps for gfx950: sfsn_gsn_train_step_fwd / _bwd.
//
// The reference trains with nn.BatchNorm1d INSIDE the cell in training mode (efficient_spiking_neuron.py:123,149-150): every
// time step normalises the membrane with the statistics of THAT step over all rows of the layer and updates the running
// statistics, and the backward pass goes through the triangle surrogate of the spike (:94-101).  The rows of a step are
// therefore coupled (one reduction over all R rows per neuron and step), which is why this is not the inference scan: a step
// is one launch, a workgroup owns 16 neurons (both gates) for ALL rows, keeps the step's pre-normalisation membranes in LDS,
// reduces over the rows inside the workgroup (no atomics, fixed order: bit-stable run to run) and finishes the step.  The
// time-parallel products (x . W_ih^T, the weight gradients, dL/dx) and the one sequential product of the backward pass
// (dL/dh_{t-1} = dz_t . W_hh) are plain library GEMMs on the host side (spiking_fullsubnet_amd/training.py).
//
// Forward, per row r and neuron j (NEURON:132-153, same association as the reference):
//     rec   = sum_k h_prev[r][k] * W_hh[g*H + j][k]              (fp32, k ascending)
//     pre_f = (z[r][j] + bias[j]) + rec_f ;  pre_g = (z[r][gH + j] + bias[H + j]) + rec_g
//     f = 1 / (1 + exp(-pre_f)) ;  c' = f * c_prev + (1 - f) * pre_g
//     mean_j = mean_r c' ; var_j = mean_r (c' - mean_j)^2 ; xhat = (c' - mean_j) * rsqrt(var_j + eps) ; u = xhat * gamma_j + beta_j
//     running_mean = (1 - m) running_mean + m mean ; running_var = (1 - m) running_var + m var R / (R - 1)
//     h = (u >= 0) ;  carry (h, u)
// Backward (given dL/dh_t = upstream + recurrent part, dL/dc_t from step t+1):
//     du = dh * max(0, 1 - |u|) + dc                               Triangle.backward, gamma = 1
//     dgamma_j += sum_r du xhat ; dbeta_j += sum_r du ; dc' = gamma_j invstd_j / R * (R du - sum_r du - xhat sum_r du xhat)
//     df = dc' (c_prev - pre_g) ; dc_prev = dc' f ; dpre_g = dc' (1 - f) ; dpre_f = df f (1 - f)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

#include "sfsn.h"

#define TR_THREADS 256
#define TR_TILE 16
#define TR_PF 4  // rows per thread whose step inputs are prefetched ahead of a wait (up to 64 rows per block; more rows: direct loads)
__device__ __forceinline__ float tr_pick(const float (&a)[TR_PF], int i) {  // a[i] without a scratch-memory array (i < TR_PF)
    return i == 0 ? a[0] : i == 1 ? a[1] : i == 2 ? a[2] : a[3];
}

static inline int hip_ok_tr(hipError_t e) { return e == hipSuccess ? SFSN_OK : SFSN_EHIP; }

// block reduction of 16 per-neuron partial sums held by threads (rsub = tid / 16 in [0, 16), j = tid % 16): red[rsub][j] -> total in
// every thread of column j.  Fixed order.
__device__ __forceinline__ float reduce16(float v, float (*red)[TR_TILE], int rsub, int j) {
    red[rsub][j] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TR_THREADS / TR_TILE; ++i) s += red[i][j];
    __syncthreads();
    return s;
}


// ---- several workgroups per neuron tile (round 3b): a launch is a grid of (H / 16 neuron tiles) x (RB row blocks); the row
// blocks of a tile exchange 2 x 16 partial sums (+ a row count) per step through `scratch`.  Transport: DATA-TAGGED GRANULES
// (MI355X_MICROARCH.md, handoff-1to1): a value travels as one naturally aligned 8-byte {value, tag} written by ONE write-through
// store, tag = the step's epoch (1, 2, ... over the steps issued on this scratch buffer; the buffer starts zeroed) -- data and
// "ready" arrive together, so there is no flag, no drained store queue, no arrival counter.  Every granule of a tile is polled
// by exactly one thread of every reading workgroup (relaxed agent loads, s_sleep between polls, bounded), the values land in
// LDS, one barrier.  The first form (sc1 payload, release fence, per-tile arrival counter, one polling lane, agent loads of the
// payload) cost ~5 of a step launch's 16 us.  Every block merges the partials in the same fixed order, so all blocks of a tile
// (and every run) get the same bits.  All blocks of a launch are resident (a few hundred 256-thread blocks with little LDS).
#define TR_PART 36   // LDS floats per (tile, row block): [0] rows, [2..17] first partial per neuron, [18..33] second
#define TR_PARTG 40  // granules (8 bytes) per (tile, row block) in `scratch`, same indices
__device__ __forceinline__ bool tr_exchange(float* scratch, unsigned* /*counters*/, unsigned* err, int tile, int rb, int RB, unsigned epoch,
                                            float p0, float p1, float nrows, int rsub, int j, float* out /* [RB][TR_PART] in LDS */) {
    unsigned long long* gran = reinterpret_cast<unsigned long long*>(scratch) + (size_t)tile * 16 * TR_PARTG;  // [16][TR_PARTG]
    unsigned long long* mine = gran + (size_t)rb * TR_PARTG;
    const unsigned long long tag = (unsigned long long)epoch << 32;
    if (rsub == 0) {
        __hip_atomic_store(mine + 2 + j, tag | __float_as_uint(p0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 18 + j, tag | __float_as_uint(p1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (j == 0) __hip_atomic_store(mine, tag | __float_as_uint(nrows), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int good = 1;
    for (int i = threadIdx.x; i < RB * 34; i += TR_THREADS) {
        const int b = i / 34, k = i - b * 34;
        if (k == 1) continue;
        const unsigned long long* g = gran + (size_t)b * TR_PARTG + k;
        unsigned long long v = 0;
        for (unsigned spins = 0;; ++spins) {
            v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(v >> 32) == epoch) break;
            if (spins > 300000u) {  // ~0.3 s: a launch whose blocks are not all resident must not hang the device
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        out[b * TR_PART + k] = __uint_as_float((unsigned)v);
    }
    return __syncthreads_and(good) != 0;
}

struct TrainFwdParams {
    const float* z;       // [R][G*H]
    const float* w_hh;    // [G*H][H]
    const float* bias;    // [2H]
    const float* h_prev;  // [R][H]
    const float* c_prev;  // [R][H]
    const float* bn_w;    // [H] or null
    const float* bn_b;
    float* running_mean;  // [H] or null
    float* running_var;
    float* spikes;        // [R][H]
    float* u;             // [R][H]
    float* xhat;          // [R][H] (bn only)
    float* f;             // [R][H]
    float* g;             // [R][H]  pre_g
    float* invstd;        // [H] (bn only)
    float momentum, eps;
    int R, H, shared, use_bn;
    int RB, rpb;          // row blocks per neuron tile, rows per block
    unsigned epoch;       // 1-based step count of this layer call (the arrival counters are monotonic)
    float* scratch;       // [H/16][RB][TR_PART] partials
    unsigned* counters;   // [H/16] arrivals, then [1] error word
};

__global__ __launch_bounds__(TR_THREADS) void gsn_train_step_fwd_kernel(const TrainFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char tr_smem[];
    const int H = p.H, G = p.shared ? 1 : 2;
    const int tile = blockIdx.x, rb = blockIdx.y;
    const int r_lo = rb * p.rpb, r_hi = (r_lo + p.rpb < p.R) ? r_lo + p.rpb : p.R, nr = r_hi - r_lo;
    float* wt = reinterpret_cast<float*>(tr_smem);                       // [G][16][H + 1]
    float* cbuf = wt + (size_t)G * TR_TILE * (H + 1);                     // [rpb][16] pre-normalisation membranes of my rows
    float* parts = cbuf + (size_t)p.rpb * TR_TILE;                        // [RB][TR_PART]
    unsigned* hb = reinterpret_cast<unsigned*>(parts + (size_t)p.RB * TR_PART);  // [rpb][H / 4] h_{t-1} of my rows, a byte per neuron
    __shared__ float red[TR_THREADS / TR_TILE][TR_TILE];
    const int tid = threadIdx.x, j = tid & 15, rsub = tid >> 4;
    const int n0 = tile * TR_TILE, nj = n0 + j;
    // the last spikes of my rows: whole rows, 16 bytes per request, all requests of a thread in flight together (the round-3a
    // kernel read them element by element inside the product loop: 224 dependent-latency loads per row -- most of its 15.9 us)
    const int H4 = H >> 2;
    for (int i = tid; i < nr * H4; i += TR_THREADS) {
        const int rr = i / H4, c4 = i - rr * H4;
        const float4 h = *reinterpret_cast<const float4*>(p.h_prev + (size_t)(r_lo + rr) * H + 4 * c4);
        hb[rr * H4 + c4] = (h.x != 0.f ? 1u : 0u) | (h.y != 0.f ? 0x100u : 0u) | (h.z != 0.f ? 0x10000u : 0u) | (h.w != 0.f ? 0x1000000u : 0u);
    }
    for (int i = tid; i < G * TR_TILE * H; i += TR_THREADS) {
        const int gi = i / (TR_TILE * H), rem = i - gi * TR_TILE * H, jj = rem / H, k = rem - jj * H;
        wt[(gi * TR_TILE + jj) * (H + 1) + k] = p.w_hh[((size_t)gi * H + n0 + jj) * H + k];
    }
    __syncthreads();
    const float bf = p.bias[nj], bg = p.bias[H + nj];
    const float* wf = wt + (size_t)j * (H + 1);
    const float* wg = wt + (size_t)((G - 1) * TR_TILE + j) * (H + 1);
    float sum = 0.f;
    for (int r = r_lo + rsub; r < r_hi; r += TR_THREADS / TR_TILE) {
        const unsigned* hp = hb + (size_t)(r - r_lo) * H4;
        float rf = 0.f, rg = 0.f;
        for (int k4 = 0; k4 < H4; ++k4) {  // (h is 0 / 1: fma(h, w, acc) in k order, as before)
            const unsigned hw = hp[k4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float hk = (float)((hw >> (8 * e)) & 1u);
                rf = __builtin_fmaf(hk, wf[4 * k4 + e], rf);
                if (G == 2) rg = __builtin_fmaf(hk, wg[4 * k4 + e], rg);
            }
        }
        if (G == 1) rg = rf;
        const float zf = p.z[(size_t)r * G * H + nj], zg = p.z[(size_t)r * G * H + (G - 1) * H + nj];
        const float pre_f = (zf + bf) + rf;
        const float pre_g = (zg + bg) + rg;
        const float f = 1.0f / (1.0f + expf(-pre_f));
        const float a = f * p.c_prev[(size_t)r * H + nj];
        const float b = (1.0f - f) * pre_g;
        const float cy = a + b;
        cbuf[(r - r_lo) * TR_TILE + j] = cy;
        p.f[(size_t)r * H + nj] = f;
        p.g[(size_t)r * H + nj] = pre_g;
        sum += cy;
    }
    if (!p.use_bn) {
        __syncthreads();
        for (int r = r_lo + rsub; r < r_hi; r += TR_THREADS / TR_TILE) {
            const float cy = cbuf[(r - r_lo) * TR_TILE + j];
            p.u[(size_t)r * H + nj] = cy;
            p.spikes[(size_t)r * H + nj] = cy >= 0.f ? 1.f : 0.f;
        }
        return;
    }
    // statistics of this step over ALL rows of the layer: per row block (count, mean, sum of squared deviations), merged pairwise in
    // block order (the parallel-variance formula: exact in real arithmetic, a few ulp from a two-pass evaluation in fp32)
    const float tot_b = reduce16(sum, red, rsub, j);
    const float mean_b = nr > 0 ? tot_b / (float)nr : 0.f;
    float sq = 0.f;
    for (int r = r_lo + rsub; r < r_hi; r += TR_THREADS / TR_TILE) {
        const float d = cbuf[(r - r_lo) * TR_TILE + j] - mean_b;
        sq = __builtin_fmaf(d, d, sq);
    }
    const float m2_b = reduce16(sq, red, rsub, j);
    float mean = mean_b, m2 = m2_b;
    if (p.RB > 1) {
        if (!tr_exchange(p.scratch, p.counters, p.counters + gridDim.x, tile, rb, p.RB, p.epoch, mean_b, m2_b, (float)nr, rsub, j, parts)) return;
        float cnt = 0.f;
        mean = 0.f; m2 = 0.f;
        for (int b = 0; b < p.RB; ++b) {
            const float nb = parts[b * TR_PART], mb = parts[b * TR_PART + 2 + j], qb = parts[b * TR_PART + 18 + j];
            if (nb > 0.f) {
                const float tot = cnt + nb, delta = mb - mean;
                mean = mean + delta * (nb / tot);
                m2 = m2 + qb + delta * delta * (cnt * nb / tot);
                cnt = tot;
            }
        }
    }
    const float var = m2 / (float)p.R;
    const float invstd = 1.0f / sqrtf(var + p.eps);
    const float gam = p.bn_w[nj], bet = p.bn_b[nj];
    for (int r = r_lo + rsub; r < r_hi; r += TR_THREADS / TR_TILE) {
        const float xh = (cbuf[(r - r_lo) * TR_TILE + j] - mean) * invstd;
        const float uu = xh * gam + bet;
        p.xhat[(size_t)r * H + nj] = xh;
        p.u[(size_t)r * H + nj] = uu;
        p.spikes[(size_t)r * H + nj] = uu >= 0.f ? 1.f : 0.f;
    }
    if (rsub == 0 && rb == 0) {
        p.invstd[nj] = invstd;
        if (p.running_mean) {
            const float unb = p.R > 1 ? var * ((float)p.R / (float)(p.R - 1)) : var;
            p.running_mean[nj] = (1.0f - p.momentum) * p.running_mean[nj] + p.momentum * mean;
            p.running_var[nj] = (1.0f - p.momentum) * p.running_var[nj] + p.momentum * unb;
        }
    }
}

struct TrainBwdParams {
    const float* dz_next;  // [R][G*H] or null: d_z of step t+1 -- the recurrent part of dL/dh_t is formed HERE (dz_next . W_hh)
    const float* w_hh;     // [G*H][H]
    const float* dh_up;    // [R][H] or null
    const float* dh_rec;   // [R][H] or null (a caller-made recurrent part instead of dz_next)
    const float* dc_next;  // [R][H] or null
    const float* u;
    const float* xhat;
    const float* f;
    const float* g;
    const float* c_prev;
    const float* invstd;
    const float* bn_w;
    float* d_gates;        // [R][2H]: d pre_f | d pre_g
    float* d_z;            // [R][G*H]: shared: d pre_f + d pre_g; unshared: = d_gates
    float* dc_prev;        // [R][H]
    float* d_bn_w;         // [H] +=
    float* d_bn_b;         // [H] +=
    int R, H, shared, use_bn;
    int RB, rpb;
    unsigned epoch;
    float* scratch;
    unsigned* counters;
};

__global__ __launch_bounds__(TR_THREADS) void gsn_train_step_bwd_kernel(const TrainBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char tr_smem[];
    __shared__ float red[TR_THREADS / TR_TILE][TR_TILE];
    const int H = p.H, GH = (p.shared ? 1 : 2) * p.H;
    const int tile = blockIdx.x, rb = blockIdx.y;
    const int r_lo = rb * p.rpb, r_hi = (r_lo + p.rpb < p.R) ? r_lo + p.rpb : p.R;
    const int tid = threadIdx.x, j = tid & 15, rsub = tid >> 4;
    const int nj = tile * TR_TILE + j;
    float* dbuf = reinterpret_cast<float*>(tr_smem);   // [rpb][16] du of my rows
    float* parts = dbuf + (size_t)p.rpb * TR_TILE;      // [RB][TR_PART]
    float* wcol = parts + (size_t)p.RB * TR_PART;       // [G*H][16]: the columns of W_hh that feed my 16 neurons of h_t
    float* dzb = wcol + (size_t)GH * TR_TILE;           // [rpb][G*H]: d_z of step t+1, my rows (whole rows, 16 bytes per request)
    if (p.dz_next) {
        const int G4 = GH >> 2, nrw = r_hi - r_lo;
        for (int i = tid; i < nrw * G4; i += TR_THREADS) {
            const int rr = i / G4, c4 = i - rr * G4;
            *reinterpret_cast<float4*>(dzb + (size_t)rr * GH + 4 * c4) = *reinterpret_cast<const float4*>(p.dz_next + (size_t)(r_lo + rr) * GH + 4 * c4);
        }
        for (int i = tid; i < GH * TR_TILE; i += TR_THREADS) {
            const int nn = i / TR_TILE, jj = i - nn * TR_TILE;
            wcol[i] = p.w_hh[(size_t)nn * H + tile * TR_TILE + jj];
        }
        __syncthreads();
    }
    float s1 = 0.f, s2 = 0.f;
    for (int r = r_lo + rsub; r < r_hi; r += TR_THREADS / TR_TILE) {
        const size_t o = (size_t)r * H + nj;
        float dh = 0.f;
        if (p.dh_up) dh += p.dh_up[o];
        if (p.dh_rec) dh += p.dh_rec[o];
        if (p.dz_next) {  // dL/dh_t through step t+1's recurrent product: sum_n dz_{t+1}[r][n] W_hh[n][my neuron]
            const float* dz = dzb + (size_t)(r - r_lo) * GH;
            float acc = 0.f;
            for (int nn = 0; nn < GH; ++nn) acc = __builtin_fmaf(dz[nn], wcol[nn * TR_TILE + j], acc);
            dh += acc;
        }
        const float uu = p.u[o];
        const float tri = fmaxf(0.f, 1.0f - fabsf(uu));
        float du = dh * tri;
        if (p.dc_next) du += p.dc_next[o];
        dbuf[(r - r_lo) * TR_TILE + j] = du;
        if (p.use_bn) {
            s1 += du;
            s2 = __builtin_fmaf(du, p.xhat[o], s2);
        }
    }
    float k1 = 0.f, k2 = 0.f, scale = 1.f;
    if (p.use_bn) {
        k1 = reduce16(s1, red, rsub, j);
        k2 = reduce16(s2, red, rsub, j);
        if (p.RB > 1) {
            if (!tr_exchange(p.scratch, p.counters, p.counters + gridDim.x, tile, rb, p.RB, p.epoch, k1, k2, (float)(r_hi - r_lo), rsub, j, parts)) return;
            k1 = 0.f; k2 = 0.f;
            for (int b = 0; b < p.RB; ++b) {
                k1 += parts[b * TR_PART + 2 + j];
                k2 += parts[b * TR_PART + 18 + j];
            }
        }
        scale = p.bn_w[nj] * p.invstd[nj] / (float)p.R;
        if (rsub == 0 && rb == 0) {
            p.d_bn_w[nj] += k2;
            p.d_bn_b[nj] += k1;
        }
    } else {
        __syncthreads();
    }
    for (int r = r_lo + rsub; r < r_hi; r += TR_THREADS / TR_TILE) {
        const size_t o = (size_t)r * H + nj;
        const float du = dbuf[(r - r_lo) * TR_TILE + j];
        const float dcy = p.use_bn ? scale * ((float)p.R * du - k1 - p.xhat[o] * k2) : du;
        const float f = p.f[o], g = p.g[o], cp = p.c_prev[o];
        const float df = dcy * (cp - g);
        const float dpf = df * f * (1.0f - f);
        const float dpg = dcy * (1.0f - f);
        p.dc_prev[o] = dcy * f;
        p.d_gates[(size_t)r * 2 * H + nj] = dpf;
        p.d_gates[(size_t)r * 2 * H + H + nj] = dpg;
        if (p.shared) p.d_z[o] = dpf + dpg;
    }
}

// rows per workgroup: enough row blocks to put most compute units to work, at least 16 rows each; every block of a launch must be
// resident (they wait for each other): a layer call stays below ~160 blocks of 256 threads
static int train_wg_target() {
#ifdef SFSN_EXPERIMENTS
    static const int v = getenv("SFSN_TRAIN_WGS") ? atoi(getenv("SFSN_TRAIN_WGS")) : 160;
    return v;
#else
    return 160;  // (220 / 160 / 110 / 70 measured at B = 64, three groups in one launch: 9.4 / 8.3 / 8.1 / 8.3 ms forward, 13.6 / 12.5 / 13.1 / 14.7 backward per 200 steps x 2 layers)
#endif
}
static void train_geometry(int R, int H, int G, int* RB, int* rpb, int target = 0) {
    const int tiles = H / TR_TILE;
    int rb = (target > 0 ? target : train_wg_target()) / tiles;
    if (rb < 1) rb = 1;
    int per = 16;
    for (;; ++rb) {
        per = (R + rb - 1) / rb;
        if (per < 16) per = 16;
        per = (per + 15) & ~15;
        // many rows: more row blocks (up to the 16 the exchange buffers hold) before the backward launch's LDS -- W_hh columns and the
        // d_z rows of a block, (G H + 4) x (16 + rows) floats -- would not fit
        const size_t lds_b = ((size_t)3 * per * TR_TILE + (size_t)16 * TR_PART + (size_t)(G * H + 4) * (TR_TILE + per)) * sizeof(float);
        if (lds_b <= 150 * 1024 || rb >= 16) break;
    }
    *rpb = per;
    *RB = (R + per - 1) / per;
}

extern "C" size_t sfsn_train_scratch_bytes(int H) {
    if (H <= 0 || H % TR_TILE) return 0;
    const int tiles = H / TR_TILE;
    return ((size_t)tiles * 16 * TR_PARTG) * sizeof(unsigned long long) + ((size_t)tiles + 4) * sizeof(unsigned);  // granules (RB <= 16) + [unused] + error word
}

// Will both step kernels take (R, H)?  The forward and the backward step have different LDS needs (the backward one stages H x
// (16 + rows per block) floats of W_hh columns and d_z rows): a layer call checks BOTH before its first forward launch, so that a
// forward pass cannot succeed where the backward pass would be refused (round-3 advisor finding: R ~ 2048).
extern "C" int sfsn_gsn_train_check(int R, int H, int shared) {
    if (R <= 0 || H <= 0) return SFSN_EINVAL;
    if (H % TR_TILE != 0 || H > SFSN_MAX_HIDDEN) return SFSN_EUNSUPPORTED;
    int RB, rpb;
    const int G = shared ? 1 : 2;
    train_geometry(R, H, G, &RB, &rpb);
    if (RB > 16) return SFSN_EUNSUPPORTED;
    // (the one-launch layer calls carry the membrane / its gradient and the packed new spikes in LDS on top of the step kernels' buffers)
    const size_t lds_f = ((size_t)G * TR_TILE * (H + 4) + (size_t)(2 + G) * rpb * TR_TILE + (size_t)RB * TR_PART) * sizeof(float) + (size_t)rpb * H + (size_t)rpb * TR_TILE;
    const size_t lds_b = ((size_t)3 * rpb * TR_TILE + (size_t)RB * TR_PART + (size_t)(G * H + 4) * (TR_TILE + rpb)) * sizeof(float);
    return (lds_f > 150 * 1024 || lds_b > 150 * 1024) ? SFSN_EUNSUPPORTED : SFSN_OK;
}

extern "C" int sfsn_gsn_train_step_fwd(const float* z, const float* w_hh, const float* bias, const float* h_prev, const float* c_prev,
                                       const float* bn_w, const float* bn_b, float* running_mean, float* running_var, float momentum,
                                       float eps, int R, int H, int shared, float* spikes, float* u, float* xhat, float* f, float* g,
                                       float* invstd, void* scratch, unsigned epoch, void* stream) {
    if (!z || !w_hh || !bias || !h_prev || !c_prev || !spikes || !u || !f || !g || R <= 0 || H <= 0) return SFSN_EINVAL;
    if (H % TR_TILE != 0 || H > SFSN_MAX_HIDDEN) return SFSN_EUNSUPPORTED;
    const int use_bn = bn_w != nullptr;
    if (use_bn && (!bn_b || !xhat || !invstd)) return SFSN_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return SFSN_EINVAL;
    const int G = shared ? 1 : 2, tiles = H / TR_TILE;
    TrainFwdParams p;
    train_geometry(R, H, shared ? 1 : 2, &p.RB, &p.rpb);
    if (p.RB > 16) return SFSN_EUNSUPPORTED;  // (more than 16 x 220 / tiles x ... rows per layer and step)
    if (p.RB > 1 && use_bn && (!scratch || epoch == 0)) return SFSN_EINVAL;
    const size_t lds = ((size_t)G * TR_TILE * (H + 1) + (size_t)p.rpb * TR_TILE + (size_t)p.RB * TR_PART) * sizeof(float) + (size_t)p.rpb * H;
    if (lds > 150 * 1024) return SFSN_EUNSUPPORTED;
    p.z = z; p.w_hh = w_hh; p.bias = bias; p.h_prev = h_prev; p.c_prev = c_prev; p.bn_w = bn_w; p.bn_b = bn_b;
    p.running_mean = running_mean; p.running_var = running_var; p.spikes = spikes; p.u = u; p.xhat = xhat; p.f = f; p.g = g;
    p.invstd = invstd; p.momentum = momentum; p.eps = eps; p.R = R; p.H = H; p.shared = shared; p.use_bn = use_bn;
    p.epoch = epoch; p.scratch = static_cast<float*>(scratch);
    p.counters = scratch ? reinterpret_cast<unsigned*>(static_cast<float*>(scratch) + (size_t)tiles * 16 * TR_PARTG * 2) : nullptr;
    auto kern = gsn_train_step_fwd_kernel;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return SFSN_EHIP;
    hipLaunchKernelGGL(kern, dim3(tiles, p.RB), dim3(TR_THREADS), lds, static_cast<hipStream_t>(stream), p);
    return hip_ok_tr(hipGetLastError());
}

extern "C" int sfsn_gsn_train_step_bwd(const float* dz_next, const float* w_hh, const float* dh_up, const float* dh_rec, const float* dc_next,
                                       const float* u, const float* xhat, const float* f, const float* g, const float* c_prev,
                                       const float* invstd, const float* bn_w, int R, int H, int shared, float* d_gates, float* d_z,
                                       float* dc_prev, float* d_bn_w, float* d_bn_b, void* scratch, unsigned epoch, void* stream) {
    if (!u || !f || !g || !c_prev || !d_gates || !dc_prev || R <= 0 || H <= 0) return SFSN_EINVAL;
    if (H % TR_TILE != 0 || H > SFSN_MAX_HIDDEN) return SFSN_EUNSUPPORTED;
    const int use_bn = bn_w != nullptr;
    if (use_bn && (!xhat || !invstd || !d_bn_w || !d_bn_b)) return SFSN_EINVAL;
    if (shared && !d_z) return SFSN_EINVAL;
    if (dz_next && !w_hh) return SFSN_EINVAL;
    const int tiles = H / TR_TILE;
    TrainBwdParams p;
    train_geometry(R, H, shared ? 1 : 2, &p.RB, &p.rpb);
    if (p.RB > 16) return SFSN_EUNSUPPORTED;
    if (p.RB > 1 && use_bn && (!scratch || epoch == 0)) return SFSN_EINVAL;
    const size_t lds = ((size_t)p.rpb * TR_TILE + (size_t)p.RB * TR_PART + (dz_next ? (size_t)(shared ? 1 : 2) * H * (TR_TILE + p.rpb) : 0)) * sizeof(float);
    if (lds > 150 * 1024) return SFSN_EUNSUPPORTED;
    p.dz_next = dz_next; p.w_hh = w_hh;
    p.dh_up = dh_up; p.dh_rec = dh_rec; p.dc_next = dc_next; p.u = u; p.xhat = xhat; p.f = f; p.g = g; p.c_prev = c_prev;
    p.invstd = invstd; p.bn_w = bn_w; p.d_gates = d_gates; p.d_z = d_z; p.dc_prev = dc_prev; p.d_bn_w = d_bn_w; p.d_bn_b = d_bn_b;
    p.R = R; p.H = H; p.shared = shared; p.use_bn = use_bn;
    p.epoch = epoch; p.scratch = static_cast<float*>(scratch);
    p.counters = scratch ? reinterpret_cast<unsigned*>(static_cast<float*>(scratch) + (size_t)tiles * 16 * TR_PARTG * 2) : nullptr;
    auto kern = gsn_train_step_bwd_kernel;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return SFSN_EHIP;
    hipLaunchKernelGGL(kern, dim3(tiles, p.RB), dim3(TR_THREADS), lds, static_cast<hipStream_t>(stream), p);
    return hip_ok_tr(hipGetLastError());
}

// ---- a whole layer call in ONE launch (round 4) ---------------------------------------------------------------------------
// The step launches above cost 15.6 us (forward) / 10.9 us (backward) for ~2 us of arithmetic: every step paid a launch ramp, a
// reload of its W_hh tile into LDS and a host-side enqueue (16,000 launches per training step of baseline_m: the host could not
// issue them faster than ~14 us each, so the three sub-band groups did not even overlap on streams of their own).  Here the
// workgroups of a layer call stay resident for all T steps: the weight tile is loaded once, the carried membrane (forward) and
// its gradient (backward) stay in LDS, and what a step needs from OTHER workgroups travels through the L2:
//   * the BatchNorm partial sums between the row blocks of a neuron tile: the tagged granules of tr_exchange, epoch = step + 1;
//   * forward: h_{t-1} of my rows for ALL neurons (each of the H / 16 tile workgroups of the row block wrote 16 of them): four
//     spikes + the step's epoch per 32-bit word, ONE write-through (sc1) store each, into a two-slot buffer [t & 1][R][H / 4];
//     readers poll the words themselves (tr_read_tagged: data and "ready" arrive together -- the first form, words + a per-row-block
//     counter behind a drained store queue, spent 5.9 of a 14.8 us step on publish / wait / read).  Two slots suffice: a
//     workgroup publishes h_{t+1} only after it has read h_t from every tile of its row block, which they published after reading h_{t-1};
//   * backward: d_z of step t+1 of my rows (all G*H products, fp32: no room for a tag): the API tensor itself, written with sc1
//     stores; a per-row-block counter counts the tile workgroups that have published a step (their stores drained by vmcnt before
//     the add); readers poll the counter (one thread), then read with sc1 loads.
// All workgroups of the launch must be resident (train_geometry keeps the grid under 220 blocks of 256 threads); every spin is
// bounded (error word, the host raises).  Arithmetic and its order are the step kernels', value for value.
__device__ __forceinline__ bool tr_wait_counter(const unsigned* cnt, unsigned want, unsigned* err) {
    int good = 1;
    if (threadIdx.x == 0) {
        for (unsigned spins = 0;; ++spins) {
            if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) break;
            if (spins > 600000u) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return __syncthreads_and(good) != 0;
}
__device__ __forceinline__ void tr_publish(unsigned* cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my write-through stores of this step have reached the coherent level
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
typedef int tr_v4i __attribute__((ext_vector_type(4)));
typedef float tr_v4f __attribute__((ext_vector_type(4)));
// n 16-byte pieces of coherent (sc1) global memory -> LDS, all threads of the workgroup, four loads per thread in flight
// Four coherent 16-byte loads and the wait for them in ONE asm block: the compiler must not touch the destination registers between a
// load's issue and the s_waitcnt (with the wait as a statement of its own, tied to the registers by "+v", hipcc copied the
// destinations BEFORE the wait -- a read of registers with a load outstanding, which the hardware does not interlock).
__device__ __forceinline__ void tr_load4x16_sc1(tr_v4i& v0, tr_v4i& v1, tr_v4i& v2, tr_v4i& v3, const void* p0, const void* p1, const void* p2,
                                                const void* p3) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc1\n\t"
        "global_load_dwordx4 %1, %5, off sc1\n\t"
        "global_load_dwordx4 %2, %6, off sc1\n\t"
        "global_load_dwordx4 %3, %7, off sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
        : "memory");
}
// (rows of `ppr` pieces in the source, `dpr` >= ppr pieces apart in LDS)
__device__ __forceinline__ void tr_copy16_sc1(void* lds_dst, const void* src, int n, int ppr, int dpr) {
    const char* s8 = static_cast<const char*>(src);
    tr_v4i* d = static_cast<tr_v4i*>(lds_dst);
    const int pad = dpr - ppr;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * TR_THREADS) {
        tr_v4i v0, v1, v2, v3;
        const int i1 = i0 + TR_THREADS, i2 = i0 + 2 * TR_THREADS, i3 = i0 + 3 * TR_THREADS;
        // (pieces past the end: a harmless second read of this thread's first piece)
        tr_load4x16_sc1(v0, v1, v2, v3, s8 + (size_t)i0 * 16, s8 + (size_t)(i1 < n ? i1 : i0) * 16, s8 + (size_t)(i2 < n ? i2 : i0) * 16,
                        s8 + (size_t)(i3 < n ? i3 : i0) * 16);
        d[i0 + (i0 / ppr) * pad] = v0;
        if (i1 < n) d[i1 + (i1 / ppr) * pad] = v1;
        if (i2 < n) d[i2 + (i2 / ppr) * pad] = v2;
        if (i3 < n) d[i3 + (i3 / ppr) * pad] = v3;
    }
}

// Forward spike exchange, data-tagged: a 32-bit word carries four spikes (bits 0..3) and the step's epoch (bits 4..31), written by
// ONE write-through store -- data and "ready" arrive together, as in tr_exchange: no drained store queue, no counter, no publish
// barrier on the writer's side; the reader polls the words it needs (16-byte pieces = the 16 neurons of one tile for one row) until
// all four carry the epoch it waits for, and unpacks them to the byte-per-neuron form the product reads.  n16 pieces, all threads.
__device__ __forceinline__ unsigned tr_unpack4(unsigned w) { return (w & 1u) | ((w & 2u) << 7) | ((w & 4u) << 14) | ((w & 8u) << 21); }
__device__ __forceinline__ bool tr_read_tagged(unsigned* lds_dst, const unsigned* src, int n16, unsigned epoch, unsigned* err) {
    int good = 1;
    for (int i0 = threadIdx.x; i0 < n16; i0 += 4 * TR_THREADS) {
        tr_v4i v[4];
        unsigned pend = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = tr_v4i{0, 0, 0, 0};
            if (i0 + k * TR_THREADS < n16) pend |= 1u << k;
        }
        const unsigned* q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = src + 4 * (size_t)((pend & (1u << k)) ? i0 + k * TR_THREADS : i0);  // (absent pieces: my first one again)
        for (unsigned spins = 0; pend; ++spins) {
            tr_load4x16_sc1(v[0], v[1], v[2], v[3], q[0], q[1], q[2], q[3]);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((pend & (1u << k)) && ((unsigned)v[k].x >> 4) == epoch && ((unsigned)v[k].y >> 4) == epoch && ((unsigned)v[k].z >> 4) == epoch &&
                    ((unsigned)v[k].w >> 4) == epoch)
                    pend &= ~(1u << k);
            if (!pend) break;
            if (spins > 600000u) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i0 + k * TR_THREADS < n16) {
                tr_v4i o;
                o.x = (int)tr_unpack4((unsigned)v[k].x); o.y = (int)tr_unpack4((unsigned)v[k].y);
                o.z = (int)tr_unpack4((unsigned)v[k].z); o.w = (int)tr_unpack4((unsigned)v[k].w);
                reinterpret_cast<tr_v4i*>(lds_dst)[i0 + k * TR_THREADS] = o;
            }
    }
    return __syncthreads_and(good) != 0;
}

#ifdef SFSN_EXPERIMENTS
#define TR_STAMP(k) do { if (x.prof && tile == 0 && rb == 0 && threadIdx.x == 0) { const long long n_ = (long long)wall_clock64(); x.prof[k] += n_ - tr_t0; tr_t0 = n_; } } while (0)
#else
#define TR_STAMP(k) do { } while (0)
#endif
struct TrainSeqExtra {
    int T;
    unsigned* hx;     // forward: [2][R][H / 4] packed spikes (write-through)
    unsigned* rbcnt;  // [RB] publish counters of the row blocks
    long long* prof;  // -DSFSN_EXPERIMENTS: per-phase wall-clock ticks of workgroup (0, 0) (null otherwise)
    float* gran2;     // the partial-sum granules of the ODD steps (the even steps use p.scratch): a row block may write step s+1's
                      // partials while a slower one has not read step s's yet -- without a launch boundary between the steps that
                      // needs two slots (step s+2's are written only after every block's step s+1 partials were read, which they
                      // wrote after reading all of step s)
};

// One launch serves up to TR_MAXG layer calls of the same (T, H, gate sharing) -- the sub-band groups of a model, which are independent
// of one another: their workgroups sit side by side in ONE grid (workgroup -> (call, tile, row block) by a prefix table), so the
// groups overlap without a second queue.  (Three launches on three streams did overlap -- until one of them was never dispatched:
// reproducibly within a few hundred iterations a queued launch stayed at zero started workgroups behind finished ones,
// scripts/dbg_train_hang.py.  One grid has no such dependence on how the runtime maps streams to hardware queues.)
#define TR_MAXG 8
struct TrainSeqFwdMulti {
    TrainFwdParams p[TR_MAXG];
    TrainSeqExtra x[TR_MAXG];
    int wg_end[TR_MAXG];  // exclusive prefix end of each call's workgroups in the flat grid
    int n;
};
struct TrainSeqBwdMulti {
    TrainBwdParams p[TR_MAXG];
    TrainSeqExtra x[TR_MAXG];
    int wg_end[TR_MAXG];
    int n;
};

__device__ __forceinline__ void train_seq_fwd_body(const TrainFwdParams& p, const TrainSeqExtra& x, const int tile, const int rb, const int tiles) {
    extern __shared__ __attribute__((aligned(16))) char tr_smem[];
    const int H = p.H, G = p.shared ? 1 : 2;
    const int r_lo = rb * p.rpb, r_hi = (r_lo + p.rpb < p.R) ? r_lo + p.rpb : p.R, nr = r_hi - r_lo;
    const int WLD = H + 4;                                                // (rows 16-byte aligned: the product reads 4 k per request)
    float* wt = reinterpret_cast<float*>(tr_smem);                       // [G][16][WLD]
    float* cbuf = wt + (size_t)G * TR_TILE * WLD;                         // [rpb][16] pre-normalisation membranes of my rows
    float* cprev = cbuf + (size_t)p.rpb * TR_TILE;                        // [rpb][16] the carried membrane u_{t-1} (my rows, my neurons)
    float* recb = cprev + (size_t)p.rpb * TR_TILE;                        // [rpb][G][16] the recurrent products of this step
    float* parts = recb + (size_t)p.rpb * G * TR_TILE;                    // [RB][TR_PART]
    unsigned* hb = reinterpret_cast<unsigned*>(parts + (size_t)p.RB * TR_PART);  // [rpb][H / 4] h_{t-1} of my rows, a byte per neuron
    unsigned* sb = hb + (size_t)p.rpb * (H >> 2);                         // [rpb][4] my 16 new spikes per row, packed
    __shared__ float red[TR_THREADS / TR_TILE][TR_TILE];
    const int tid = threadIdx.x, j = tid & 15, rsub = tid >> 4;
    const int wave = tid >> 6, lj = tid & 15, kq = (tid & 63) >> 4;
    const int n0 = tile * TR_TILE, nj = n0 + j;
    const int H4 = H >> 2;
    unsigned* err = p.counters + tiles;
    for (int i = tid; i < G * TR_TILE * H; i += TR_THREADS) {
        const int gi = i / (TR_TILE * H), rem = i - gi * TR_TILE * H, jj = rem / H, k = rem - jj * H;
        wt[(gi * TR_TILE + jj) * WLD + k] = p.w_hh[((size_t)gi * H + n0 + jj) * H + k];
    }
    if (tid == 0) __hip_atomic_fetch_add(x.rbcnt + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (diagnostics: workgroups started)
    // initial state: zero (MODEL:100-106), or -- a layer call cut into chunks of frames (sfsn.h: h0 / c0) -- the spikes and the
    // membrane of the frame before this call's first
    for (int i = tid; i < p.rpb * TR_TILE; i += TR_THREADS) {
        const int r = r_lo + (i >> 4);
        cprev[i] = (p.c_prev && r < r_hi) ? p.c_prev[(size_t)r * H + n0 + (i & 15)] : 0.f;
    }
    for (int i = tid; i < p.rpb * H4; i += TR_THREADS) {
        unsigned w = 0u;
        const int rr = i / H4, r = r_lo + rr;
        if (p.h_prev && r < r_hi) {
            const tr_v4f h4 = *reinterpret_cast<const tr_v4f*>(p.h_prev + (size_t)r * H + 4 * (i - rr * H4));
            w = (h4[0] != 0.f ? 1u : 0u) | (h4[1] != 0.f ? 0x100u : 0u) | (h4[2] != 0.f ? 0x10000u : 0u) | (h4[3] != 0.f ? 0x1000000u : 0u);
        }
        hb[i] = w;
    }
    const float bf = p.bias[nj], bg = p.bias[H + nj];
    const float gam = p.use_bn ? p.bn_w[nj] : 1.f, bet = p.use_bn ? p.bn_b[nj] : 0.f;
    const bool stat_owner = rsub == 0 && rb == 0;
    float rmean = (stat_owner && p.running_mean) ? p.running_mean[nj] : 0.f, rvar = (stat_owner && p.running_mean) ? p.running_var[nj] : 0.f;
    const size_t RH = (size_t)p.R * H, RG = (size_t)p.R * G * H;
    __syncthreads();
    long long tr_t0 = (long long)wall_clock64();
    (void)tr_t0;
    for (int t = 0; t < x.T; ++t) {
        // this step's input terms of my (row, neuron) pairs do not depend on anybody: requested before the wait for h_{t-1}
        const float* z = p.z + (size_t)t * RG;
        float zfp[TR_PF], zgp[TR_PF];
#pragma unroll
        for (int i = 0; i < TR_PF; ++i) {
            const int r = r_lo + rsub + i * (TR_THREADS / TR_TILE);
            const int rc = r < r_hi ? r : r_lo;
            zfp[i] = z[(size_t)rc * G * H + nj];
            zgp[i] = z[(size_t)rc * G * H + (G - 1) * H + nj];
        }
        if (t > 0) {  // h_{t-1} of my rows, 16 neurons from each tile workgroup of my row block: tagged words of epoch t
            const unsigned* src = x.hx + ((size_t)((t - 1) & 1) * p.R + r_lo) * H4;
            if (!tr_read_tagged(hb, src, (nr * H4) >> 2, (unsigned)t, err)) return;  // (H / 4 words per row, H % 16 == 0: whole 16-byte pieces)
            TR_STAMP(1);
        }
        float *o_f = p.f + (size_t)t * RH, *o_g = p.g + (size_t)t * RH, *o_u = p.u + (size_t)t * RH, *o_s = p.spikes + (size_t)t * RH;
        float* o_x = p.xhat ? p.xhat + (size_t)t * RH : nullptr;
        // the recurrent products of my rows x my 16 neurons on the matrix pipe (v_mfma_f32_16x16x4_f32: fp32 products, fp32
        // accumulation; the scalar loop of the step kernel took 7 us per row and thread -- 22 of a step's 34 us at 48 rows per
        // block).  A = h (0 / 1) of 16 rows, B = the weight rows of my neurons; k runs as (16 s + 4 kq + i): one packed word of h
        // and one 16-byte piece of a weight row feed four instructions.  Wave w takes the row tiles w, w + 4, ...
        for (int mt = wave; mt < (p.rpb >> 4); mt += TR_THREADS / 64) {
            tr_v4f accf = {0.f, 0.f, 0.f, 0.f}, accg = {0.f, 0.f, 0.f, 0.f};
            const unsigned* hrow = hb + (size_t)(mt * 16 + lj) * H4;
            const float* wfr = wt + (size_t)lj * WLD;
            const float* wgr = wt + (size_t)((G - 1) * TR_TILE + lj) * WLD;
            for (int s4 = 0; s4 < (H >> 4); ++s4) {
                const unsigned hw = hrow[4 * s4 + kq];
                const tr_v4f wv = *reinterpret_cast<const tr_v4f*>(wfr + 16 * s4 + 4 * kq);
                tr_v4f wgv = wv;
                if (G == 2) wgv = *reinterpret_cast<const tr_v4f*>(wgr + 16 * s4 + 4 * kq);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float hk = (float)((hw >> (8 * e)) & 1u);
                    accf = __builtin_amdgcn_mfma_f32_16x16x4f32(hk, wv[e], accf, 0, 0, 0);
                    if (G == 2) accg = __builtin_amdgcn_mfma_f32_16x16x4f32(hk, wgv[e], accg, 0, 0, 0);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {  // D: lane (lj, kq) holds rows 4 kq + e of the tile, neuron lj
                recb[(size_t)((mt * 16 + 4 * kq + e) * G) * TR_TILE + lj] = accf[e];
                if (G == 2) recb[(size_t)((mt * 16 + 4 * kq + e) * G + 1) * TR_TILE + lj] = accg[e];
            }
        }
        __syncthreads();
        float sum = 0.f;
        int pi = 0;
        for (int r = r_lo + rsub; r < r_hi; r += TR_THREADS / TR_TILE, ++pi) {
            const float rf = recb[(size_t)((r - r_lo) * G) * TR_TILE + j];
            const float rg = recb[(size_t)((r - r_lo) * G + (G - 1)) * TR_TILE + j];
            float zf, zg;
            if (pi < TR_PF) {
                zf = tr_pick(zfp, pi);
                zg = tr_pick(zgp, pi);
            } else {
                zf = z[(size_t)r * G * H + nj];
                zg = z[(size_t)r * G * H + (G - 1) * H + nj];
            }
            const float pre_f = (zf + bf) + rf;
            const float pre_g = (zg + bg) + rg;
            const float f = 1.0f / (1.0f + expf(-pre_f));
            const float a = f * cprev[(r - r_lo) * TR_TILE + j];
            const float b = (1.0f - f) * pre_g;
            const float cy = a + b;
            cbuf[(r - r_lo) * TR_TILE + j] = cy;
            o_f[(size_t)r * H + nj] = f;
            o_g[(size_t)r * H + nj] = pre_g;
            sum += cy;
        }
        float mean = 0.f, invstd = 1.f, var = 0.f;
        TR_STAMP(2);
        if (p.use_bn) {
            const float tot_b = reduce16(sum, red, rsub, j);
            const float mean_b = nr > 0 ? tot_b / (float)nr : 0.f;
            float sq = 0.f;
            for (int r = r_lo + rsub; r < r_hi; r += TR_THREADS / TR_TILE) {
                const float d = cbuf[(r - r_lo) * TR_TILE + j] - mean_b;
                sq = __builtin_fmaf(d, d, sq);
            }
            const float m2_b = reduce16(sq, red, rsub, j);
            float m2 = m2_b;
            mean = mean_b;
            TR_STAMP(3);
            if (p.RB > 1) {
                if (!tr_exchange((t & 1) ? x.gran2 : p.scratch, p.counters, err, tile, rb, p.RB, (unsigned)(t + 1), mean_b, m2_b, (float)nr, rsub, j, parts)) return;
                TR_STAMP(4);
                float cnt = 0.f;
                mean = 0.f; m2 = 0.f;
                for (int b = 0; b < p.RB; ++b) {
                    const float nb = parts[b * TR_PART], mb = parts[b * TR_PART + 2 + j], qb = parts[b * TR_PART + 18 + j];
                    if (nb > 0.f) {
                        const float tot = cnt + nb, delta = mb - mean;
                        mean = mean + delta * (nb / tot);
                        m2 = m2 + qb + delta * delta * (cnt * nb / tot);
                        cnt = tot;
                    }
                }
            }
            var = m2 / (float)p.R;
            invstd = 1.0f / sqrtf(var + p.eps);
        } else {
            __syncthreads();
        }
        for (int r = r_lo + rsub; r < r_hi; r += TR_THREADS / TR_TILE) {
            const float cy = cbuf[(r - r_lo) * TR_TILE + j];
            float uu = cy;
            if (p.use_bn) {
                const float xh = (cy - mean) * invstd;
                uu = xh * gam + bet;
                o_x[(size_t)r * H + nj] = xh;
            }
            o_u[(size_t)r * H + nj] = uu;
            const bool fire = uu >= 0.f;
            o_s[(size_t)r * H + nj] = fire ? 1.f : 0.f;
            cprev[(r - r_lo) * TR_TILE + j] = uu;
            reinterpret_cast<unsigned char*>(sb)[(r - r_lo) * TR_TILE + j] = fire ? 1 : 0;
        }
        if (p.use_bn && stat_owner) {
            p.invstd[(size_t)t * H + nj] = invstd;
            if (p.running_mean) {
                const float unb = p.R > 1 ? var * ((float)p.R / (float)(p.R - 1)) : var;
                // momentum < 0: nn.BatchNorm1d(momentum=None), the cumulative moving average -- factor 1 / num_batches_tracked AFTER this
                // step's increment (torch/nn/modules/batchnorm.py: exponential_average_factor = 1.0 / float(num_batches_tracked));
                // the caller passes -(num_batches_tracked before the call + 1)
                const float mom = p.momentum >= 0.0f ? p.momentum : 1.0f / (-p.momentum + (float)t);
                rmean = (1.0f - mom) * rmean + mom * mean;
                rvar = (1.0f - mom) * rvar + mom * unb;
            }
        }
        __syncthreads();
        TR_STAMP(5);
        if (t + 1 < x.T) {  // my 16 spikes of every row -> the slot of step t: four tagged words per row, write-through
            unsigned* dst = x.hx + ((size_t)(t & 1) * p.R + r_lo) * H4 + tile * 4;
            const unsigned tag = (unsigned)(t + 1) << 4;
            for (int i = tid; i < nr * 4; i += TR_THREADS) {
                const unsigned w = sb[i];
                __hip_atomic_store(dst + (size_t)(i >> 2) * H4 + (i & 3), tag | (w & 1u) | ((w >> 7) & 2u) | ((w >> 14) & 4u) | ((w >> 21) & 8u),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();  // (sb is rewritten by the next step's output loop)
            TR_STAMP(6);
        }
    }
    if (p.use_bn && stat_owner && p.running_mean) {
        p.running_mean[nj] = rmean;
        p.running_var[nj] = rvar;
    }
}

// backward: steps t = T-1 ... 0 in one launch.  p.dh_up / u / xhat / f / g / invstd / d_gates / d_z point at the [T]-leading tensors;
// c_prev of step t is u[t-1] (zero at t = 0); the carried membrane gradient stays in LDS.
__global__ __launch_bounds__(TR_THREADS) void gsn_train_seq_fwd_kernel(const TrainSeqFwdMulti m) {
    int g = 0;
    while (g + 1 < m.n && (int)blockIdx.x >= m.wg_end[g]) ++g;
    const int li = (int)blockIdx.x - (g ? m.wg_end[g - 1] : 0), tiles = m.p[g].H / TR_TILE;
    train_seq_fwd_body(m.p[g], m.x[g], li % tiles, li / tiles, tiles);
}

__device__ __forceinline__ void train_seq_bwd_body(const TrainBwdParams& p, const TrainSeqExtra& x, const int tile, const int rb, const int tiles) {
    extern __shared__ __attribute__((aligned(16))) char tr_smem[];
    __shared__ float red[TR_THREADS / TR_TILE][TR_TILE];
    const int H = p.H, G = p.shared ? 1 : 2, GH = G * p.H;
    const int r_lo = rb * p.rpb, r_hi = (r_lo + p.rpb < p.R) ? r_lo + p.rpb : p.R, nr = r_hi - r_lo;
    const int tid = threadIdx.x, j = tid & 15, rsub = tid >> 4;
    const int nj = tile * TR_TILE + j;
    float* dbuf = reinterpret_cast<float*>(tr_smem);   // [rpb][16] du of my rows
    float* dcn = dbuf + (size_t)p.rpb * TR_TILE;        // [rpb][16] dL/dc carried from step t+1 (my rows, my neurons)
    const int DLD = GH + 4;                             // (row stride of the two matrix operands in LDS: 16-byte aligned, off the bank period)
    float* recb = dcn + (size_t)p.rpb * TR_TILE;        // [rpb][16] dL/dh_t through step t+1's recurrent product
    float* parts = recb + (size_t)p.rpb * TR_TILE;      // [RB][TR_PART]
    float* wcol = parts + (size_t)p.RB * TR_PART;       // [16][DLD]: the columns of W_hh that feed my 16 neurons of h_t, one row per neuron
    float* dzb = wcol + (size_t)TR_TILE * DLD;          // [rpb][DLD]: d_z of step t+1, my rows
    unsigned* err = p.counters + tiles;
    const int wave = tid >> 6, lj = tid & 15, kq = (tid & 63) >> 4;
    for (int i = tid; i < GH * TR_TILE; i += TR_THREADS) {
        const int nn = i / TR_TILE, jj = i - nn * TR_TILE;
        wcol[(size_t)jj * DLD + nn] = p.w_hh[(size_t)nn * H + tile * TR_TILE + jj];
    }
    for (int i = tid; i < p.rpb * DLD; i += TR_THREADS) dzb[i] = 0.f;
    if (tid == 0) __hip_atomic_fetch_add(x.rbcnt + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (diagnostics: workgroups started)
    // A layer call cut into chunks of frames (sfsn.h: dc_in / dc_out / has_prev): `later` = the frames behind this call's last one have
    // been processed by an earlier launch -- their d_z sits behind this call's in the same tensor (frame T: plain loads, the launch
    // boundary ordered it) and dL/dc carried out of them is dc_next [R][H]
    const bool later = p.dc_next != nullptr;
    for (int i = tid; i < p.rpb * TR_TILE; i += TR_THREADS) {
        const int r = r_lo + (i >> 4);
        dcn[i] = (later && r < r_hi) ? p.dc_next[(size_t)r * H + tile * TR_TILE + (i & 15)] : 0.f;
    }
    const size_t RH = (size_t)p.R * H, RG = (size_t)p.R * GH;
    float* dzs = p.shared ? p.d_z : p.d_gates;  // the gradient of the (shared or per-gate) products: [T][R][G*H]
    const float gam = p.use_bn ? p.bn_w[nj] : 1.f;
    float acc_w = 0.f, acc_b = 0.f;
    const bool stat_owner = rsub == 0 && rb == 0;
    if (p.use_bn && stat_owner) { acc_w = p.d_bn_w[nj]; acc_b = p.d_bn_b[nj]; }
    __syncthreads();
    long long tr_t0 = (long long)wall_clock64();
    (void)tr_t0;
    for (int t = x.T - 1; t >= 0; --t) {
        const int s = x.T - 1 - t;
        const float *i_u = p.u + (size_t)t * RH, *i_f = p.f + (size_t)t * RH, *i_g = p.g + (size_t)t * RH, *i_up = p.dh_up + (size_t)t * RH;
        const float* i_x = p.xhat ? p.xhat + (size_t)t * RH : nullptr;
        const float* i_cp = t ? p.u + (size_t)(t - 1) * RH : p.c_prev;  // (c_prev: the membrane of the frame before a chunk's first, or null)
        // this step's own inputs (forward-saved tensors, the upstream gradient) do not depend on anybody: requested before the wait
        float upp[TR_PF], uup[TR_PF], xhp[TR_PF];
#pragma unroll
        for (int i = 0; i < TR_PF; ++i) {
            const int r = r_lo + rsub + i * (TR_THREADS / TR_TILE);
            const size_t o = (size_t)(r < r_hi ? r : r_lo) * H + nj;
            upp[i] = i_up[o];
            uup[i] = i_u[o];
            xhp[i] = p.use_bn ? i_x[o] : 0.f;
        }
        if (s > 0 || later) {  // d_z of step t+1 of my rows: every tile workgroup of my row block has published it (or an earlier launch wrote it)
            if (s > 0 && !tr_wait_counter(x.rbcnt + rb, (unsigned)s * (unsigned)tiles, err)) return;
            TR_STAMP(0);
            const float* src = dzs + (size_t)(t + 1) * RG + (size_t)r_lo * GH;
            tr_copy16_sc1(dzb, src, (nr * GH) >> 2, GH >> 2, DLD >> 2);
            __syncthreads();
            TR_STAMP(1);
            // sum_n dz_{t+1}[r][n] W_hh[n][my neuron] on the matrix pipe (fp32 products and accumulation; 26 of the scalar step's 44 us)
            for (int mt = wave; mt < (p.rpb >> 4); mt += TR_THREADS / 64) {
                tr_v4f acc = {0.f, 0.f, 0.f, 0.f};
                const float* dzr = dzb + (size_t)(mt * 16 + lj) * DLD;
                const float* wr = wcol + (size_t)lj * DLD;
                for (int s4 = 0; s4 < (GH >> 4); ++s4) {
                    const tr_v4f av = *reinterpret_cast<const tr_v4f*>(dzr + 16 * s4 + 4 * kq);
                    const tr_v4f bv = *reinterpret_cast<const tr_v4f*>(wr + 16 * s4 + 4 * kq);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) recb[(size_t)(mt * 16 + 4 * kq + e) * TR_TILE + lj] = acc[e];
            }
            __syncthreads();
        }
        float s1 = 0.f, s2 = 0.f;
        int pi = 0;
        for (int r = r_lo + rsub; r < r_hi; r += TR_THREADS / TR_TILE, ++pi) {
            const size_t o = (size_t)r * H + nj;
            float dh = 0.f;
            dh += pi < TR_PF ? tr_pick(upp, pi) : i_up[o];
            if (s > 0 || later) dh += recb[(size_t)(r - r_lo) * TR_TILE + j];  // dL/dh_t through step t+1's recurrent product
            const float uu = pi < TR_PF ? tr_pick(uup, pi) : i_u[o];
            const float tri = fmaxf(0.f, 1.0f - fabsf(uu));
            float du = dh * tri;
            if (s > 0 || later) du += dcn[(r - r_lo) * TR_TILE + j];
            dbuf[(r - r_lo) * TR_TILE + j] = du;
            if (p.use_bn) {
                s1 += du;
                s2 = __builtin_fmaf(du, pi < TR_PF ? tr_pick(xhp, pi) : i_x[o], s2);
            }
        }
        // ... and what the last loop of the step reads (forget gate, cell-gate pre-activation, previous membrane): under the exchange
        float fp[TR_PF], gp[TR_PF], cpp[TR_PF];
#pragma unroll
        for (int i = 0; i < TR_PF; ++i) {
            const int r = r_lo + rsub + i * (TR_THREADS / TR_TILE);
            const size_t o = (size_t)(r < r_hi ? r : r_lo) * H + nj;
            fp[i] = i_f[o];
            gp[i] = i_g[o];
            cpp[i] = i_cp ? i_cp[o] : 0.f;
        }
        float k1 = 0.f, k2 = 0.f, scale = 1.f;
        TR_STAMP(2);
        if (p.use_bn) {
            k1 = reduce16(s1, red, rsub, j);
            k2 = reduce16(s2, red, rsub, j);
            TR_STAMP(3);
            if (p.RB > 1) {
                if (!tr_exchange((s & 1) ? x.gran2 : p.scratch, p.counters, err, tile, rb, p.RB, (unsigned)(s + 1), k1, k2, (float)nr, rsub, j, parts)) return;
                TR_STAMP(4);
                k1 = 0.f; k2 = 0.f;
                for (int b = 0; b < p.RB; ++b) {
                    k1 += parts[b * TR_PART + 2 + j];
                    k2 += parts[b * TR_PART + 18 + j];
                }
            }
            scale = gam * p.invstd[(size_t)t * H + nj] / (float)p.R;
            if (stat_owner) { acc_w += k2; acc_b += k1; }
        } else {
            __syncthreads();
        }
        float* o_dg = p.d_gates + (size_t)t * p.R * 2 * H;
        float* o_dz = p.shared ? p.d_z + (size_t)t * RH : nullptr;
        pi = 0;
        for (int r = r_lo + rsub; r < r_hi; r += TR_THREADS / TR_TILE, ++pi) {
            const size_t o = (size_t)r * H + nj;
            const float du = dbuf[(r - r_lo) * TR_TILE + j];
            const float xh = p.use_bn ? (pi < TR_PF ? tr_pick(xhp, pi) : i_x[o]) : 0.f;
            const float dcy = p.use_bn ? scale * ((float)p.R * du - k1 - xh * k2) : du;
            const float f = pi < TR_PF ? tr_pick(fp, pi) : i_f[o], g = pi < TR_PF ? tr_pick(gp, pi) : i_g[o];
            const float cp = pi < TR_PF ? tr_pick(cpp, pi) : (i_cp ? i_cp[o] : 0.f);
            const float df = dcy * (cp - g);
            const float dpf = df * f * (1.0f - f);
            const float dpg = dcy * (1.0f - f);
            dcn[(r - r_lo) * TR_TILE + j] = dcy * f;
            // what the next step's workgroups read goes out write-through (the per-gate tensor when the products are not shared)
            if (p.shared) {
                o_dg[(size_t)r * 2 * H + nj] = dpf;
                o_dg[(size_t)r * 2 * H + H + nj] = dpg;
                __hip_atomic_store(reinterpret_cast<unsigned*>(o_dz + o), __float_as_uint(dpf + dpg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(reinterpret_cast<unsigned*>(o_dg + (size_t)r * 2 * H + nj), __float_as_uint(dpf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(reinterpret_cast<unsigned*>(o_dg + (size_t)r * 2 * H + H + nj), __float_as_uint(dpg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        TR_STAMP(5);
        if (t > 0) tr_publish(x.rbcnt + rb);
        else __syncthreads();
        TR_STAMP(6);
    }
    if (p.use_bn && stat_owner) { p.d_bn_w[nj] = acc_w; p.d_bn_b[nj] = acc_b; }
    if (p.dc_prev) {  // dL/dc carried out of this call's first frame: what the launch of the frames before it starts from
        for (int r = r_lo + rsub; r < r_hi; r += TR_THREADS / TR_TILE) p.dc_prev[(size_t)r * H + nj] = dcn[(r - r_lo) * TR_TILE + j];
    }
}

__global__ __launch_bounds__(TR_THREADS) void gsn_train_seq_bwd_kernel(const TrainSeqBwdMulti m) {
    int g = 0;
    while (g + 1 < m.n && (int)blockIdx.x >= m.wg_end[g]) ++g;
    const int li = (int)blockIdx.x - (g ? m.wg_end[g - 1] : 0), tiles = m.p[g].H / TR_TILE;
    train_seq_bwd_body(m.p[g], m.x[g], li % tiles, li / tiles, tiles);
}

// ---- a whole layer call: one launch for all T steps (the kernels above).  Tensors as for the step entries with a leading [T]; zero
// initial state.  `scratch`: sfsn_train_seq_scratch_bytes(R, H) bytes, ZEROED by the caller before the call (the packed-spike
// slots, the row blocks' publish counters, then the step entries' layout: granules, error word in the last four words).
// -DSFSN_EXPERIMENTS and SFSN_TRAIN_PROF=1: where a step of workgroup (0, 0) spends its time (synchronises after every layer call)
#ifdef SFSN_EXPERIMENTS
#include <cstdio>
#include <cstdlib>
static long long* seq_prof_begin() {
    static long long* buf = nullptr;
    if (!getenv("SFSN_TRAIN_PROF")) return nullptr;
    if (!buf && hipMalloc(&buf, 8 * sizeof(long long)) != hipSuccess) return nullptr;
    (void)hipMemset(buf, 0, 8 * sizeof(long long));
    return buf;
}
static void seq_prof_end(long long* buf, const char* what, int T, int R, int H, int RB, hipStream_t st) {
    if (!buf) return;
    long long h[8];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "[train prof] %s T=%d R=%d H=%d RB=%d us/step: wait %.2f load %.2f product %.2f reduce %.2f exchange %.2f output %.2f publish %.2f\n", what, T, R, H, RB,
            h[0] / 100.0 / T, h[1] / 100.0 / T, h[2] / 100.0 / T, h[3] / 100.0 / T, h[4] / 100.0 / T, h[5] / 100.0 / T, h[6] / 100.0 / T);
}
#else
static long long* seq_prof_begin() { return nullptr; }
static void seq_prof_end(long long*, const char*, int, int, int, int, hipStream_t) {}
#endif
static size_t seq_gran_bytes(int H) { return (size_t)(H / TR_TILE) * 16 * TR_PARTG * sizeof(unsigned long long); }
static size_t seq_hx_bytes(int R, int H) { return (((size_t)2 * R * (H / 4) + 32) * sizeof(unsigned) + 15) & ~(size_t)15; }  // (+ [16] started, [17] finished workgroups)
static size_t seq_head_bytes(int R, int H) { return seq_hx_bytes(R, H) + seq_gran_bytes(H); }  // [hx | rbcnt | odd-step granules]

extern "C" size_t sfsn_train_seq_scratch_bytes(int R, int H) {
    if (R <= 0 || H <= 0 || H % TR_TILE) return 0;
    return seq_head_bytes(R, H) + sfsn_train_scratch_bytes(H);
}

static size_t seq_lds_fwd(int G, int H, int RB, int rpb) {
    return ((size_t)G * TR_TILE * (H + 4) + (size_t)(2 + G) * rpb * TR_TILE + (size_t)RB * TR_PART) * sizeof(float) + (size_t)rpb * H + (size_t)rpb * TR_TILE;
}
static size_t seq_lds_bwd(int G, int H, int RB, int rpb) {
    return ((size_t)3 * rpb * TR_TILE + (size_t)RB * TR_PART + (size_t)(G * H + 4) * (TR_TILE + rpb)) * sizeof(float);
}

// all workgroups of a launch must be resident together: blocks per compute unit at this LDS size x compute units
static int seq_slots(const void* kern, size_t lds) {
    // (asked before every launch: the last few answers are remembered per (device, kernel, LDS size) -- the occupancy query costs tens of
    //  microseconds, a training step makes sixteen launches)
    struct Memo { int dev; const void* kern; size_t lds; int slots; };
    static thread_local Memo memo[48];
    static thread_local int n_memo = 0;
    int dev = 0, cus = 0, per = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (lds > 64 * 1024 && hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    for (int i = 0; i < n_memo; ++i)
        if (memo[i].dev == dev && memo[i].kern == kern && memo[i].lds == lds) return memo[i].slots;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kern, TR_THREADS, lds) != hipSuccess) return -1;
    memo[n_memo < 48 ? n_memo++ : 47] = Memo{dev, kern, lds, per * cus};
    return per * cus;
}

// Geometry of a multi-call launch in ONE direction (dir 0 forward, 1 backward): rows per block / row blocks of every call, the launch's
// LDS, its workgroups.  Every call starts from the single call's geometry (~160 workgroups); when the launch would not be resident
// (more calls side by side than the chip has slots: the layers of several stacks in one grid, GSNStackTrainFn) all calls take fewer,
// larger row blocks -- the step time hardly depends on the rows per block (220 / 160 / 110 / 70 workgroups per call measured 9.4 / 8.3 /
// 8.1 / 8.3 ms forward) -- until it is, or until a block's rows no longer fit the LDS.  A chunked call's state travels through global
// memory, so the geometry may differ from launch to launch and between the directions.
static int seq_dir_geometry(const int* R, int n, int H, int shared, int dir, int* RBs, int* rpbs, int* wgs, size_t* lds) {
    if (!R || n <= 0 || n > TR_MAXG || H <= 0) return SFSN_EINVAL;
    if (H % TR_TILE != 0 || H > SFSN_MAX_HIDDEN) return SFSN_EUNSUPPORTED;
    const int G = shared ? 1 : 2, tiles = H / TR_TILE;
    for (int i = 0; i < n; ++i)
        if (R[i] <= 0) return SFSN_EINVAL;
    const void* kern = dir ? reinterpret_cast<const void*>(gsn_train_seq_bwd_kernel) : reinterpret_cast<const void*>(gsn_train_seq_fwd_kernel);
    static const int targets[] = {0, 128, 96, 80, 64, 48, 40, 32, 24, 16};
    for (int target : targets) {
        int total = 0;
        size_t l = 0;
        bool fits_lds = true;
        for (int i = 0; i < n; ++i) {
            train_geometry(R[i], H, G, &RBs[i], &rpbs[i], target);
            if (RBs[i] > 16) return SFSN_EUNSUPPORTED;
            const size_t li = dir ? seq_lds_bwd(G, H, RBs[i], rpbs[i]) : seq_lds_fwd(G, H, RBs[i], rpbs[i]);
            if (li > 150 * 1024) fits_lds = false;
            l = li > l ? li : l;
            total += tiles * RBs[i];
        }
        if (!fits_lds) return SFSN_EUNSUPPORTED;  // (train_geometry already took as many row blocks as the LDS needs: fewer cannot fit either)
        const int slots = seq_slots(kern, l);
        if (slots < 0) return SFSN_EHIP;
        *wgs = total; *lds = l;
        if (total <= slots) return SFSN_OK;
    }
    return SFSN_EUNSUPPORTED;
}

// Will the per-step launches (sfsn_gsn_train_step_fwd / _bwd: round 3's kernels, what training.py falls back to when the one-launch
// layer call cannot hold its workgroups resident, and what the eval-mode-BatchNorm path always runs) take (R, H)?  Their own LDS
// formulas (smaller than the one-launch kernels': no carried membrane, no packed spike rows), and -- with more than one row block --
// the same residency condition per step launch (the row blocks of a step exchange their partial sums inside the launch).
extern "C" int sfsn_gsn_train_step_check(int R, int H, int shared) {
    if (R <= 0 || H <= 0) return SFSN_EINVAL;
    if (H % TR_TILE != 0 || H > SFSN_MAX_HIDDEN) return SFSN_EUNSUPPORTED;
    int RB, rpb;
    const int G = shared ? 1 : 2, tiles = H / TR_TILE;
    train_geometry(R, H, G, &RB, &rpb);
    if (RB > 16) return SFSN_EUNSUPPORTED;
    const size_t lds_f = ((size_t)G * TR_TILE * (H + 1) + (size_t)rpb * TR_TILE + (size_t)RB * TR_PART) * sizeof(float) + (size_t)rpb * H;
    const size_t lds_b = ((size_t)rpb * TR_TILE + (size_t)RB * TR_PART + (size_t)G * H * (TR_TILE + rpb)) * sizeof(float);
    if (lds_f > 150 * 1024 || lds_b > 150 * 1024) return SFSN_EUNSUPPORTED;
    if (RB > 1) {
        const int sf = seq_slots(reinterpret_cast<const void*>(gsn_train_step_fwd_kernel), lds_f), sb = seq_slots(reinterpret_cast<const void*>(gsn_train_step_bwd_kernel), lds_b);
        if (sf < 0 || sb < 0) return SFSN_EHIP;
        if (tiles * RB > sf || tiles * RB > sb) return SFSN_EUNSUPPORTED;
    }
    return SFSN_OK;
}

// SFSN_OK when ONE launch per direction can hold the workgroups of all n layer calls (rows R[i], same H / gate sharing) resident
// together; SFSN_EUNSUPPORTED otherwise (issue the calls one after the other, or in smaller sets).  Needs the device.
extern "C" int sfsn_gsn_train_multi_check(const int* R, int n, int H, int shared) {
    int RBs[TR_MAXG], rpbs[TR_MAXG], wgs;
    size_t lds;
    if (n > TR_MAXG) return SFSN_EINVAL;
    const int rc = seq_dir_geometry(R, n, H, shared, 0, RBs, rpbs, &wgs, &lds);
    if (rc != SFSN_OK) return rc;
    return seq_dir_geometry(R, n, H, shared, 1, RBs, rpbs, &wgs, &lds);
}

extern "C" int sfsn_gsn_train_seq_fwd_multi(const SfsnTrainSeqFwd* c, int n, int T, int H, int shared, void* stream) {
    if (!c || n <= 0 || n > TR_MAXG || T <= 0 || H <= 0) return SFSN_EINVAL;
    int Rs[TR_MAXG];
    for (int i = 0; i < n; ++i) {
        if (!c[i].z || !c[i].w_hh || !c[i].bias || !c[i].spikes || !c[i].u || !c[i].f || !c[i].g || !c[i].scratch) return SFSN_EINVAL;
        if (c[i].bn_w && (!c[i].bn_b || !c[i].xhat || !c[i].invstd)) return SFSN_EINVAL;
        if ((c[i].bn_w != nullptr) != (c[0].bn_w != nullptr)) return SFSN_EINVAL;
        if ((c[i].running_mean == nullptr) != (c[i].running_var == nullptr)) return SFSN_EINVAL;
        if ((c[i].h0 == nullptr) != (c[i].c0 == nullptr) || (reinterpret_cast<uintptr_t>(c[i].h0) & 15)) return SFSN_EINVAL;
        Rs[i] = c[i].R;
    }
    int wgs, RBs[TR_MAXG], rpbs[TR_MAXG];
    size_t lds;
    int rc = seq_dir_geometry(Rs, n, H, shared, 0, RBs, rpbs, &wgs, &lds);
    if (rc != SFSN_OK) return rc;
    auto kern = gsn_train_seq_fwd_kernel;
    const int tiles = H / TR_TILE;
    TrainSeqFwdMulti m;
    m.n = n;
    int end = 0;
    for (int i = 0; i < n; ++i) {
        TrainFwdParams& p = m.p[i];
        TrainSeqExtra& x = m.x[i];
        const int R = c[i].R;
        p.RB = RBs[i]; p.rpb = rpbs[i];
        char* base = static_cast<char*>(c[i].scratch);
        float* step_scr = reinterpret_cast<float*>(base + seq_head_bytes(R, H));
        p.z = c[i].z; p.w_hh = c[i].w_hh; p.bias = c[i].bias; p.h_prev = c[i].h0; p.c_prev = c[i].c0; p.bn_w = c[i].bn_w; p.bn_b = c[i].bn_b;
        p.running_mean = c[i].running_mean; p.running_var = c[i].running_var; p.spikes = c[i].spikes; p.u = c[i].u; p.xhat = c[i].xhat;
        p.f = c[i].f; p.g = c[i].g; p.invstd = c[i].invstd; p.momentum = c[i].momentum; p.eps = c[i].eps; p.R = R; p.H = H; p.shared = shared;
        p.use_bn = c[i].bn_w != nullptr; p.epoch = 0; p.scratch = step_scr;
        p.counters = reinterpret_cast<unsigned*>(step_scr + (size_t)tiles * 16 * TR_PARTG * 2);
        x.T = c[i].T > 0 ? c[i].T : T;
        x.hx = reinterpret_cast<unsigned*>(base);
        x.rbcnt = x.hx + (size_t)2 * R * (H / 4);
        x.gran2 = reinterpret_cast<float*>(base + seq_hx_bytes(R, H));
        x.prof = (n == 1) ? seq_prof_begin() : nullptr;
        end += tiles * p.RB;
        m.wg_end[i] = end;
    }
    hipLaunchKernelGGL(kern, dim3(end), dim3(TR_THREADS), lds, static_cast<hipStream_t>(stream), m);
    if (n == 1) seq_prof_end(m.x[0].prof, "fwd", T, c[0].R, H, m.p[0].RB, static_cast<hipStream_t>(stream));
    return hip_ok_tr(hipGetLastError());
}

extern "C" int sfsn_gsn_train_seq_bwd_multi(const SfsnTrainSeqBwd* c, int n, int T, int H, int shared, void* stream) {
    if (!c || n <= 0 || n > TR_MAXG || T <= 0 || H <= 0) return SFSN_EINVAL;
    int Rs[TR_MAXG];
    for (int i = 0; i < n; ++i) {
        if (!c[i].w_hh || !c[i].dh_up || !c[i].u || !c[i].f || !c[i].g || !c[i].d_gates || (shared && !c[i].d_z) || !c[i].scratch) return SFSN_EINVAL;
        if (c[i].bn_w && (!c[i].xhat || !c[i].invstd || !c[i].d_bn_w || !c[i].d_bn_b)) return SFSN_EINVAL;
        if ((c[i].bn_w != nullptr) != (c[0].bn_w != nullptr)) return SFSN_EINVAL;
        Rs[i] = c[i].R;
    }
    int wgs, RBs[TR_MAXG], rpbs[TR_MAXG];
    size_t lds;
    int rc = seq_dir_geometry(Rs, n, H, shared, 1, RBs, rpbs, &wgs, &lds);
    if (rc != SFSN_OK) return rc;
    auto kern = gsn_train_seq_bwd_kernel;
    const int tiles = H / TR_TILE;
    TrainSeqBwdMulti m;
    m.n = n;
    int end = 0;
    for (int i = 0; i < n; ++i) {
        TrainBwdParams& p = m.p[i];
        TrainSeqExtra& x = m.x[i];
        const int R = c[i].R;
        p.RB = RBs[i]; p.rpb = rpbs[i];
        char* base = static_cast<char*>(c[i].scratch);
        float* step_scr = reinterpret_cast<float*>(base + seq_head_bytes(R, H));
        p.dz_next = nullptr; p.w_hh = c[i].w_hh; p.dh_up = c[i].dh_up; p.dh_rec = nullptr; p.dc_next = c[i].dc_in; p.u = c[i].u; p.xhat = c[i].xhat;
        p.f = c[i].f; p.g = c[i].g; p.c_prev = c[i].has_prev ? c[i].u - (size_t)c[i].R * H : nullptr; p.invstd = c[i].invstd; p.bn_w = c[i].bn_w;
        p.d_gates = c[i].d_gates; p.d_z = c[i].d_z; p.dc_prev = c[i].dc_out; p.d_bn_w = c[i].d_bn_w; p.d_bn_b = c[i].d_bn_b;
        p.R = R; p.H = H; p.shared = shared; p.use_bn = c[i].bn_w != nullptr; p.epoch = 0; p.scratch = step_scr;
        p.counters = reinterpret_cast<unsigned*>(step_scr + (size_t)tiles * 16 * TR_PARTG * 2);
        x.T = c[i].T > 0 ? c[i].T : T;
        x.hx = reinterpret_cast<unsigned*>(base);
        x.rbcnt = x.hx + (size_t)2 * R * (H / 4);
        x.gran2 = reinterpret_cast<float*>(base + seq_hx_bytes(R, H));
        x.prof = (n == 1) ? seq_prof_begin() : nullptr;
        end += tiles * p.RB;
        m.wg_end[i] = end;
    }
    hipLaunchKernelGGL(kern, dim3(end), dim3(TR_THREADS), lds, static_cast<hipStream_t>(stream), m);
    if (n == 1) seq_prof_end(m.x[0].prof, "bwd", T, c[0].R, H, m.p[0].RB, static_cast<hipStream_t>(stream));
    return hip_ok_tr(hipGetLastError());
}

// one layer call (= the multi entries with n = 1, ABI-13 call shape)
extern "C" int sfsn_gsn_train_seq_fwd(const float* z, const float* w_hh, const float* bias, const float* bn_w, const float* bn_b,
                                      float* running_mean, float* running_var, float momentum, float eps, int T, int R, int H,
                                      int shared, const float* /*zero: unused since ABI 14*/, float* spikes, float* u, float* xhat, float* f, float* g,
                                      float* invstd, void* scratch, void* stream) {
    SfsnTrainSeqFwd c;
    c.h0 = nullptr; c.c0 = nullptr; c.T = 0;
    c.z = z; c.w_hh = w_hh; c.bias = bias; c.bn_w = bn_w; c.bn_b = bn_b; c.running_mean = running_mean; c.running_var = running_var;
    c.momentum = momentum; c.eps = eps; c.R = R; c.spikes = spikes; c.u = u; c.xhat = xhat; c.f = f; c.g = g; c.invstd = invstd; c.scratch = scratch;
    return sfsn_gsn_train_seq_fwd_multi(&c, 1, T, H, shared, stream);
}

// d_gates [T][R][2H], d_z [T][R][H] (shared) or NULL; d_bn_w / d_bn_b are accumulated into (+=) as by the step entry
extern "C" int sfsn_gsn_train_seq_bwd(const float* w_hh, const float* dh_up, const float* u, const float* xhat, const float* f,
                                      const float* g, const float* invstd, const float* bn_w, int T, int R, int H, int shared,
                                      const float* /*zero: unused since ABI 14*/, float* d_gates, float* d_z, float* /*dc_work: unused since ABI 14*/,
                                      float* d_bn_w, float* d_bn_b, void* scratch, void* stream) {
    SfsnTrainSeqBwd c;
    c.dc_in = nullptr; c.dc_out = nullptr; c.has_prev = 0; c.T = 0;
    c.w_hh = w_hh; c.dh_up = dh_up; c.u = u; c.xhat = xhat; c.f = f; c.g = g; c.invstd = invstd; c.bn_w = bn_w; c.R = R;
    c.d_gates = d_gates; c.d_z = d_z; c.d_bn_w = d_bn_w; c.d_bn_b = d_bn_b; c.scratch = scratch;
    return sfsn_gsn_train_seq_bwd_multi(&c, 1, T, H, shared, stream);
}
