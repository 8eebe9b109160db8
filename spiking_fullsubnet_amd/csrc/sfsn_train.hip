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

#include "sfsn.h"

#define TR_THREADS 256
#define TR_TILE 16

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
            if (spins > 2000000u) {  // ~1 s: a launch whose blocks are not all resident must not hang the device
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

// rows per workgroup: enough row blocks to put ~all compute units to work, at least 16 rows each; every block of a launch must be
// resident (they wait for each other): the grid stays below ~220 blocks of 256 threads with < 40 KB of LDS
static void train_geometry(int R, int H, int* RB, int* rpb) {
    const int tiles = H / TR_TILE;
    int rb = 220 / tiles;
    if (rb < 1) rb = 1;
    int per = (R + rb - 1) / rb;
    if (per < 16) per = 16;
    per = (per + 15) & ~15;
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
    train_geometry(R, H, &RB, &rpb);
    if (RB > 16) return SFSN_EUNSUPPORTED;
    const int G = shared ? 1 : 2;
    const size_t lds_f = ((size_t)G * TR_TILE * (H + 1) + (size_t)rpb * TR_TILE + (size_t)RB * TR_PART) * sizeof(float) + (size_t)rpb * H;
    const size_t lds_b = ((size_t)rpb * TR_TILE + (size_t)RB * TR_PART + (size_t)G * H * (TR_TILE + rpb)) * sizeof(float);
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
    train_geometry(R, H, &p.RB, &p.rpb);
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
    train_geometry(R, H, &p.RB, &p.rpb);
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

// ---- a whole layer call: the T step launches enqueued from here (the interpreter's share of a step launch was ~10 us of the
// 15 it took at small batches).  Tensors as for the step entries with a leading [T]; zero initial state (`zero` = [R][H] zeros).
extern "C" int sfsn_gsn_train_seq_fwd(const float* z, const float* w_hh, const float* bias, const float* bn_w, const float* bn_b,
                                      float* running_mean, float* running_var, float momentum, float eps, int T, int R, int H,
                                      int shared, const float* zero, float* spikes, float* u, float* xhat, float* f, float* g,
                                      float* invstd, void* scratch, void* stream) {
    if (T <= 0 || !zero) return SFSN_EINVAL;
    const size_t RH = (size_t)R * H, RG = (size_t)R * (shared ? 1 : 2) * H;
    for (int t = 0; t < T; ++t) {
        const int rc = sfsn_gsn_train_step_fwd(z + t * RG, w_hh, bias, t ? spikes + (t - 1) * RH : zero, t ? u + (t - 1) * RH : zero, bn_w, bn_b,
                                               running_mean, running_var, momentum, eps, R, H, shared, spikes + t * RH, u + t * RH,
                                               xhat ? xhat + t * RH : nullptr, f + t * RH, g + t * RH, invstd ? invstd + (size_t)t * H : nullptr,
                                               scratch, (unsigned)(t + 1), stream);
        if (rc != SFSN_OK) return rc;
    }
    return SFSN_OK;
}

// d_gates [T][R][2H], d_z [T][R][H] (shared) or NULL; dc_work [2][R][H] scratch for the carried membrane gradient
extern "C" int sfsn_gsn_train_seq_bwd(const float* w_hh, const float* dh_up, const float* u, const float* xhat, const float* f,
                                      const float* g, const float* invstd, const float* bn_w, int T, int R, int H, int shared,
                                      const float* zero, float* d_gates, float* d_z, float* dc_work, float* d_bn_w, float* d_bn_b,
                                      void* scratch, void* stream) {
    if (T <= 0 || !zero || !dc_work || !d_gates || (shared && !d_z)) return SFSN_EINVAL;
    const size_t RH = (size_t)R * H, RG = (size_t)R * (shared ? 1 : 2) * H;
    float* dzs = shared ? d_z : d_gates;  // the gradient of the (shared or per-gate) products: [T][R][G*H]
    for (int t = T - 1; t >= 0; --t) {
        const bool last = t == T - 1;
        const int rc = sfsn_gsn_train_step_bwd(last ? nullptr : dzs + (t + 1) * RG, w_hh, dh_up + t * RH, nullptr,
                                               last ? nullptr : dc_work + ((t + 1) & 1) * RH, u + t * RH, xhat ? xhat + t * RH : nullptr,
                                               f + t * RH, g + t * RH, t ? u + (t - 1) * RH : zero, invstd ? invstd + (size_t)t * H : nullptr, bn_w,
                                               R, H, shared, d_gates + (size_t)t * R * 2 * H, shared ? d_z + t * RH : nullptr,
                                               dc_work + (t & 1) * RH, d_bn_w, d_bn_b, scratch, (unsigned)(T - t), stream);
        if (rc != SFSN_OK) return rc;
    }
    return SFSN_OK;
}
