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
};

__global__ __launch_bounds__(TR_THREADS) void gsn_train_step_fwd_kernel(const TrainFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char tr_smem[];
    const int H = p.H, R = p.R, G = p.shared ? 1 : 2;
    float* wt = reinterpret_cast<float*>(tr_smem);                       // [G][16][H + 1]
    float* cbuf = wt + (size_t)G * TR_TILE * (H + 1);                     // [R][16] pre-normalisation membranes
    __shared__ float red[TR_THREADS / TR_TILE][TR_TILE];
    const int tid = threadIdx.x, j = tid & 15, rsub = tid >> 4;
    const int n0 = blockIdx.x * TR_TILE, nj = n0 + j;
    for (int i = tid; i < G * TR_TILE * H; i += TR_THREADS) {
        const int gi = i / (TR_TILE * H), rem = i - gi * TR_TILE * H, jj = rem / H, k = rem - jj * H;
        wt[(gi * TR_TILE + jj) * (H + 1) + k] = p.w_hh[((size_t)gi * H + n0 + jj) * H + k];
    }
    __syncthreads();
    const float bf = p.bias[nj], bg = p.bias[H + nj];
    const float* wf = wt + (size_t)j * (H + 1);
    const float* wg = wt + (size_t)((G - 1) * TR_TILE + j) * (H + 1);
    float sum = 0.f;
    for (int r = rsub; r < R; r += TR_THREADS / TR_TILE) {
        const float* hp = p.h_prev + (size_t)r * H;
        float rf = 0.f, rg = 0.f;
        for (int k = 0; k < H; ++k) {
            const float hk = hp[k];
            rf = __builtin_fmaf(hk, wf[k], rf);
            if (G == 2) rg = __builtin_fmaf(hk, wg[k], rg);
        }
        if (G == 1) rg = rf;
        const float zf = p.z[(size_t)r * G * H + nj], zg = p.z[(size_t)r * G * H + (G - 1) * H + nj];
        const float pre_f = (zf + bf) + rf;
        const float pre_g = (zg + bg) + rg;
        const float f = 1.0f / (1.0f + expf(-pre_f));
        const float a = f * p.c_prev[(size_t)r * H + nj];
        const float b = (1.0f - f) * pre_g;
        const float cy = a + b;
        cbuf[r * TR_TILE + j] = cy;
        p.f[(size_t)r * H + nj] = f;
        p.g[(size_t)r * H + nj] = pre_g;
        sum += cy;
    }
    if (!p.use_bn) {
        __syncthreads();
        for (int r = rsub; r < R; r += TR_THREADS / TR_TILE) {
            const float cy = cbuf[r * TR_TILE + j];
            p.u[(size_t)r * H + nj] = cy;
            p.spikes[(size_t)r * H + nj] = cy >= 0.f ? 1.f : 0.f;
        }
        return;
    }
    const float mean = reduce16(sum, red, rsub, j) / (float)R;
    float sq = 0.f;
    for (int r = rsub; r < R; r += TR_THREADS / TR_TILE) {
        const float d = cbuf[r * TR_TILE + j] - mean;
        sq = __builtin_fmaf(d, d, sq);
    }
    const float var = reduce16(sq, red, rsub, j) / (float)R;
    const float invstd = 1.0f / sqrtf(var + p.eps);
    const float gam = p.bn_w[nj], bet = p.bn_b[nj];
    for (int r = rsub; r < R; r += TR_THREADS / TR_TILE) {
        const float xh = (cbuf[r * TR_TILE + j] - mean) * invstd;
        const float uu = xh * gam + bet;
        p.xhat[(size_t)r * H + nj] = xh;
        p.u[(size_t)r * H + nj] = uu;
        p.spikes[(size_t)r * H + nj] = uu >= 0.f ? 1.f : 0.f;
    }
    if (rsub == 0) {
        p.invstd[nj] = invstd;
        if (p.running_mean) {
            const float unb = R > 1 ? var * ((float)R / (float)(R - 1)) : var;
            p.running_mean[nj] = (1.0f - p.momentum) * p.running_mean[nj] + p.momentum * mean;
            p.running_var[nj] = (1.0f - p.momentum) * p.running_var[nj] + p.momentum * unb;
        }
    }
}

struct TrainBwdParams {
    const float* dh_up;    // [R][H] or null
    const float* dh_rec;   // [R][H] or null
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
};

__global__ __launch_bounds__(TR_THREADS) void gsn_train_step_bwd_kernel(const TrainBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char tr_smem[];
    float* dbuf = reinterpret_cast<float*>(tr_smem);  // [R][16] du
    __shared__ float red[TR_THREADS / TR_TILE][TR_TILE];
    const int H = p.H, R = p.R;
    const int tid = threadIdx.x, j = tid & 15, rsub = tid >> 4;
    const int nj = blockIdx.x * TR_TILE + j;
    float s1 = 0.f, s2 = 0.f;
    for (int r = rsub; r < R; r += TR_THREADS / TR_TILE) {
        const size_t o = (size_t)r * H + nj;
        float dh = 0.f;
        if (p.dh_up) dh += p.dh_up[o];
        if (p.dh_rec) dh += p.dh_rec[o];
        const float uu = p.u[o];
        const float tri = fmaxf(0.f, 1.0f - fabsf(uu));
        float du = dh * tri;
        if (p.dc_next) du += p.dc_next[o];
        dbuf[r * TR_TILE + j] = du;
        if (p.use_bn) {
            s1 += du;
            s2 = __builtin_fmaf(du, p.xhat[o], s2);
        }
    }
    float k1 = 0.f, k2 = 0.f, scale = 1.f;
    if (p.use_bn) {
        k1 = reduce16(s1, red, rsub, j);
        k2 = reduce16(s2, red, rsub, j);
        scale = p.bn_w[nj] * p.invstd[nj] / (float)R;
        if (rsub == 0) {
            p.d_bn_w[nj] += k2;
            p.d_bn_b[nj] += k1;
        }
    } else {
        __syncthreads();
    }
    for (int r = rsub; r < R; r += TR_THREADS / TR_TILE) {
        const size_t o = (size_t)r * H + nj;
        const float du = dbuf[r * TR_TILE + j];
        const float dcy = p.use_bn ? scale * ((float)R * du - k1 - p.xhat[o] * k2) : du;
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

extern "C" int sfsn_gsn_train_step_fwd(const float* z, const float* w_hh, const float* bias, const float* h_prev, const float* c_prev,
                                       const float* bn_w, const float* bn_b, float* running_mean, float* running_var, float momentum,
                                       float eps, int R, int H, int shared, float* spikes, float* u, float* xhat, float* f, float* g,
                                       float* invstd, void* stream) {
    if (!z || !w_hh || !bias || !h_prev || !c_prev || !spikes || !u || !f || !g || R <= 0 || H <= 0) return SFSN_EINVAL;
    if (H % TR_TILE != 0 || H > SFSN_MAX_HIDDEN) return SFSN_EUNSUPPORTED;
    const int use_bn = bn_w != nullptr;
    if (use_bn && (!bn_b || !xhat || !invstd)) return SFSN_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return SFSN_EINVAL;
    const int G = shared ? 1 : 2;
    const size_t lds = ((size_t)G * TR_TILE * (H + 1) + (size_t)R * TR_TILE) * sizeof(float);
    if (lds > 150 * 1024) return SFSN_EUNSUPPORTED;  // rows of a layer per step: ~2000 at H = 320 (one workgroup holds them all)
    TrainFwdParams p;
    p.z = z; p.w_hh = w_hh; p.bias = bias; p.h_prev = h_prev; p.c_prev = c_prev; p.bn_w = bn_w; p.bn_b = bn_b;
    p.running_mean = running_mean; p.running_var = running_var; p.spikes = spikes; p.u = u; p.xhat = xhat; p.f = f; p.g = g;
    p.invstd = invstd; p.momentum = momentum; p.eps = eps; p.R = R; p.H = H; p.shared = shared; p.use_bn = use_bn;
    auto kern = gsn_train_step_fwd_kernel;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return SFSN_EHIP;
    hipLaunchKernelGGL(kern, dim3(H / TR_TILE), dim3(TR_THREADS), lds, static_cast<hipStream_t>(stream), p);
    return hip_ok_tr(hipGetLastError());
}

extern "C" int sfsn_gsn_train_step_bwd(const float* dh_up, const float* dh_rec, const float* dc_next, const float* u, const float* xhat,
                                       const float* f, const float* g, const float* c_prev, const float* invstd, const float* bn_w, int R,
                                       int H, int shared, float* d_gates, float* d_z, float* dc_prev, float* d_bn_w, float* d_bn_b,
                                       void* stream) {
    if (!u || !f || !g || !c_prev || !d_gates || !dc_prev || R <= 0 || H <= 0) return SFSN_EINVAL;
    if (H % TR_TILE != 0 || H > SFSN_MAX_HIDDEN) return SFSN_EUNSUPPORTED;
    const int use_bn = bn_w != nullptr;
    if (use_bn && (!xhat || !invstd || !d_bn_w || !d_bn_b)) return SFSN_EINVAL;
    if (shared && !d_z) return SFSN_EINVAL;
    const size_t lds = (size_t)R * TR_TILE * sizeof(float);
    if (lds > 150 * 1024) return SFSN_EUNSUPPORTED;
    TrainBwdParams p;
    p.dh_up = dh_up; p.dh_rec = dh_rec; p.dc_next = dc_next; p.u = u; p.xhat = xhat; p.f = f; p.g = g; p.c_prev = c_prev;
    p.invstd = invstd; p.bn_w = bn_w; p.d_gates = d_gates; p.d_z = d_z; p.dc_prev = dc_prev; p.d_bn_w = d_bn_w; p.d_bn_b = d_bn_b;
    p.R = R; p.H = H; p.shared = shared; p.use_bn = use_bn;
    auto kern = gsn_train_step_bwd_kernel;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return SFSN_EHIP;
    hipLaunchKernelGGL(kern, dim3(H / TR_TILE), dim3(TR_THREADS), lds, static_cast<hipStream_t>(stream), p);
    return hip_ok_tr(hipGetLastError());
}
