// pair_role.hip -- the FUSED3 role (csrc/sfsn_scan3i_dev.h) alone, input complete before the launch (FLG = 0), against a plain
// per-(row, neuron) restatement of the same arithmetic on the same packed weights: every spike of every frame compared, then timed.
// A fast loop for working on the role (the library takes 90 s to build; this file 10 s).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include -I../../spiking_fullsubnet_amd/csrc -o pair_role.bin pair_role.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdint.h>
#include <vector>

#include "sfsn.h"
#include "sfsn_scan3i_dev.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#ifndef HH
#define HH 224
#endif
#ifndef FLGV
#define FLGV 0  // 1: the gated form of the role (ring depth, sc1 loads, stop word) with no producer to wait for
#endif
constexpr int H = HH, NT = H / 16, KS = (H + 63) / 64, HP = KS * 64;
constexpr int TL = ((H & 63) != 0 && (H & 63) <= 32) ? 1 : 0;

struct Args {
    Scan3iRole rl;
    unsigned* err;
    int T;
};

template <int FLG>
__global__ __launch_bounds__(1024) void role_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Scan3iRole rl = a.rl;
    rl.row0 = blockIdx.x * 8;
    StackLink lk;
    lk.in = nullptr; lk.n_in = 0; lk.out = nullptr; lk.err = a.err; lk.lag = 0; lk.dbg = nullptr;
    scan3i_role<KS, TL, 3, FLG>(rl, lk, smem, a.T, H, NT);
}

// the same arithmetic, one thread per neuron, one workgroup per row
__device__ __forceinline__ int wdigit(const int8_t* w, int d, int neuron, int k) {
    const int tile = neuron >> 4, n = neuron & 15, ks = k >> 6, q = (k >> 4) & 3, j = k & 15;
    return w[((((size_t)d * NT + tile) * KS + ks) * 64 + q * 16 + n) * 16 + j];
}
__global__ __launch_bounds__(256) void ref_kernel(Args a, float* spikes_ref) {
    __shared__ int8_t h[2][HP];
    const Scan3iRole& rl = a.rl;
    const int r = blockIdx.x, j = threadIdx.x;
    for (int i = j; i < 2 * HP; i += 256) (&h[0][0])[i] = 0;
    __syncthreads();
    float c = 0.f;
    for (int t = 0; t < a.T; ++t) {
        float y = 0.f;
        if (j < H) {
            int si = 0, sr = 0;
            const int8_t* s = rl.spikes_in + ((size_t)t * rl.R + r) * HP;
            for (int k = 0; k < H; ++k) {
                const int wi = (wdigit(rl.w_ih, 2, j, k) << 16) + (wdigit(rl.w_ih, 1, j, k) << 8) + wdigit(rl.w_ih, 0, j, k);
                const int wh = (wdigit(rl.w_hh, 2, j, k) << 16) + (wdigit(rl.w_hh, 1, j, k) << 8) + wdigit(rl.w_hh, 0, j, k);
                si += s[k] ? wi : 0;
                sr += h[t & 1][k] ? wh : 0;
            }
            const float z = __builtin_fmaf((float)si, rl.w_ih_dq[j], rl.bias[j]);
            const float pre_f = __builtin_fmaf((float)sr, rl.w_dq[j], z);
            const float pre_g = pre_f + (rl.bias[H + j] - rl.bias[j]);
            const float f = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre_f * -1.44269504088896341f));
            const float m = __builtin_fmaf(f, c - pre_g, pre_g);
            y = __builtin_fmaf(m, rl.bn_alpha[j], rl.bn_beta[j]);
            c = y;
            h[(t & 1) ^ 1][j] = y >= 0.f ? 1 : 0;
            spikes_ref[((size_t)t * rl.R + r) * H + j] = y >= 0.f ? 1.f : 0.f;
        }
        __syncthreads();
    }
}

template <class T_>
static T_* dev(const std::vector<T_>& v) {
    T_* d; CK(hipMalloc(&d, v.size() * sizeof(T_))); CK(hipMemcpy(d, v.data(), v.size() * sizeof(T_), hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 40, R = argc > 2 ? atoi(argv[2]) : 832;
    srand(7);
    auto packed = [&]() {
        std::vector<int8_t> w((size_t)3 * NT * KS * 1024);
        for (size_t i = 0; i < w.size(); ++i) {
            const int ks = (int)((i / 1024) % KS), lane = (int)((i % 1024) / 16), k = ks * 64 + (lane / 16) * 16 + (int)(i % 16);
            const int d = (int)(i / ((size_t)NT * KS * 1024));
            w[i] = k < H ? (int8_t)((rand() & 0xff) - 128) : 0;
            if (d == 2) w[i] = (int8_t)(w[i] / 8);  // (a weight of ~2^-4 at dq = 2^-23: realistic pre-activations)
        }
        return w;
    };
    std::vector<float> dq(H, 1.0f / 8388608.f), bias(2 * H), al(H), be(H);
    for (int j = 0; j < H; ++j) { bias[j] = 0.05f * ((j % 11) - 5); bias[H + j] = 0.03f * ((j % 7) - 3); al[j] = 1.0f + 0.01f * (j % 13); be[j] = 0.02f * ((j % 9) - 4); }
    std::vector<int8_t> sin((size_t)T * R * HP);
    for (size_t i = 0; i < sin.size(); ++i) sin[i] = ((int)(i % HP) < H && (rand() & 3) == 0) ? 1 : 0;
    Args a{};
    a.T = T;
    std::vector<int8_t> hwi = packed(), hwh = packed();
    { unsigned c1 = 0, c2 = 0; for (size_t i = 0; i < hwi.size(); ++i) { c1 = c1 * 31 + (unsigned)(uint8_t)hwi[i]; c2 = c2 * 31 + (unsigned)(uint8_t)hwh[i]; } printf("weight checksums %08x %08x\n", c1, c2); }
    a.rl.spikes_in = dev(sin); a.rl.w_ih = dev(hwi); a.rl.w_hh = dev(hwh);
    a.rl.w_ih_dq = dev(dq); a.rl.w_dq = dev(dq); a.rl.bias = dev(bias); a.rl.bn_alpha = dev(al); a.rl.bn_beta = dev(be);
    CK(hipMalloc(&a.rl.h_state, (size_t)R * H * 4)); CK(hipMalloc(&a.rl.c_state, (size_t)R * H * 4));
    CK(hipMalloc(&a.rl.spikes_f32, (size_t)T * R * H * 4)); CK(hipMalloc(&a.rl.spikes_i8, (size_t)T * R * HP));
    CK(hipMemset(a.rl.spikes_i8, 0, (size_t)T * R * HP));
    a.rl.R = R;
    CK(hipMalloc(&a.err, 64)); CK(hipMemset(a.err, 0, 64));
    float* ref; CK(hipMalloc(&ref, (size_t)T * R * H * 4));
    const int grid = (R + 7) / 8, lds = Scan3iCfg<KS, FLGV>::lds_bytes(NT);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(role_kernel<FLGV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    auto launch = [&]() {
        CK(hipMemset(a.rl.h_state, 0, (size_t)R * H * 4)); CK(hipMemset(a.rl.c_state, 0, (size_t)R * H * 4));
        hipLaunchKernelGGL(role_kernel<FLGV>, dim3(grid), dim3(1024), lds, 0, a);
    };
    float* ref2; CK(hipMalloc(&ref2, (size_t)T * R * H * 4));
    hipLaunchKernelGGL(ref_kernel, dim3(R), dim3(256), 0, 0, a, ref2);
    CK(hipDeviceSynchronize());
    launch();
    hipLaunchKernelGGL(ref_kernel, dim3(R), dim3(256), 0, 0, a, ref);
    CK(hipDeviceSynchronize());
    {
        std::vector<float> r1((size_t)T * R * H), r2((size_t)T * R * H);
        CK(hipMemcpy(r1.data(), ref, r1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(r2.data(), ref2, r2.size() * 4, hipMemcpyDeviceToHost));
        size_t d = 0; for (size_t i = 0; i < r1.size(); ++i) d += r1[i] != r2[i];
        unsigned cs = 0; for (size_t i = 0; i < sin.size(); ++i) cs = cs * 31 + (unsigned)sin[i];
        unsigned co = 0; for (size_t i = 0; i < r1.size(); ++i) co = co * 31 + (r1[i] != 0.f);
        size_t f0 = 0; for (size_t i = 0; i < (size_t)R * H; ++i) f0 += r1[i] != 0.f;
        printf("reference before / after the role launch: %zu differing spikes; input checksum %08x; reference checksum %08x, frame-0 spikes %zu\n", d, cs, co, f0);
    }
    std::vector<float> g((size_t)T * R * H), w((size_t)T * R * H);
    CK(hipMemcpy(g.data(), a.rl.spikes_f32, g.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(w.data(), ref, w.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, ones = 0;
    int first_t = -1;
    for (int t = 0; t < T; ++t) {
        size_t bt = 0;
        for (size_t i = 0; i < (size_t)R * H; ++i) { bt += g[(size_t)t * R * H + i] != w[(size_t)t * R * H + i]; ones += w[(size_t)t * R * H + i] != 0.f; }
        if (bt && first_t < 0) {
            first_t = t;
            printf("first differing frame %d: %zu spikes; examples (row, neuron, got, want):", t, bt);
            int shown = 0;
            for (size_t i = 0; i < (size_t)R * H && shown < 12; ++i)
                if (g[(size_t)t * R * H + i] != w[(size_t)t * R * H + i]) { printf(" (%zu,%zu,%g,%g)", i / H, i % H, g[(size_t)t * R * H + i], w[(size_t)t * R * H + i]); ++shown; }
            printf("\n");
        }
        bad += bt;
    }
    printf("H=%d KS=%d TL=%d T=%d R=%d grid=%d lds=%d: %zu differing spikes of %zu (rate %.3f)%s\n", H, KS, TL, T, R, grid, lds, bad, g.size(), (double)ones / g.size(),
           bad ? "  ** MISMATCH **" : "  identical");
    if (argc > 3) {
        const int TT = atoi(argv[3]);
        std::vector<int8_t> big((size_t)TT * R * HP);
        for (size_t i = 0; i < big.size(); ++i) big[i] = ((int)(i % HP) < H && (rand() & 3) == 0) ? 1 : 0;
        Args b = a;
        b.T = TT; b.rl.spikes_in = dev(big);
        CK(hipMalloc(&b.rl.spikes_f32, (size_t)TT * R * H * 4)); CK(hipMalloc(&b.rl.spikes_i8, (size_t)TT * R * HP));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemset(b.rl.h_state, 0, (size_t)R * H * 4)); CK(hipMemset(b.rl.c_state, 0, (size_t)R * H * 4));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(role_kernel<FLGV>, dim3(grid), dim3(1024), lds, 0, b);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("timing: T=%d, %d workgroups: %.3f ms = %.3f us per step\n", TT, grid, best, best * 1e3 / TT);
    }
    return 0;
}
