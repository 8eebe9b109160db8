// Do MFMA (v_mfma_i32_16x16x64_i8) and VALU work of DIFFERENT waves of one SIMD overlap on gfx950?
// Each wave runs `iters` x [NM MFMAs ; NV dependent-free fp32 FMAs]; 16 waves per CU (4 per SIMD), one workgroup per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int NM, int NV, int SKEW>
__global__ __launch_bounds__(1024) void k(int iters, float* out, int* outi) {
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)threadIdx.x};
    v4i acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0;
    float f0 = threadIdx.x, f1 = 1.f, f2 = 2.f, f3 = 3.f;
    const int wave = threadIdx.x >> 6;
    if (SKEW && ((wave >> 2) & 1)) {  // half of each SIMD's waves start with the VALU phase
#pragma unroll
        for (int j = 0; j < NV / 4; ++j) {
            f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0001f, 0.5f);
            f2 = __builtin_fmaf(f2, 1.0001f, 0.5f); f3 = __builtin_fmaf(f3, 1.0001f, 0.5f);
        }
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NM / 3; ++j) {
            acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc2, 0, 0, 0);
        }
        asm volatile("" : "+v"(acc0), "+v"(acc1), "+v"(acc2));
#pragma unroll
        for (int j = 0; j < NV / 4; ++j) {
            f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0001f, 0.5f);
            f2 = __builtin_fmaf(f2, 1.0001f, 0.5f); f3 = __builtin_fmaf(f3, 1.0001f, 0.5f);
        }
        asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
    }
    out[blockIdx.x * 1024 + threadIdx.x] = f0 + f1 + f2 + f3;
    outi[blockIdx.x * 1024 + threadIdx.x] = acc0.x + acc1.y + acc2.z;
}

template <int NM, int NV, int SKEW>
void run(const char* name, float* out, int* outi) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NM, NV, SKEW>), dim3(256), dim3(1024), 0, 0, 100, out, outi);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NV, SKEW>), dim3(256), dim3(1024), 0, 0, iters, out, outi);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s NM=%2d NV=%3d skew=%d : %8.1f ns per iteration (4 waves/SIMD)\n", name, NM, NV, SKEW, ms * 1e6 / iters);
}

int main() {
    float* out; int* outi;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&outi, 256 * 1024 * 4);
    run<12, 0, 0>("mfma only", out, outi);
    run<0, 36, 0>("valu only", out, outi);
    run<12, 36, 0>("mfma then valu, waves in phase", out, outi);
    run<12, 36, 1>("mfma then valu, half skewed", out, outi);
    run<12, 72, 0>("mfma then 2x valu, in phase", out, outi);
    run<12, 72, 1>("mfma then 2x valu, skewed", out, outi);
    run<24, 36, 0>("2x mfma then valu, in phase", out, outi);
    return 0;
}
