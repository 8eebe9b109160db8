// sfsn_feat_dev.h -- device helpers of the feature prologue / deep-filter epilogue shared by sfsn_kernels.hip and sfsn_hop.hip
// (the streaming hop kernel reproduces the offline kernels' arithmetic expression by expression).  gfx950 only.
#ifndef SFSN_FEAT_DEV_H
#define SFSN_FEAT_DEV_H
#include <hip/hip_runtime.h>

#include <type_traits>

__device__ __forceinline__ int reflect_bin(int f, int nf) { return f < 0 ? -f : (f > nf - 1 ? 2 * (nf - 1) - f : f); }

// |re + i im| without libm's range-scaling hypot (~25 instructions): STFT magnitudes of audio are nowhere near the
// fp32 overflow / underflow of re^2 + im^2; v_sqrt_f32 is good to 1 ulp (parity tolerance is 1e-4 relative).
__device__ __forceinline__ float fast_abs2(float re, float im) { return __builtin_amdgcn_sqrtf(__builtin_fmaf(re, re, im * im)); }

__device__ __forceinline__ float compress_mag(float re, float im, float fdrc) {
    const float m = fast_abs2(re, im);                               // torch.abs(complex)
    return fdrc == 0.5f ? __builtin_amdgcn_sqrtf(m) : powf(m, fdrc);  // ATen evaluates pow(x, 0.5) as sqrt
}

// Wave-wide sum, result broadcast to every lane.  DPP row shifts / broadcasts (VALU speed) instead of
// __shfl_xor, which lowers to ds_bpermute_b32 (an LDS round trip per step: the LayerNorm rows were latency bound).
__device__ __forceinline__ float wave_sum(float v) {
    auto dpp_add = [](float x, auto ctrl, auto rmask) __attribute__((always_inline)) {
        const int shifted = __builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(rmask)::value, 0xf, true);
        return x + __int_as_float(shifted);
    };
    v = dpp_add(v, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});  // row_shr:1
    v = dpp_add(v, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});  // row_shr:2
    v = dpp_add(v, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});  // row_shr:4
    v = dpp_add(v, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});  // row_shr:8  -> lane 15 of a row = row sum
    v = dpp_add(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});  // row_bcast:15 into rows 1,3
    v = dpp_add(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});  // row_bcast:31 into rows 2,3 -> lane 63 = total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

#endif
