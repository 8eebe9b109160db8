// sfsn_fft_dev.h -- the 512-point real transform one wave computes for one frame, shared by the offline STFT / inverse STFT
// kernels (sfsn_fft.hip) and the streaming hop (sfsn_hop.hip): the same instructions, hence the same bits.  gfx950 only.
#ifndef SFSN_FFT_DEV_H
#define SFSN_FFT_DEV_H
#include <hip/hip_runtime.h>

#define FFT_N 256       // complex points per frame
#define FFT_TT 16       // frames per workgroup
#define FFT_NFFT 512
#define FFT_F 257

struct Twiddles {  // per lane: stage twiddles w[s][r-1] = exp(-+ 2 pi i k r / (4 Ns)), k = lane mod Ns, Ns = 4, 16, 64
    float2 w[3][3];
};

// unit[m] = exp(-2 pi i m / 512), m < 512, computed once per workgroup (one sincospif per thread or two) into LDS; every
// twiddle of the transform is an entry of it: 13 sincospif calls per lane would cost more than the transform itself.
__device__ __forceinline__ void fill_unit_table(float2* unit, int tid, int nthreads) {
    for (int m = tid; m < FFT_NFFT; m += nthreads) {
        float sn, cs;
        sincospif((float)m / 256.0f, &sn, &cs);
        unit[m] = make_float2(cs, -sn);
    }
}

template <bool INV>
__device__ __forceinline__ float2 unit_at(const float2* unit, int m) {
    const float2 u = unit[m & (FFT_NFFT - 1)];
    return INV ? make_float2(u.x, -u.y) : u;
}

template <bool INV>
__device__ __forceinline__ Twiddles make_twiddles(const float2* unit, int lane) {
    Twiddles t;
    const int ns[3] = {4, 16, 64};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int k = lane & (ns[s] - 1);
#pragma unroll
        for (int r = 1; r < 4; ++r) t.w[s][r - 1] = unit_at<INV>(unit, k * r * (FFT_NFFT / (4 * ns[s])));  // 2 pi k r / (4 Ns)
    }
    return t;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

template <bool INV>
__device__ __forceinline__ void fft4(float2 (&v)[4]) {
    const float2 a = make_float2(v[0].x + v[2].x, v[0].y + v[2].y), b = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
    const float2 c = make_float2(v[1].x + v[3].x, v[1].y + v[3].y), d = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
    v[0] = make_float2(a.x + c.x, a.y + c.y);
    v[2] = make_float2(a.x - c.x, a.y - c.y);
    const float2 p = make_float2(b.x + d.y, b.y - d.x), m = make_float2(b.x - d.y, b.y + d.x);  // b - i d, b + i d
    v[1] = INV ? m : p;
    v[3] = INV ? p : m;
}

// v[r] = in[lane + 64 r] on entry, Z[lane + 64 r] on exit (unnormalised).  buf: 256 float2 of LDS owned by this wave.
template <bool INV>
__device__ __forceinline__ void fft256(float2 (&v)[4], float2* buf, int lane, const Twiddles& tw) {
    // stage Ns = 1: no twiddle; outputs of butterfly j are 4 j + r
    fft4<INV>(v);
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[4 * lane + r] = v[r];
    const int ns[3] = {4, 16, 64};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = buf[lane + 64 * r];
#pragma unroll
        for (int r = 1; r < 4; ++r) v[r] = cmul(v[r], tw.w[s][r - 1]);
        fft4<INV>(v);
        if (s < 2) {
            const int k = lane & (ns[s] - 1), j0 = (lane - k) * 4 + k;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; ++r) buf[j0 + r * ns[s]] = v[r];
        }
    }
}


// Forward split step: v = Z[lane + 64 r] of the 256-point transform of z[n] = x[2n] + i x[2n+1] (x already windowed) ->
// X[k], k = lane + 64 r, of the 512-point real transform; lane 0 also gets the Nyquist bin.  wk[r] = exp(-2 pi i k / 512).
__device__ __forceinline__ void rfft512_split(const float2 (&v)[4], float2* buf, int lane, const float2 (&wk)[4], float2 (&X)[4], float2& nyq) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[lane + 64 * r] = v[r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = lane + 64 * r;
        const float2 zm = buf[(FFT_N - k) & (FFT_N - 1)], zk = v[r];
        const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        const float2 o = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
        const float2 wo = cmul(wk[r], o);
        X[r] = make_float2(e.x + wo.x, e.y + wo.y);
        if (k == 0) nyq = make_float2(zk.x - zk.y, 0.0f);
    }
}

// Inverse pre-split: X[k] and X[256 - k] (k = lane + 64 r) -> Z[k], the input of the inverse 256-point transform whose output
// pairs are (x[2n], x[2n+1]) * 256.  wk[r] = exp(+2 pi i k / 512).  (C2R semantics: Im of DC and Nyquist ignored.)
__device__ __forceinline__ float2 irfft512_presplit(float2 xk, float2 xm, int k, float2 wk) {
    if (k == 0) { xk.y = 0.0f; xm.y = 0.0f; }
    const float2 e = make_float2(0.5f * (xk.x + xm.x), 0.5f * (xk.y - xm.y));
    const float2 p = make_float2(0.5f * (xk.x - xm.x), 0.5f * (xk.y + xm.y));
    const float2 o = cmul(p, wk);
    return make_float2(e.x - o.y, e.y + o.x);  // Z = E + i O
}

#endif
