// The two edges of the path on the device: STFT and inverse STFT for the 512-point Hann / hop-128 analysis every
// reference config uses (audiozen/acoustics/audio_feature.py:236-347 = torch.stft(center=True, pad_mode="constant",
// window=hann(n_fft)) and torch.istft(window=hann(n_fft), length=L)); SURVEY.md 8(f) rank 2.
//
// One wave transforms one frame.  A real 512-point transform is a 256-point complex one on z[n] = x[2n] + i x[2n+1]
// plus a split step; the 256-point transform is four radix-4 Stockham stages -- lane j owns butterfly j of every stage
// (four points in registers), the exchange between stages goes through 2 KB of LDS private to the wave (LDS operations
// of one wave execute in order: all reads of a stage are served before its writes land, no barrier).  The first stage
// reads its points straight from the waveform (coalesced float2 per lane), the last leaves Z[j + 64 r] in lane j.
// Spectra are [B][F][T] with T contiguous (torch layout): a workgroup stages its 16 frames in LDS and writes rows of
// 16 frames per bin, so both the waveform side and the spectrum side are coalesced.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sfsn.h"

#include "sfsn_fft_dev.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// STFT: wave [B][L] -> X [B][257][T][2]; frame t covers samples [t*hop - 256, t*hop + 256), zeros outside [0, L)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ wave, const float* __restrict__ window,
                                                    float* __restrict__ X, int B, int L, int T, int hop) {
    __shared__ float2 fbuf[4][FFT_N];
    __shared__ float2 stage[FFT_F][FFT_TT + 1];
    const int b = blockIdx.y, t0 = blockIdx.x * FFT_TT;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ float2 unit[FFT_NFFT];
    fill_unit_table(unit, tid, 256);
    __syncthreads();
    const Twiddles tw = make_twiddles<false>(unit, lane);
    float2 win[4], wk[4];  // window at my samples; split-step twiddle exp(-2 pi i k / 512) at my bins k = lane + 64 r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = 2 * (lane + 64 * r);
        win[r] = make_float2(window[n], window[n + 1]);
        wk[r] = unit_at<false>(unit, lane + 64 * r);
    }
    const float* wrow = wave + (size_t)b * L;
    for (int ft = wv; ft < FFT_TT; ft += 4) {
        const int t = t0 + ft;
        if (t >= T) break;  // wave-uniform
        const int s0 = t * hop - FFT_NFFT / 2;
        float2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = s0 + 2 * (lane + 64 * r);
            const float x0 = (s >= 0 && s < L) ? wrow[s] : 0.0f, x1 = (s + 1 >= 0 && s + 1 < L) ? wrow[s + 1] : 0.0f;
            v[r] = make_float2(x0 * win[r].x, x1 * win[r].y);
        }
        float2* buf = fbuf[wv];
        fft256<false>(v, buf, lane, tw);
        float2 Xk[4], nyq = make_float2(0.0f, 0.0f);
        rfft512_split(v, buf, lane, wk, Xk, nyq);
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[lane + 64 * r][ft] = Xk[r];
        if (lane == 0) stage[FFT_N][ft] = nyq;
    }
    __syncthreads();
    const int nt = (T - t0 < FFT_TT) ? T - t0 : FFT_TT;
    for (int idx = tid; idx < FFT_F * FFT_TT; idx += 256) {
        const int f = idx >> 4, ft = idx & (FFT_TT - 1);
        if (ft < nt) *reinterpret_cast<float2*>(X + (((size_t)b * FFT_F + f) * T + t0 + ft) * 2) = stage[f][ft];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// inverse STFT: X [B][257][T][2] -> wave [B][length].  torch.istft: frames = irfft(X) * window, overlap-added at
// t*hop, divided by the overlap-added squared window, the first n_fft/2 samples dropped, `length` samples kept.
// A workgroup produces the padded samples [t0*hop, (t0+16)*hop): it needs frames t0-3 .. t0+15.
// ---------------------------------------------------------------------------------------------------------------------
#define IFFT_HALO 3
#define IFFT_WAVES 8
#define IFFT_NFR (FFT_TT + IFFT_HALO)
#define IFFT_XS_BYTES (FFT_F * (IFFT_NFR + 2) * 8)  // spectrum tile (>= the IFFT_NFR * 2 KB of time-domain frames that replace it)
#define IFFT_LDS (IFFT_WAVES * FFT_N * 8 + FFT_NFFT * 8 + FFT_NFFT * 4 + IFFT_XS_BYTES)
__global__ __launch_bounds__(IFFT_WAVES * 64) void istft_kernel(const float* __restrict__ X, const float* __restrict__ window,
                                                                float* __restrict__ wave, int B, int T, int hop, int length) {
    extern __shared__ __attribute__((aligned(16))) char ifft_smem[];
    constexpr int NFR = IFFT_NFR, FPW = (NFR + IFFT_WAVES - 1) / IFFT_WAVES;  // frames per wave
    float2(*fbuf)[FFT_N] = reinterpret_cast<float2(*)[FFT_N]>(ifft_smem);  // per-wave exchange
    float2* unit = reinterpret_cast<float2*>(ifft_smem + IFFT_WAVES * FFT_N * 8);
    float* wl = reinterpret_cast<float*>(ifft_smem + IFFT_WAVES * FFT_N * 8 + FFT_NFFT * 8);  // window
    char* region = ifft_smem + IFFT_WAVES * FFT_N * 8 + FFT_NFFT * 8 + FFT_NFFT * 4;
    // spectrum tile; row stride 21 float2 = 42 dwords: the 64 bins a wave reads for one frame land on distinct banks.
    // Once every wave holds its transformed frames in registers the same memory takes the time-domain frames.
    float2(*xs)[NFR + 2] = reinterpret_cast<float2(*)[NFR + 2]>(region);
    float(*tbuf)[FFT_NFFT] = reinterpret_cast<float(*)[FFT_NFFT]>(region);
    static_assert(IFFT_NFR * FFT_NFFT * 4 <= IFFT_XS_BYTES, "time-domain frames must fit in the spectrum tile");
    const int b = blockIdx.y, t0 = blockIdx.x * FFT_TT;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tfirst = t0 - IFFT_HALO;
    {   // spectrum tile: 19 frames (152 B) per bin; eight independent requests per thread in flight (a load -> store loop
        // would pay one memory round trip per iteration)
        const int c = tid & 31, t = tfirst + c;
        const bool live = c < NFR && t >= 0 && t < T;
        const int tc = live ? t : (t < 0 ? 0 : T - 1);
        const float* src = X + ((size_t)b * FFT_F * T + tc) * 2;
        for (int f0 = tid >> 5; f0 < FFT_F; f0 += 8 * (IFFT_WAVES * 2)) {  // my bins: f0, f0+16, ..., f0+112
            float2 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int f = f0 + i * (IFFT_WAVES * 2);
                if (f > FFT_F - 1) f = FFT_F - 1;
                v[i] = *reinterpret_cast<const float2*>(src + (size_t)f * T * 2);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int f = f0 + i * (IFFT_WAVES * 2);
                if (f < FFT_F && c < NFR) xs[f][c] = live ? v[i] : make_float2(0.0f, 0.0f);
            }
        }
    }
    fill_unit_table(unit, tid, IFFT_WAVES * 64);
    for (int i = tid; i < FFT_NFFT; i += IFFT_WAVES * 64) wl[i] = window[i];
    __syncthreads();
    const Twiddles tw = make_twiddles<true>(unit, lane);
    float2 win[4], wk[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = 2 * (lane + 64 * r);
        win[r] = make_float2(wl[n], wl[n + 1]);
        wk[r] = unit_at<true>(unit, lane + 64 * r);  // exp(+2 pi i k / 512)
    }
    float2 res[FPW][4];
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
        const int c = wv + i * IFFT_WAVES, t = tfirst + c;
        if (c >= NFR || t < 0 || t >= T) continue;  // wave-uniform
        float2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = lane + 64 * r;
            v[r] = irfft512_presplit(xs[k][c], xs[FFT_N - k][c], k, wk[r]);
        }
        // v holds Z[lane + 64 r]: exactly the input order of the first stage
        fft256<true>(v, fbuf[wv], lane, tw);
        const float sc = 1.0f / (float)FFT_N;
#pragma unroll
        for (int r = 0; r < 4; ++r) res[i][r] = make_float2(v[r].x * sc * win[r].x, v[r].y * sc * win[r].y);
    }
    __syncthreads();  // every wave has read what it needs of the spectrum tile
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
        const int c = wv + i * IFFT_WAVES, t = tfirst + c;
        if (c >= NFR || t < 0 || t >= T) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<float2*>(&tbuf[c][2 * (lane + 64 * r)]) = res[i][r];
    }
    __syncthreads();
    // overlap-add: padded sample n = t0*hop + i gets frames t = floor(n/hop) - q, q = 0..(512/hop - 1), that exist
    const int nover = FFT_NFFT / hop;
    for (int i = tid; i < FFT_TT * hop; i += IFFT_WAVES * 64) {
        const int n = t0 * hop + i, m = n - FFT_NFFT / 2;
        if (m < 0 || m >= length) continue;
        const int tq = n / hop;
        float acc = 0.0f, env = 0.0f;
        for (int q = nover - 1; q >= 0; --q) {  // ascending frame index, as a sequential overlap-add visits them
            const int t = tq - q, off = n - t * hop;
            if (t < 0 || t >= T || off >= FFT_NFFT) continue;
            const float w = wl[off];
            acc += tbuf[t - tfirst][off];
            env += w * w;
        }
        wave[(size_t)b * length + m] = env > 1e-11f ? acc / env : 0.0f;
    }
}

inline int hip_rc(hipError_t e) { return e == hipSuccess ? SFSN_OK : SFSN_EHIP; }

}  // namespace

extern "C" int sfsn_stft(const float* wave, int B, int L, int n_fft, int hop, const float* window, float* stft_ri, int T, void* stream) {
    if (!wave || !window || !stft_ri || B <= 0 || L <= 0 || hop <= 0 || T <= 0) return SFSN_EINVAL;
    if (n_fft != FFT_NFFT) return SFSN_EUNSUPPORTED;
    if (T != 1 + L / hop || (reinterpret_cast<uintptr_t>(stft_ri) & 7)) return SFSN_EINVAL;
    hipLaunchKernelGGL(stft_kernel, dim3((T + FFT_TT - 1) / FFT_TT, B), dim3(256), 0, static_cast<hipStream_t>(stream), wave, window,
                       stft_ri, B, L, T, hop);
    return hip_rc(hipGetLastError());
}

extern "C" int sfsn_istft(const float* stft_ri, int B, int T, int n_fft, int hop, const float* window, float* wave, int length,
                          void* stream) {
    if (!wave || !window || !stft_ri || B <= 0 || T <= 0 || hop <= 0 || length <= 0) return SFSN_EINVAL;
    if (n_fft != FFT_NFFT || FFT_NFFT % hop != 0 || FFT_NFFT / hop > IFFT_HALO + 1) return SFSN_EUNSUPPORTED;
    if (length > (T - 1) * hop + FFT_NFFT / 2 || (reinterpret_cast<uintptr_t>(stft_ri) & 7)) return SFSN_EINVAL;
    const int nblk = (length + FFT_NFFT / 2 + FFT_TT * hop - 1) / (FFT_TT * hop);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(istft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, IFFT_LDS) != hipSuccess)
        return SFSN_EHIP;
    hipLaunchKernelGGL(istft_kernel, dim3(nblk, B), dim3(IFFT_WAVES * 64), IFFT_LDS, static_cast<hipStream_t>(stream), stft_ri, window,
                       wave, B, T, hop, length);
    return hip_rc(hipGetLastError());
}
