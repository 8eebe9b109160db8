// store_retire.hip -- round 4: how long does a wave's global store take to retire (vmcnt), and how many bytes per microsecond can
// ONE wave push through its 63-deep memory queue?  The IO-wave scans (sfsn_scan3_dev.h) move a frame's spikes with one or two
// storer waves per workgroup; their step time at 8 rows per workgroup (0.69-0.9 us) was attributed to "stores per frame x retire
// time / 63".  This measures the retire time directly.
//
// Each workgroup: wave 0 issues N stores of 1 KiB (64 lanes x dwordx4) to consecutive 1 KiB blocks of its own region, frame stride
// STRIDE bytes between groups of PER stores (a frame), then waits vmcnt(0).  Reported: clk for (a) one store + wait (latency), (b) N
// stores back to back + wait(0) (throughput at queue depth <= 63), (c) N stores with a counted wait keeping Q in flight.
// MODE 0 plain, 1 sc1, 2 nt, 3 sc0 sc1.   WGS workgroups side by side (1 = idle chip, 208 = the pair launch's footprint).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o store_retire.bin store_retire.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ void st16(float* base, unsigned off, v4f d) {
    if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(off), "v"(d), "s"(base) : "memory");
    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(off), "v"(d), "s"(base) : "memory");
    if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(off), "v"(d), "s"(base) : "memory");
    if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" ::"v"(off), "v"(d), "s"(base) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, long long* res, int N, int PER, long long stride, long long wg_stride, int reps) {
    const int lane = threadIdx.x;
    float* base = out + (size_t)blockIdx.x * (wg_stride / 4);
    v4f d = {1.f * lane, 2.f, 3.f, (float)blockIdx.x};
    long long lat = 0, thr = 0, q16 = 0;
    for (int r = 0; r < reps; ++r) {
        float* b = base + (size_t)r * (size_t)((N / PER + 1) * (stride / 4)) * 3;
        // (a) latency of one store
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long t0 = (long long)wall_clock64();
        st16<MODE>(b, lane * 16, d);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long t1 = (long long)wall_clock64();
        lat += t1 - t0;
        // (b) N stores, wait at the end
        b += (N / PER + 1) * (stride / 4);
        t0 = (long long)wall_clock64();
        for (int i = 0; i < N; ++i) {
            float* f = b + (size_t)(i / PER) * (stride / 4);
            st16<MODE>(f, (unsigned)((i % PER) * 1024 + lane * 16), d);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t1 = (long long)wall_clock64();
        thr += t1 - t0;
        // (c) N stores, at most 16 in flight
        b += (N / PER + 1) * (stride / 4);
        t0 = (long long)wall_clock64();
        for (int i = 0; i < N; ++i) {
            float* f = b + (size_t)(i / PER) * (stride / 4);
            st16<MODE>(f, (unsigned)((i % PER) * 1024 + lane * 16), d);
            asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t1 = (long long)wall_clock64();
        q16 += t1 - t0;
    }
    if (lane == 0) { res[blockIdx.x * 3] = lat / reps; res[blockIdx.x * 3 + 1] = thr / reps; res[blockIdx.x * 3 + 2] = q16 / reps; }
}

int main(int argc, char** argv) {
    int N = argc > 1 ? atoi(argv[1]) : 504, PER = argc > 2 ? atoi(argv[2]) : 7, reps = 4;
    long long stride = argc > 3 ? atoll(argv[3]) : 745472;  // bytes between frames (832 rows x 224 x 4)
    size_t per_wg = (size_t)(N / PER + 1) * stride * 3 * reps;
    const int wgs_list[3] = {1, 52, 208};
    for (int wi = 0; wi < 3; ++wi) {
        int WGS = wgs_list[wi];
        // workgroups write interleaved row blocks of the same frames (like the scan): wg_stride = PER KiB inside a frame
        long long wg_stride = PER * 1024;
        if ((long long)WGS * wg_stride > stride) { printf("skip WGS %d\n", WGS); continue; }
        float* out; long long* res;
        CK(hipMalloc(&out, per_wg + (size_t)WGS * wg_stride + (1 << 20)));
        CK(hipMalloc(&res, WGS * 3 * sizeof(long long)));
        for (int mode = 0; mode < 4; ++mode) {
            for (int it = 0; it < 2; ++it) {
                if (mode == 0) k<0><<<WGS, 64>>>(out, res, N, PER, stride, wg_stride, reps);
                if (mode == 1) k<1><<<WGS, 64>>>(out, res, N, PER, stride, wg_stride, reps);
                if (mode == 2) k<2><<<WGS, 64>>>(out, res, N, PER, stride, wg_stride, reps);
                if (mode == 3) k<3><<<WGS, 64>>>(out, res, N, PER, stride, wg_stride, reps);
                CK(hipDeviceSynchronize());
            }
            std::vector<long long> h(WGS * 3);
            CK(hipMemcpy(h.data(), res, WGS * 3 * sizeof(long long), hipMemcpyDeviceToHost));
            double a = 0, b = 0, c = 0;
            for (int i = 0; i < WGS; ++i) { a += h[i * 3]; b += h[i * 3 + 1]; c += h[i * 3 + 2]; }
            a /= WGS; b /= WGS; c /= WGS;
            // wall_clock64: 100 MHz
            printf("WGS %3d mode %d (%s): one store %.2f us; %d stores back to back %.2f us = %.3f us per store, %.1f KB/us per wave; <=16 in flight %.3f us per store\n",
                   WGS, mode, mode == 0 ? "plain" : mode == 1 ? "sc1" : mode == 2 ? "nt" : "sc0 sc1", a / 100.0, N, b / 100.0, b / 100.0 / N,
                   N / (b / 100.0), c / 100.0 / N);
        }
        CK(hipFree(out)); CK(hipFree(res));
    }
    return 0;
}
