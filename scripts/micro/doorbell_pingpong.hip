// Doorbell floor for a resident kernel (round 3): host rings a word, a resident wave answers into pinned host memory.
//   variant 0: doorbell in pinned host memory, polled by the GPU over PCIe (relaxed system-scope loads)
//   variant 1: doorbell in fine-grained DEVICE memory written by the CPU through the BAR (if the platform maps it), polled locally
// Also: ring -> NW workgroups polling -> last one answers (fan-out cost), and a 512-byte payload read from host memory after the ring.
// hipcc --offload-arch=gfx950 -O2 -o /tmp/doorbell scripts/micro/doorbell_pingpong.hip && /tmp/doorbell
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void resident(const unsigned* bell, unsigned* ack, unsigned* fan, const float* payload, int read_payload, unsigned n_hops, unsigned long long limit) {
    const unsigned long long t0 = wall_clock64();
    for (unsigned k = 1; k <= n_hops; ++k) {
        unsigned v;
        for (;;) {
            v = __hip_atomic_load(bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v >= k) break;
            if (wall_clock64() - t0 > limit) return;
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        float s = 0.f;
        if (read_payload) s = payload[threadIdx.x & 127];
        // fan-in: the last workgroup to arrive answers
        unsigned arrived = 0;
        if (threadIdx.x == 0) arrived = __hip_atomic_fetch_add(fan, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        arrived = __builtin_amdgcn_readfirstlane(arrived);
        if (arrived == k * gridDim.x && threadIdx.x == 0) {
            if (s == 12345.f) ack[1] = 1;
            __hip_atomic_store(ack, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

static void run(const char* name, volatile unsigned* bell_host_view, unsigned* bell_dev, int nwg, int read_payload) {
    unsigned *ack, *fan;
    float* payload;
    CK(hipHostMalloc((void**)&ack, 64, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&payload, 512, hipHostMallocDefault));
    CK(hipMalloc((void**)&fan, 64));
    CK(hipMemset(fan, 0, 64));
    ack[0] = 0; ack[1] = 0;
    *bell_host_view = 0;
    const unsigned n = 20000;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipLaunchKernelGGL(resident, dim3(nwg), dim3(64), 0, st, bell_dev, ack, fan, payload, read_payload, n, 300000000ull /* 3 s */);
    std::vector<double> lat;
    volatile unsigned* a = ack;
    for (unsigned k = 1; k <= n; ++k) {
        payload[k & 127] = (float)k;
        auto t0 = std::chrono::steady_clock::now();
        __atomic_store_n((unsigned*)bell_host_view, k, __ATOMIC_RELEASE);
        while (*a != k) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 1.0) { printf("%s: timeout at %u\n", name, k); goto out; }
        }
        lat.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
out:
    CK(hipStreamSynchronize(st));
    if (!lat.empty()) {
        std::sort(lat.begin(), lat.end());
        printf("%-46s wgs %3d payload %d: ring->ack p50 %.2f us  p99 %.2f  min %.2f\n", name, nwg, read_payload, lat[lat.size() / 2], lat[lat.size() * 99 / 100], lat[0]);
    }
    CK(hipFree(fan)); CK(hipHostFree(ack)); CK(hipHostFree(payload));
}

int main() {
    unsigned* hb;
    CK(hipHostMalloc((void**)&hb, 64, hipHostMallocDefault));
    for (int nwg : {1, 23}) for (int pl : {0, 1}) run("doorbell in pinned host memory (GPU polls PCIe)", hb, hb, nwg, pl);
    unsigned* db = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&db, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
        CK(hipMemset(db, 0, 4096));
        CK(hipDeviceSynchronize());
        signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
        if (sigsetjmp(jb, 1) == 0) {
            volatile unsigned* v = db;
            *v = 0;  // CPU store into device memory through the BAR
            unsigned r = *v;
            printf("CPU can write fine-grained device memory directly (read back %u)\n", r);
            for (int nwg : {1, 23}) for (int pl : {0, 1}) run("doorbell in device memory (CPU writes the BAR)", db, db, nwg, pl);
        } else {
            printf("CPU store into fine-grained device memory faults: not mapped for the host on this platform\n");
        }
    }
    unsigned* mb = nullptr;
    e = hipMallocManaged((void**)&mb, 4096);
    printf("hipMallocManaged: %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
        int dev = 0;
        hipMemAdvise(mb, 4096, hipMemAdviseSetPreferredLocation, dev);
        hipMemAdvise(mb, 4096, hipMemAdviseSetAccessedBy, hipCpuDeviceId);
        CK(hipMemset(mb, 0, 4096));
        CK(hipDeviceSynchronize());
        if (sigsetjmp(jb, 1) == 0) {
            for (int nwg : {1, 23}) run("doorbell in managed memory (preferred: device)", mb, mb, nwg, 0);
        } else {
            printf("managed memory: fault\n");
        }
    }
    return 0;
}
