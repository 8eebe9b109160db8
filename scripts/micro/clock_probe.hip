// clock_probe.hip -- round 5: what does the shader clock do while bench.py's timed region runs?
//
// The round-4 review: every scan kernel runs 1.3-1.7x slower inside the twelve-forwards-in-flight region than alone on the chip, and
// nobody has measured the chip's clock there.  MI355X_MICROARCH.md ("DVFS give-back"): the part clocks to its power budget, and
// `s_memtime` ticks at the shader clock while `s_memrealtime` ticks at a constant 100 MHz.  This probe is ONE wave per workgroup that
// samples both counters every `interval` real-time ticks (sleeping in between: it burns nothing) for `n` samples and writes
// {memtime, memrealtime, hw_id, xcc_id} per sample.  shader MHz of an interval = d(memtime) / d(memrealtime) * 100.  Launched on its own
// stream BEFORE the work under test, so that its workgroups hold their slots (a 16-wave x 128-register scan workgroup leaves no room
// for another wave on its CU).  A second figure that does not trust `s_memtime`: a fixed dependent SALU chain timed on the 100 MHz
// clock (relative clock, polluted by issue arbitration when other waves share the SIMD -- reported next to the first).
//
// Build (shared object, loaded by scripts/diag_region_r05.py through ctypes):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o clock_probe.so clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, int n, int interval) {
    if (threadIdx.x != 0) return;
    unsigned long long* o = out + (size_t)blockIdx.x * n * 4;
    unsigned hw = 0, xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long next = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
        unsigned long long rt;
        do {
            __builtin_amdgcn_s_sleep(32);
            rt = __builtin_amdgcn_s_memrealtime();
        } while (rt < next);
        next = rt + interval;
        // the dependent scalar chain: 2048 s_add_u32 on one register, bracketed by the 100 MHz clock
        unsigned acc = (unsigned)rt;
        unsigned long long c0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long m0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int k = 0; k < 32; ++k)
            asm volatile(
                ".rept 64\n s_add_u32 %0, %0, 1\n .endr\n"
                : "+s"(acc)
                :
                : "scc");  // (s_add_u32 writes SCC: without the clobber the loop's own compare was overwritten and it never ended)
        unsigned long long m1 = __builtin_readcyclecounter();
        unsigned long long c1 = __builtin_amdgcn_s_memrealtime();
        o[4 * i + 0] = m0;
        o[4 * i + 1] = c0;
        o[4 * i + 2] = ((unsigned long long)(unsigned)(m1 - m0) << 32) | (unsigned)(c1 - c0);
        o[4 * i + 3] = ((unsigned long long)xcc << 32) | hw | ((unsigned long long)(acc & 1) << 63);
    }
}

extern "C" int clock_probe_launch(unsigned long long* out, int wgs, int n, int interval, void* stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(wgs), dim3(64), 0, (hipStream_t)stream, out, n, interval);
    return (int)hipGetLastError();
}
