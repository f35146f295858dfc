// Microbenchmark: how many work-groups per microsecond does an MI355X start?  (profiles/r03_dispatch_rate.txt)
// Not part of the product path.
//
// k_primary_mfma and k_shadow_mfma run one wave per work-group (rt_params.h) with waves of 15 to 200 us; the per-wave
// timeline (tools/wave_times.py) shows at most ~85 wave starts per microsecond while wave slots stand empty.  This
// program measures that rate in isolation: N work-groups of `threads` threads that use `lds` bytes of LDS and a
// register budget that allows `occ` waves per SIMD, each spinning for T microseconds of wall clock; the launch takes
//     max(N * T / slots, N / rate) + a start-up term,
// so for short T the time per work-group is 1 / rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// REGS: 0 = as few registers as the compiler needs; 1 = the wave really holds 128 (OCC 4) / 168 (OCC 3) VGPRs, as the render
// kernels do (an asm statement that names a high register); 2 = ... and a few bytes of scratch per lane (a spill slot)
template <int LDS_WORDS, int OCC, int REGS>
__global__ void __attribute__((amdgpu_waves_per_eu(OCC, OCC))) k_spin(unsigned ticks, unsigned *sink) {
    __shared__ unsigned s[LDS_WORDS > 0 ? LDS_WORDS : 1];
    const unsigned long long t0 = wall_clock64();
    if (LDS_WORDS > 0) s[threadIdx.x % (LDS_WORDS > 0 ? LDS_WORDS : 1)] = threadIdx.x;
    if (REGS >= 1) { if (OCC >= 4) asm volatile("v_mov_b32 v127, 0" ::: "v127"); else asm volatile("v_mov_b32 v160, 0" ::: "v160"); }
    volatile unsigned priv[REGS >= 2 ? 8 : 1];
    if (REGS >= 2) priv[threadIdx.x & 7] = threadIdx.x;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks == 0xffffffffu) sink[blockIdx.x] = s[0] + (REGS >= 2 ? priv[ticks & 7] : 0u);
}

template <int LDS_WORDS, int OCC, int REGS = 0>
static void run(const char *name, int threads, unsigned *sink) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 1 << 16;
    printf("%s: %d work-groups of %d thread(s), %d bytes of LDS, %d wave(s) per SIMD allowed\n", name, N, threads, LDS_WORDS * 4, OCC);
    for (unsigned us : {0u, 2u, 5u, 10u, 20u, 40u, 80u}) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_spin<LDS_WORDS, OCC, REGS>), dim3(N), dim3(threads), 0, 0, us * 100u, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double slots = 1024.0 * OCC / ((threads + 63) / 64);      // work-groups resident at once (1024 SIMDs)
        const double ideal_us = (double)N * us / slots;
        printf("   spin %3u us: launch %8.1f us = %6.1f work-groups/us (%.1f waves/us); slots alone would need %8.1f us -> slots %4.0f %% full\n",
               us, best * 1e3, N / (best * 1e3), N * ((threads + 63) / 64) / (best * 1e3), ideal_us, best > 0 ? 100.0 * ideal_us / (best * 1e3) : 0.0);
    }
}

int main() {
    unsigned *sink; CK(hipMalloc(&sink, (1 << 16) * sizeof(unsigned)));
    run<0, 8>("no LDS, 8 waves/SIMD", 64, sink);
    run<2048, 4>("8 KiB LDS, 4 waves/SIMD (k_primary_mfma's shape)", 64, sink);
    run<2816, 3>("11 KiB LDS, 3 waves/SIMD (k_shadow_mfma's shape)", 64, sink);
    run<2048, 4, 1>("8 KiB LDS, 128 VGPRs held (4 waves/SIMD)", 64, sink);
    run<2048, 4, 2>("8 KiB LDS, 128 VGPRs held, scratch in use", 64, sink);
    run<2816, 3, 1>("11 KiB LDS, 168 VGPRs held (3 waves/SIMD)", 64, sink);
    run<2048, 4>("8 KiB LDS, 4 waves/SIMD, 2 waves per work-group", 128, sink);
    run<2048, 4>("8 KiB LDS, 4 waves/SIMD, 4 waves per work-group", 256, sink);
    CK(hipFree(sink));
    return 0;
}
