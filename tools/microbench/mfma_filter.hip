// Microbenchmark: the ray x leaf linear forms as a bf16 MFMA (v_mfma_f32_32x32x16_bf16) with the min3/max3 post-pass
// on the VALU.  One wave = 256 rays (8 B operands of 32 rays), one A block = 10 leaves x 3 forms.  Prints cycles per
// (wave, leaf).  Not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int i32x4;

template <int WAVES_EU>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES_EU, WAVES_EU)))
k_mfma(const i32x4 *__restrict__ ablocks, int ngroups, const i32x4 *__restrict__ rays, unsigned long long *out) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    bf16x8 B[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) { i32x4 v = rays[(wave * 8 + b) * 64 + lane]; B[b] = __builtin_bit_cast(bf16x8, v); }
    unsigned long long hits = 0;
    i32x4 a_next = ablocks[lane & 31];
    for (int g = 0; g < ngroups; ++g) {
        const bf16x8 A = __builtin_bit_cast(bf16x8, a_next);
        a_next = ablocks[(g + 1) * 32 + (lane & 31)];
        unsigned pend = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            f32x16 acc = {0};
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[b], acc, 0, 0, 0);
            const float z0 = __builtin_fminf(__builtin_fminf(acc[0], acc[1]), acc[2]);
            const float z1 = __builtin_fminf(__builtin_fminf(acc[3], acc[4]), acc[5]);
            const float z2 = __builtin_fminf(__builtin_fminf(acc[6], acc[7]), acc[8]);
            const float z3 = __builtin_fminf(__builtin_fminf(acc[9], acc[10]), acc[11]);
            const float z4 = __builtin_fminf(__builtin_fminf(acc[12], acc[13]), acc[14]);
            const float mx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(z0, z1), z2), __builtin_fmaxf(z3, z4));
            if (mx >= 0.f) pend |= 1u << b;
        }
        if (pend) hits += __popc(pend);
    }
    if (hits) atomicAdd(out, hits);
}

static unsigned short bf16(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main() {
    CK(hipSetDevice(0));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ngroups = 1600, nwaves = 256 * 4 * 24;   // 24 waves per SIMD in total
    std::vector<unsigned short> a((size_t)(ngroups + 1) * 32 * 8), r((size_t)nwaves * 8 * 64 * 8);
    srand(1);
    for (size_t i = 0; i < a.size(); i += 8) {          // rows: 3 coefficients hi/mid + constant -1 (never passes ... mostly)
        for (int k = 0; k < 6; ++k) a[i + k] = bf16((rand() / (float)RAND_MAX - 0.5f) * (k & 1 ? 0.004f : 1.f));
        a[i + 6] = bf16(-0.9f + ((i / 8) % 977 == 0 ? 1.5f : 0.f)); a[i + 7] = 0;
    }
    for (size_t i = 0; i < r.size(); i += 8) {
        const bool hi = ((i / 8) % 64) < 32;
        for (int k = 0; k < 6; ++k) r[i + k] = bf16((rand() / (float)RAND_MAX - 0.5f) * (hi ? 1.f : 0.004f));
        r[i + 6] = r[i + 7] = hi ? bf16(1.f) : 0;
    }
    void *da, *dr; unsigned long long *dout;
    CK(hipMalloc(&da, a.size() * 2)); CK(hipMalloc(&dr, r.size() * 2)); CK(hipMalloc(&dout, 8));
    CK(hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, r.data(), r.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 3; ++variant) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(dout, 0, 8));
            CK(hipEventRecord(e0));
            if (variant == 0) k_mfma<2><<<nwaves / 4, 256>>>((const i32x4 *)da, ngroups, (const i32x4 *)dr, dout);
            else if (variant == 1) k_mfma<4><<<nwaves / 4, 256>>>((const i32x4 *)da, ngroups, (const i32x4 *)dr, dout);
            else k_mfma<5><<<nwaves / 4, 256>>>((const i32x4 *)da, ngroups, (const i32x4 *)dr, dout);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h; CK(hipMemcpy(&h, dout, 8, hipMemcpyDeviceToHost));
            const double cyc = ms * 1e-3 * 2.33e9 * 1024 / ((double)nwaves * ngroups * 10);
            printf("waves/EU %d: %.3f ms  %.1f cycles per (wave, leaf) at 2.33 GHz  [%.1f per MFMA]  events %llu\n",
                   variant == 0 ? 2 : (variant == 1 ? 4 : 5), ms, cyc, cyc * 10 / 8, h);
        }
    }
    return 0;
}
