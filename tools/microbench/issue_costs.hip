// Microbenchmark: what one wave64 vector instruction costs a gfx950 SIMD in issue cycles, for the instructions that
// dominate k_primary_mfma / k_shadow_mfma (profiles/r03_issue_costs.txt; bench.py weights its issue fraction with it).
// Not part of the product path.
//
// Method: W waves per SIMD (W = 1, 2, 4, 8; every CU gets W blocks of 256 threads = one wave per SIMD each) run a loop
// of 8 independent copies of ONE instruction, 1024 loops x 32 instructions; every wave stamps s_memtime (the shader
// clock, MI355X_MICROARCH.md "s_memtime tick = shader cycle") and the 100 MHz wall clock around its loop, which gives
// the clock the chip really ran at; the launch is timed with HIP events.  Reported per instruction and W: SIMD cycles
// per wave-instruction = launch time x clock / (W x instructions per wave) — the whole launch, so it does not matter
// whether the W waves of a SIMD really were resident together.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

constexpr int LOOPS = 1024, PER_LOOP = 32;   // (a wave runs for tens of microseconds: all W waves of a SIMD are resident together)

struct Stamp { unsigned long long cyc, wall; };

// 8 independent chains, 4 rounds per loop body
#define BODY8(INSTR) INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)
#define BODY32(INSTR) BODY8(INSTR) BODY8(INSTR) BODY8(INSTR) BODY8(INSTR)

#define KERNEL_BEGIN(name) \
__global__ void __launch_bounds__(256) name(Stamp *out, float seed) { \
    float a[8], x = seed + 1e-6f * threadIdx.x, y = 0.5f + x; \
    double d[8], dx = (double)x, dy = 1.0 + dx; \
    int n[8], ix = (int)threadIdx.x + 7, iy = ix * 3; \
    f2 p[8], px = {x, y}; \
    unsigned long long sm[8]; \
    for (int k = 0; k < 8; ++k) { a[k] = x + k; d[k] = dx + k; n[k] = ix + k; p[k] = f2{x + k, y + k}; sm[k] = 0; } \
    __shared__ float lds[1024]; lds[threadIdx.x] = x; __syncthreads(); \
    const unsigned la = (threadIdx.x & 255u) * 4u, la2 = (threadIdx.x & 63u) * 8u; (void)la; (void)la2; (void)dy; (void)iy; (void)px; \
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64(); \
    for (int it = 0; it < LOOPS; ++it) {
#define KERNEL_END \
    } \
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64(); \
    float s = 0; for (int k = 0; k < 8; ++k) s += a[k] + (float)d[k] + (float)n[k] + p[k][0] + p[k][1] + (float)(sm[k] & 1ull); \
    if (s == 12345.678f) lds[0] = s; \
    if ((threadIdx.x & 63u) == 0u) { out[(blockIdx.x * 256 + threadIdx.x) >> 6] = Stamp{c1 - c0, w1 - w0}; } \
}

#define I_FMA32(k)   asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(x), "v"(y));
#define I_PKFMA(k)   asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[k]) : "v"(px));
#define I_ADD32(k)   asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[k]) : "v"(x));
#define I_FMA64(k)   asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[k]) : "v"(dx), "v"(dy));
#define I_ADD64(k)   asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[k]) : "v"(dx));
#define I_MUL64(k)   asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d[k]) : "v"(dx));
#define I_MAX3I(k)   asm volatile("v_max3_i32 %0, %1, %2, %0" : "+v"(n[k]) : "v"(ix), "v"(iy));
#define B_CMPI asm volatile("v_cmp_gt_i32 %0, %8, %9\nv_cmp_gt_i32 %1, %8, %9\nv_cmp_gt_i32 %2, %8, %9\nv_cmp_gt_i32 %3, %8, %9\nv_cmp_gt_i32 %4, %8, %9\nv_cmp_gt_i32 %5, %8, %9\nv_cmp_gt_i32 %6, %8, %9\nv_cmp_gt_i32 %7, %8, %9" : "=s"(sm[0]), "=s"(sm[1]), "=s"(sm[2]), "=s"(sm[3]), "=s"(sm[4]), "=s"(sm[5]), "=s"(sm[6]), "=s"(sm[7]) : "v"(n[0]), "v"(ix));
#define B_CMPF asm volatile("v_cmp_ge_f32 %0, %8, %9\nv_cmp_ge_f32 %1, %8, %9\nv_cmp_ge_f32 %2, %8, %9\nv_cmp_ge_f32 %3, %8, %9\nv_cmp_ge_f32 %4, %8, %9\nv_cmp_ge_f32 %5, %8, %9\nv_cmp_ge_f32 %6, %8, %9\nv_cmp_ge_f32 %7, %8, %9" : "=s"(sm[0]), "=s"(sm[1]), "=s"(sm[2]), "=s"(sm[3]), "=s"(sm[4]), "=s"(sm[5]), "=s"(sm[6]), "=s"(sm[7]) : "v"(a[0]), "v"(x));
#define B_CMPD asm volatile("v_cmp_ge_f64 %0, %8, %9\nv_cmp_ge_f64 %1, %8, %9\nv_cmp_ge_f64 %2, %8, %9\nv_cmp_ge_f64 %3, %8, %9\nv_cmp_ge_f64 %4, %8, %9\nv_cmp_ge_f64 %5, %8, %9\nv_cmp_ge_f64 %6, %8, %9\nv_cmp_ge_f64 %7, %8, %9" : "=s"(sm[0]), "=s"(sm[1]), "=s"(sm[2]), "=s"(sm[3]), "=s"(sm[4]), "=s"(sm[5]), "=s"(sm[6]), "=s"(sm[7]) : "v"(d[0]), "v"(dx));
#define I_MBCNT(k)   asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(n[k]) : "v"(ix));
#define I_RSQ32(k)   asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k]));
#define I_RCP64(k)   asm volatile("v_rcp_f64 %0, %0" : "+v"(d[k]));
#define I_RSQ64(k)   asm volatile("v_rsq_f64 %0, %0" : "+v"(d[k]));
#define I_SQRT64(k)  asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[k]));
#define I_DIVSC(k)   asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(d[k]) : "v"(dx) : "vcc");
#define I_DIVFM(k)   asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(dx), "v"(dy) : "vcc");
#define I_DIVFX(k)   asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(dx), "v"(dy));
#define I_CVT_FD(k)  asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[k]) : "v"(d[k]));
#define I_CVT_DF(k)  asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(a[k]));
#define I_CNDM(k)    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[k]) : "v"(ix));
#define I_MOV(k)     asm volatile("v_mov_b32 %0, %1" : "=v"(n[k]) : "v"(ix));
#define I_LSHLOR(k)  asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(n[k]) : "v"(ix));
#define B_RFL { int t_[8]; asm volatile("v_readfirstlane_b32 %0, %8\nv_readfirstlane_b32 %1, %8\nv_readfirstlane_b32 %2, %8\nv_readfirstlane_b32 %3, %8\nv_readfirstlane_b32 %4, %8\nv_readfirstlane_b32 %5, %8\nv_readfirstlane_b32 %6, %8\nv_readfirstlane_b32 %7, %8" : "=s"(t_[0]), "=s"(t_[1]), "=s"(t_[2]), "=s"(t_[3]), "=s"(t_[4]), "=s"(t_[5]), "=s"(t_[6]), "=s"(t_[7]) : "v"(n[0])); for (int k = 0; k < 8; ++k) sm[k] = (unsigned)t_[k]; }
#define B_DSRD asm volatile("ds_read_b32 %0, %8\nds_read_b32 %1, %8\nds_read_b32 %2, %8\nds_read_b32 %3, %8\nds_read_b32 %4, %8\nds_read_b32 %5, %8\nds_read_b32 %6, %8\nds_read_b32 %7, %8\ns_waitcnt lgkmcnt(0)" : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7]) : "v"(la) : "memory");
#define B_DSWR asm volatile("ds_write_b32 %0, %1\nds_write_b32 %0, %2\nds_write_b32 %0, %3\nds_write_b32 %0, %4\nds_write_b32 %0, %5\nds_write_b32 %0, %6\nds_write_b32 %0, %7\nds_write_b32 %0, %8\ns_waitcnt lgkmcnt(0)" :: "v"(la), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
#define B_DSRD64 asm volatile("ds_read_b64 %0, %8\nds_read_b64 %1, %8\nds_read_b64 %2, %8\nds_read_b64 %3, %8\nds_read_b64 %4, %8\nds_read_b64 %5, %8\nds_read_b64 %6, %8\nds_read_b64 %7, %8\ns_waitcnt lgkmcnt(0)" : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]) : "v"(la2) : "memory");
#define X4(B) B B B B

KERNEL_BEGIN(k_fma32)  BODY32(I_FMA32)  KERNEL_END
KERNEL_BEGIN(k_pkfma)  BODY32(I_PKFMA)  KERNEL_END
KERNEL_BEGIN(k_add32)  BODY32(I_ADD32)  KERNEL_END
KERNEL_BEGIN(k_fma64)  BODY32(I_FMA64)  KERNEL_END
KERNEL_BEGIN(k_add64)  BODY32(I_ADD64)  KERNEL_END
KERNEL_BEGIN(k_mul64)  BODY32(I_MUL64)  KERNEL_END
KERNEL_BEGIN(k_max3i)  BODY32(I_MAX3I)  KERNEL_END
KERNEL_BEGIN(k_cmpi) X4(B_CMPI) KERNEL_END
KERNEL_BEGIN(k_cmpf) X4(B_CMPF) KERNEL_END
KERNEL_BEGIN(k_cmpd) X4(B_CMPD) KERNEL_END
KERNEL_BEGIN(k_mbcnt)  BODY32(I_MBCNT)  KERNEL_END
KERNEL_BEGIN(k_rsq32)  BODY32(I_RSQ32)  KERNEL_END
KERNEL_BEGIN(k_rcp64)  BODY32(I_RCP64)  KERNEL_END
KERNEL_BEGIN(k_rsq64)  BODY32(I_RSQ64)  KERNEL_END
KERNEL_BEGIN(k_sqrt64) BODY32(I_SQRT64) KERNEL_END
KERNEL_BEGIN(k_divsc)  BODY32(I_DIVSC)  KERNEL_END
KERNEL_BEGIN(k_divfm)  BODY32(I_DIVFM)  KERNEL_END
KERNEL_BEGIN(k_divfx)  BODY32(I_DIVFX)  KERNEL_END
KERNEL_BEGIN(k_cvtfd)  BODY32(I_CVT_FD) KERNEL_END
KERNEL_BEGIN(k_cvtdf)  BODY32(I_CVT_DF) KERNEL_END
KERNEL_BEGIN(k_cndm)   BODY32(I_CNDM)   KERNEL_END
KERNEL_BEGIN(k_mov)    BODY32(I_MOV)    KERNEL_END
KERNEL_BEGIN(k_lshlor) BODY32(I_LSHLOR) KERNEL_END
KERNEL_BEGIN(k_rfl) X4(B_RFL) KERNEL_END
KERNEL_BEGIN(k_dsrd) X4(B_DSRD) KERNEL_END
KERNEL_BEGIN(k_dsrd64) X4(B_DSRD64) KERNEL_END
KERNEL_BEGIN(k_dswr) X4(B_DSWR) KERNEL_END

// the real thing, compiled from C: one IEEE double division and one square root (what leaf_isect / to_light are made of)
__global__ void __launch_bounds__(256) k_div64(Stamp *out, float seed) {
    double d[8], dx = 1.000001 + (double)seed + 1e-9 * threadIdx.x;
    for (int k = 0; k < 8; ++k) d[k] = dx + k;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < LOOPS; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) { d[k] = d[k] / dx; asm volatile("" : "+v"(d[k])); }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    double s = 0; for (int k = 0; k < 8; ++k) s += d[k];
    if (s == 12345.678) out[0].cyc = 0;
    if ((threadIdx.x & 63u) == 0u) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = Stamp{c1 - c0, w1 - w0};
}
__global__ void __launch_bounds__(256) k_sqrtdiv64(Stamp *out, float seed) {
    double d[8], dx = 1.000001 + (double)seed + 1e-9 * threadIdx.x;
    for (int k = 0; k < 8; ++k) d[k] = dx + k;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < LOOPS; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) { d[k] = sqrt(d[k] + dx); asm volatile("" : "+v"(d[k])); }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    double s = 0; for (int k = 0; k < 8; ++k) s += d[k];
    if (s == 12345.678) out[0].cyc = 0;
    if ((threadIdx.x & 63u) == 0u) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = Stamp{c1 - c0, w1 - w0};
}
// one MFMA + F vector instructions per step (F = 0, 4, 6, 8, 10): what an MFMA holds of the vector issue port
template <int F>
__global__ void __launch_bounds__(256) k_mfma_mix(Stamp *out, float seed) {
    f16x8 A, B;
    for (int k = 0; k < 8; ++k) { A[k] = (_Float16)(seed + k); B[k] = (_Float16)(1.f + k); }
    f32x16 acc0 = {0}, acc1 = {0};
    int n[8], ix = (int)threadIdx.x + 7, iy = ix * 3;
    for (int k = 0; k < 8; ++k) n[k] = ix + k;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < LOOPS; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < F; ++k) asm volatile("v_max3_i32 %0, %1, %2, %0" : "+v"(n[k & 7]) : "v"(ix), "v"(iy));
            __builtin_amdgcn_sched_barrier(0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < F; ++k) asm volatile("v_max3_i32 %0, %1, %2, %0" : "+v"(n[k & 7]) : "v"(ix), "v"(iy));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = acc0[0] + acc1[3]; for (int k = 0; k < 8; ++k) s += (float)n[k];
    if (s == 12345.678f) out[0].cyc = 0;
    if ((threadIdx.x & 63u) == 0u) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = Stamp{c1 - c0, w1 - w0};
}

typedef void (*kern_t)(Stamp *, float);
struct Test { const char *name; kern_t fn; int per_wave; const char *cls; };

int main() {
    CK(hipSetDevice(0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# device %s, %d CUs; %d loops x %d instructions per wave; cycles = s_memtime, wall = 100 MHz wall_clock64\n", prop.name, cus, LOOPS, PER_LOOP);
    printf("# SIMD cycles per wave64 instruction at W waves per SIMD = launch time (HIP events) x shader clock / (W x instructions per wave);\n");
    printf("# shader clock = s_memtime ticks / wall-clock time of the waves' loops; 'alone' = one wave's own loop, cycles per instruction\n");
    printf("%-26s %-8s %8s %8s %8s %8s %8s   %s\n", "instruction", "class", "alone", "W=1", "W=2", "W=4", "W=8", "clock GHz (W=4)");
    const int N = LOOPS * PER_LOOP;
    const Test tests[] = {
        {"v_fma_f32", k_fma32, N, "fp32"}, {"v_pk_fma_f32", k_pkfma, N, "fp32"}, {"v_add_f32", k_add32, N, "fp32"},
        {"v_fma_f64", k_fma64, N, "fp64"}, {"v_add_f64", k_add64, N, "fp64"}, {"v_mul_f64", k_mul64, N, "fp64"},
        {"v_max3_i32", k_max3i, N, "int"}, {"v_cmp_gt_i32 (sgpr dst)", k_cmpi, N, "int"}, {"v_cmp_ge_f32 (sgpr dst)", k_cmpf, N, "rest"},
        {"v_cmp_ge_f64 (sgpr dst)", k_cmpd, N, "fp64"}, {"v_mbcnt_lo_u32_b32", k_mbcnt, N, "int"},
        {"v_rsq_f32", k_rsq32, N, "trans"}, {"v_rcp_f64", k_rcp64, N, "trans"}, {"v_rsq_f64", k_rsq64, N, "trans"}, {"v_sqrt_f64", k_sqrt64, N, "trans"},
        {"v_div_scale_f64", k_divsc, N, "fp64"}, {"v_div_fmas_f64", k_divfm, N, "fp64"}, {"v_div_fixup_f64", k_divfx, N, "fp64"},
        {"v_cvt_f32_f64", k_cvtfd, N, "fp64"}, {"v_cvt_f64_f32", k_cvtdf, N, "fp64"},
        {"v_cndmask_b32", k_cndm, N, "rest"}, {"v_mov_b32", k_mov, N, "rest"}, {"v_lshl_or_b32", k_lshlor, N, "int"}, {"v_readfirstlane_b32", k_rfl, N, "rest"},
        {"ds_read_b32", k_dsrd, N, "lds"}, {"ds_read_b64", k_dsrd64, N, "lds"}, {"ds_write_b32", k_dswr, N, "lds"},
        {"a / b (f64, C)", k_div64, N, "fp64"}, {"sqrt(a + b) (f64, C)", k_sqrtdiv64, N, "fp64"},
        {"mfma_32x32x16_f16 alone", k_mfma_mix<0>, LOOPS * 32, "mfma"},
        {"mfma + 4 v_max3_i32", k_mfma_mix<4>, LOOPS * 32, "mfma"}, {"mfma + 6 v_max3_i32", k_mfma_mix<6>, LOOPS * 32, "mfma"},
        {"mfma + 8 v_max3_i32", k_mfma_mix<8>, LOOPS * 32, "mfma"}, {"mfma + 10 v_max3_i32", k_mfma_mix<10>, LOOPS * 32, "mfma"},
    };
    Stamp *out; CK(hipMalloc(&out, sizeof(Stamp) * cus * 8 * 4));
    std::vector<Stamp> h((size_t)cus * 8 * 4);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Test &t : tests) {
        double res[4] = {0, 0, 0, 0}, solo = 0, ghz = 0;
        for (int wi = 0; wi < 4; ++wi) {
            const int W = 1 << wi, blocks = cus * W;
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {     // the last repetition counts (clocks ramped, code cached)
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, 0.999f);
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            CK(hipMemcpy(h.data(), out, sizeof(Stamp) * blocks * 4, hipMemcpyDeviceToHost));
            double cyc = 0, wall = 0;
            for (int k = 0; k < blocks * 4; ++k) { cyc += (double)h[k].cyc; wall += (double)h[k].wall; }
            cyc /= blocks * 4; wall /= blocks * 4;
            const double clock = cyc / (wall * 10.0);                 // shader cycles per ns while the loop ran
            // the whole launch: every SIMD issued W x per_wave instructions in (launch time x clock) cycles
            res[wi] = (double)ms * 1e6 * clock / ((double)W * t.per_wave);
            if (W == 1) solo = cyc / t.per_wave;
            if (W == 4) ghz = clock;
        }
        printf("%-26s %-8s %8.2f %8.2f %8.2f %8.2f %8.2f   %.2f\n", t.name, t.cls, solo, res[0], res[1], res[2], res[3], ghz);
    }
    printf("# rows 'mfma + F v_max3_i32': cycles per (MFMA + F vector instructions) step\n");
    return 0;
}
