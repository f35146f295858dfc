// rt_device.h — device code shared by the product kernels (rt_kernels.hip) and the cross-check build's vector-ALU
// kernels (rt_kernels_valu.hip): the exact IEEE-double path with the reference's expression order (mkray, cast,
// planeRayIsect and the inside tests, intersectBB guard chains: Main.hs:107-127, 152-174, 226-277, 363-385, 506-533), pixel
// arithmetic and pointLight (Main.hs:445-487), the scalar-record helpers of the FP32 filters, counters.
//
// Measurement switches that CHANGE RESULTS (-DRT_NO_GUARDS, -DRT_X_NOSCAN, -DRT_X_SCANONLY, -DRT_MF_NOSLOW, -DRT_NO_DEPTH_SKIP,
// -DRT_NO_TALK) are honoured only together with -DRT_MEASUREMENT_BUILD, and rt_version() then says so.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

#include "rt_params.h"

#if !defined(RT_MEASUREMENT_BUILD) && (defined(RT_NO_GUARDS) || defined(RT_X_NOSCAN) || defined(RT_X_SCANONLY) || defined(RT_MF_NOSLOW) || \
                                       defined(RT_NO_DEPTH_SKIP) || defined(RT_NO_TALK))
#error "result-changing measurement switches need -DRT_MEASUREMENT_BUILD (rt_version() reports such a build)"
#endif

using namespace rt;

typedef float f2 __attribute__((ext_vector_type(2)));
// Wave-uniform, read-only scene data is addressed through the constant address space, which makes
// hipcc fetch it with scalar loads (s_load_dwordx8 ...) no matter what else the kernel stores.
#define RT_CONST __attribute__((address_space(4)))
typedef const rt::FilterRec RT_CONST *CRecPtr;
typedef const int32_t RT_CONST *CIntPtr;

#ifndef RT_SHADOW_CLIP
#define RT_SHADOW_CLIP 0       // 1: shadow filter tests "not beyond the ray origin" per leaf (6 more VALU per 4 rays)
#endif
#ifndef RT_PREFETCH
#define RT_PREFETCH 1          // double-buffer the leaf records in SGPRs
#endif
// occupancy (waves per SIMD) the register allocator must reach for the two brute-force kernels
#ifndef RT_WAVES_EU_PRIMARY
#define RT_WAVES_EU_PRIMARY 8
#endif
#ifndef RT_WAVES_EU_SHADOW
#define RT_WAVES_EU_SHADOW 6
#endif
#define RT_HOT(n) __attribute__((amdgpu_waves_per_eu(n, 8))) __launch_bounds__(RT_WAVES_PER_BLOCK * 64)

// RT_TALK(P): is the scene in the Talk.hs dialect?  (-DRT_NO_TALK compiles the dialect out: measurement only.)
#ifdef RT_NO_TALK
#define RT_TALK(P) false
#else
#define RT_TALK(P) ((P).talk != 0)
#endif

namespace {

// ------------------------------------------------------------------------------------------
// Exact path (IEEE double, reference expression order)

struct RayD { double px, py, pz, dx, dy, dz, rx, ry, rz; };

__device__ __forceinline__ double clampd(double d, double eps) { return fabs(d) <= eps ? eps : d; } // Main.hs:112

// mkray position direction — Main.hs:107-115
__device__ __forceinline__ RayD mkray(double px, double py, double pz, double dx, double dy, double dz, double eps) {
    RayD r;
    r.px = px; r.py = py; r.pz = pz;
    r.dx = clampd(dx, eps); r.dy = clampd(dy, eps); r.dz = clampd(dz, eps);
    r.rx = 1.0 / r.dx; r.ry = 1.0 / r.dy; r.rz = 1.0 / r.dz;
    return r;
}

// jittered supersampling offsets (extension): spp == 1 -> (0, 0) = the reference
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}
__device__ __forceinline__ void sample_offset(const RenderParams &P, int x, int y, double &ox, double &oy) {
    if (P.spp <= 1) { ox = 0; oy = 0; return; }
    uint32_t h = mix32(P.seed_lo ^ mix32(P.seed_hi + 0x9e3779b9u));
    h = mix32(h ^ (uint32_t)x);
    h = mix32(h ^ ((uint32_t)y + 0x9e3779b9u));
    h = mix32(h ^ ((uint32_t)P.sample + 0x7f4a7c15u));
    const uint32_t h2 = mix32(h ^ 0x68bc21ebu);
    const double u1 = (double)(h >> 8) * (1.0 / 16777216.0), u2 = (double)(h2 >> 8) * (1.0 / 16777216.0);
    const int k = (int)floor(sqrt((double)P.spp) + 0.5);
    if (k * k == P.spp) {
        const int sx = P.sample % k, sy = P.sample / k;
        ox = ((double)sx + u1) / (double)k - 0.5;
        oy = ((double)sy + u2) / (double)k - 0.5;
    } else { ox = u1 - 0.5; oy = u2 - 0.5; }
}

// cast camera x y — Main.hs:528-533
template <bool MAY_TALK = true>
__device__ __forceinline__ RayD camera_ray(const RenderParams &P, int x, int y) {
    double ox, oy;
    sample_offset(P, x, y, ox, oy);
    const double posX = P.cam.scaleX * ((double)x + ox) + P.cam.dX;
    const double posY = P.cam.scaleY * ((double)y + oy) + P.cam.dY;
    const double n = sqrt(posX * posX + posY * posY + P.cam.d * P.cam.d);
    if (MAY_TALK && RT_TALK(P)) {                       // Talk.hs:66-67, 152: scalar (recip (norm v)) v
        const double rn = 1.0 / n;
        return mkray(posX, posY, 0.0, rn * posX, rn * posY, rn * P.cam.d, P.eps);
    }
    return mkray(posX, posY, 0.0, posX / n, posY / n, P.cam.d / n, P.eps);
}

__device__ __forceinline__ void stat_add(const RenderParams &P, int k, unsigned long long v) {
    const unsigned wave = (blockIdx.x + blockIdx.y * gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    atomicAdd(&P.counters[(size_t)(wave % RT_STAT_SLOTS) * 16 + k], v);
}
#ifdef RT_WAVE_TIMES
// measurement builds: one record per wave {kernel | light << 8 | hardware id << 16, wave, start, end (100 MHz wall clock),
// blocks scanned, record tests, exact evaluations, phases}, in a place of its own (`slot`: the launch's work-group number,
// offset per kernel) — a counter that hands out places is ONE address for every wave of the frame, and the L2 serialises
// atomics on one address at 12 ns each: 83 waves per microsecond, which is about the rate these kernels end waves at
// (the first version of this tool measured its own counter).  Records that were not written have end = 0.
__device__ __forceinline__ void wave_record(const RenderParams &P, unsigned kernel, unsigned long long slot, unsigned wave, unsigned long long t0,
                                            unsigned nblocks, unsigned nrec, unsigned nexact, const unsigned long long (&ph)[3]) {
    if (!P.dbg || (threadIdx.x & 63u) != 0u) return;
    const unsigned long long t1 = wall_clock64();
    const unsigned long long k = slot;
    if (k >= (1ull << 20)) return;
    unsigned long long *r = P.dbg + 8 + 8 * k;
    // where the wave ran: HW_ID (wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]) and XCC_ID [3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
    r[0] = (unsigned long long)kernel | ((unsigned long long)(hw & 0xffffu) << 16) | ((unsigned long long)(xcc & 0xfu) << 32);
    r[1] = wave; r[2] = t0; r[3] = t1;
    r[4] = (unsigned long long)nblocks | ((unsigned long long)nrec << 16) | ((unsigned long long)nexact << 40);
    r[5] = ph[0]; r[6] = ph[1]; r[7] = ph[2];      // phase boundaries: after the ray set-up, after bound + guard proofs, after the scan
}
#define RT_WAVE_T0 const unsigned long long wave_t0 = wall_clock64(); unsigned long long wave_ph[3] = {0, 0, 0};
#define RT_WAVE_PHASE(k) wave_ph[k] = wall_clock64();
#else
#define RT_WAVE_T0
#define RT_WAVE_PHASE(k)
#endif

__device__ __forceinline__ unsigned num_slots(const RenderParams &P) {
    if (!P.count_ptr) return (unsigned)P.qcap;
    const unsigned c = *P.count_ptr;
    return c < (unsigned)P.lcap ? c : (unsigned)P.lcap;
}
// the ray of a slot: `cast camera x y` at level 0, `mkray (ipos + 1e-5 rdir) rdir` (Main.hs:552) above
// A level's ray list in memory: runs of 64 slots, structure-of-arrays inside a run (8 fields x 64 x 8 bytes = 4 KiB per
// run: origin, direction, reflectivity, parent slot), so that a wave reading or writing the rays of a run touches 512
// contiguous bytes per field.  (As an array of 72-byte structures every field access of a wave scattered over 36 cache
// lines: k_final spent 0.4 of its 0.47 ms per chunk writing the reflected rays.)  RayRec is the value in registers.
__device__ __forceinline__ size_t ray_index(unsigned slot, int field) { return ((size_t)(slot >> 6) * 8 + (size_t)field) * 64 + (slot & 63u); }
__device__ __forceinline__ uint32_t ray_parent(const RayRec *rays, unsigned slot) {
    return (uint32_t)reinterpret_cast<const unsigned long long *>(rays)[ray_index(slot, 7)];
}
__device__ __forceinline__ RayRec load_ray(const RayRec *rays, unsigned slot) {
    const double *p = reinterpret_cast<const double *>(rays);
    RayRec R;
    R.ox = p[ray_index(slot, 0)]; R.oy = p[ray_index(slot, 1)]; R.oz = p[ray_index(slot, 2)];
    R.dx = p[ray_index(slot, 3)]; R.dy = p[ray_index(slot, 4)]; R.dz = p[ray_index(slot, 5)];
    R.rho = p[ray_index(slot, 6)];
    R.parent = ray_parent(rays, slot); R._pad = 0u;
    return R;
}
__device__ __forceinline__ void store_ray(RayRec *rays, unsigned slot, const RayRec &R) {
    double *p = reinterpret_cast<double *>(rays);
    p[ray_index(slot, 0)] = R.ox; p[ray_index(slot, 1)] = R.oy; p[ray_index(slot, 2)] = R.oz;
    p[ray_index(slot, 3)] = R.dx; p[ray_index(slot, 4)] = R.dy; p[ray_index(slot, 5)] = R.dz;
    p[ray_index(slot, 6)] = R.rho;
    reinterpret_cast<unsigned long long *>(rays)[ray_index(slot, 7)] = (unsigned long long)R.parent;
}

__device__ __forceinline__ RayD slot_ray(const RenderParams &P, int slot) {
    if (P.rays) {
        const RayRec R = load_ray(P.rays, (unsigned)slot);
        return mkray(R.ox, R.oy, R.oz, R.dx, R.dy, R.dz, P.eps);
    }
    const int ly = slot / P.w;
    return camera_ray(P, slot - ly * P.w, P.row0 + ly);
}

// planeRayIsect + the inside test of rectangle (Main.hs:234-242) / triangle (Main.hs:374-383).
// KIND is the leaf type; L is wave-uniform in the filter loops (scalar loads).
template <int KIND>
__device__ __forceinline__ bool leaf_isect(const LeafD &L, const RayD &r, double &ix, double &iy, double &iz, double &t, bool talk) {
    if (KIND == LEAF_SPHERE) {                                                         // sphere, Main.hs:395-411
        const double ox = r.px - L.p0x, oy = r.py - L.p0y, oz = r.pz - L.p0z;          // :396
        const double sb = 2 * (r.dx * ox + r.dy * oy + r.dz * oz);                     // :398
        const double sc = (ox * ox + oy * oy + oz * oz) - L.c0 * L.c0;                 // :399
        const double delta = sb * sb - 4 * sc;                                         // :400
        if (!(delta >= 0)) return false;                                               // :401
        double ts;
        if (delta > 0) {
            const double sq = sqrt(delta);
            const double t1 = (-sb + sq) / 2, t2 = (-sb - sq) / 2;                     // :404-405
            const double m1 = t1 <= 0 ? 0.0 : t1, m2 = t2 <= 0 ? 0.0 : t2;             // max t 0
            ts = m1 <= m2 ? m1 : m2;                                                   // min, :406
        } else ts = -sb / 2;                                                           // :407
        if (!(ts > 0)) return false;                                                   // :408
        t = ts;
        ix = r.px + ts * r.dx; iy = r.py + ts * r.dy; iz = r.pz + ts * r.dz;           // :409
        return true;
    }
    const double a = L.nx, b = L.ny, c = L.nz;
    const double frac = (a * r.dx + b * r.dy + c * r.dz) / r.dz;                       // Main.hs:269
    const double z = (-L.d - a * r.px - b * r.py + (frac - c) * r.pz) / frac;          // :270
    const double x = (r.dx * (z - r.pz) / r.dz) + r.px;                                // :271
    const double y = (r.dy * (z - r.pz) / r.dz) + r.py;                                // :272
    if (fabs(frac) <= 0.00001) return false;                                           // :273
    if (talk) {                       // Talk.hs:121-122: ray_direction *@ (i - ray_position) >= 0; key = squared distance, :192-193
        const double vx = x - r.px, vy = y - r.py, vz = z - r.pz;
        if (!(r.dx * vx + r.dy * vy + r.dz * vz >= 0)) return false;
        t = vx * vx + vy * vy + vz * vz;
    } else {
        t = (z - r.pz) * r.rz;                                                         // :256, 122-126
        if (!(t >= 0)) return false;                                                   // :257
    }
    ix = x; iy = y; iz = z;
    const double wx = x - L.p0x, wy = y - L.p0y, wz = z - L.p0z;
    if (KIND == LEAF_RECT) {
        const double dw = wx * L.e1x + wy * L.e1y + wz * L.e1z;                        // :239
        const double dh = wx * L.e2x + wy * L.e2y + wz * L.e2z;                        // :240
        return 0 <= dw && dw <= L.c0 && 0 <= dh && dh <= L.c1;                         // :241
    } else {
        const double wv = wx * L.e2x + wy * L.e2y + wz * L.e2z;                        // :377
        const double wu = wx * L.e1x + wy * L.e1y + wz * L.e1z;                        // :378
        const double rr = (L.c0 * wv - L.c2 * wu) / L.c3;                              // :379
        const double ss = (L.c0 * wu - L.c1 * wv) / L.c3;                              // :380
        return rr >= 0 && ss >= 0 && rr + ss <= 1;                                     // :382
    }
}

// intersectBB — Main.hs:152-174 — for every mappend node above the leaf (Main.hs:209).
// The reference ORs six face tests; the result does not depend on their order.  A ray that meets a box leaves it
// through one of the three faces it points at, with t >= 0, so those are tried first, and the remaining tests are
// skipped once every active lane of the wave has found a face (wave-uniform branches): about two face tests per
// box instead of six, the same six when in doubt.
#ifndef RT_GUARD_EARLY
#define RT_GUARD_EARLY 1
#endif
__device__ __forceinline__ bool face_xy(const GuardD &B, double zf, const RayD &r) {       // isectXY
    const double t = (zf - r.pz) * r.rz;
    const double a = r.px + t * r.dx, c = r.py + t * r.dy;
    return t >= 0 && B.x0 <= a && a <= B.x1 && B.y0 <= c && c <= B.y1;
}
__device__ __forceinline__ bool face_xz(const GuardD &B, double yf, const RayD &r) {       // isectXZ
    const double t = (yf - r.py) * r.ry;
    const double a = r.px + t * r.dx, c = r.pz + t * r.dz;
    return t >= 0 && B.x0 <= a && a <= B.x1 && B.z0 <= c && c <= B.z1;
}
__device__ __forceinline__ bool face_yz(const GuardD &B, double xf, const RayD &r) {       // isectYZ
    const double t = (xf - r.px) * r.rx;
    const double a = r.py + t * r.dy, c = r.pz + t * r.dz;
    return t >= 0 && B.y0 <= a && a <= B.y1 && B.z0 <= c && c <= B.z1;
}
// `passed`: bit g set = box g (< 64) and every box above it are known to let this ray through (tile_guard_mask)
template <bool EARLY = (RT_GUARD_EARLY != 0)>
__device__ __forceinline__ bool guards_ok(const GuardD *__restrict__ G, int g, const RayD &r, unsigned long long passed = 0ull) {
    // the faces the ray points at: it leaves the box through one of these three
    const bool nz = r.dz < 0, ny = r.dy < 0, nx = r.dx < 0;
    (void)nz; (void)ny; (void)nx;
#ifdef RT_NO_GUARDS
    return true;        // timing experiments only
#endif
    while (g >= 0) {
        if (g < 64 && ((passed >> g) & 1ull)) return true;
        const GuardD B = G[g];
        bool any;
        if (EARLY) {
        any = face_xy(B, nz ? B.z0 : B.z1, r);
        if (__builtin_amdgcn_ballot_w64(!any)) {
            any |= face_xz(B, ny ? B.y0 : B.y1, r);
            if (__builtin_amdgcn_ballot_w64(!any)) {
                any |= face_yz(B, nx ? B.x0 : B.x1, r);
                if (__builtin_amdgcn_ballot_w64(!any)) {
                    any |= face_xy(B, nz ? B.z1 : B.z0, r);
                    any |= face_xz(B, ny ? B.y1 : B.y0, r);
                    any |= face_yz(B, nx ? B.x1 : B.x0, r);
                }
            }
        }
        } else {
            any = face_xy(B, B.z0, r); any |= face_xz(B, B.y0, r); any |= face_yz(B, B.x0, r);
            any |= face_xy(B, B.z1, r); any |= face_xz(B, B.y1, r); any |= face_yz(B, B.x1, r);
        }
        if (!any) return false;
        g = B.parent;
    }
    return true;
}

// ------------------------------------------------------------------------------------------
// FP32 filter: 4 rays of one lane against one leaf record (wave-uniform -> SGPR operands)

__device__ __forceinline__ CRecPtr as_const(const FilterRec *p) { return (CRecPtr)(const void *)p; }
__device__ __forceinline__ CIntPtr as_const(const int32_t *p) { return (CIntPtr)(const void *)p; }
template <int N = 10>
__device__ __forceinline__ FilterRec load_rec(CRecPtr p) {
    FilterRec r;
#pragma unroll
    for (int k = 0; k < 16; ++k) r.c[k] = k < N ? p->c[k] : 0.f;
    return r;
}
// p, but not before record r has arrived: orders a prefetch after the wait for the record in use
__device__ __forceinline__ CRecPtr after(CRecPtr p, const FilterRec &r) {
    asm volatile("" : "+s"(p) : "s"(r.c[0]));
    return p;
}
__device__ __forceinline__ f2 splat(float v) { f2 r = {v, v}; return r; }
__device__ __forceinline__ f2 pkfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

struct Dir4 { f2 x[2], y[2], z[2]; };   // directions of the lane's 4 rays from the apex

// Sphere, apex form: z = (D.oc)^2 - |D|^2 cs; dd = |D|^2 of the lane's rays.
__device__ __forceinline__ bool filter4_sphere(const FilterRec &R, const Dir4 &D, const f2 dd[2], float z[4]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const f2 b = pkfma(splat(R.c[0]), D.x[p], pkfma(splat(R.c[1]), D.y[p], splat(R.c[2]) * D.z[p]));
        const f2 zz = pkfma(b, b, -(dd[p] * splat(R.c[3])));
        z[2 * p] = zz[0]; z[2 * p + 1] = zz[1];
    }
    const float mx = __builtin_fmaxf(__builtin_fmaxf(z[0], z[1]), __builtin_fmaxf(z[2], z[3]));
    return mx >= R.c[9];
}
__device__ __forceinline__ void dir_norms(const Dir4 &D, f2 dd[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) dd[p] = pkfma(D.x[p], D.x[p], pkfma(D.y[p], D.y[p], D.z[p] * D.z[p]));
}

__device__ __forceinline__ void to_light(bool talk, const LightD &Lt, double hx, double hy, double hz, double &tx, double &ty, double &tz) {
    const double lx = Lt.px - hx, ly = Lt.py - hy, lz = Lt.pz - hz;                    // Main.hs:470
    const double ln = sqrt(lx * lx + ly * ly + lz * lz);
    if (talk) { const double rn = 1.0 / ln; tx = rn * lx; ty = rn * ly; tz = rn * lz; return; }   // Talk.hs:239, 66-67
    tx = lx / ln; ty = ly / ln; tz = lz / ln;
}

// Haskell `round` (half-even) then fromInteger to Word8 (wraps) — Main.hs:451-456
__device__ __forceinline__ int round8(double v) {
    const double r = __builtin_rint(v);
    if (!(fabs(r) < 9.0e18)) return 0;
    return (int)((long long)r & 0xff);
}
__device__ __forceinline__ double hs_max0(double v) { return 0.0 <= v ? v : 0.0; }    // max 0 v

// RT_OPT_RGB24: pixel `idx` of a 3-bytes-per-pixel buffer.  Lanes l, l + 1, l + 2, l + 3 of a wave hold the pixels
// idx, .., idx + 3 with idx % 4 == 0 when `packed` (the image width is a multiple of 4: every row, every 8-pixel patch
// row and every chunk then starts on such an index, and a group of four is inside the image or outside as a whole): the
// twelve bytes of a group are three aligned dwords, dword k = (colour k >> 8k) | (colour k + 1 << (24 - 8k)), stored by
// lanes 0 .. 2 of the group.  Otherwise three byte stores per pixel.  Call with EVERY lane of the wave (`act`: this lane
// has a pixel).
__device__ __forceinline__ void store_rgb24(uint8_t *out, long long idx, uint32_t rgb, bool act, bool packed) {
    if (!packed) {
        if (act) { uint8_t *o = out + 3 * idx; o[0] = (uint8_t)rgb; o[1] = (uint8_t)(rgb >> 8); o[2] = (uint8_t)(rgb >> 16); }
        return;
    }
    const uint32_t next = __shfl_down(rgb, 1);
    const unsigned k = (unsigned)idx & 3u;
    if (act && k < 3u) *reinterpret_cast<uint32_t *>(out + 3 * (idx - (long long)k) + 4 * k) = (rgb >> (8u * k)) | (next << (24u - 8u * k));
}

// x ** specularity (Main.hs:483).  A DELIBERATE departure from the reference's expression: materials carry small whole
// exponents (1, 20, 100, 400 in the reference's scenes) and for whole n <= 512 the power is taken by squaring.  Every
// squaring doubles the relative error of its operand, so the result is off by up to about n * 2^-53 (4e-14 at n = 400),
// where libm's pow is within one rounding of the real power: an 8-bit colour differs from the reference's only if
// f * 255 lies within ~n * 3e-14 of a rounding boundary (never seen: test_whole_and_fractional_specular_exponents, the
// fuzzers).  Larger and fractional exponents go through pow.
#ifndef RT_POWI
#define RT_POWI 1
#endif
__device__ __forceinline__ double pow_spec(double x, double sp) {
#if RT_POWI
    const int n = (int)sp;
    if ((double)n == sp && n >= 0 && n <= 512) {
        double r = 1.0, b = x;
        for (int k = n; k != 0; k >>= 1) { if (k & 1) r *= b; b *= b; }
        return r;
    }
#endif
    return pow(x, sp);
}

// What a point light adds to a pixel it reaches: pointLight (Main.hs:476-487) or Talk.hs's pointLight (Talk.hs:244-248).
// k_final adds it for the lights that reach the pixel; k_setup evaluates the same function first and casts no shadow
// ray where it is (0, 0, 0) — a black material, a grazing light: the shadow test could not change the pixel.
// Hit records live as two arrays: points and normals (k_shadow_mfma gathers the points of 5.6 M shadow rays per frame, and a
// gather moves whole lines: the normals came along for nothing)
__device__ __forceinline__ HitRec load_hit_point(const RenderParams &P, size_t pix) {
    const HitVec p = P.hitpos[pix];
    return HitRec{p.x, p.y, p.z, 0, 0, 0};
}
// The normal of a planar leaf with the geometric normal is the leaf's own (the same doubles): it is not stored per pixel
// (24 of 48 bytes per pixel that k_setup would write and k_final read back; the leaves stay in the L2).
__device__ __forceinline__ bool normal_is_the_leafs(const LeafD &L) { return L.kind != LEAF_SPHERE && L.nmode == RT_NORMAL_GEOMETRIC; }
__device__ __forceinline__ HitRec load_hit(const RenderParams &P, size_t pix, const LeafD &L) {
    const HitVec p = P.hitpos[pix];
    if (normal_is_the_leafs(L)) return HitRec{p.x, p.y, p.z, L.nx, L.ny, L.nz};
    const HitVec n = P.hitnrm[pix];
    return HitRec{p.x, p.y, p.z, n.x, n.y, n.z};
}
__device__ __forceinline__ void store_hit(const RenderParams &P, size_t pix, const HitRec &H, bool leafs_normal) {
    P.hitpos[pix] = HitVec{H.px, H.py, H.pz};
    if (!leafs_normal) P.hitnrm[pix] = HitVec{H.nx, H.ny, H.nz};
}
__device__ __forceinline__ void light_terms(const RenderParams &P, const LightD &Lt, const MaterialD &M, const HitRec &H,
                                            const RayD &ray, int &r, int &g, int &b) {
    double tx, ty, tz;
    to_light(RT_TALK(P), Lt, H.px, H.py, H.pz, tx, ty, tz);
    const double lndot = tx * H.nx + ty * H.ny + tz * H.nz;
    if (RT_TALK(P)) {                                                                  // Talk.hs:244-248
        const double v = hs_max0(lndot) * Lt.diffuse;
        const double f = (1.0 <= v) ? 1.0 : v;                                         // min 1 v
        r = round8(f * (double)M.diffuse[0]); g = round8(f * (double)M.diffuse[1]); b = round8(f * (double)M.diffuse[2]);
        return;
    }
    const double k2 = 2 * lndot;
    const double rx = tx - k2 * H.nx, ry = ty - k2 * H.ny, rz = tz - k2 * H.nz;
    const double fDiffuse = Lt.diffuse * hs_max0(lndot);
    const double fSpecular = (lndot <= 0) ? 0.0
        : Lt.specular * pow_spec(hs_max0(rx * ray.dx + ry * ray.dy + rz * ray.dz), M.specularity);
    r = min(255, round8(fDiffuse * (double)M.diffuse[0]) + round8(fSpecular * (double)M.specular[0]));
    g = min(255, round8(fDiffuse * (double)M.diffuse[1]) + round8(fSpecular * (double)M.specular[1]));
    b = min(255, round8(fDiffuse * (double)M.diffuse[2]) + round8(fSpecular * (double)M.specular[2]));
}

// spheres between a light and the shadow-ray origins (no own-leaf exclusion: a sphere is not planar)
__device__ __forceinline__ void shadow_sphere_loop(const RenderParams &P, CRecPtr recs, const Dir4 &D, const uint32_t pix[4],
                                                const LightD &Lt, bool need[4], unsigned &nexact) {
    f2 dd[2];
    dir_norms(D, dd);
    CIntPtr gids = as_const(P.sph_gid);
    for (int i = 0; i < P.nsph; ++i) {
        const FilterRec cur = load_rec(recs + i);
        float z[4];
        if (filter4_sphere(cur, D, dd, z)) {
            const int gid = gids[i];
            const LeafD &L = P.leaves[gid];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if ((z[r] >= cur.c[9]) & need[r]) {
                    uint32_t pr = pix[r];
                    asm volatile("" : "+v"(pr));
                    const HitRec H = load_hit_point(P, pr);
                    double tx, ty, tz;
                    to_light(false, Lt, H.px, H.py, H.pz, tx, ty, tz);
                    const RayD ray = mkray(H.px + 0.00001 * tx, H.py + 0.00001 * ty, H.pz + 0.00001 * tz, tx, ty, tz, P.eps);
                    double ix, iy, iz, t;
                    ++nexact;
                    if (leaf_isect<LEAF_SPHERE>(L, ray, ix, iy, iz, t, false) && guards_ok<false>(P.guards, L.guard, ray)) {
                        const double tl = (Lt.pz - ray.pz) * ray.rz;
                        if (t <= tl) need[r] = false;
                    }
                }
            }
        }
    }
}

} // namespace
