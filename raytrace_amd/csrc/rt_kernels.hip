// rt_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the render path and the render entry points.
//
// Replaces parallelTrace / computeP (reference src/Main.hs:623-637) and everything its element
// function calls: cast (506-533), trace (541-557), intersect world (196-217, 226-277, 363-385),
// light lights (436-492).
//
// Design (DESIGN.md has the long form):
//   * Brute force: every ray is tested against every leaf, through a chain of conservative filters in front of the
//     exact test.  All rays of a pass leave a common apex (the eye; a light for shadow rays), so a leaf is seen from the
//     apex under a bounding cone (stage 0, one linear form per pair) and its exact shape is three linear forms of the
//     ray direction (stage 1, rt_scene.cpp).
//   * Matrix-core kernels (k_primary_mfma, k_shadow_mfma; the default): stage 0 for 32 leaves x 32 rays is one
//     v_mfma_f32_32x32x16_f16; survivors go through per-wave LDS queues to stage 1 (FP32) and to the exact stage, 64
//     entries at a time.
//   * Vector-ALU kernels (k_primary, k_shadow; RT_FILTER=valu): stage 1 for every pair.  One wavefront owns a 32 x 8
//     pixel tile, a lane 4 pixels as two float2 pairs (v_pk_fma_f32); leaf records are wave-uniform and are read with
//     scalar loads into SGPRs: no VGPRs, no LDS bandwidth, no bank conflicts.
//   * (ray, leaf) pairs that pass the filters — a handful per ray — are evaluated in IEEE double with the reference's
//     own expression order (planeLineIsect etc.), which decides the hit, the tie-break (left-most leaf,
//     Main.hs:213-214) and the colour exactly.  The file is compiled with -ffp-contract=off; the FP32 loops use
//     explicit fma builtins.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <type_traits>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rt_scene.h"

using namespace rt;

typedef float f2 __attribute__((ext_vector_type(2)));
// Wave-uniform, read-only scene data is addressed through the constant address space, which makes
// hipcc fetch it with scalar loads (s_load_dwordx8 ...) no matter what else the kernel stores.
#define RT_CONST __attribute__((address_space(4)))
typedef const rt::FilterRec RT_CONST *CRecPtr;
typedef const int32_t RT_CONST *CIntPtr;

#define RT_TILE_W 32
#define RT_TILE_H 8
// One wave per work-group: the wave slots of a 4-wave group only came free when its slowest wave had finished, and the
// waves of these kernels take between 10 and 200 us; measured on the benchmark frame, a third of the slots sat empty
// (tools/wave_times.py), 9 % of the frame time.
#ifndef RT_WAVES_PER_BLOCK
#define RT_WAVES_PER_BLOCK 1
#endif
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

struct HitRec;
struct RenderParams {
    const LeafD *leaves;
    const AuxD *aux;
    const GuardD *guards;
    const MaterialD *mats;
    const LightD *lights;
    const FilterRec *tri_rec;
    const FilterRec *rect_rec;
    const int32_t *tri_gid;
    const int32_t *rect_gid;
    const FilterRec *sph_rec;
    const GenRec *gen_sph;
    const int32_t *sph_gid;
    const uint16_t *mf_rows;        // MFMA row blocks, per apex (rt_types.h)
    const int32_t *mf_gid;          // MFMA stream position -> leaf index, per apex
    const float *mf_cone;           // bounding cone of every block of the stream, per apex (rt_scene.cpp: block_cone)
    const FilterRec *mf_rec;        // FP32 records in MFMA stream order, per apex (c[14] = kind, c[15] = leaf index)
    unsigned long long *counters;   // [0] shadow rays cast, [1] exact evals (primary), [2] exact evals (shadow)
    int32_t *hit_id;                // band-local output, indexed out_off + pix
    uint8_t *rgba;                  // band-local output (spp == 1) or nullptr
    uint32_t *accum;                // band-local 3 x u32 sums (spp > 1) or nullptr
    HitRec *hitrec;                 // per chunk pixel: exact hit point and shading normal
    uint32_t *queue;                // npoint blocks of qcap shadow jobs (chunk pixel indices)
    unsigned *qcount;               // shadow jobs per point light
    uint32_t *qsparse;              // k_setup's output: per light, RT_SETUP_THREADS slots per block, the block's jobs first
    unsigned *tcount;               // jobs of every (light, k_setup block); k_queue_offsets turns them into offsets in place
    int setup_blocks;               // k_setup blocks of this chunk
    uint32_t *lit;                  // per chunk pixel and group of 32 point lights: bit l set = light 32 g + l reaches it
    // RT_NORMAL_CALLBACK leaves (an arbitrary fnorm, Main.hs:956-965): k_cb_gather lists the hit points on them, the host
    // evaluates the caller's function, k_setup reads the normals back
    uint32_t *cb_slot;              // per slot of the chunk / level: index into the list
    double *cb_xy;                  // per list entry: the arguments x, y of fnorm
    int32_t *cb_group;              // per list entry: which callback
    const double *cb_normal;        // per list entry: fnorm x y
    unsigned *cb_count;             // entries in the list
    unsigned long long *dbg;        // -DRT_WAVE_TIMES builds only (tools/wave_times.py): per-wave timeline records
    // bounce levels (reflective scenes, trace depth > 1; Main.hs:546-556).  A "slot" is a pixel of the chunk at
    // level 0 and an entry of the level's ray list above it.
    const RayRec *rays;             // this level's rays (nullptr at level 0: camera rays)
    const unsigned *count_ptr;      // number of slots of this level (nullptr at level 0: qcap)
    RayRec *next_rays;              // where k_final appends the reflected rays (nullptr: none)
    unsigned *next_count;
    uint32_t *colour;               // per-slot colour of this level (nullptr: write RGBA8 / accum directly)
    const GenRec *gen_tri;          // general-ray filter records, in a space-filling order of the leaves (rt_scene.cpp: spatial_order)
    const GenRec *gen_rect;
    const int32_t *gen_tri_gid;     // position in that order -> leaf index
    const int32_t *gen_rect_gid;
    const float *gen_tri_sph;       // bounding sphere of every block of 32 consecutive records: centre xyz, radius (inf: always scan)
    const float *gen_rect_sph;
    const uint16_t *gen_tri_rows;   // the blocks' matrix-core rows: 2 K steps x 64 x 16 bytes per block (lane (m, h): K-half h of leaf m), + one padding block
    const uint16_t *gen_rect_rows;
    const float *gen_tri_ssph;      // super-blocks of 32 blocks: bounding sphere = the frame their leaves' rows are written in
    const float *gen_rect_ssph;
    CameraD cam;
    double eps;
    int ntri, nrect, nsph, nlights, nleaves, npoint;
    int mf_groups;                  // blocks of MF_LEAVES planar leaves per apex (+ 2 of padding)
    int cull;                       // 0: every wave scans every block (RT_CULL=0)
    int nguard_cached;              // guards [0, nguard_cached) are tested per tile (tile_guard_mask); 0: off
    int w, h;
    int row0, nrows;                // this chunk: first absolute image row, rows
    long long out_off;              // index of the chunk's first pixel inside the band-local outputs
    int qcap;                       // pixels in the chunk = capacity of one shadow queue
    int lcap;                       // capacity of a bounce level's ray list: the chunk's pixels padded to whole 32 x 8 tiles
    int spp, sample;
    int talk;                       // 1: the rules of src/Talk.hs (rt_scene_set_dialect)
    uint32_t seed_lo, seed_hi;
    int tiles_x, tiles_y;
    uint16_t point_light[32];       // index in `lights` of the k-th point light of the current group of 32
    uint32_t *lit_w;                // the current group's word of `lit` (= lit + group * lit_stride)
    long long lit_stride;           // slots per group of 32 point lights in `lit`
};

struct HitRec { double px, py, pz, nx, ny, nz; };

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

// Statistics counters: RT_STAT_SLOTS copies of the 8 counters, one 128-byte line each, picked by wave number.  Every
// wave adds to them once, and atomics on ONE address are serialised by the L2 (12 ns each: 0.4 ms for the 32 400 waves
// of a 4K frame, which is what a single set of counters cost); rt_get_stats sums the copies.
constexpr int RT_STAT_SLOTS = 256;
// Workspace::qcount: [0, 32) shadow jobs per point light; [32, 32 + RT_MAX_LEVELS] rays per bounce level; then the
// number of hit points waiting for a normal callback
constexpr int RT_QCOUNT_CB = 32 + RT_MAX_LEVELS + 1, RT_QCOUNT_CLEAR = (RT_QCOUNT_CB + 1 + 3) / 4 * 4;   // (cleared as whole 16-byte units: one fill kernel)
// ... then, NOT cleared between renders: [RT_QCOUNT_HIST, + 64) per point light of a group {wall-clock ticks, waves} of a
// sample of its k_shadow_mfma waves, halved at every launch (a moving average); [RT_QCOUNT_ORDER, + 32) the order the
// lights' waves are dispatched in, costliest light first (k_queue_offsets).
constexpr size_t RT_COUNTER_BYTES = (size_t)RT_STAT_SLOTS * 16 * sizeof(unsigned long long);
constexpr int RT_QCOUNT_HIST = RT_QCOUNT_CLEAR, RT_QCOUNT_ORDER = RT_QCOUNT_HIST + 64, RT_QCOUNT_WORDS = RT_QCOUNT_ORDER + 32;
__device__ __forceinline__ void stat_add(const RenderParams &P, int k, unsigned long long v) {
    const unsigned wave = (blockIdx.x + blockIdx.y * gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    atomicAdd(&P.counters[(size_t)(wave % RT_STAT_SLOTS) * 16 + k], v);
}
#ifdef RT_WAVE_TIMES
// measurement builds: one record per wave {kernel | light << 8, wave, start, end (100 MHz wall clock), blocks scanned,
// record tests, exact evaluations, hardware id}; dbg[0] counts the records
__device__ __forceinline__ void wave_record(const RenderParams &P, unsigned kernel, unsigned wave, unsigned long long t0,
                                            unsigned nblocks, unsigned nrec, unsigned nexact, const unsigned long long (&ph)[3]) {
    if (!P.dbg || (threadIdx.x & 63u) != 0u) return;
    const unsigned long long t1 = wall_clock64();
    const unsigned long long k = atomicAdd(&P.dbg[0], 1ull);
    if (k >= (1ull << 20)) return;
    unsigned long long *r = P.dbg + 8 + 8 * k;
    r[0] = kernel; r[1] = wave; r[2] = t0; r[3] = t1;
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

#ifdef RT_VALU_FILTER   // the FP32 vector-ALU kernel family: cross-check build only (libraytrace_gpu_valu.so)
// Returns z[r] for the lane's 4 rays: ray r passes the filter iff z[r] >= R.c[9] (= -margin).
//   triangle : z = min(q1*N, q2*N, (1-q1-q2)*N [, N - Kq])      q1, q2 = barycentrics r, s
//   rectangle: z = min(N - max(|a'N|, |b'N|)     [, N - Kq])     a', b' in [-1, 1]
// CLIP adds the "not beyond the ray origin" test u* = K/N <= 1 used by shadow rays.
template <int KIND, bool CLIP>
__device__ __forceinline__ bool filter4(const FilterRec &R, const Dir4 &D, float z[4]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const f2 q1 = pkfma(splat(R.c[0]), D.x[p], pkfma(splat(R.c[1]), D.y[p], splat(R.c[2]) * D.z[p]));
        const f2 q2 = pkfma(splat(R.c[3]), D.x[p], pkfma(splat(R.c[4]), D.y[p], splat(R.c[5]) * D.z[p]));
        const f2 n = pkfma(splat(R.c[6]), D.x[p], pkfma(splat(R.c[7]), D.y[p], splat(R.c[8]) * D.z[p]));
        f2 zz;
        if (KIND == LEAF_TRI) {
            const f2 w = CLIP ? n - q1 - q2 : n;      // without the clip the third form already is W
            zz = f2{__builtin_fminf(__builtin_fminf(q1[0], q2[0]), w[0]), __builtin_fminf(__builtin_fminf(q1[1], q2[1]), w[1])};
        } else {
            const f2 m = f2{__builtin_fmaxf(__builtin_fabsf(q1[0]), __builtin_fabsf(q2[0])),
                            __builtin_fmaxf(__builtin_fabsf(q1[1]), __builtin_fabsf(q2[1]))};
            zz = n - m;
        }
        if (CLIP) {
            const f2 c = n - splat(R.c[10]);
            zz = f2{__builtin_fminf(zz[0], c[0]), __builtin_fminf(zz[1], c[1])};
        }
        z[2 * p] = zz[0]; z[2 * p + 1] = zz[1];
    }
    const float mx = __builtin_fmaxf(__builtin_fmaxf(z[0], z[1]), __builtin_fmaxf(z[2], z[3]));
    return mx >= R.c[9];
}

#endif // RT_VALU_FILTER
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

#ifdef RT_VALU_FILTER   // the FP32 vector-ALU kernel family: cross-check build only (libraytrace_gpu_valu.so)
// Eye-apex variant: the forms are affine in the pixel's viewport position (posX, posY); the lane's 4
// pixels share posY (SHARED_Y, spp == 1), so the posY part of a form is computed once per lane.
//   triangle : z = min(U, V, W)          rectangle: z = N - max(|A|, |B|)
struct Eye4 { f2 x[2]; f2 y[2]; float ys; };

template <int KIND, bool SHARED_Y>
__device__ __forceinline__ bool filter4_eye(const FilterRec &R, const Eye4 &E, float z[4]) {
    f2 q[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (SHARED_Y) {
            const float b = __builtin_fmaf(R.c[3 * k + 1], E.ys, R.c[3 * k + 2]);
#pragma unroll
            for (int p = 0; p < 2; ++p) q[k][p] = pkfma(splat(R.c[3 * k]), E.x[p], splat(b));
        } else {
#pragma unroll
            for (int p = 0; p < 2; ++p) q[k][p] = pkfma(splat(R.c[3 * k]), E.x[p], pkfma(splat(R.c[3 * k + 1]), E.y[p], splat(R.c[3 * k + 2])));
        }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (KIND == LEAF_TRI) {
            z[2 * p] = __builtin_fminf(__builtin_fminf(q[0][p][0], q[1][p][0]), q[2][p][0]);
            z[2 * p + 1] = __builtin_fminf(__builtin_fminf(q[0][p][1], q[1][p][1]), q[2][p][1]);
        } else {
            const f2 m = f2{__builtin_fmaxf(__builtin_fabsf(q[0][p][0]), __builtin_fabsf(q[1][p][0])),
                            __builtin_fmaxf(__builtin_fabsf(q[0][p][1]), __builtin_fabsf(q[1][p][1]))};
            const f2 zz = q[2][p] - m;
            z[2 * p] = zz[0]; z[2 * p + 1] = zz[1];
        }
    }
    const float mx = __builtin_fmaxf(__builtin_fmaxf(z[0], z[1]), __builtin_fmaxf(z[2], z[3]));
    return mx >= R.c[9];
}

__device__ __forceinline__ void tile_coords(const RenderParams &P, int &x0, int &y, bool &row_ok) {
    const int wave = (int)(blockIdx.x * RT_WAVES_PER_BLOCK + (threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int tx = wave % P.tiles_x, ty = wave / P.tiles_x;
    x0 = tx * RT_TILE_W + (lane & 7) * 4;
    const int ly = ty * RT_TILE_H + (lane >> 3);
    row_ok = ly < P.nrows;
    y = P.row0 + ly;             // absolute image row
}

// ------------------------------------------------------------------------------------------
// Kernel 1: nearest hit of the primary rays -> hit_id
//   replaces `intersect world (cast camera x y)` of trace (Main.hs:543) for every pixel of the chunk.
//   One wave = 32 x 8 pixel tile, 4 pixels per lane.  Apex = the eye (all primary rays pass through
//   (0, 0, -d)), D = (posX, posY, d).

template <int KIND, bool SHARED_Y>
__device__ __forceinline__ void primary_loop(const RenderParams &P, CRecPtr recs, CIntPtr gids, int n, const Eye4 &D,
                                             const bool act[4], int x0, int y, double bt[4], int bid[4],
                                             unsigned &nexact) {
    if (n <= 0) return;
    auto body = [&](const FilterRec &cur, int i) {
        float z[4];
        if (filter4_eye<KIND, SHARED_Y>(cur, D, z)) {
            const float negm = cur.c[9];
            const int gid = gids[i];
            const LeafD &L = P.leaves[gid];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if ((z[r] >= negm) & act[r]) {
                    int xr = x0 + r;
                    asm volatile("" : "+v"(xr));   // keep the exact-path ray set-up inside this rare branch
                    const RayD ray = camera_ray<false>(P, xr, y);
                    double ix, iy, iz, t;
                    ++nexact;
                    if (leaf_isect<KIND>(L, ray, ix, iy, iz, t, false) && guards_ok<false>(P.guards, L.guard, ray)) {
                        // nearest wins; equal t -> the left-most leaf of the tree (Main.hs:213-214)
                        if (t < bt[r] || (t == bt[r] && gid < bid[r]) || bid[r] < 0) { bt[r] = t; bid[r] = gid; }
                    }
                }
            }
        }
    };
#if RT_PREFETCH
    // Two record buffers in SGPRs, used in turn (no copies).  Scalar loads return out of order, so the only wait
    // there is waits for ALL of them: the load of record i + 1 is issued right after the wait for record i (the
    // asm ties its address to a register of record i) and has the whole test of record i to arrive.
    FilterRec a = load_rec(recs);
    int i = 0;
    for (; i + 1 < n; i += 2) {
        const FilterRec b = load_rec(after(recs + i + 1, a));
        body(a, i);
        a = load_rec(after(recs + i + 2, b));             // arrays are padded: i + 2 may be one past the end
        body(b, i + 1);
    }
    if (i < n) body(a, i);
#else
    for (int i = 0; i < n; ++i) body(load_rec(recs + i), i);
#endif
}

// spheres seen from the eye (few per scene: plain loop, generic apex form)
__device__ __forceinline__ void primary_sphere_loop(const RenderParams &P, const Eye4 &E, const bool act[4], int x0, int y,
                                                 double bt[4], int bid[4], unsigned &nexact) {
    Dir4 D;
    const float dz = (float)P.cam.d;
#pragma unroll
    for (int p = 0; p < 2; ++p) { D.x[p] = E.x[p]; D.y[p] = E.y[p]; D.z[p] = f2{dz, dz}; }
    f2 dd[2];
    dir_norms(D, dd);
    CRecPtr recs = as_const(P.sph_rec);
    CIntPtr gids = as_const(P.sph_gid);
    for (int i = 0; i < P.nsph; ++i) {
        const FilterRec cur = load_rec(recs + i);
        float z[4];
        if (filter4_sphere(cur, D, dd, z)) {
            const int gid = gids[i];
            const LeafD &L = P.leaves[gid];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if ((z[r] >= cur.c[9]) & act[r]) {
                    int xr = x0 + r;
                    asm volatile("" : "+v"(xr));
                    const RayD ray = camera_ray<false>(P, xr, y);
                    double ix, iy, iz, t;
                    ++nexact;
                    if (leaf_isect<LEAF_SPHERE>(L, ray, ix, iy, iz, t, false) && guards_ok<false>(P.guards, L.guard, ray)) {
                        if (t < bt[r] || (t == bt[r] && gid < bid[r]) || bid[r] < 0) { bt[r] = t; bid[r] = gid; }
                    }
                }
            }
        }
    }
}

template <bool SHARED_Y>
__global__ void RT_HOT(RT_WAVES_EU_PRIMARY) k_primary(const RenderParams P) {
    int x0, y; bool row_ok;
    tile_coords(P, x0, y, row_ok);
    bool act[4];
    Eye4 D;
    {
        float dx[4], dy[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            act[r] = row_ok && (x0 + r) < P.w;
            double ox, oy;
            sample_offset(P, x0 + r, y, ox, oy);
            dx[r] = (float)(P.cam.scaleX * ((double)(x0 + r) + ox) + P.cam.dX);
            dy[r] = (float)(P.cam.scaleY * ((double)y + oy) + P.cam.dY);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            D.x[p] = f2{dx[2 * p], dx[2 * p + 1]};
            D.y[p] = f2{dy[2 * p], dy[2 * p + 1]};
        }
        D.ys = dy[0];
    }
    double bt[4]; int bid[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { bt[r] = INFINITY; bid[r] = -1; }
    unsigned nexact = 0;
    primary_loop<LEAF_TRI, SHARED_Y>(P, as_const(P.tri_rec), as_const(P.tri_gid), P.ntri, D, act, x0, y, bt, bid, nexact);
    primary_loop<LEAF_RECT, SHARED_Y>(P, as_const(P.rect_rec), as_const(P.rect_gid), P.nrect, D, act, x0, y, bt, bid, nexact);
    if (P.nsph > 0) primary_sphere_loop(P, D, act, x0, y, bt, bid, nexact);
    const long long row = P.out_off + (long long)(y - P.row0) * P.w;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (act[r]) P.hit_id[row + (x0 + r)] = bid[r];
    for (int o = 32; o > 0; o >>= 1) nexact += __shfl_down(nexact, o);
    if ((threadIdx.x & 63) == 0 && nexact) stat_add(P, 1, (unsigned long long)nexact);
}

#endif // RT_VALU_FILTER
// ------------------------------------------------------------------------------------------
// Kernel 2: per hit pixel, the exact hit point and shading normal; shadow jobs for every point light
//   with L.N > 0, compacted into one queue per light (wave-aggregated atomics).
//   When L.N <= 0 both Phong terms are exactly 0 (Main.hs:480-484), so the shadow ray cannot change
//   the pixel and is not cast.

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

#ifndef RT_SETUP_THREADS
#define RT_SETUP_THREADS 512
#endif
#ifndef RT_WAVES_EU_SETUP
#define RT_WAVES_EU_SETUP 4
#endif
__global__ void __attribute__((amdgpu_waves_per_eu(RT_WAVES_EU_SETUP, 8))) __launch_bounds__(RT_SETUP_THREADS) k_setup(const RenderParams P) {
    // shadow jobs of the block per light: per-wave counts -> offsets inside the block's run
    __shared__ unsigned s_cnt[32][RT_SETUP_THREADS / 64];
    // level 0: a block takes a 32 x 16 pixel tile, so that the shadow rays it queues (256 consecutive entries make one
    // wave of k_shadow_mfma) leave from neighbouring pixels and form a narrow bundle; bounce levels: slots in order
    int pix = (int)(blockIdx.x * RT_SETUP_THREADS + threadIdx.x);
    bool act = (unsigned)pix < num_slots(P);
    if (!P.rays) {
        // a wave takes 8 x 8 pixels of the tile (objects are blobs: fewer waves straddle an outline than with 32 x 2 strips)
        const int tx = (int)blockIdx.x % P.tiles_x, ty = (int)blockIdx.x / P.tiles_x;
        const int wq = (int)(threadIdx.x >> 6), lq = (int)(threadIdx.x & 63);
        const int x = tx * 32 + (wq & 3) * 8 + (lq & 7), y = ty * (RT_SETUP_THREADS / 32) + (wq >> 2) * 8 + (lq >> 3);
        act = x < P.w && y < P.nrows;
        pix = act ? y * P.w + x : 0;
    }
    const int id = act ? P.hit_id[P.out_off + pix] : -1;
    HitRec H = {0, 0, 0, 0, 0, 0};
    RayD ray = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int mat = 0;
    if (id >= 0) {
        const LeafD L = P.leaves[id];
        ray = slot_ray(P, pix);
        mat = L.mat;
        double ix = 0, iy = 0, iz = 0, t;
        // only the hit point is wanted, and planeRayIsect (Main.hs:264-277) finds it the same way for a triangle as for a
        // rectangle: one pass for a wave that holds both
        if (L.kind != LEAF_SPHERE) leaf_isect<LEAF_RECT>(L, ray, ix, iy, iz, t, RT_TALK(P));
        else leaf_isect<LEAF_SPHERE>(L, ray, ix, iy, iz, t, RT_TALK(P));
        double nx = L.nx, ny = L.ny, nz = L.nz;
        if (L.kind == LEAF_SPHERE) {                                                   // scalar (recip radius) (isect - center), :410
            const double rr = 1.0 / L.c0;
            nx = rr * (ix - L.p0x); ny = rr * (iy - L.p0y); nz = rr * (iz - L.p0z);
        }
        if (L.nmode == RT_NORMAL_MANTLE) {                                             // Main.hs:769-771
            const AuxD A = P.aux[L.aux];
            nx = A.a[0] + (ix - A.a[3]) * A.a[6];
            nz = A.a[2] + (iz - A.a[5]) * A.a[8];
            ny = 0;
        } else if (L.nmode == RT_NORMAL_GAUSSIAN) {                                    // Main.hs:959-963, 580-584
            const AuxD A = P.aux[L.aux];
            const double gx = A.a[0] * (ix - A.a[1]), gy = A.a[0] * (iz - A.a[3]);
            const double f = exp(-(gx * gx + gy * gy) * 4);
            const double dfdx = -8 * gx * f, dfdy = -8 * gy * f;
            // normalize (MkV3D 1 dfdx 0 *# MkV3D 0 dfdy 1)
            const double vx = dfdx * 1 - 0 * dfdy, vy = 0 * 0 - 1 * 1, vz = 1 * dfdy - 0 * dfdx;
            const double vn = sqrt(vx * vx + vy * vy + vz * vz);
            const double ux = vx / vn, uy = vy / vn, uz = vz / vn;
            const double dt = nx * ux + ny * uy + nz * uz;
            const double sg = dt > 0 ? 1.0 : (dt < 0 ? -1.0 : dt);                      // signum
            nx = sg * ux; ny = sg * uy; nz = sg * uz;
        } else if (L.nmode == RT_NORMAL_CALLBACK) {                                    // Main.hs:959-963 with the caller's fnorm
            const double *un = P.cb_normal + 3 * (size_t)P.cb_slot[pix];               // n' = fnorm' x y, evaluated on the host
            const double ux = un[0], uy = un[1], uz = un[2];
            const double dt = nx * ux + ny * uy + nz * uz;
            const double sg = dt > 0 ? 1.0 : (dt < 0 ? -1.0 : dt);                      // signum
            nx = sg * ux; ny = sg * uy; nz = sg * uz;
        }
        H.px = ix; H.py = iy; H.pz = iz; H.nx = nx; H.ny = ny; H.nz = nz;
        P.hitrec[pix] = H;
    }
    if (act) P.lit_w[pix] = 0u;
    const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6);
    uint32_t needmask = 0;
    int pl = 0;
    for (; pl < P.npoint; ++pl) {                     // the (up to 32) point lights of this group
        const LightD &Lt = P.lights[P.point_light[pl]];
        bool need = false;
        if (id >= 0) {
            double tx, ty, tz;
            to_light(RT_TALK(P), Lt, H.px, H.py, H.pz, tx, ty, tz);
            const double lndot = tx * H.nx + ty * H.ny + tz * H.nz;                    // Main.hs:476
            if (!(lndot <= 0)) {        // ... and only where the light would add something to the pixel (light_terms != 0):
                const MaterialD M = P.mats[mat];
                const double fd = RT_TALK(P) ? ((1.0 <= hs_max0(lndot) * Lt.diffuse) ? 1.0 : hs_max0(lndot) * Lt.diffuse)
                                             : Lt.diffuse * hs_max0(lndot);
                // the diffuse term decides almost always; the specular term (a pow) is looked at only behind a zero one
                need = (round8(fd * (double)M.diffuse[0]) | round8(fd * (double)M.diffuse[1]) | round8(fd * (double)M.diffuse[2])) != 0;
                if (!need && !RT_TALK(P) && (M.specular[0] | M.specular[1] | M.specular[2]) != 0) {
                    int r, g, b;
                    light_terms(P, Lt, M, H, ray, r, g, b);
                    need = (r | g | b) != 0;
                }
            }
        }
        if (need) needmask |= 1u << pl;
        const unsigned long long m = __ballot(need);
        if (lane == 0) s_cnt[pl][wv] = (unsigned)__popcll(m);
    }
    __syncthreads();
    if ((int)threadIdx.x < P.npoint) {
        unsigned run = 0;
        for (int k = 0; k < RT_SETUP_THREADS / 64; ++k) { const unsigned c = s_cnt[threadIdx.x][k]; s_cnt[threadIdx.x][k] = run; run += c; }
        P.tcount[(size_t)threadIdx.x * (size_t)P.setup_blocks + blockIdx.x] = run;
    }
    __syncthreads();
    // the block's jobs go to its own slots; k_queue_offsets + k_queue_compact then line the blocks' runs up in block
    // order, so that the 256 consecutive jobs of a k_shadow_mfma wave come from the same or from neighbouring tiles
    for (pl = 0; pl < P.npoint; ++pl) {
        const bool need = (needmask >> pl) & 1u;
        const unsigned long long m = __ballot(need);
        if (need) P.qsparse[((size_t)pl * (size_t)P.setup_blocks + blockIdx.x) * RT_SETUP_THREADS + s_cnt[pl][wv] + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)pix;
    }
}

// Exclusive prefix sums of one light's per-block job counts (in place), the light's total into qcount and the
// shadow-ray counter.  One block of 1024 threads per light.
__global__ void __launch_bounds__(1024) k_queue_offsets(const RenderParams P) {
    __shared__ unsigned s_w[16];
    const int pl = (int)blockIdx.x, t = (int)threadIdx.x, lane = t & 63, wv = t >> 6;
    unsigned *cnt = P.tcount + (size_t)pl * (size_t)P.setup_blocks;
    const int n = P.setup_blocks;
    // 1024 consecutive counts at a time (coalesced; the next chunk's are loaded while this one is scanned): a scan inside
    // every wave, the waves' totals through LDS, the running total of the chunks before in a register
    unsigned carry = 0;
    unsigned c_next = t < n ? cnt[t] : 0u;
    for (int base = 0; base < n; base += 1024) {
        const int k = base + t;
        const unsigned c = c_next;
        c_next = k + 1024 < n ? cnt[k + 1024] : 0u;
        unsigned incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(incl, o); if (lane >= o) incl += up; }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        unsigned before = 0, total = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) { const unsigned v = s_w[j]; total += v; if (j < wv) before += v; }
        if (k < n) cnt[k] = carry + before + incl - c;
        carry += total;
        __syncthreads();
    }
    if (t == 1023) {
        P.qcount[pl] = carry;
        if (carry) stat_add(P, 0, (unsigned long long)carry);
    }
    // The order k_shadow_mfma takes the lights in: the light whose waves took longest on average (a sample of the waves
    // of the renders before this one) first.  The kernel ends when its last wave ends, and with 3 waves per SIMD that
    // tail is a fifth of its time; started last, the cheap lights' waves make a shorter one.  The order decides when a
    // wave runs, never what it computes; without a history (first render, rt_render_device_noalloc) it is 0, 1, 2, ...
    if (pl == 0 && t < 32) {
        unsigned *hist = P.qcount + RT_QCOUNT_HIST, *order = P.qcount + RT_QCOUNT_ORDER;
        const unsigned ticks = hist[2 * t], waves = hist[2 * t + 1];
        const float cost = (t < P.npoint && waves > 0u) ? (float)ticks / (float)waves : 0.f;
        unsigned rank = 0;
        for (int j = 0; j < P.npoint; ++j) {
            const float cj = __shfl(cost, j);
            rank += (cj > cost || (cj == cost && j < t)) ? 1u : 0u;
        }
        if (t < P.npoint) order[rank] = (unsigned)t;
        hist[2 * t] = ticks >> 1; hist[2 * t + 1] = waves >> 1;
    }
}

// Moves every k_setup block's run of jobs to its place in the light's dense queue.
#ifndef RT_COMPACT_THREADS
#define RT_COMPACT_THREADS 128
#endif
__global__ void __launch_bounds__(RT_COMPACT_THREADS) k_queue_compact(const RenderParams P) {
    const int pl = (int)blockIdx.y;
    const size_t slot = (size_t)pl * (size_t)P.setup_blocks + blockIdx.x;
    const unsigned off = P.tcount[slot];
    const unsigned end = (int)blockIdx.x + 1 < P.setup_blocks ? P.tcount[slot + 1] : P.qcount[pl];
    // (a run holds a third of RT_SETUP_THREADS jobs on average: a thread per possible job was mostly idle waves)
    for (unsigned i = threadIdx.x; i < end - off; i += RT_COMPACT_THREADS)
        P.queue[(size_t)pl * (size_t)P.qcap + off + i] = P.qsparse[slot * RT_SETUP_THREADS + i];
}

// ------------------------------------------------------------------------------------------
// Kernel 3: shadow rays of one point light (blockIdx.y), 4 jobs per lane, brute force over all leaves.
//   replaces `intersect world ray_to_light` + the t <= distance test of pointLight (Main.hs:472-474).
//   Apex = the light, D = shadow-ray origin - light; the filter's clip keeps leaves between them.

#ifdef RT_VALU_FILTER   // the FP32 vector-ALU kernel family: cross-check build only (libraytrace_gpu_valu.so)
template <int KIND>
__device__ __forceinline__ void shadow_loop(const RenderParams &P, CRecPtr recs, CIntPtr gids, int n, const Dir4 &D,
                                            const uint32_t pix[4], const int own[4], const LightD &Lt, bool need[4],
                                            unsigned &nexact) {
    constexpr int NLOAD = RT_SHADOW_CLIP ? 11 : 10;   // the hot test reads c[0..9]; Kq and N stay in memory
    if (n <= 0) return;
    auto body = [&](const FilterRec &cur, int i) {
        float z[4];
        if (filter4<KIND, RT_SHADOW_CLIP != 0>(cur, D, z)) {
            const float negm = cur.c[9];
#ifdef RT_COUNT_EVENTS
            if ((threadIdx.x & 63) == __builtin_amdgcn_readfirstlane(threadIdx.x & 63)) stat_add(P, 3, 1ull);
#endif
            bool cand[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) cand[r] = (z[r] >= negm) & need[r];
#if !RT_SHADOW_CLIP
            // The fast filter tests the whole line from the light through the shadow-ray origin.  Keep only the
            // rays whose SEGMENT light..origin can meet the plane: N.D >= Kq (Main.hs:474 needs t <= distance).
            // Kq and N are fetched here, on the rare path, so the loop carries 10 SGPRs per record, not 14.
            const float kq = recs[i].c[10], nx = recs[i].c[11], ny = recs[i].c[12], nz = recs[i].c[13];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const f2 nd = pkfma(splat(nx), D.x[p], pkfma(splat(ny), D.y[p], splat(nz) * D.z[p]));
                cand[2 * p] &= nd[0] >= kq;
                cand[2 * p + 1] &= nd[1] >= kq;
            }
#endif
            if (cand[0] | cand[1] | cand[2] | cand[3]) {
                const int gid = gids[i];
                const LeafD &L = P.leaves[gid];
#ifdef RT_COUNT_EVENTS
                if ((threadIdx.x & 63) == __builtin_amdgcn_readfirstlane(threadIdx.x & 63)) stat_add(P, 4, 1ull);
#endif
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // the hit leaf itself lies 1e-5 behind the shadow-ray origin (Main.hs:472): t < 0
                    if (cand[r] & (own[r] != gid)) {
                        uint32_t pr = pix[r];
                        asm volatile("" : "+v"(pr));   // keep the exact-path ray set-up inside this rare branch
                        const HitRec H = P.hitrec[pr];
                        double tx, ty, tz;
                        to_light(false, Lt, H.px, H.py, H.pz, tx, ty, tz);
                        const RayD ray = mkray(H.px + 0.00001 * tx, H.py + 0.00001 * ty, H.pz + 0.00001 * tz,
                                               tx, ty, tz, P.eps);                     // Main.hs:472
                        double ix, iy, iz, t;
                        ++nexact;
                        if (leaf_isect<KIND>(L, ray, ix, iy, iz, t, false) && guards_ok<false>(P.guards, L.guard, ray)) {
                            const double tl = (Lt.pz - ray.pz) * ray.rz;               // rayDistance ray lpos
                            const bool occ = false ? ((Lt.px - ix) * tx + (Lt.py - iy) * ty + (Lt.pz - iz) * tz >= 0)   // Talk.hs:242
                                                    : (t <= tl);                        // Main.hs:474
                            if (occ) need[r] = false;                                  // -> black
                        }
                    }
                }
            }
        }
    };
#if RT_PREFETCH
    FilterRec a = load_rec<NLOAD>(recs);                   // two SGPR buffers used in turn, see primary_loop
    int i = 0;
    for (; i + 1 < n; i += 2) {
        const FilterRec b = load_rec<NLOAD>(after(recs + i + 1, a));
        body(a, i);
        a = load_rec<NLOAD>(after(recs + i + 2, b));
        body(b, i + 1);
    }
    if (i < n) body(a, i);
#else
    for (int i = 0; i < n; ++i) body(load_rec<NLOAD>(recs + i), i);
#endif
}

#endif // RT_VALU_FILTER
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
                    const HitRec H = P.hitrec[pr];
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

#ifdef RT_VALU_FILTER   // the FP32 vector-ALU kernel family: cross-check build only (libraytrace_gpu_valu.so)
__global__ void RT_HOT(RT_WAVES_EU_SHADOW) k_shadow(const RenderParams P) {
    const int pl = (int)blockIdx.y;
    const unsigned count = P.qcount[pl];
    const unsigned wave = blockIdx.x * RT_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (wave * 256u >= count) return;
    const int lane = (int)(threadIdx.x & 63);
    // the pl-th point light
    const int li = P.point_light[pl];
    const LightD &Lt = P.lights[li];
    const uint32_t *q = P.queue + (size_t)pl * (size_t)P.qcap;
    uint32_t pix[4]; bool need[4]; int own[4];
    float ddx[4], ddy[4], ddz[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned j = wave * 256u + (unsigned)(r * 64 + lane);
        need[r] = j < count;
        pix[r] = need[r] ? q[j] : 0u;
        own[r] = need[r] ? P.hit_id[P.out_off + pix[r]] : -1;
        ddx[r] = ddy[r] = ddz[r] = 0.f;
        if (need[r]) {
            const HitRec H = P.hitrec[pix[r]];
            double tx, ty, tz;
            to_light(false, Lt, H.px, H.py, H.pz, tx, ty, tz);
            ddx[r] = (float)((H.px + 0.00001 * tx) - Lt.px);
            ddy[r] = (float)((H.py + 0.00001 * ty) - Lt.py);
            ddz[r] = (float)((H.pz + 0.00001 * tz) - Lt.pz);
        }
    }
    Dir4 D;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        D.x[p] = f2{ddx[2 * p], ddx[2 * p + 1]};
        D.y[p] = f2{ddy[2 * p], ddy[2 * p + 1]};
        D.z[p] = f2{ddz[2 * p], ddz[2 * p + 1]};
    }
    unsigned nexact = 0;
    // wave-uniform by construction; say so, or the record loads become per-lane vector loads
    const size_t ro = (size_t)__builtin_amdgcn_readfirstlane(Lt.rec_offset);
    shadow_loop<LEAF_TRI>(P, as_const(P.tri_rec + ro * (size_t)P.ntri), as_const(P.tri_gid), P.ntri, D, pix, own, Lt, need, nexact);
    shadow_loop<LEAF_RECT>(P, as_const(P.rect_rec + ro * (size_t)P.nrect), as_const(P.rect_gid), P.nrect, D, pix, own, Lt, need, nexact);
    if (P.nsph > 0) shadow_sphere_loop(P, as_const(P.sph_rec + ro * (size_t)P.nsph), D, pix, Lt, need, nexact);
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (need[r]) atomicOr(&P.lit_w[pix[r]], 1u << pl);
    for (int o = 32; o > 0; o >>= 1) nexact += __shfl_down(nexact, o);
    if (lane == 0 && nexact) stat_add(P, 2, (unsigned long long)nexact);
}

#endif // RT_VALU_FILTER
// ------------------------------------------------------------------------------------------
// Matrix-core versions of kernels 1 and 3.
//
// First stage: every ray against the bounding cone of every planar leaf, ONE linear form per pair,
//     a . D^ - cos(theta) + margin > 0        (rt_scene.cpp: make_cone_row),
// which is a matrix product (leaves x 4) x (4 x rays) and runs on v_mfma_f32_32x32x16_f16: the A operand is a block
// of 32 leaves (rows = 2^10 (a, margin - cos theta) as FP16 hi + mid pairs), the B operand is 32 rays (2^10 (D^, 1)
// split the same way: lanes 0-31 hold the hi parts, lanes 32-63 the mid parts), and the K = 16 products of a row and
// a column are exactly (a_hi + a_mid)(D_hi + D_mid) over x, y, z plus the constant.  A lane then owns one ray
// (column) and 16 leaves (registers): 8 v_max3_f32 per 32 rays x 32 leaves, beside a 32-cycle MFMA.
// A wave carries 8 B operands (256 rays).  Pairs that pass go to the wave's candidate queues (below): FP32 record
// test (exact shape, segment clip), then the exact FP64 evaluation.

// The guard boxes [0, P.nguard_cached) are the ones above the most leaves, parents first (rt_scene.cpp: order_guards).
// For the 32 x 8 pixel tile at (x0, y0): bit g of the result = EVERY camera ray of the tile certainly passes box g and
// every box above it, so the chain walk of a leaf may stop there.  "Certainly": the ray (posX, posY, 0) + t (posX, posY, d)/n
// crosses the plane z = zf at t = zf n / d, in the point (posX, posY)(1 + zf/d), which is linear in the pixel position;
// if that point lies inside the face [x0, x1] x [y0, y1] for the whole tile, with a margin 10^6 times the rounding
// errors of intersectBB's own arithmetic (Main.hs:157-161), isectXY of that face holds and the `or` (Main.hs:172) is True.
// One pixel of slack around the tile covers the sample jitter.  One lane per box; the ancestors by pointer jumping.
__device__ __forceinline__ unsigned long long tile_guard_mask(const RenderParams &P, int x0, int y0) {
    const int lane = (int)(threadIdx.x & 63u);
    bool ok = false;
    int anc = -1;
    if (lane < P.nguard_cached) {
        const GuardD B = P.guards[lane];
        anc = B.parent;
        const double xa = P.cam.scaleX * (double)(x0 - 1) + P.cam.dX, xb = P.cam.scaleX * (double)(x0 + RT_TILE_W) + P.cam.dX;
        const double ya = P.cam.scaleY * (double)(y0 - 1) + P.cam.dY, yb = P.cam.scaleY * (double)(y0 + RT_TILE_H) + P.cam.dY;
        const double XL = fmin(xa, xb), XH = fmax(xa, xb), YL = fmin(ya, yb), YH = fmax(ya, yb);
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const double zf = f ? B.z0 : B.z1;
            const double q = zf / P.cam.d;                         // t >= 0 (Main.hs:161)  <=>  q >= 0
            const double sc = 1.0 + q;
            const double lx = XL * sc, hx = XH * sc, ly = YL * sc, hy = YH * sc;
            const double mx = 1e-9 * (fabs(B.x0) + fabs(B.x1) + fabs(lx) + fabs(hx)) + 1e-300;
            const double my = 1e-9 * (fabs(B.y0) + fabs(B.y1) + fabs(ly) + fabs(hy)) + 1e-300;
            ok |= q > 1e-9 && B.x0 + mx <= lx && hx <= B.x1 - mx && B.y0 + my <= ly && hy <= B.y1 - my;
        }
    }
#pragma unroll
    for (int it = 0; it < 6; ++it) {                               // ok[g] &= ok[anc[g]]; anc[g] = anc[anc[g]]
        const int src = anc >= 0 ? anc : lane;
        const int aok = __shfl((int)ok, src), aanc = __shfl(anc, src);
        if (anc >= 0) { ok = ok && aok != 0; anc = aanc; }
    }
    return __builtin_amdgcn_ballot_w64(ok);
}

// The same for the shadow rays of one wave (Main.hs:472: origin o = hit + 1e-5 L^, direction L^ towards the light at lp):
// all origins lie in the box [lo, hi].  A ray crosses the plane a = f of a guard box at the fraction
// u = (f - o_a) / (lp_a - o_a) of the way to the light (u > 0: ahead of the origin), which is monotone in o_a, and the
// other two coordinates of the crossing, o_b + (lp_b - o_b) u, are bilinear in (o_b, u): their extremes over the box
// of origins are at its corners.  If for some face every crossing lies inside the face, with a margin far above the
// rounding errors of intersectBB's arithmetic AND of this function's own, that face's test holds for every ray of the wave.
// The proof itself runs in FP32 (it is 6 faces x 2 divisions per box and wave: 4.8 us of a 79 us wave in FP64, 6 % of
// the kernel).  Its error budget, with S_a = |f| + |lp_a| + |lo_a| + |hi_a|: the four inputs are rounded to FP32 and
// subtracted, so n and d are off by < 3e-7 S_a; `tiny` = 1e-5 S_a keeps the sign of n certain; with |d| > `far` >= 1e-2 S_a
// the fraction u is off by < 3e-5 (1 + u) + 3e-7 u (the reciprocal), and a crossing coordinate by
// < 3.2e-5 (|lp_b| + |o_b|)(1 + u); the margin m is 1e-4 of that scale plus the box's own.  The box of origins was
// rounded to FP32 as well (6e-8 of its coordinates: inside `tiny` and m).  A light within 1 % of the scale of a face's
// plane, or a crossing within 1e-4 of the scale of the face's rim, is simply not proven: those pairs walk their chain.
__device__ __forceinline__ unsigned long long bundle_guard_mask(const RenderParams &P, const LightD &Lt, const float (&lo)[3], const float (&hi)[3]) {
    const int lane = (int)(threadIdx.x & 63u);
    bool ok = false;
    int anc = -1;
    if (lane < P.nguard_cached) {
        const GuardD B = P.guards[lane];
        anc = B.parent;
        const float b0[3] = {(float)B.x0, (float)B.y0, (float)B.z0}, b1[3] = {(float)B.x1, (float)B.y1, (float)B.z1};
        const float lp[3] = {(float)Lt.px, (float)Lt.py, (float)Lt.pz};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float f = s ? b1[a] : b0[a];
                const float n0 = f - lo[a], n1 = f - hi[a], d0 = lp[a] - lo[a], d1 = lp[a] - hi[a];
                const float S = __builtin_fabsf(f) + __builtin_fabsf(lp[a]) + __builtin_fabsf(lo[a]) + __builtin_fabsf(hi[a]);
                const float tiny = 1e-5f * S + 1e-30f;
                // every origin strictly on one side of the face and of the light's coordinate, and the face ahead; the light
                // is farther than the 1e-5 that separates the origin from the hit point the direction was taken from
                const float far = 1e-2f * S + 1e-4f;
                bool face = (n0 > tiny && n1 > tiny && d0 > far && d1 > far) || (n0 < -tiny && n1 < -tiny && d0 < -far && d1 < -far);
                const float u0 = n0 * __builtin_amdgcn_rcpf(d0), u1 = n1 * __builtin_amdgcn_rcpf(d1);
                const float ul = __builtin_fminf(u0, u1), uh = __builtin_fmaxf(u0, u1);
#pragma unroll
                for (int k = 1; k < 3; ++k) {
                    const int b = (a + k) % 3;
                    const float v00 = lo[b] + (lp[b] - lo[b]) * ul, v01 = lo[b] + (lp[b] - lo[b]) * uh;
                    const float v10 = hi[b] + (lp[b] - hi[b]) * ul, v11 = hi[b] + (lp[b] - hi[b]) * uh;
                    const float vl = __builtin_fminf(__builtin_fminf(v00, v01), __builtin_fminf(v10, v11));
                    const float vh = __builtin_fmaxf(__builtin_fmaxf(v00, v01), __builtin_fmaxf(v10, v11));
                    const float m = 1e-4f * (__builtin_fabsf(b0[b]) + __builtin_fabsf(b1[b])
                                             + (__builtin_fabsf(lp[b]) + __builtin_fabsf(lo[b]) + __builtin_fabsf(hi[b])) * (1.0f + uh)) + 1e-30f;
                    // (a NaN anywhere fails a comparison: not proven)
                    face = face && b0[b] + m <= vl && vh <= b1[b] - m;
                }
                ok |= face;
            }
        }
    }
#pragma unroll
    for (int it = 0; it < 6; ++it) {                               // ok[g] &= ok[anc[g]]; anc[g] = anc[anc[g]]
        const int src = anc >= 0 ? anc : lane;
        const int aok = __shfl((int)ok, src), aanc = __shfl(anc, src);
        if (anc >= 0) { ok = ok && aok != 0; anc = aanc; }
    }
    return __builtin_amdgcn_ballot_w64(ok);
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int i32x4;

// B fragment of one ray for lane half h: 2^10 D^ = v: h = 0 -> (x_hi, x_hi, y_hi, y_hi, z_hi, z_hi, w, w) with w = 2^10,
// h = 1 -> the mid parts and w = 0.  An inactive ray is all zero: every form is then 0, which is not > 0.
__device__ __forceinline__ f16x8 ray_fragment(float x, float y, float z, bool active, int h) {
    const float s = 1024.f * __builtin_amdgcn_rsqf(__builtin_fmaf(x, x, __builtin_fmaf(y, y, z * z)));
    const float vx = x * s, vy = y * s, vz = z * s;
    _Float16 hx = (_Float16)vx, hy = (_Float16)vy, hz = (_Float16)vz, hw = (_Float16)1024.f;
    if (h) {
        hx = (_Float16)(vx - (float)hx);
        hy = (_Float16)(vy - (float)hy);
        hz = (_Float16)(vz - (float)hz);
        hw = (_Float16)0.f;
    }
    if (!active) hx = hy = hz = hw = (_Float16)0.f;
    return f16x8{hx, hx, hy, hy, hz, hz, hw, hw};
}

// The scan is software-pipelined.  One step = one MFMA (32 rays x the 32 leaves of a block) and 8 v_max3_f32 over
// its 16 results per lane.  The MFMA of step k + 1 is issued BEFORE the VALU work on step k, into a second
// accumulator, so the matrix pipe (32 cycles per MFMA) and the vector pipe (8 cycles of issue hold per MFMA + 4
// per VALU instruction) of the SIMD run side by side inside one wave and no s_nop separates an MFMA from the read of
// its result.  There is one branch per block of 8 steps.  Only the per-step maxima m[0..7] are kept: the rare path
// re-issues the MFMA of a step that has a positive lane (bit-identical) and derives the 16-bit pass mask.
#ifndef RT_MF_SCHED
#define RT_MF_SCHED 1
#endif
// "Some form is > 0" is tested on the bit patterns with v_max3_i32: a positive float is a positive integer, and an
// integer max needs no NaN canonicalisation of its inputs (v_max3_f32 does: two more instructions per step).  A NaN
// with a clear sign bit counts as a candidate, which is the safe side.
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }
__device__ __forceinline__ int step_max(const f32x16 acc) {
    int a[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = __float_as_int(acc[q]);
    const int m0 = max3i(a[0], a[1], a[2]), m1 = max3i(a[3], a[4], a[5]), m2 = max3i(a[6], a[7], a[8]);
    const int m3 = max3i(a[9], a[10], a[11]), m4 = max3i(a[12], a[13], a[14]);
    return max(max3i(m0, m1, m2), max3i(m3, m4, a[15]));
}
__device__ __forceinline__ f32x16 mfma0(const i32x4 a, const f16x8 b) {
    const f32x16 zero = {0};
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), b, zero, 0, 0, 0);
}
__device__ __forceinline__ unsigned step_bits(const f32x16 acc) {
    unsigned bits = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) bits |= (__float_as_int(acc[q]) > 0 ? 1u : 0u) << q;
    return bits;
}
// Candidate queues of one wave, in LDS.  The first stage appends one entry per (ray, leaf) pair that passes; entries
// are taken off 64 at a time, so that every lane of the wave works on one: stage1 (the FP32 record test) sends its
// survivors to a second queue, whose entries go through the exact FP64 evaluation, again 64 at a time.  Without the
// queues those two stages ran on the few lanes that happened to hold a candidate (5 % of the lanes on average).
//   q1 entry = (row of the leaf in its block << 8) | ray of the wave (0..255), 16 bits; q1 only ever holds entries of the
//   block whose records are staged in `rec` (it is emptied before the next block with candidates is looked at);
//   q2 entry = (leaf index << 9) | (leaf kind << 8) | ray: everything the exact stage needs, so it survives the block.
// The counts are wave-uniform.
// `rec`: the FP32 records of the 32 leaves of the current block, structure-of-arrays (c[k] of row r at rec[32 k + r]).
// They arrive with ONE coalesced 2 KiB load per block (32 bytes per lane) instead of a 64-byte gather per candidate:
// a gather of 64 different lines costs the CU's address path about 64 cycles per instruction, and with four of them
// per batch of 64 candidates that path, shared by all waves of the CU, was what the kernels were waiting for.  Read
// back, lanes with different rows hit different LDS banks and lanes with the same row share a broadcast.
constexpr int RT_Q1_CAP = 16 * 64 + 64, RT_Q2_CAP = 128;      // q1: one ray block can add 16 x 64 entries to the 63 that may wait
struct WaveQueues {
    uint16_t *q1; uint32_t *q2;
    unsigned n1, n2;
    unsigned nrec;                  // FP32 record tests so far (statistics)
    float *rec;                     // 16 x 32 floats
#ifdef RT_WAVE_TIMES
    unsigned long long t_s1 = 0, t_ex = 0, t_bits = 0, t_loop = 0;   // 100 MHz ticks spent in stage 1, exact, bit extraction, queueing loop
#endif
};
#ifdef RT_WAVE_TIMES
#define RT_TICK(var) const unsigned long long var = wall_clock64();
#define RT_TOCK(acc, var) acc += wall_clock64() - var;
#else
#define RT_TICK(var)
#define RT_TOCK(acc, var)
#endif
__device__ __forceinline__ void q_push(uint32_t *q, unsigned &n, bool pred, uint32_t entry) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(pred);
    if (pred) q[n + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = entry;
    n += (unsigned)__popcll(m);
}
// Works q1 down to `keep` entries; flush2: also the exact stage's last partial batch (end of the scan).
template <class S1, class EX>
__device__ __forceinline__ void q_drain(WaveQueues &Q, unsigned keep, bool flush2, S1 &&stage1, EX &&exact) {
    const unsigned lane = threadIdx.x & 63u;
    while (Q.n1 > keep) {
        const unsigned take = Q.n1 < 64u ? Q.n1 : 64u;
        Q.n1 -= take;
        Q.nrec += take;
        const bool have = lane < take;
        const uint32_t entry = have ? (uint32_t)Q.q1[Q.n1 + lane] : 0u;
        uint32_t entry2 = 0u;
        RT_TICK(ts)
        const bool pass = have && stage1(entry, entry2);
        q_push(Q.q2, Q.n2, pass, entry2);
        RT_TOCK(Q.t_s1, ts)
        if (Q.n2 >= 64u) {
            Q.n2 -= 64u;
            RT_TICK(te)
            exact(Q.q2[Q.n2 + lane]);
            RT_TOCK(Q.t_ex, te)
        }
    }
    if (flush2 && Q.n2 > 0u) {
        const bool have = lane < Q.n2;
        const uint32_t entry = have ? Q.q2[lane] : 0u;
        Q.n2 = 0u;
        RT_TICK(te)
        if (have) exact(entry);
        RT_TOCK(Q.t_ex, te)
    }
}
// Scans the n blocks (a multiple of 3) whose indices are list[0..n-1]; list[n] and list[n + 1] must name a block too (the
// prefetch runs two blocks ahead).  Blocks are 32 x i32x4 at the wave-uniform address `base`.  Lane (c, h) owns ray
// column c and 16 of the 32 leaves of every block.
// The A fragments live in a ring of three registers (in use, next, in flight) and the loop body is written out three
// times, so a fragment is loaded straight into the register it is used from: no copies, two blocks of prefetch.
template <class S1, class EX>
__device__ __forceinline__ void mfma_scan(const i32x4 *base, const FilterRec *recs, const uint16_t *list, int n, const f16x8 (&B)[8], WaveQueues &Q, int c, int h,
                                          S1 &&stage1, EX &&exact) {
    auto idx = [&](int k) { return __builtin_amdgcn_readfirstlane((int)list[k]); };      // the same for every lane
    auto blk = [&](int g) { return (base + (size_t)g * 32)[c]; };      // uniform base + lane offset
    int g0 = idx(0), g1 = idx(1), g2;
    i32x4 a0 = blk(g0), a1 = blk(g1), a2;
    f32x16 acc = mfma0(a0, B[0]);
    // one block: 8 steps on fragment a_use; the last step issues the first MFMA of the next block (fragment a_nxt)
    auto block = [&](const i32x4 &a_use, const i32x4 &a_nxt, int g) {
        int m[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const f32x16 nxt = b < 7 ? mfma0(a_use, B[b + 1]) : mfma0(a_nxt, B[0]);
#if RT_MF_SCHED
            __builtin_amdgcn_sched_barrier(0);      // the MFMA of the next step goes first ...
#endif
            m[b] = step_max(acc);
#if RT_MF_SCHED
            __builtin_amdgcn_sched_barrier(0);      // ... and nothing of the next step moves up into this one
#endif
            acc = nxt;
        }
        const int mx = max3i(max3i(m[0], m[1], m[2]), max3i(m[3], m[4], m[5]), max(m[6], m[7]));
#ifdef RT_MF_NOSLOW
        if (mx == 0x7fffffff)
#endif
        if (__builtin_amdgcn_ballot_w64(mx > 0)) {       // wave-uniform: every lane takes part in the MFMAs below
            // the records of this block's 32 leaves: 2 KiB, 32 bytes per lane (lane l: half l & 1 of record l >> 1)
            const int ln = (int)(threadIdx.x & 63u);
            const float4 *rsrc = reinterpret_cast<const float4 *>(recs + (size_t)g * MF_LEAVES) + 2 * ln;
            const float4 r0 = rsrc[0], r1 = rsrc[1];
            RT_TICK(tb)
            unsigned mb = 0;                        // ray blocks with a candidate
#pragma unroll
            for (int b = 0; b < 8; ++b) mb |= (__builtin_amdgcn_ballot_w64(m[b] > 0) != 0ull ? 1u : 0u) << b;
            {
                float *dst = Q.rec + (ln & 1) * 8 * 32 + (ln >> 1);
                dst[0 * 32] = r0.x; dst[1 * 32] = r0.y; dst[2 * 32] = r0.z; dst[3 * 32] = r0.w;
                dst[4 * 32] = r1.x; dst[5 * 32] = r1.y; dst[6 * 32] = r1.z; dst[7 * 32] = r1.w;
                __builtin_amdgcn_wave_barrier();
            }
            RT_TOCK(Q.t_bits, tb)
            RT_TICK(tl)
            // The MFMA of a ray block with candidates is issued again (bit-identical) and each of its 16 accumulator
            // registers is compared on its own: the compare IS the ballot (a lane mask in scalar registers), so the lanes
            // that pass append their entries with one prefix count, no per-lane bit vectors and no loop over their bits.
            // Register q of lane half h is row (q & 3) + 8 (q >> 2) + 4 h of the block; entry = (row << 8) | ray.
            const uint32_t lane_part = ((uint32_t)(4 * h) << 8) | (uint32_t)c;
#pragma unroll 1
            for (int b = 0; b < 8; ++b) {           // (not unrolled: one copy of the candidate stages)
                if (!((mb >> b) & 1u)) continue;
                f16x8 Bb;
                switch (b) {                        // wave-uniform
                case 0: Bb = B[0]; break; case 1: Bb = B[1]; break; case 2: Bb = B[2]; break; case 3: Bb = B[3]; break;
                case 4: Bb = B[4]; break; case 5: Bb = B[5]; break; case 6: Bb = B[6]; break; default: Bb = B[7]; break;
                }
                const f32x16 pa = mfma0(a_use, Bb);
                const uint32_t eb = lane_part + (uint32_t)(b * 32);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const bool p = __float_as_int(pa[q]) > 0;
                    const unsigned long long pm = __builtin_amdgcn_ballot_w64(p);
                    if (pm) {                       // wave-uniform
                        if (p) Q.q1[Q.n1 + __builtin_amdgcn_mbcnt_hi((unsigned)(pm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pm, 0u))] =
                            (uint16_t)(eb + (uint32_t)(((q & 3) + 8 * (q >> 2)) << 8));
                        Q.n1 += (unsigned)__popcll(pm);
                    }
                }
                q_drain(Q, 63u, false, stage1, exact);
            }
            q_drain(Q, 0u, false, stage1, exact);       // q1 is empty before `rec` is overwritten
            __builtin_amdgcn_wave_barrier();
#ifndef RT_MF_KEEP_ACC
            // the first MFMA of the next block once more (bit-identical): its 16 result registers then need not live through
            // the candidate stages above, which is where the kernels run out of registers
            acc = mfma0(a_nxt, B[0]);
#endif
            RT_TOCK(Q.t_loop, tl)
        }
    };
    for (int k = 0; k < n; k += 3) {
        g2 = idx(k + 2); a2 = blk(g2); block(a0, a1, g0);
        g0 = idx(k + 3); a0 = blk(g0); block(a1, a2, g1);
        g1 = idx(k + 4); a1 = blk(g1); block(a2, a0, g2);
    }
}
// Bound of a wave's ray directions, in LDS (the same for every lane): a capsule on the unit sphere = the directions
// within rho of the great-circle arc from p to q.  Primary tiles are cones (p = q); the shadow rays of a pixel tile leave
// from points strung out along the tile's view pyramid, which the light sees as an arc.
//   [0..2] p  [3..5] q  [6..8] n = p x q / |p x q|  [9..11] e1 = n x p  [12..14] e2 = q x n  [15] cos rho  [16] sin rho
//   [17] 1: test blocks against the bound (0: scan every block)  [18] 1: p != q
constexpr int RT_BUNDLE_FLOATS = 20;
__device__ __forceinline__ void store_bundle(float *bun, const float p[3], const float q[3], bool arc, float cr, bool cull) {
    float n[3] = {0.f, 0.f, 0.f}, e1[3] = {0.f, 0.f, 0.f}, e2[3] = {0.f, 0.f, 0.f};
    if (arc) {
        const float cx = p[1] * q[2] - p[2] * q[1], cy = p[2] * q[0] - p[0] * q[2], cz = p[0] * q[1] - p[1] * q[0];
        const float rn = __builtin_amdgcn_rsqf(__builtin_fmaf(cx, cx, __builtin_fmaf(cy, cy, cz * cz)));
        n[0] = cx * rn; n[1] = cy * rn; n[2] = cz * rn;
        e1[0] = n[1] * p[2] - n[2] * p[1]; e1[1] = n[2] * p[0] - n[0] * p[2]; e1[2] = n[0] * p[1] - n[1] * p[0];
        e2[0] = q[1] * n[2] - q[2] * n[1]; e2[1] = q[2] * n[0] - q[0] * n[2]; e2[2] = q[0] * n[1] - q[1] * n[0];
    }
    cr -= 1e-6f;                                                  // on the wide side
    const float sr = __builtin_sqrtf(__builtin_fmaxf(0.f, (1.f - cr) * (1.f + cr)));
    if ((threadIdx.x & 63u) == 0u) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { bun[k] = p[k]; bun[3 + k] = q[k]; bun[6 + k] = n[k]; bun[9 + k] = e1[k]; bun[12 + k] = e2[k]; }
        bun[15] = cr; bun[16] = sr;
        bun[17] = (cull && cr >= 0.1f) ? 1.f : 0.f;               // a NaN or a bound wider than 84 degrees: scan every block
        bun[18] = arc ? 1.f : 0.f;
    }
}
// The first stage over all `groups` blocks of one apex: every lane tests one block's bounding cone (`cones`, 8 floats
// per block, rt_scene.cpp: block_cone) against the wave's bound; the blocks that may hold a leaf for some ray of the
// wave go to `list` (LDS, RT_LIST_CAP + 8 entries), which mfma_scan works off RT_LIST_CAP blocks at a time.  Block
// `groups` is the first padding block (never rows): it fills the list up to a multiple of 3 and feeds the prefetch.
constexpr int RT_LIST_CAP = 192;
template <class S1, class EX>
__device__ __forceinline__ void culled_scan(const i32x4 *base, const FilterRec *recs, const float *cones, int groups, const float *bun, uint16_t *list,
                                            const f16x8 (&B)[8], WaveQueues &Q, int c, int h, unsigned &nblocks, S1 &&stage1, EX &&exact) {
    const int lane = (int)(threadIdx.x & 63u);
    int n = 0;                                                    // wave-uniform
    // the wave's bound, once, as scalars (it is the same for every lane)
    float bn[19];
#pragma unroll
    for (int k = 0; k < 19; ++k) bn[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bun[k])));
    const bool test = bn[17] != 0.f, arc = bn[18] != 0.f;
    const float cr = bn[15], sr = bn[16];
    const float4 *cones4 = reinterpret_cast<const float4 *>(cones);
    // Four rounds of 64 blocks at a time: their cones are loaded together (one trip to the L2 instead of four; with one
    // round per trip the 8 rounds of a 500-block stream were 4 us of a 40 us wave), then tested and listed in order.
    for (int g0 = 0; g0 < groups; g0 += 256) {
        float4 ka[4], kb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = g0 + 64 * u + lane;
            const int gg = g < groups ? g : 0;
            ka[u] = cones4[2 * gg]; kb[u] = cones4[2 * gg + 1];          // {axis xyz, cos T}, {sin T, code, 0, 0}
        }
        unsigned long long mk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = g0 + 64 * u + lane;
            bool pass = g < groups;
            if (pass && test) {
                const float4 k = ka[u];
                const float sT = kb[u].x, code = kb[u].y;              // code 1: never, 2: always
                // the block's cone (axis k.xyz, half-angle T) meets the capsule iff its axis lies within T + rho of the arc
                const float cc = __builtin_fmaf(cr, k.w, -(sr * sT)) - 4e-6f;         // cos(T + rho), low side
                const float ss = __builtin_fmaf(sr, k.w, cr * sT) + 4e-6f;            // sin(T + rho), high side
                bool geo = cc <= 0.f;                                                  // T + rho >= 90 degrees
                geo |= __builtin_fmaf(bn[0], k.x, __builtin_fmaf(bn[1], k.y, bn[2] * k.z)) >= cc;
                if (arc) {
                    geo |= __builtin_fmaf(bn[3], k.x, __builtin_fmaf(bn[4], k.y, bn[5] * k.z)) >= cc;
                    // between the end points (taken generously: beyond them the great circle is nearer than the arc)
                    const bool between = __builtin_fmaf(bn[9], k.x, __builtin_fmaf(bn[10], k.y, bn[11] * k.z)) >= -1e-3f
                                       && __builtin_fmaf(bn[12], k.x, __builtin_fmaf(bn[13], k.y, bn[14] * k.z)) >= -1e-3f;
                    geo |= between && __builtin_fabsf(__builtin_fmaf(bn[6], k.x, __builtin_fmaf(bn[7], k.y, bn[8] * k.z))) <= ss;
                }
                pass = code == 2.f || (code == 0.f && geo);
            }
            mk[u] = __builtin_amdgcn_ballot_w64(pass);
        }
#pragma unroll 1
        for (int u = 0; u < 4 && g0 + 64 * u < groups; ++u) {          // (not unrolled: one copy of the scan)
            const unsigned long long m = u == 0 ? mk[0] : (u == 1 ? mk[1] : (u == 2 ? mk[2] : mk[3]));
            const int g = g0 + 64 * u + lane;
            if ((m >> lane) & 1ull) list[n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (uint16_t)g;
            n += (int)__popcll(m);
            if (n > RT_LIST_CAP - 64 || g0 + 64 * u + 64 >= groups) {
                nblocks += (unsigned)n;
                if (n > 0) {
                    if (lane < 5) list[n + lane] = (uint16_t)groups;  // up to a multiple of 3, then two blocks of prefetch
                    __builtin_amdgcn_wave_barrier();
                    mfma_scan(base, recs, list, (n + 2) / 3 * 3, B, Q, c, h, stage1, exact);
                    __builtin_amdgcn_wave_barrier();
                }
                n = 0;
            }
        }
    }
    q_drain(Q, 0u, true, stage1, exact);
}
// FP32 record test of ONE ray (the second stage, rare path): the exact shape of the leaf, same margins as filter4.
__device__ __forceinline__ bool record_test(const FilterRec *R, int kind, float x, float y, float z) {
    const float q1 = __builtin_fmaf(R->c[0], x, __builtin_fmaf(R->c[1], y, R->c[2] * z));
    const float q2 = __builtin_fmaf(R->c[3], x, __builtin_fmaf(R->c[4], y, R->c[5] * z));
    const float n = __builtin_fmaf(R->c[6], x, __builtin_fmaf(R->c[7], y, R->c[8] * z));
    const float zz = kind == LEAF_TRI ? __builtin_fminf(__builtin_fminf(q1, q2), n)
                                      : n - __builtin_fmaxf(__builtin_fabsf(q1), __builtin_fabsf(q2));
    return zz >= R->c[9];
}

// the same on a record staged in LDS (WaveQueues::rec), row = the leaf's row in its block
__device__ __forceinline__ bool record_test_staged(const float *rec, int row, int kind, float x, float y, float z) {
    const float *R = rec + row;
    const float q1 = __builtin_fmaf(R[0 * 32], x, __builtin_fmaf(R[1 * 32], y, R[2 * 32] * z));
    const float q2 = __builtin_fmaf(R[3 * 32], x, __builtin_fmaf(R[4 * 32], y, R[5 * 32] * z));
    const float n = __builtin_fmaf(R[6 * 32], x, __builtin_fmaf(R[7 * 32], y, R[8 * 32] * z));
    const float zz = kind == LEAF_TRI ? __builtin_fminf(__builtin_fminf(q1, q2), n)
                                      : n - __builtin_fmaxf(__builtin_fabsf(q1), __builtin_fabsf(q2));
    return zz >= R[9 * 32];
}

// Waves per SIMD the two matrix-core kernels are compiled for.  Measured (profiles/README.md, round 2): the shadow
// kernel at 4 (128 VGPRs) spills its scan and loses 60 %; the primary kernel at 4 spills 120 B per lane in the exact
// stage only and gains 7 %.
#ifndef RT_WAVES_EU_MF_SHADOW
#define RT_WAVES_EU_MF_SHADOW 3
#endif
#ifndef RT_WAVES_EU_MF_PRIMARY
#define RT_WAVES_EU_MF_PRIMARY 4
#endif

// Kernel 3 on the matrix cores.  Lane (c = lane & 31, h = lane >> 5): ray column c of each of the wave's 8 ray
// blocks, half of the leaves of each leaf block.
__global__ void RT_HOT(RT_WAVES_EU_MF_SHADOW) k_shadow_mfma(const RenderParams P) {
    RT_WAVE_T0
    const int pl = (int)P.qcount[RT_QCOUNT_ORDER + blockIdx.y];      // costliest light first (k_queue_offsets)
    const unsigned count = P.qcount[pl];
    const unsigned wave = blockIdx.x * RT_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (wave * 256u >= count) return;
    const unsigned long long t_start = wall_clock64();
    const int lane = (int)(threadIdx.x & 63), c = lane & 31, h = lane >> 5;
    const int li = P.point_light[pl];
    const LightD &Lt = P.lights[li];
    const uint32_t *q = P.queue + (size_t)pl * (size_t)P.qcap;
    // per ray of the wave: D (FP32), the queue entry, the leaf it starts on, "an occluder was found"
    __shared__ float s_d[RT_WAVES_PER_BLOCK][3][256];
    __shared__ uint32_t s_pix[RT_WAVES_PER_BLOCK][256];
    __shared__ int s_own[RT_WAVES_PER_BLOCK][256];
    __shared__ uint32_t s_occ[RT_WAVES_PER_BLOCK][256];
    __shared__ uint16_t s_q1[RT_WAVES_PER_BLOCK][RT_Q1_CAP];
    __shared__ uint32_t s_q2[RT_WAVES_PER_BLOCK][RT_Q2_CAP];
    __shared__ uint16_t s_list[RT_WAVES_PER_BLOCK][RT_LIST_CAP + 8];
    __shared__ float s_bun[RT_WAVES_PER_BLOCK][RT_BUNDLE_FLOATS];
    __shared__ float s_rec[RT_WAVES_PER_BLOCK][16 * 32];
    const int wv = (int)(threadIdx.x >> 6);
    f16x8 B[8];
    float ux[8], uy[8], uz[8];              // unit directions of the lane's 8 rays (0: no ray), for the bundle cone
    float olo[3] = {INFINITY, INFINITY, INFINITY}, ohi[3] = {-INFINITY, -INFINITY, -INFINITY};   // box of the ray origins (for the guard proofs)
    // lane half h sets up the rays of blocks 4h .. 4h + 3 (hit record, FP64 direction to the light) and leaves them in
    // LDS; then every lane picks up column c of all 8 blocks
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = 4 * h + r;
        const unsigned j = wave * 256u + (unsigned)(b * 32 + c);
        const bool ok = j < count;
        float dx = 0.f, dy = 0.f, dz = 0.f;
        uint32_t pix = 0u;
        if (ok) {
            pix = q[j];
            const HitRec H = P.hitrec[pix];
            double tx, ty, tz;
            to_light(RT_TALK(P), Lt, H.px, H.py, H.pz, tx, ty, tz);
            const double o[3] = {H.px + 0.00001 * tx, H.py + 0.00001 * ty, H.pz + 0.00001 * tz};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float of = (float)o[k];
                olo[k] = __builtin_fminf(olo[k], of); ohi[k] = __builtin_fmaxf(ohi[k], of);
                if (of != of) olo[k] = of;                 // a NaN origin: nothing is certain
            }
            dx = (float)(o[0] - Lt.px);
            dy = (float)(o[1] - Lt.py);
            dz = (float)(o[2] - Lt.pz);
        }
        const int e = b * 32 + c;
        s_d[wv][0][e] = dx; s_d[wv][1][e] = dy; s_d[wv][2][e] = dz;
        s_pix[wv][e] = pix;
        s_own[wv][e] = ok ? P.hit_id[P.out_off + pix] : -1;
        s_occ[wv][e] = ok ? 0u : 1u;          // a ray that does not exist counts as occluded: never lit
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int e = b * 32 + c;
        const float dx = s_d[wv][0][e], dy = s_d[wv][1][e], dz = s_d[wv][2][e];
        const bool ok = s_occ[wv][e] == 0u;
        B[b] = ray_fragment(dx, dy, dz, ok, h);
        const float rn = ok ? __builtin_amdgcn_rsqf(__builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz))) : 0.f;
        ux[b] = dx * rn; uy[b] = dy * rn; uz[b] = dz * rn;
    }
    __builtin_amdgcn_wave_barrier();
    RT_WAVE_PHASE(0)
    // bound of the wave's directions (both lane halves hold the same rays): p = the ray farthest from the mean direction,
    // q = the ray farthest from p, rho = the largest distance of a ray from the arc p..q
    {
        bool some[8], bad = false;
        float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            some[b] = (ux[b] != 0.f) | (uy[b] != 0.f) | (uz[b] != 0.f);
            bad |= (ux[b] != ux[b]) | (uy[b] != uy[b]) | (uz[b] != uz[b]);
            ax += ux[b]; ay += uy[b]; az += uz[b];
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { ax += __shfl_xor(ax, o); ay += __shfl_xor(ay, o); az += __shfl_xor(az, o); }
        // the ray of the wave with the smallest dot product against (vx, vy, vz)
        auto farthest = [&](float vx, float vy, float vz, float out[3]) {
            float best = 3.0e38f, bx = 0.f, by = 0.f, bz = 0.f;      // (the mean direction comes unnormalised)
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const float dt = __builtin_fmaf(vx, ux[b], __builtin_fmaf(vy, uy[b], vz * uz[b]));
                if (some[b] && dt < best) { best = dt; bx = ux[b]; by = uy[b]; bz = uz[b]; }
            }
            float m = best;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) m = __builtin_fminf(m, __shfl_xor(m, o));
            const unsigned long long who = __builtin_amdgcn_ballot_w64(best == m);
            const int src = who ? __ffsll((long long)who) - 1 : 0;
            out[0] = __shfl(bx, src); out[1] = __shfl(by, src); out[2] = __shfl(bz, src);
        };
        float p[3], q[3];
        farthest(ax, ay, az, p);
        farthest(p[0], p[1], p[2], q);
        const float cx = p[1] * q[2] - p[2] * q[1], cy = p[2] * q[0] - p[0] * q[2], cz = p[0] * q[1] - p[1] * q[0];
        const float dpq = p[0] * q[0] + p[1] * q[1] + p[2] * q[2];
        // an arc between 0.06 and 168 degrees; otherwise a cone around p
        const bool arc = __builtin_fmaf(cx, cx, __builtin_fmaf(cy, cy, cz * cz)) > 1e-6f && dpq > -0.98f;
        if (!arc) { q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; }
        float nn[3] = {0.f, 0.f, 0.f}, e1[3] = {0.f, 0.f, 0.f}, e2[3] = {0.f, 0.f, 0.f};
        if (arc) {
            const float rn = __builtin_amdgcn_rsqf(__builtin_fmaf(cx, cx, __builtin_fmaf(cy, cy, cz * cz)));
            nn[0] = cx * rn; nn[1] = cy * rn; nn[2] = cz * rn;
            e1[0] = nn[1] * p[2] - nn[2] * p[1]; e1[1] = nn[2] * p[0] - nn[0] * p[2]; e1[2] = nn[0] * p[1] - nn[1] * p[0];
            e2[0] = q[1] * nn[2] - q[2] * nn[1]; e2[1] = q[2] * nn[0] - q[0] * nn[2]; e2[2] = q[0] * nn[1] - q[1] * nn[0];
        }
        float cmin = 1.f;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const float dp = __builtin_fmaf(p[0], ux[b], __builtin_fmaf(p[1], uy[b], p[2] * uz[b]));
            const float dq = __builtin_fmaf(q[0], ux[b], __builtin_fmaf(q[1], uy[b], q[2] * uz[b]));
            float cd = __builtin_fmaxf(dp, dq);                  // never farther from the arc than from its end points
            // strictly between the end points: the distance to the great circle
            const bool between = arc && __builtin_fmaf(e1[0], ux[b], __builtin_fmaf(e1[1], uy[b], e1[2] * uz[b])) >= 1e-3f
                                     && __builtin_fmaf(e2[0], ux[b], __builtin_fmaf(e2[1], uy[b], e2[2] * uz[b])) >= 1e-3f;
            const float dn = __builtin_fmaf(nn[0], ux[b], __builtin_fmaf(nn[1], uy[b], nn[2] * uz[b]));
            if (between) cd = __builtin_fmaxf(cd, __builtin_sqrtf(__builtin_fmaxf(0.f, __builtin_fmaf(-dn, dn, 1.f))) - 2e-7f);
            if (some[b]) cmin = __builtin_fminf(cmin, cd);
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) cmin = __builtin_fminf(cmin, __shfl_xor(cmin, o));
        bad = __builtin_amdgcn_ballot_w64(bad) != 0ull;
        store_bundle(s_bun[wv], p, q, arc, cmin, P.cull != 0 && !bad);
    }
    __builtin_amdgcn_wave_barrier();
    unsigned long long passed = 0ull;
    if (P.nguard_cached > 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float l2 = __shfl_xor(olo[k], o), h2 = __shfl_xor(ohi[k], o);
                olo[k] = (l2 != l2 || olo[k] != olo[k]) ? NAN : __builtin_fminf(olo[k], l2);     // a NaN origin: nothing is certain
                ohi[k] = __builtin_fmaxf(ohi[k], h2);
            }
        passed = bundle_guard_mask(P, Lt, olo, ohi);
    }
    RT_WAVE_PHASE(1)
    const size_t ro = (size_t)__builtin_amdgcn_readfirstlane(Lt.rec_offset);
    const size_t gstride = (size_t)(P.mf_groups + 2);
    const i32x4 *blocks = reinterpret_cast<const i32x4 *>(P.mf_rows) + ro * gstride * 32;
    const FilterRec *recs = P.mf_rec + ro * gstride * MF_LEAVES;
    unsigned nexact = 0, nblocks = 0;
    WaveQueues Q = {s_q1[wv], s_q2[wv], 0u, 0u, 0u, s_rec[wv]};
    culled_scan(blocks, recs, P.mf_cone + ro * gstride * 8, P.mf_groups, s_bun[wv], s_list[wv], B, Q, c, h, nblocks,
        [&](uint32_t entry, uint32_t &entry2) -> bool {           // FP32 record test of one (ray, leaf) pair
            const int e = (int)(entry & 255u), row = (int)((entry >> 8) & 31u);
            const float *R = Q.rec + row;
            if (s_occ[wv][e]) return false;
            const float dx = s_d[wv][0][e], dy = s_d[wv][1][e], dz = s_d[wv][2][e];
            const int kind = __float_as_int(R[14 * 32]), gid = __float_as_int(R[15 * 32]);
            if (gid < 0 || !record_test_staged(Q.rec, row, kind, dx, dy, dz)) return false;
            // only leaves between the light and the origin: N.D >= Kq (Main.hs:474 needs t <= distance)
            if (!(__builtin_fmaf(R[11 * 32], dx, __builtin_fmaf(R[12 * 32], dy, R[13 * 32] * dz)) >= R[10 * 32])) return false;
            entry2 = ((uint32_t)gid << 9) | ((uint32_t)kind << 8) | (uint32_t)e;
            // the hit leaf itself lies 1e-5 behind the shadow-ray origin (Main.hs:472): t < 0
            return s_own[wv][e] != gid;
        },
        [&](uint32_t entry) {                   // exact: intersect world ray_to_light, t <= distance (Main.hs:472-474)
            const int e = (int)(entry & 255u);
#ifdef RT_X_SCANONLY      // measurement only (wrong pixels): what the kernel costs without its exact stage and without the early-out of occluded rays
            s_own[wv][e] = (int)(entry >> 9); ++nexact; return;
#endif
            const int kind = (int)((entry >> 8) & 1u), gid = (int)(entry >> 9);
            const LeafD &L = P.leaves[gid];
            const HitRec H = P.hitrec[s_pix[wv][e]];
            double tx, ty, tz;
            to_light(RT_TALK(P), Lt, H.px, H.py, H.pz, tx, ty, tz);
            const RayD ray = mkray(H.px + 0.00001 * tx, H.py + 0.00001 * ty, H.pz + 0.00001 * tz, tx, ty, tz, P.eps);   // Main.hs:472
            double ix, iy, iz, tt;
            ++nexact;
            const bool hit = kind == LEAF_TRI ? leaf_isect<LEAF_TRI>(L, ray, ix, iy, iz, tt, RT_TALK(P)) : leaf_isect<LEAF_RECT>(L, ray, ix, iy, iz, tt, RT_TALK(P));
            if (hit && guards_ok(P.guards, L.guard, ray, passed)) {
                const double tl = (Lt.pz - ray.pz) * ray.rz;                       // rayDistance ray lpos
                const bool occ = RT_TALK(P) ? ((Lt.px - ix) * tx + (Lt.py - iy) * ty + (Lt.pz - iz) * tz >= 0)   // Talk.hs:242
                                        : (tt <= tl);                               // Main.hs:474
                if (occ) s_occ[wv][e] = 1u;                                        // -> black
            }
        });
    __builtin_amdgcn_wave_barrier();
    RT_WAVE_PHASE(2)
    // lane (c, h) finishes the rays (4h + r, c): spheres with the scalar-record loop, then the lit bits
    uint32_t pix[4]; bool nd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = (4 * h + r) * 32 + c;
        nd[r] = s_occ[wv][e] == 0u;
        pix[r] = s_pix[wv][e];
    }
    if (P.nsph > 0) {
        Dir4 D;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int e0 = (4 * h + 2 * p) * 32 + c, e1 = e0 + 32;
            D.x[p] = f2{s_d[wv][0][e0], s_d[wv][0][e1]};
            D.y[p] = f2{s_d[wv][1][e0], s_d[wv][1][e1]};
            D.z[p] = f2{s_d[wv][2][e0], s_d[wv][2][e1]};
        }
        shadow_sphere_loop(P, as_const(P.sph_rec + ro * (size_t)P.nsph), D, pix, Lt, nd, nexact);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (nd[r]) atomicOr(&P.lit_w[pix[r]], 1u << pl);
    for (int o = 32; o > 0; o >>= 1) nexact += __shfl_down(nexact, o);
    if (lane == 0 && nexact) stat_add(P, 2, (unsigned long long)nexact);
    if (lane == 0) stat_add(P, 7, (unsigned long long)nblocks);
    if (lane == 0 && Q.nrec) stat_add(P, 4, (unsigned long long)Q.nrec);
    if (lane == 0 && (wave & 7u) == 3u) {       // every eighth wave reports how long it took (100 MHz ticks): the lights' order next time
        atomicAdd(&P.qcount[RT_QCOUNT_HIST + 2 * pl], (unsigned)(wall_clock64() - t_start));
        atomicAdd(&P.qcount[RT_QCOUNT_HIST + 2 * pl + 1], 1u);
    }
#ifdef RT_WAVE_TIMES
    wave_ph[2] = (Q.t_s1 & 0xffff) | ((Q.t_ex & 0xffff) << 16) | ((Q.t_bits & 0xffff) << 32) | ((Q.t_loop & 0xffff) << 48);
    wave_record(P, 1u | ((unsigned)pl << 8), wave, wave_t0, nblocks, Q.nrec, nexact, wave_ph);
#endif
}

// Kernel 1 on the matrix cores.  One wave = 32 x 8 pixel tile; ray block b = tile row b, column c = lane & 31.
// The nearest hit of every pixel of the tile lives in LDS: t as the bit pattern of the double (t >= 0, so the
// patterns order like the values) and the leaf index.
__global__ void RT_HOT(RT_WAVES_EU_MF_PRIMARY) k_primary_mfma(const RenderParams P) {
    RT_WAVE_T0
    __shared__ unsigned long long s_bt[RT_WAVES_PER_BLOCK][256];
    __shared__ int s_bid[RT_WAVES_PER_BLOCK][256];
    __shared__ uint16_t s_q1[RT_WAVES_PER_BLOCK][RT_Q1_CAP];
    __shared__ uint32_t s_q2[RT_WAVES_PER_BLOCK][RT_Q2_CAP];
    __shared__ uint16_t s_list[RT_WAVES_PER_BLOCK][RT_LIST_CAP + 8];
    __shared__ float s_bun[RT_WAVES_PER_BLOCK][RT_BUNDLE_FLOATS];
    __shared__ float s_rec[RT_WAVES_PER_BLOCK][16 * 32];
    const int wv = (int)(threadIdx.x >> 6);
    const int wave = (int)(blockIdx.x * RT_WAVES_PER_BLOCK) + wv;
    const int lane = (int)(threadIdx.x & 63), c = lane & 31, h = lane >> 5;
    const int tx = wave % P.tiles_x, ty = wave / P.tiles_x;
    if (ty >= P.tiles_y) return;
    const int x0 = tx * RT_TILE_W;
    const int ly0 = ty * RT_TILE_H;
    // Ray e = 32 b + c of the wave is the pixel (ex(e), ey(e)) of the tile: ray block b is an 8 x 4 patch (four patches side
    // by side, two rows of them), not a 32 x 1 strip — a leaf covers a blob of pixels, a blob touches fewer patches than
    // strips, and only ray blocks with a candidate get their MFMA re-issued and their 16 registers compared.
    auto ex = [](int e) { return ((e >> 5) & 3) * 8 + (e & 7); };
    auto ey = [](int e) { return (e >> 7) * 4 + ((e >> 3) & 3); };
    unsigned long long *bt = s_bt[wv];
    int *bid = s_bid[wv];
    f16x8 B[8];
    const float dzf = (float)P.cam.d;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int x = x0 + ex(b * 32 + c), ly = ly0 + ey(b * 32 + c);
        const bool ok = x < P.w && ly < P.nrows;
        double ox, oy;
        sample_offset(P, x, P.row0 + ly, ox, oy);
        const float dx = (float)(P.cam.scaleX * ((double)x + ox) + P.cam.dX);
        const float dy = (float)(P.cam.scaleY * ((double)(P.row0 + ly) + oy) + P.cam.dY);
        B[b] = ray_fragment(dx, dy, dzf, ok, h);
        if (h == 0) { bt[b * 32 + c] = ~0ull; bid[b * 32 + c] = -1; }
    }
    __builtin_amdgcn_wave_barrier();
    RT_WAVE_PHASE(0)
    const i32x4 *blocks = reinterpret_cast<const i32x4 *>(P.mf_rows);          // apex 0 = the eye
    // bundle cone of the tile: axis through its centre, half-angle to the farthest corner (the directions inside a cone
    // cut the image plane in a convex region, so the corners decide); one pixel of slack all round covers the jitter
    {
        const float cx = (float)(P.cam.scaleX * ((double)x0 + 15.5) + P.cam.dX);
        const float cy = (float)(P.cam.scaleY * ((double)(P.row0 + ly0) + 3.5) + P.cam.dY);
        const float hx = (float)(17.5 * fabs(P.cam.scaleX)), hy = (float)(5.5 * fabs(P.cam.scaleY));
        const float cn = __builtin_amdgcn_rsqf(__builtin_fmaf(cx, cx, __builtin_fmaf(cy, cy, dzf * dzf)));
        const float axis[3] = {cx * cn, cy * cn, dzf * cn};
        float cmin = 1.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float qx = cx + ((k & 1) ? hx : -hx), qy = cy + ((k & 2) ? hy : -hy);
            const float qn = __builtin_amdgcn_rsqf(__builtin_fmaf(qx, qx, __builtin_fmaf(qy, qy, dzf * dzf)));
            cmin = __builtin_fminf(cmin, (axis[0] * qx + axis[1] * qy + axis[2] * dzf) * qn);
        }
        store_bundle(s_bun[wv], axis, axis, false, cmin, P.cull != 0);
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned long long passed = P.nguard_cached > 0 ? tile_guard_mask(P, x0, P.row0 + ly0) : 0ull;
    RT_WAVE_PHASE(1)
    unsigned nexact = 0, nblocks = 0;
    WaveQueues Q = {s_q1[wv], s_q2[wv], 0u, 0u, 0u, s_rec[wv]};
#ifdef RT_X_NOSCAN
    if (P.w < 0)
#endif
    culled_scan(blocks, P.mf_rec, P.mf_cone, P.mf_groups, s_bun[wv], s_list[wv], B, Q, c, h, nblocks,
        [&](uint32_t entry, uint32_t &entry2) -> bool {           // FP32 record test of one (pixel, leaf) pair
            const int e = (int)(entry & 255u), row = (int)((entry >> 8) & 31u);
            const float *R = Q.rec + row;                                          // apex 0 = the eye
            const int xe = x0 + ex(e), ye = P.row0 + ly0 + ey(e);
            const int kind = __float_as_int(R[14 * 32]), gid = __float_as_int(R[15 * 32]);
            double ox, oy;
            sample_offset(P, xe, ye, ox, oy);
            const float px = (float)(P.cam.scaleX * ((double)xe + ox) + P.cam.dX);
            const float py = (float)(P.cam.scaleY * ((double)ye + oy) + P.cam.dY);
            entry2 = ((uint32_t)gid << 9) | ((uint32_t)kind << 8) | (uint32_t)e;
            // eye records are affine in (posX, posY): the z column already holds c_z * d
            return gid >= 0 && record_test_staged(Q.rec, row, kind, px, py, 1.0f);
        },
        [&](uint32_t entry) {                   // exact: intersect leaf (cast camera x y), Main.hs:543
            const int e = (int)(entry & 255u);
#ifdef RT_X_SCANONLY      // measurement only (wrong pixels): what the kernel costs without its exact stage
            bid[e] = (int)(entry >> 9); ++nexact; return;
#endif
            const int kind = (int)((entry >> 8) & 1u), gid = (int)(entry >> 9);
            const LeafD &L = P.leaves[gid];
            const RayD ray = camera_ray(P, x0 + ex(e), P.row0 + ly0 + ey(e));
            double ix, iy, iz, tt;
            ++nexact;
            const bool hit = (kind == LEAF_TRI ? leaf_isect<LEAF_TRI>(L, ray, ix, iy, iz, tt, RT_TALK(P)) : leaf_isect<LEAF_RECT>(L, ray, ix, iy, iz, tt, RT_TALK(P)))
                             && guards_ok(P.guards, L.guard, ray, passed);
#ifdef RT_COUNT_EVENTS
            if (hit) stat_add(P, 3, 1ull);
#endif
            // Nearest wins; equal t -> the left-most leaf of the tree (Main.hs:213-214).  Several lanes may hold hits of
            // the same pixel: the wave runs in lockstep and its LDS operations complete in order, so (1) all read the
            // old t, (2) all fold their t in with an atomic min, (3) the lanes whose t is the new minimum reset the
            // leaf index if the minimum moved, (4) and fold their leaf index in with an atomic min.
            // t = -0.0 passes `t >= 0` (a leaf in the viewport plane) and must order as 0, not as the largest pattern
            const unsigned long long tb = (unsigned long long)__double_as_longlong(tt + 0.0);
            unsigned long long old = 0;
            if (hit) { old = bt[e]; atomicMin(&bt[e], tb); }
            __builtin_amdgcn_wave_barrier();
            const bool win = hit && bt[e] == tb;
            if (win && tb < old) bid[e] = 0x7fffffff;
            __builtin_amdgcn_wave_barrier();
            if (win) atomicMin(&bid[e], gid);
        });
    __builtin_amdgcn_wave_barrier();
    RT_WAVE_PHASE(2)
    // the lane finishes rays (4h + r) * 32 + c: the table, then spheres
    double mt[4]; int mid[4]; bool act[4]; int px[4], py[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = (4 * h + r) * 32 + c;
        mid[r] = bid[e];
        mt[r] = mid[r] >= 0 ? __longlong_as_double((long long)bt[e]) : (double)INFINITY;
        px[r] = x0 + ex(e); py[r] = ly0 + ey(e);
        act[r] = px[r] < P.w && py[r] < P.nrows;
    }
    if (P.nsph > 0) {
        Dir4 D; f2 dd[2];
        float dx[4], dy[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double ox, oy;
            sample_offset(P, px[r], P.row0 + py[r], ox, oy);
            dx[r] = (float)(P.cam.scaleX * ((double)px[r] + ox) + P.cam.dX);
            dy[r] = (float)(P.cam.scaleY * ((double)(P.row0 + py[r]) + oy) + P.cam.dY);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) { D.x[p] = f2{dx[2 * p], dx[2 * p + 1]}; D.y[p] = f2{dy[2 * p], dy[2 * p + 1]}; D.z[p] = f2{dzf, dzf}; }
        dir_norms(D, dd);
        CRecPtr recs = as_const(P.sph_rec);
        CIntPtr gids = as_const(P.sph_gid);
        for (int i = 0; i < P.nsph; ++i) {
            const FilterRec cur = load_rec(recs + i);
            float z[4];
            if (filter4_sphere(cur, D, dd, z)) {
                const int gid = gids[i];
                const LeafD &L = P.leaves[gid];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if ((z[r] >= cur.c[9]) & act[r]) {
                        int yr = P.row0 + py[r];
                        asm volatile("" : "+v"(yr));
                        const RayD ray = camera_ray(P, px[r], yr);
                        double ix, iy, iz, tt;
                        ++nexact;
                        if (leaf_isect<LEAF_SPHERE>(L, ray, ix, iy, iz, tt, RT_TALK(P)) && guards_ok(P.guards, L.guard, ray)) {
                            if (tt < mt[r] || (tt == mt[r] && gid < mid[r]) || mid[r] < 0) { mt[r] = tt; mid[r] = gid; }
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (act[r]) P.hit_id[P.out_off + (long long)py[r] * P.w + px[r]] = mid[r];
    for (int o = 32; o > 0; o >>= 1) nexact += __shfl_down(nexact, o);
    if (lane == 0 && nexact) stat_add(P, 1, (unsigned long long)nexact);
    if (lane == 0) stat_add(P, 6, (unsigned long long)nblocks);
    if (lane == 0 && Q.nrec) stat_add(P, 4, (unsigned long long)Q.nrec);
#ifdef RT_WAVE_TIMES
    wave_ph[2] = (Q.t_s1 & 0xffff) | ((Q.t_ex & 0xffff) << 16) | ((Q.t_bits & 0xffff) << 32) | ((Q.t_loop & 0xffff) << 48);
    wave_record(P, 0u, (unsigned)wave, wave_t0, nblocks, Q.nrec, nexact, wave_ph);
#endif
}

// ------------------------------------------------------------------------------------------
// Kernel 4: `light lights hit ray` (Main.hs:545, 436-441) = pointLight (467-487) + ambient (490-492),
//   saturating 8-bit sums, coalesced RGBA8 store.

// Launched with 256 threads, or with 1024 when reflected rays are handed to a next level (one atomic per BLOCK then).
#ifndef RT_WAVES_EU_FINAL
#define RT_WAVES_EU_FINAL 4
#endif
__global__ void __attribute__((amdgpu_waves_per_eu(RT_WAVES_EU_FINAL, 8))) __launch_bounds__(1024) k_final(const RenderParams P) {
    // level 0: a block takes a 32 x 8 (32 x 32) pixel tile, a wave 8 x 8 pixels of it; bounce levels: slots in order
    int pix = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    bool act = (unsigned)pix < num_slots(P);
    if (!P.rays) {
        const int tx = (int)blockIdx.x % P.tiles_x, ty = (int)blockIdx.x / P.tiles_x, wq = (int)(threadIdx.x >> 6);
        const int x = tx * 32 + (wq & 3) * 8 + (int)(threadIdx.x & 7), y = ty * (int)(blockDim.x >> 5) + (wq >> 2) * 8 + (int)((threadIdx.x & 63) >> 3);
        act = x < P.w && y < P.nrows;
        pix = act ? y * P.w + x : 0;
    }
    const int id = act ? P.hit_id[P.out_off + pix] : -1;
    int cr = 0, cg = 0, cb = 0;
    bool spawn = false;
    RayRec child = {0, 0, 0, 0, 0, 0, 0, 0u, 0u};
    if (id >= 0) {
        const MaterialD M = P.mats[P.leaves[id].mat];
        const HitRec H = P.hitrec[pix];
        uint32_t lit = 0u;
        const RayD ray = slot_ray(P, pix);
        int pl = 0;
        for (int li = 0; li < P.nlights; ++li) {
            const LightD &Lt = P.lights[li];
            if (Lt.kind == 1) {                                                        // ambient, Main.hs:490-492
                cr = min(255, cr + round8(Lt.f * (double)M.diffuse[0]));
                cg = min(255, cg + round8(Lt.f * (double)M.diffuse[1]));
                cb = min(255, cb + round8(Lt.f * (double)M.diffuse[2]));
                continue;
            }
            if ((pl & 31) == 0) lit = P.lit[(long long)(pl >> 5) * P.lit_stride + pix];   // the next group of 32 point lights
            if ((lit >> (pl & 31)) & 1u) {                                             // lit: Main.hs:476-487 / Talk.hs:244-248
                int r, g, b;
                light_terms(P, Lt, M, H, ray, r, g, b);
                cr = min(255, cr + r); cg = min(255, cg + g); cb = min(255, cb + b);
            }
            ++pl;
        }
        if (P.next_rays) {                                                             // reflection, Main.hs:546-556
            const double r = H.nx * ray.dx + H.ny * ray.dy + H.nz * ray.dz;            // :549
            if (M.reflectivity > 0 && r < 0) {                                         // :553
                const double k2 = 2 * r;
                const double qx = ray.dx - k2 * H.nx, qy = ray.dy - k2 * H.ny, qz = ray.dz - k2 * H.nz;   // rdir, :550
                child.ox = H.px + 0.00001 * qx; child.oy = H.py + 0.00001 * qy; child.oz = H.pz + 0.00001 * qz;   // :552
                child.dx = qx; child.dy = qy; child.dz = qz;
                child.rho = M.reflectivity; child.parent = (uint32_t)pix;
                spawn = true;
            }
        }
    }
    if (P.next_rays) {
        // The reflected rays of a wave (8 x 8 pixels at level 0) take a run of 64 slots of the next level, lane for lane
        // (one atomic per wave); lanes that reflect nothing leave a padding entry (parent = ~0).  Kept together like
        // this, the 64 rays k_general bounds at a time leave one patch of one surface: a narrow bundle.
        // The runs of a block are claimed with ONE atomic (a wave each cost 0.4 ms per chunk: 50 000 atomics on one address
        // are serialised in the L2) and stay together in the list.
        __shared__ unsigned s_run[17];
        const int lane = (int)(threadIdx.x & 63), wq = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
        const unsigned long long m = __ballot(spawn);
        if (lane == 0) s_run[wq] = m ? 1u : 0u;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned total = 0;
            for (int k = 0; k < nw; ++k) { const unsigned c = s_run[k]; s_run[k] = total; total += c; }
            s_run[16] = total ? atomicAdd(P.next_count, 64u * total) : 0u;
        }
        __syncthreads();
        if (m) {
            const unsigned base = s_run[16] + 64u * s_run[wq];
            if (!spawn) { child.dz = 1.0; child.parent = 0xffffffffu; }
            store_ray(P.next_rays, base + (unsigned)lane, child);
        }
    }
    if (!act) return;
    const uint32_t rgb = (uint32_t)cr | ((uint32_t)cg << 8) | ((uint32_t)cb << 16);
    if (P.colour) {
        P.colour[pix] = rgb;
    } else if (P.rgba) {
        reinterpret_cast<uint32_t *>(P.rgba)[P.out_off + pix] = rgb | 0xff000000u;     // 256 B per wave, coalesced
    } else {
        uint32_t *a = P.accum + 3 * (P.out_off + pix);
        a[0] += (uint32_t)cr; a[1] += (uint32_t)cg; a[2] += (uint32_t)cb;
    }
}

// ------------------------------------------------------------------------------------------
// Reflected rays (bounce levels >= 1): nearest hit of rays with arbitrary origins.
//   replaces `intersect world reflected` inside trace (n-1) (Main.hs:543, 552-554).

typedef const rt::GenRec RT_CONST *CGenPtr;
__device__ __forceinline__ CGenPtr as_const(const GenRec *p) { return (CGenPtr)(const void *)p; }

// Record test of the bounce levels: z >= negm  <=>  the ray may hit the leaf: q_k*a = e_k*a - b*f_k and t*a = -b must all
// carry the sign of a (rt_scene.cpp: make_gen_rec); filter_gen_staged below evaluates it on a record staged in LDS.

// Bound of one run of 64 reflected rays (one per lane): the origins lie within ro of c, the directions within the cone
// (a, cos rho) — the same in every lane.  Reflected rays of neighbouring pixels off the same mirror leave from
// neighbouring points in similar directions, so most blocks of leaves (32 consecutive records of a space-filling
// order, with their bounding sphere) can be skipped for the whole run: a point p of the sphere (C, R) can only be met
// by a ray of the run if the angle between C - c and a is at most rho + asin((R + ro) / |C - c|).
struct RunBound { float cx, cy, cz, ro, ax, ay, az, cr, sr; int mode; };     // mode 0: no rays, 1: test blocks, 2: scan every block

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ RunBound run_bound(bool act, float ox, float oy, float oz, float dx, float dy, float dz, bool cull) {
    RunBound b = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long m = __builtin_amdgcn_ballot_w64(act);
    if (!m) return b;
    b.mode = 2;
    if (!cull) return b;
    const float cnt = (float)__popcll(m);
    const float dd = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
    const float rn = __builtin_amdgcn_rsqf(dd);
    const float ux = dx * rn, uy = dy * rn, uz = dz * rn;
    bool bad = act && !(dd > 1e-30f && dd < 1e30f && __builtin_fabsf(ox) < 1e18f && __builtin_fabsf(oy) < 1e18f && __builtin_fabsf(oz) < 1e18f);   // NaN, inf, zero direction
    if (__builtin_amdgcn_ballot_w64(bad)) return b;
    b.cx = wave_sum(act ? ox : 0.f) / cnt; b.cy = wave_sum(act ? oy : 0.f) / cnt; b.cz = wave_sum(act ? oz : 0.f) / cnt;
    const float ex = ox - b.cx, ey = oy - b.cy, ez = oz - b.cz;
    const float ro = wave_max(act ? __builtin_sqrtf(__builtin_fmaf(ex, ex, __builtin_fmaf(ey, ey, ez * ez))) : 0.f);
    // (the rays were rounded from double to float: 2^-23 of their coordinates covers that twice over)
    b.ro = ro * 1.00001f + 1.2e-7f * (__builtin_fabsf(b.cx) + __builtin_fabsf(b.cy) + __builtin_fabsf(b.cz) + ro) + 1e-30f;
    const float sx = wave_sum(act ? ux : 0.f), sy = wave_sum(act ? uy : 0.f), sz = wave_sum(act ? uz : 0.f);
    const float sn = __builtin_fmaf(sx, sx, __builtin_fmaf(sy, sy, sz * sz));
    if (!(sn > 1e-6f * cnt * cnt)) return b;                     // directions all over the sphere
    const float sr = __builtin_amdgcn_rsqf(sn);
    b.ax = sx * sr; b.ay = sy * sr; b.az = sz * sr;
    const float cmin = -wave_max(act ? -__builtin_fmaf(b.ax, ux, __builtin_fmaf(b.ay, uy, b.az * uz)) : -1.f);
    b.cr = cmin - 1e-5f;
    if (!(b.cr > 0.05f)) return b;                               // wider than 87 degrees: not worth a test
    b.sr = __builtin_sqrtf(__builtin_fmaxf(0.f, (1.f - b.cr) * (1.f + b.cr)));
    b.mode = 1;
    return b;
}
__device__ __forceinline__ bool run_meets(const RunBound &b, float Cx, float Cy, float Cz, float R) {
    if (b.mode != 1) return b.mode == 2;
    const float vx = Cx - b.cx, vy = Cy - b.cy, vz = Cz - b.cz;
    const float d2 = __builtin_fmaf(vx, vx, __builtin_fmaf(vy, vy, vz * vz));
    const float rr = R + b.ro;
    if (!(d2 > rr * rr * 1.0001f)) return true;                  // the origins reach into the sphere (or a NaN / inf sphere)
    const float dist = __builtin_sqrtf(d2);
    const float sa = __builtin_fminf(1.f, rr / dist * 1.00001f + 1e-6f);
    const float ca = __builtin_sqrtf(__builtin_fmaxf(0.f, (1.f - sa) * (1.f + sa)));
    const float c = __builtin_fmaf(b.cr, ca, -(b.sr * sa)) - 1e-5f;              // cos(rho + alpha), low side
    return __builtin_fmaf(vx, b.ax, __builtin_fmaf(vy, b.ay, vz * b.az)) >= dist * c;
}

// The same test on a record staged in LDS (WaveQueues::rec: c[k] of row r at rec[32 k + r]); R = rec + row.
template <int KIND>
__device__ __forceinline__ float filter_gen_staged(const float *R, float ox, float oy, float oz, float dx, float dy, float dz, float &a, float &b) {
    const float n0 = R[0 * 32], n1 = R[1 * 32], n2 = R[2 * 32];
    const float g0 = R[4 * 32], g1 = R[5 * 32], g2 = R[6 * 32];
    const float k0 = R[8 * 32], k1 = R[9 * 32], k2 = R[10 * 32];
    a = __builtin_fmaf(n0, dx, __builtin_fmaf(n1, dy, n2 * dz));
    b = __builtin_fmaf(n0, ox, __builtin_fmaf(n1, oy, __builtin_fmaf(n2, oz, R[3 * 32])));
    const float e1 = __builtin_fmaf(g0, ox, __builtin_fmaf(g1, oy, __builtin_fmaf(g2, oz, R[7 * 32])));
    const float f1 = __builtin_fmaf(g0, dx, __builtin_fmaf(g1, dy, g2 * dz));
    const float e2 = __builtin_fmaf(k0, ox, __builtin_fmaf(k1, oy, __builtin_fmaf(k2, oz, R[11 * 32])));
    const float f2_ = __builtin_fmaf(k0, dx, __builtin_fmaf(k1, dy, k2 * dz));
    const float u = __builtin_fmaf(e1, a, -(b * f1));
    const float v = __builtin_fmaf(e2, a, -(b * f2_));
    if (KIND == LEAF_TRI) {
        const float w = a - u - v, t = -b;
        const float lo = __builtin_fminf(__builtin_fminf(__builtin_fminf(u, v), w), t);
        const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(u, v), w), t);
        return __builtin_fmaxf(lo, -hi);
    }
    const float q = __builtin_fabsf(a) - __builtin_fmaxf(__builtin_fabsf(u), __builtin_fabsf(v));
    return __builtin_fminf(q, -(b * a));
}

// Kernel of the bounce levels.  One wave = one run of 64 reflected rays (what one wave of k_final left: one patch of one
// surface).  Lane (c = lane & 31, h = lane >> 5) looks after ray column c of the run's two ray blocks (rays c and 32 + c).
//   1. the run's bound against the bounding sphere of every block of 32 leaves (a lane per block);
//   2. for the blocks that are left, stage 0 on the matrix cores: the line of every ray against the bounding sphere of
//      every leaf of the block, one linear form per pair (rt_scene.cpp: sphere_row), 32 leaves x 32 rays per MFMA; the
//      ray's vector depends on the block (the point of the line nearest the block's centre), so the B operand is rebuilt
//      per block: ~25 vector instructions per ray block, against the 20 per LEAF of the scalar-record loop this replaces;
//   3. pairs that pass go through the wave's LDS queues like the candidates of the other two matrix-core kernels: the
//      FP32 record test of the exact shape on the block's records staged in LDS, then the exact FP64 evaluation, both 64
//      pairs at a time on all lanes; the nearest hit of every ray lives in LDS as (bits of t, leaf).
// Ray flags (LDS): 0 = no ray (padding of k_final, end of the list), 1 = a ray, 2 = a ray whose direction or origin is
// not a usable number (NaN, inf, zero): it skips stage 0 and meets every record, as it did before there was a stage 0.
template <class EX>
__device__ __forceinline__ void q2_push(WaveQueues &Q, bool cand, uint32_t entry, EX &&exact) {
    if (!__builtin_amdgcn_ballot_w64(cand)) return;                // wave-uniform
    q_push(Q.q2, Q.n2, cand, entry);
    if (Q.n2 >= 64u) { Q.n2 -= 64u; exact(Q.q2[Q.n2 + (threadIdx.x & 63u)]); }
}

struct GenRays {                 // the two rays of a lane's column, FP32: origin, unit direction; flag as above
    float o[2][3], u[2][3];
    int flag[2];
    f16x8 B1[2];                 // step 1 of the ray's vector for this lane's K-half: hi / mid parts of d_i d_j x 2^10 (the same for every frame)
    _Float16 d5h[2], d5m[2];     // hi / mid of dy dz x 2^10 (they sit in step 0)
};

// Step 0 of the vector of ray block b's rays in the frame (centre C, radius 1024 / invs), for this lane's K-half: the parts
// that depend on the frame (rt_scene.cpp: sphere_row has the slot layout).  Everything is scaled by 2^10.
__device__ __forceinline__ f16x8 gen_ray_fragment(const GenRays &G, int b, int h, float Cx, float Cy, float Cz, float invs) {
    const float px = G.o[b][0] - Cx, py = G.o[b][1] - Cy, pz = G.o[b][2] - Cz;
    const float sd = __builtin_fmaf(px, G.u[b][0], __builtin_fmaf(py, G.u[b][1], pz * G.u[b][2]));
    // q = the point of the ray's line nearest the centre, in radii; far away is far enough at 4
    const float qx = __builtin_amdgcn_fmed3f(__builtin_fmaf(-sd, G.u[b][0], px) * invs, -4096.f, 4096.f);
    const float qy = __builtin_amdgcn_fmed3f(__builtin_fmaf(-sd, G.u[b][1], py) * invs, -4096.f, 4096.f);
    const float qz = __builtin_amdgcn_fmed3f(__builtin_fmaf(-sd, G.u[b][2], pz) * invs, -4096.f, 4096.f);
    const float qq = __builtin_fmaf(qx, qx, __builtin_fmaf(qy, qy, qz * qz)) * (1.f / 1024.f);
    const float v1 = h ? qy : qq, v2 = h ? qz : qx;
    const _Float16 v1h = (_Float16)v1, v2h = (_Float16)v2;
    const _Float16 v1m = (_Float16)(v1 - (float)v1h), v2m = (_Float16)(v2 - (float)v2h);
    const _Float16 one = (_Float16)1024.f, z0 = (_Float16)0.f;
    // K-half 0: {1, 1, qq_h, qq_m, qx_h, qx_m, qx_h, D5_m}   K-half 1: {qy_h, qy_m, qy_h, qz_h, qz_m, qz_h, D5_h, 0}
    f16x8 B = {h ? v1h : one, h ? v1m : one, v1h, h ? v2h : v1m, h ? v2m : v2h, h ? v2h : v2m, h ? G.d5h[b] : v2h, h ? z0 : G.d5m[b]};
    if (G.flag[b] != 1) B = f16x8{z0, z0, z0, z0, z0, z0, z0, z0};
    return B;
}
__device__ __forceinline__ float rfl(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ f32x16 mfma2(const i32x4 a0, const f16x8 b0, const i32x4 a1, const f16x8 b1) {       // K = 32: two steps
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), b1, mfma0(a0, b0), 0, 0, 0);
}

template <int KIND, class EX>
__device__ __forceinline__ void general_blocks(const RenderParams &P, const GenRec *recs, const uint16_t *rows, const float *spheres,
                                               const float *sspheres, int n, const RunBound &rb, const GenRays &G, const double *s_ray, const float *s_rayf,
                                               const unsigned long long *s_bt, bool depth_skip, bool anybad,
                                               uint16_t *list, WaveQueues &Q, unsigned &nblocks, EX &&exact) {
    const int lane = (int)(threadIdx.x & 63), c = lane & 31, h = lane >> 5, nblk = (n + 31) / 32;
    const i32x4 *rows4 = reinterpret_cast<const i32x4 *>(rows);
    const float4 *sph4 = reinterpret_cast<const float4 *>(spheres), *ssph4 = reinterpret_cast<const float4 *>(sspheres);
    auto stage1 = [&](uint32_t entry, uint32_t &entry2) -> bool {
        const int e = (int)(entry & 63u), row = (int)((entry >> 8) & 31u);
        const float *R = Q.rec + row;
        // FP32 copy of the ray: origin and UNIT direction.  The record's margin is for a direction of unit length; a
        // reflection off a normal that is not (an `fnorm` may return any vector, Main.hs:550, 956-965) is as long as it
        // likes, and the signs the test reads do not depend on the length.
        const float ox = s_rayf[0 * 64 + e], oy = s_rayf[1 * 64 + e], oz = s_rayf[2 * 64 + e];
        const float dx = s_rayf[3 * 64 + e], dy = s_rayf[4 * 64 + e], dz = s_rayf[5 * 64 + e];
        const float rn = s_rayf[6 * 64 + e];                       // 1 / |direction|
        float a, b;
        const float z = filter_gen_staged<KIND>(R, ox, oy, oz, dx, dy, dz, a, b);
        entry2 = ((uint32_t)__float_as_int(R[15 * 32]) << 8) | ((uint32_t)KIND << 6) | (uint32_t)e;
#ifndef RT_NO_DEPTH_SKIP
        // Behind the nearest hit this ray has so far?  The plane is met at t = -b / a (in units of the ray's own direction:
        // |b| rn / |a| with the unit direction); c[13], c[14] bound the FP32 error of a and b (make_gen_rec), 2^-18 covers the
        // roundings of this line and of rn, and the last term covers rayDistance's own error (Main.hs:122-127 divides a
        // difference of z by the direction's z: 16 ulp (|oz| / |dz| + t), which is why a ray with a flat z is left alone).
        // A pair that is skipped would have lost strictly, so ties still go to the left-most leaf.
        if (depth_skip && __builtin_fabsf(dz) >= 0x1p-10f) {
            const float best = (float)__longlong_as_double((long long)s_bt[e]);        // no hit yet: NaN, skips nothing
            const float lhs = (__builtin_fabsf(b) - R[14 * 32]) * rn * (1.f - 0x1p-18f);
            const float rhs = __builtin_fmaf(0x1p-36f * __builtin_fabsf(oz), rn, best) * (__builtin_fabsf(a) + R[13 * 32]);
            if (lhs > rhs) return false;
        }
#endif
        return z >= R[12 * 32];
    };
    // Works off the blocks list[0..cnt-1]; list[cnt] names the padding block (the loads run one block ahead).
    auto scan_list = [&](int cnt) {
        int g = __builtin_amdgcn_readfirstlane((int)list[0]);
        i32x4 An0 = rows4[(size_t)g * 128 + lane], An1 = rows4[(size_t)g * 128 + 64 + lane];
        float4 SSn = ssph4[g >> 5];
        int frame = -1;
        f16x8 B0[2] = {};
        for (int k = 0; k < cnt; ++k) {                             // wave-uniform
            const i32x4 A0 = An0, A1 = An1;
            const float4 SS = SSn;
            const int gc = g;
            g = __builtin_amdgcn_readfirstlane((int)list[k + 1]);
            An0 = rows4[(size_t)g * 128 + lane]; An1 = rows4[(size_t)g * 128 + 64 + lane];
            SSn = ssph4[g >> 5];
            if ((gc >> 5) != frame) {                               // a new super-block: the rays' vectors in its frame
                frame = gc >> 5;
                RT_TICK(tb)
                const float Rs = rfl(SS.w);
                const float invs = Rs < 3.0e38f ? 1024.f * __builtin_amdgcn_rcpf(Rs) : 0.f;    // no sphere: its rows are constants
                const float Cx = rfl(SS.x), Cy = rfl(SS.y), Cz = rfl(SS.z);
                B0[0] = gen_ray_fragment(G, 0, h, Cx, Cy, Cz, invs);
                B0[1] = gen_ray_fragment(G, 1, h, Cx, Cy, Cz, invs);
                RT_TOCK(Q.t_bits, tb)
            }
            int mx0 = step_max(mfma2(A0, B0[0], A1, G.B1[0]));
            int mx1 = step_max(mfma2(A0, B0[1], A1, G.B1[1]));
            if (G.flag[0] == 2) mx0 = 1;
            if (G.flag[1] == 2) mx1 = 1;
            const unsigned mb = (__builtin_amdgcn_ballot_w64(mx0 > 0) != 0ull ? 1u : 0u) | (__builtin_amdgcn_ballot_w64(mx1 > 0) != 0ull ? 2u : 0u);
            if (mb) {                                               // wave-uniform
#ifdef RT_GEN_DEBUG
                nblocks += 0x10000u;                                // (measurement build: blocks with a candidate, in the high half)
#endif
                RT_TICK(tl)
                // the records of this block's 32 leaves: 2 KiB, 32 bytes per lane (lane l: half l & 1 of record l >> 1)
                const float4 *rsrc = reinterpret_cast<const float4 *>(recs + (size_t)gc * 32) + 2 * lane;
                const float4 r0 = rsrc[0], r1 = rsrc[1];
                float *dst = Q.rec + (lane & 1) * 8 * 32 + (lane >> 1);
                dst[0 * 32] = r0.x; dst[1 * 32] = r0.y; dst[2 * 32] = r0.z; dst[3 * 32] = r0.w;
                dst[4 * 32] = r1.x; dst[5 * 32] = r1.y; dst[6 * 32] = r1.z; dst[7 * 32] = r1.w;
                __builtin_amdgcn_wave_barrier();
                const uint32_t lane_part = ((uint32_t)(4 * h) << 8) | (uint32_t)c;
#pragma unroll 1
                for (int b = 0; b < 2; ++b) {
                    if (!((mb >> b) & 1u)) continue;
                    const f32x16 pa = mfma2(A0, b ? B0[1] : B0[0], A1, b ? G.B1[1] : G.B1[0]);   // issued again (bit-identical): the 16 results were not kept
                    const bool force = anybad && (b ? G.flag[1] : G.flag[0]) == 2;
                    const uint32_t eb = lane_part + (uint32_t)(b * 32);
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const bool p = __float_as_int(pa[q]) > 0 || force;
                        const unsigned long long pm = __builtin_amdgcn_ballot_w64(p);
                        if (pm) {                                   // wave-uniform
                            if (p) Q.q1[Q.n1 + __builtin_amdgcn_mbcnt_hi((unsigned)(pm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pm, 0u))] =
                                (uint16_t)(eb + (uint32_t)(((q & 3) + 8 * (q >> 2)) << 8));
                            Q.n1 += (unsigned)__popcll(pm);
                        }
                    }
                    q_drain(Q, 63u, false, stage1, exact);
                }
                q_drain(Q, 0u, false, stage1, exact);               // q1 is empty before `rec` is overwritten
                __builtin_amdgcn_wave_barrier();
                RT_TOCK(Q.t_loop, tl)
            }
        }
    };
    // The run's bound against the bounding sphere of every block, a lane per block, 256 blocks per trip to the L2; the
    // blocks it reaches go to a list in LDS.
    int cnt = 0;                                                    // wave-uniform
    for (int g0 = 0; g0 < nblk; g0 += 256) {
        float4 S[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = g0 + 64 * u + lane;
            S[u] = sph4[g < nblk ? g : nblk];
        }
        unsigned long long mk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = g0 + 64 * u + lane;
            mk[u] = __builtin_amdgcn_ballot_w64(g < nblk && run_meets(rb, S[u].x, S[u].y, S[u].z, S[u].w));
        }
#pragma unroll 1
        for (int u = 0; u < 4 && g0 + 64 * u < nblk; ++u) {         // (not unrolled: one copy of the scan)
            const unsigned long long m = u == 0 ? mk[0] : (u == 1 ? mk[1] : (u == 2 ? mk[2] : mk[3]));
            const int g = g0 + 64 * u + lane;
            if ((m >> lane) & 1ull) list[cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (uint16_t)g;
            cnt += (int)__popcll(m);
            if (cnt > RT_LIST_CAP - 64 || g0 + 64 * u + 64 >= nblk) {
                nblocks += (unsigned)cnt;
                if (cnt > 0) {
                    if (lane == 0) list[cnt] = (uint16_t)nblk;
                    __builtin_amdgcn_wave_barrier();
                    scan_list(cnt);
                    __builtin_amdgcn_wave_barrier();
                }
                cnt = 0;
            }
        }
    }
}

#ifndef RT_WAVES_EU_GEN
#define RT_WAVES_EU_GEN 3      // (168 registers, nothing spilled; the 12.5 KiB of LDS per wave allow no more than 3.25 anyway)
#endif
__global__ void __attribute__((amdgpu_waves_per_eu(RT_WAVES_EU_GEN, 8))) __launch_bounds__(64) k_general(const RenderParams P) {
    RT_WAVE_T0
    __shared__ unsigned long long s_bt[64];
    __shared__ int s_bid[64];
    __shared__ int s_flag[64];
    __shared__ uint16_t s_list[RT_LIST_CAP + 8];
    __shared__ uint16_t s_q1[RT_Q1_CAP];
    __shared__ uint32_t s_q2[RT_Q2_CAP];
    __shared__ double s_ray[9 * 64];
    __shared__ float s_rayf[7 * 64];
    __shared__ float s_rec[16 * 32];
    const unsigned count = num_slots(P);
    const unsigned wave = blockIdx.x;
    if (wave * 64u >= count) return;
    const int lane = (int)(threadIdx.x & 63), c = lane & 31;
    const unsigned slot = wave * 64u + (unsigned)lane;
    const bool inlist = slot < count;
    const bool act = inlist && ray_parent(P.rays, slot) != 0xffffffffu;  // (not a padding entry of k_final)
    const RayD ray = slot_ray(P, inlist ? (int)slot : 0);
    {
        const double r9[9] = {ray.px, ray.py, ray.pz, ray.dx, ray.dy, ray.dz, ray.rx, ray.ry, ray.rz};
#pragma unroll
        for (int k = 0; k < 9; ++k) s_ray[k * 64 + lane] = r9[k];
    }
    s_bt[lane] = ~0ull; s_bid[lane] = -1;
    const float ox = (float)ray.px, oy = (float)ray.py, oz = (float)ray.pz;
    const float dx = (float)ray.dx, dy = (float)ray.dy, dz = (float)ray.dz;
    const RunBound rb = run_bound(act, ox, oy, oz, dx, dy, dz, P.cull != 0);
    bool anybad;
    {
        const float dd = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
        const bool bad = act && !(dd > 1e-30f && dd < 1e30f && __builtin_fabsf(ox) < 1e18f && __builtin_fabsf(oy) < 1e18f && __builtin_fabsf(oz) < 1e18f);
        s_flag[lane] = act ? (bad ? 2 : 1) : 0;
        anybad = __builtin_amdgcn_ballot_w64(bad) != 0ull;
    }
    __builtin_amdgcn_wave_barrier();
    GenRays G;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int e = 32 * b + c;
        G.flag[b] = s_flag[e];
        const float ex = (float)s_ray[3 * 64 + e], ey = (float)s_ray[4 * 64 + e], ez = (float)s_ray[5 * 64 + e];
        const float rn = G.flag[b] == 1 ? __builtin_amdgcn_rsqf(__builtin_fmaf(ex, ex, __builtin_fmaf(ey, ey, ez * ez))) : 0.f;
        G.u[b][0] = ex * rn; G.u[b][1] = ey * rn; G.u[b][2] = ez * rn;
#pragma unroll
        for (int k = 0; k < 3; ++k) G.o[b][k] = G.flag[b] == 1 ? (float)s_ray[k * 64 + e] : 0.f;
        // d_i d_j x 2^10, split: this lane's K-half of step 1, and dy dz for step 0
        const float s10 = 1024.f;
        const float D[6] = {G.u[b][0] * G.u[b][0] * s10, G.u[b][1] * G.u[b][1] * s10, G.u[b][2] * G.u[b][2] * s10,
                            G.u[b][0] * G.u[b][1] * s10, G.u[b][0] * G.u[b][2] * s10, G.u[b][1] * G.u[b][2] * s10};
        _Float16 Dh[6], Dm[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) { Dh[t] = (_Float16)D[t]; Dm[t] = (_Float16)(D[t] - (float)Dh[t]); }
        // K-half 0: {D0_h, D0_m, D0_h, D1_h, D1_m, D1_h, D2_h, D2_m}   K-half 1: {D2_h, D3_h, D3_m, D3_h, D4_h, D4_m, D4_h, D5_h}
        G.B1[b] = (lane >> 5) ? f16x8{Dh[2], Dh[3], Dm[3], Dh[3], Dh[4], Dm[4], Dh[4], Dh[5]}
                              : f16x8{Dh[0], Dm[0], Dh[0], Dh[1], Dm[1], Dh[1], Dh[2], Dm[2]};
        G.d5h[b] = Dh[5]; G.d5m[b] = Dm[5];
    }
    {       // FP32 copies for stage 1: origin, unit direction (as it is for a ray that is no usable number), 1 / |direction|
        const bool usable = s_flag[lane] == 1;
        const float rn = usable ? __builtin_amdgcn_rsqf(__builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz))) : 1.f;
        s_rayf[0 * 64 + lane] = ox; s_rayf[1 * 64 + lane] = oy; s_rayf[2 * 64 + lane] = oz;
        s_rayf[3 * 64 + lane] = dx * rn; s_rayf[4 * 64 + lane] = dy * rn; s_rayf[5 * 64 + lane] = dz * rn;
        s_rayf[6 * 64 + lane] = rn;
        __builtin_amdgcn_wave_barrier();
    }
    const bool depth_skip = !RT_TALK(P);        // (Talk.hs:97-128 picks the nearest by squared distance, not by rayDistance)
    RT_WAVE_PHASE(0)
    RT_WAVE_PHASE(1)
    WaveQueues Q = {s_q1, s_q2, 0u, 0u, 0u, s_rec};
    unsigned nblocks = 0, nexact = 0;
    // exact: intersect leaf reflected-ray (Main.hs:543 inside trace (n-1)), for the entry's ray
    auto exact = [&](uint32_t entry) {
        const int e = (int)(entry & 63u), kind = (int)((entry >> 6) & 3u), gid = (int)(entry >> 8);
        const LeafD &L = P.leaves[gid];
        RayD r;
        r.px = s_ray[0 * 64 + e]; r.py = s_ray[1 * 64 + e]; r.pz = s_ray[2 * 64 + e];
        r.dx = s_ray[3 * 64 + e]; r.dy = s_ray[4 * 64 + e]; r.dz = s_ray[5 * 64 + e];
        r.rx = s_ray[6 * 64 + e]; r.ry = s_ray[7 * 64 + e]; r.rz = s_ray[8 * 64 + e];
        double ix, iy, iz, tt = 0;
        ++nexact;
        const bool hit = (kind == LEAF_TRI ? leaf_isect<LEAF_TRI>(L, r, ix, iy, iz, tt, RT_TALK(P))
                          : (kind == LEAF_RECT ? leaf_isect<LEAF_RECT>(L, r, ix, iy, iz, tt, RT_TALK(P)) : leaf_isect<LEAF_SPHERE>(L, r, ix, iy, iz, tt, RT_TALK(P))))
                         && guards_ok(P.guards, L.guard, r);
        // nearest wins; equal t -> the left-most leaf of the tree (Main.hs:213-214); the same four steps as in k_primary_mfma
        const unsigned long long tb = (unsigned long long)__double_as_longlong(tt + 0.0);
        unsigned long long old = 0;
        if (hit) { old = s_bt[e]; atomicMin(&s_bt[e], tb); }
        __builtin_amdgcn_wave_barrier();
        const bool win = hit && s_bt[e] == tb;
        if (win && tb < old) s_bid[e] = 0x7fffffff;
        __builtin_amdgcn_wave_barrier();
        if (win) atomicMin(&s_bid[e], gid);
    };
    general_blocks<LEAF_TRI>(P, P.gen_tri, P.gen_tri_rows, P.gen_tri_sph, P.gen_tri_ssph, P.ntri, rb, G, s_ray, s_rayf, s_bt, depth_skip, anybad, s_list, Q, nblocks, exact);
    general_blocks<LEAF_RECT>(P, P.gen_rect, P.gen_rect_rows, P.gen_rect_sph, P.gen_rect_ssph, P.nrect, rb, G, s_ray, s_rayf, s_bt, depth_skip, anybad, s_list, Q, nblocks, exact);
    if (P.nsph > 0) {           // spheres (few per scene): delta/4 = (d.o)^2 - (o.o - r^2) with o = origin - centre, Main.hs:396-400
        CGenPtr recs = as_const(P.gen_sph);
        CIntPtr gids = as_const(P.sph_gid);
        for (int i = 0; i < P.nsph; ++i) {
            CGenPtr R = recs + i;
            const float sx = ox - R->c[0], sy = oy - R->c[1], sz = oz - R->c[2];
            const float b = __builtin_fmaf(dx, sx, __builtin_fmaf(dy, sy, dz * sz));
            const float cc = __builtin_fmaf(sx, sx, __builtin_fmaf(sy, sy, __builtin_fmaf(sz, sz, -R->c[3])));
            const float z = __builtin_fmaf(b, b, -cc);
            q2_push(Q, act && z >= R->c[12], ((uint32_t)gids[i] << 8) | ((uint32_t)LEAF_SPHERE << 6) | (uint32_t)lane, exact);
        }
    }
    q_drain(Q, 0u, true, [](uint32_t, uint32_t &) { return false; }, exact);       // the last, partial batch
    __builtin_amdgcn_wave_barrier();
    if (inlist) P.hit_id[P.out_off + slot] = s_bid[lane];        // padding entries: -1 (a miss)
    for (int o2 = 32; o2 > 0; o2 >>= 1) nexact += __shfl_down(nexact, o2);
    if (lane == 0 && nexact) stat_add(P, 1, (unsigned long long)nexact);
    const unsigned nrays = (unsigned)__popcll(__builtin_amdgcn_ballot_w64(act));
    if (lane == 0) stat_add(P, 5, (unsigned long long)nrays);     // reflected rays traced
#ifdef RT_GEN_DEBUG
    nblocks >>= 16;
#endif
    if (lane == 0) stat_add(P, 8, (unsigned long long)nblocks);   // (run of 64 rays, block of 32 leaves) pairs looked at
    if (lane == 0 && Q.nrec) stat_add(P, 4, (unsigned long long)Q.nrec);
#ifdef RT_WAVE_TIMES
    wave_ph[2] = (Q.t_s1 & 0xffff) | ((Q.t_ex & 0xffff) << 16) | ((Q.t_bits & 0xffff) << 32) | ((Q.t_loop & 0xffff) << 48);
    wave_record(P, 2u, wave, wave_t0, nblocks, Q.nrec, nexact, wave_ph);
#endif
}

// colour of a level folded into the level below: addPixelRGB8 direct (scalePixelRGB8 reflectivity sub), Main.hs:545, 554
__global__ void __launch_bounds__(256) k_combine(const RayRec *__restrict__ rays, const uint32_t *__restrict__ col,
                                                 const unsigned *__restrict__ count, unsigned cap, uint32_t *__restrict__ parent_col) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    const unsigned n = *count < cap ? *count : cap;
    if (i >= n) return;
    RayRec R;
    R.parent = ray_parent(rays, i);
    if (R.parent == 0xffffffffu) return;        // padding entry
    R.rho = reinterpret_cast<const double *>(rays)[ray_index(i, 6)];
    const uint32_t c = col[i], pc = parent_col[R.parent];
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int sub = round8(R.rho * (double)((c >> (8 * k)) & 255u));
        const int sum = (int)((pc >> (8 * k)) & 255u) + sub;
        out |= (uint32_t)(sum > 255 ? 255 : sum) << (8 * k);
    }
    parent_col[R.parent] = out;
}

// level-0 colours -> RGBA8 (spp == 1) or the sample accumulator
__global__ void __launch_bounds__(256) k_store(const RenderParams P) {
    const int pix = (int)(blockIdx.x * 256 + threadIdx.x);
    if (pix >= P.qcap) return;
    const uint32_t rgb = P.colour[pix];
    if (P.rgba) reinterpret_cast<uint32_t *>(P.rgba)[P.out_off + pix] = rgb | 0xff000000u;
    else {
        uint32_t *a = P.accum + 3 * (P.out_off + pix);
        a[0] += rgb & 255u; a[1] += (rgb >> 8) & 255u; a[2] += (rgb >> 16) & 255u;
    }
}

// spp > 1: mean of the per-sample 8-bit colours, (sum + spp/2) / spp
__global__ void k_resolve(const uint32_t *__restrict__ accum, uint8_t *__restrict__ rgba, size_t npix, uint32_t spp) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const uint32_t r = (accum[3 * i] + spp / 2) / spp, g = (accum[3 * i + 1] + spp / 2) / spp, b = (accum[3 * i + 2] + spp / 2) / spp;
    reinterpret_cast<uint32_t *>(rgba)[i] = r | (g << 8) | (b << 16) | 0xff000000u;
}

// RT_NORMAL_CALLBACK (an arbitrary fnorm, Main.hs:956-965): lists the slots of this chunk / level whose nearest hit lies
// on such a leaf, with the arguments the reference passes to fnorm: (x, _, y) = scaleInv * (i - offset), Main.hs:961.
__global__ void __launch_bounds__(256) k_cb_gather(const RenderParams P) {
    const int pix = (int)(blockIdx.x * 256 + threadIdx.x);
    const bool act = (unsigned)pix < num_slots(P);
    const int id = act ? P.hit_id[P.out_off + pix] : -1;
    bool want = false;
    double fx = 0, fy = 0;
    int group = 0;
    if (id >= 0) {
        const LeafD L = P.leaves[id];
        if (L.nmode == RT_NORMAL_CALLBACK) {
            const RayD ray = slot_ray(P, pix);
            double ix = 0, iy = 0, iz = 0, t;
            if (L.kind == LEAF_RECT) leaf_isect<LEAF_RECT>(L, ray, ix, iy, iz, t, RT_TALK(P));
            else if (L.kind == LEAF_TRI) leaf_isect<LEAF_TRI>(L, ray, ix, iy, iz, t, RT_TALK(P));
            else leaf_isect<LEAF_SPHERE>(L, ray, ix, iy, iz, t, RT_TALK(P));
            const AuxD A = P.aux[L.aux];
            fx = A.a[0] * (ix - A.a[1]); fy = A.a[0] * (iz - A.a[3]);
            group = (int)A.a[4];
            want = true;
        }
    }
    const unsigned long long m = __ballot(want);
    if (!m) return;
    const int lane = (int)(threadIdx.x & 63), leader = __ffsll((long long)m) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(P.cb_count, (unsigned)__popcll(m));
    base = __shfl(base, leader);
    if (want) {
        const unsigned k = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        P.cb_slot[pix] = k;
        P.cb_xy[2 * (size_t)k] = fx; P.cb_xy[2 * (size_t)k + 1] = fy;
        P.cb_group[k] = group;
    }
}

} // namespace

// Scratch of rt_render_device_noalloc (for callers that cannot allocate: HIP graph capture, tight serving loops):
// every buffer a render needs beyond its inputs and outputs is carved out of this array instead of hipMalloc'ed.
// Each GPU the library runs on holds its own copy, at its own address, until the process ends.  On the host the name
// is only a stand-in for HIP_SYMBOL(): the device address comes from hipGetSymbolAddress (device_init), once per device.
constexpr size_t RT_STATIC_SCRATCH_BYTES = 64u << 20;
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];

namespace {

// ------------------------------------------------------------------------------------------
// Host side

// Library state per device and per thread.  Everything else lives in the (scene, device) pair: streams, events,
// uploaded records, work buffers — so distinct scenes can be rendered from different threads.
struct DeviceCtx {
    bool ready = false;
    unsigned char *scratch = nullptr;    // this device's copy of `scratch` (hipGetSymbolAddress, with the device current)
};
DeviceCtx g_ctx[RT_MAX_DEVICES];
std::mutex g_ctx_mutex;
thread_local int t_device = -1;          // the device this thread renders on (rt_init)

bool hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}
#define HIP_TRY(expr) do { if (!hip_ok((expr), #expr)) return RT_ERR_HIP; } while (0)

#ifdef RT_WAVE_TIMES
unsigned long long *g_wave_dbg = nullptr;
#endif

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Sets the library up on `device` and makes it the calling thread's device.
int device_init(int device) {
    int n = 0;
    if (!hip_ok(hipGetDeviceCount(&n), "hipGetDeviceCount") || n <= 0) {
        set_error("rt_init: no HIP device available (this library has no CPU fallback)");
        return RT_ERR_HIP;
    }
    if (device < 0 || device >= n || device >= RT_MAX_DEVICES) { set_error("rt_init: device index out of range"); return RT_ERR_ARG; }
    HIP_TRY(hipSetDevice(device));
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        DeviceCtx &C = g_ctx[device];
        if (!C.ready) {
            unsigned char *ptr = nullptr;
            if (!hip_ok(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(scratch)") || !ptr) return RT_ERR_HIP;
            C.scratch = ptr;
            C.ready = true;
        }
    }
    t_device = device;
    return RT_OK;
}
// Every entry point that touches the device starts here: the thread's device is made current again (the caller may
// have switched it, e.g. through another library).
int enter() {
    if (t_device < 0) return device_init(0);
    HIP_TRY(hipSetDevice(t_device));
    return RT_OK;
}

// The resident copy of a scene is ONE allocation (DeviceScene::arena) that the arrays are views of: a first upload is one
// hipMalloc and a run of copies on the scene's stream instead of ~25 allocations and blocking copies (a one-shot caller
// pays for it: examples/cold_start.cpp).  plan(): size of the views, 256-byte aligned; place(): view + copy.
struct UploadPlan {
    size_t total = 0;
    template <typename T> void add(const std::vector<T> &src) { total = ((total + 255) & ~(size_t)255) + sizeof(T) * (src.empty() ? 1 : src.size()); }
};
template <typename T>
int place(DeviceScene &D, size_t &off, DevBuf &dst, const std::vector<T> &src, unsigned char *staging) {
    off = (off + 255) & ~(size_t)255;
    const size_t bytes = sizeof(T) * (src.empty() ? 1 : src.size());
    dst.p = (char *)D.arena.p + off; dst.cap = bytes;
    if (!src.empty()) std::memcpy(staging + off, src.data(), sizeof(T) * src.size());
    off += bytes;
    return RT_OK;
}
// elements [first, first + count) of an array that is already resident
template <typename T>
int upload_slice(DevBuf &dst, const std::vector<T> &src, size_t first, size_t count) {
    if (count == 0) return RT_OK;
    if (!dst.p || dst.cap < sizeof(T) * (first + count) || src.size() < first + count) { set_error("upload_slice: array not resident"); return RT_ERR_ARG; }
    HIP_TRY(hipMemcpy((char *)dst.p + sizeof(T) * first, src.data() + first, sizeof(T) * count, hipMemcpyHostToDevice));
    return RT_OK;
}

template <typename T>
int ensure(T *&buf, size_t &cap, size_t need) {
    if (cap >= need && buf) return RT_OK;
    if (buf) { HIP_TRY(hipFree(buf)); buf = nullptr; cap = 0; }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&buf), sizeof(T) * (need ? need : 1)));
    cap = need;
    return RT_OK;
}

// Makes the scene resident on the thread's device for w x h renders (rt_scene_prepare_device).
// What the runtime sets up when streams are first made costs a one-shot caller milliseconds (a stream = a hardware queue:
// 8 ms for the first of a process, 4.6 ms for the two copy streams and their events; examples/cold_start.cpp): a helper
// thread makes them while the calling thread builds the host index and uploads.  The render stream is waited for in
// device_prepare; the copy tools by the first call that needs them (copy_tools), or by rt_scene_destroy.
struct Helper { std::thread th; std::mutex m; std::condition_variable cv; bool stream_ready = false; hipError_t err = hipSuccess; };
void helper_join(DeviceScene &D) {
    Helper *H = static_cast<Helper *>(D.helper);
    if (!H) return;
    if (H->th.joinable()) H->th.join();
    delete H;
    D.helper = nullptr;
}
int make_copy_tools(DeviceScene &D) {
    if (!D.copy_stream) { hipStream_t st; HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); D.copy_stream = st; }
    if (!D.copy_stream2) { hipStream_t st; HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); D.copy_stream2 = st; }
    for (void *&e : D.band_done) if (!e) { hipEvent_t ev; HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); e = ev; }
    for (void *&e : D.hits_done) if (!e) { hipEvent_t ev; HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); e = ev; }
    return RT_OK;
}

int device_prepare(rt_scene &s, DeviceScene &D, int device, int w, int h) {
    if (!D.stream && !D.helper && !host_ready(s, w, h)) {
        Helper *H = new Helper;
        D.helper = H;
        H->th = std::thread([&D, H, device] {
            hipStream_t st = nullptr;
            hipError_t e = hipSetDevice(device);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            {
                std::lock_guard<std::mutex> lock(H->m);
                if (e == hipSuccess) D.stream = st;
                H->err = e; H->stream_ready = true;
            }
            H->cv.notify_all();
            if (e == hipSuccess && make_copy_tools(D) != RT_OK) (void)hipGetLastError();     // (copy_tools() makes what is missing, and reports)
        });
    }
    D.device = device;              // (from here on rt_scene_destroy looks at this entry)
    const bool host_ok = prepare_host(s, w, h);
    if (Helper *H = static_cast<Helper *>(D.helper)) {
        std::unique_lock<std::mutex> lock(H->m);
        H->cv.wait(lock, [H] { return H->stream_ready; });
        if (H->err != hipSuccess) { const hipError_t e = H->err; lock.unlock(); helper_join(D); HIP_TRY(e); }
    }
    if (!host_ok) return RT_ERR_ARG;
    if (s.mf_groups + 2 > 65535) { set_error("rt_render: more than 2 million planar leaves are not supported (16-bit block indices)"); return RT_ERR_UNSUPPORTED; }
    if (!D.stream) { hipStream_t st; HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); D.stream = st; }
    if (D.version == s.version && D.eye_version == s.eye_version) return RT_OK;
    const double t0 = now_ms();
    int rc;
    // a render of the stale copy may still be in flight on some stream: the arrays are overwritten in place
    if (D.version != 0) HIP_TRY(hipDeviceSynchronize());
    const size_t nsph = s.sph_gid.size();
#ifdef RT_VALU_FILTER
    const size_t ntri = s.tri_gid.size(), nrect = s.rect_gid.size();
#endif
    const size_t gstride = (size_t)s.mf_groups + 2;
    if (D.version != s.version) {
        // every array of the scene, in one arena (grown with a quarter of slack so that a slightly bigger scene fits again)
        auto each = [&](auto &&f) -> int {
            int r;
            if ((r = f(D.leaves, s.leaves)) || (r = f(D.aux, s.aux)) || (r = f(D.guards, s.guards)) || (r = f(D.mats, s.mats)) ||
                (r = f(D.lights, s.lights)) || (r = f(D.tri_rec, s.tri_rec)) || (r = f(D.rect_rec, s.rect_rec)) ||
                (r = f(D.sph_rec, s.sph_rec)) || (r = f(D.sph_gid, s.sph_gid)) || (r = f(D.mf_rows, s.mf_rows)) ||
                (r = f(D.mf_gid, s.mf_gid)) || (r = f(D.mf_cone, s.mf_cone)) || (r = f(D.mf_rec, s.mf_rec)) ||
                (r = f(D.tri_gid, s.tri_gid)) || (r = f(D.rect_gid, s.rect_gid))) return r;
            if (s.reflective) {
                if ((r = f(D.gen_sph, s.gen_sph)) || (r = f(D.gen_tri, s.gen_tri)) || (r = f(D.gen_rect, s.gen_rect)) ||
                    (r = f(D.gen_tri_gid, s.gen_tri_gid)) || (r = f(D.gen_rect_gid, s.gen_rect_gid)) ||
                    (r = f(D.gen_tri_sph, s.gen_tri_sph)) || (r = f(D.gen_rect_sph, s.gen_rect_sph)) ||
                    (r = f(D.gen_tri_rows, s.gen_tri_rows)) || (r = f(D.gen_rect_rows, s.gen_rect_rows)) ||
                    (r = f(D.gen_tri_ssph, s.gen_tri_ssph)) || (r = f(D.gen_rect_ssph, s.gen_rect_ssph))) return r;
            }
            return RT_OK;
        };
        UploadPlan plan;
        (void)each([&](DevBuf &, const auto &src) { plan.add(src); return (int)RT_OK; });
        if (!D.arena.p || D.arena.cap < plan.total) {
            if (D.arena.p) { HIP_TRY(hipFree(D.arena.p)); D.arena.p = nullptr; D.arena.cap = 0; }
            const size_t cap = plan.total + plan.total / 4 + 4096;
            HIP_TRY(hipMalloc(&D.arena.p, cap));
            D.arena.cap = cap;
        }
        // the arrays are laid out in a host image of the arena and travel in ONE copy (25 copies out of pageable vectors
        // cost 8.7 ms for 10 MB: each one is staged and waited for on its own)
        size_t off = 0;
        unsigned char *image = nullptr;                   // page-locked: one DMA, no staging inside the runtime
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&image), plan.total, hipHostMallocDefault));
        rc = each([&](DevBuf &dst, const auto &src) { return place(D, off, dst, src, image); });
        if (rc == RT_OK && (!hip_ok(hipMemcpyAsync(D.arena.p, image, off, hipMemcpyHostToDevice, (hipStream_t)D.stream), "hipMemcpyAsync") ||
                            !hip_ok(hipStreamSynchronize((hipStream_t)D.stream), "hipStreamSynchronize"))) rc = RT_ERR_HIP;   // complete on return
        (void)hipHostFree(image);
        if (rc) return rc;
    } else {
        // only the camera bound grew: apex 0 (the eye) of every per-apex array
#ifdef RT_VALU_FILTER
        if ((rc = upload_slice(D.tri_rec, s.tri_rec, 0, ntri))) return rc;
        if ((rc = upload_slice(D.rect_rec, s.rect_rec, 0, nrect))) return rc;
#endif
        if ((rc = upload_slice(D.sph_rec, s.sph_rec, 0, nsph))) return rc;
        if ((rc = upload_slice(D.mf_rows, s.mf_rows, 0, gstride * 32 * 8))) return rc;
        if ((rc = upload_slice(D.mf_gid, s.mf_gid, 0, gstride * MF_LEAVES))) return rc;
        if ((rc = upload_slice(D.mf_cone, s.mf_cone, 0, gstride * 8))) return rc;
        if ((rc = upload_slice(D.mf_rec, s.mf_rec, 0, gstride * MF_LEAVES))) return rc;
    }
    D.version = s.version;
    D.eye_version = s.eye_version;
    D.stats.upload_ms = now_ms() - t0;
    return RT_OK;
}

// The streams and events of the copies to host memory (rt_render) and between devices (rt_render_multi).
int copy_tools(DeviceScene &D) {
    helper_join(D);                 // (it may be making them right now)
    return make_copy_tools(D);
}

// event pool: one event per kernel boundary, so that rt_get_stats can attribute time to each kernel
int next_event(DeviceScene &D, hipStream_t st, int kind) {
    if (D.ev_used == (int)D.evpool.size()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        D.evpool.push_back(e);
        D.evkind.push_back(0);
    }
    D.evkind[D.ev_used] = kind;
    HIP_TRY(hipEventRecord((hipEvent_t)D.evpool[D.ev_used], st));
    ++D.ev_used;
    return RT_OK;
}

constexpr size_t RT_CHUNK_PIXELS = 16u << 20;   // rows are processed in chunks of about 16 Mi pixels

int render_levels(const rt_scene *s, int depth) {
    return (s->reflective && depth > 1 && !s->talk) ? (depth < RT_MAX_LEVELS ? depth : RT_MAX_LEVELS) : 1;
}
bool levels_supported(const rt_scene *s, int depth) {
    if (s->reflective && !s->talk && depth > RT_MAX_LEVELS) {
        set_error("rt_render: trace depth > 64 with reflective materials is not supported"); return false;
    }
    return true;
}

// What one render needs beyond its inputs and outputs, in elements of each buffer.
struct WsNeeds {
    size_t hit_id = 0, accum = 0, hitrec = 0, queue = 0, qsparse = 0, tcount = 0, lit = 0, lvl = 0, cb = 0;
    int levels = 1;
};
// slots of a bounce level: k_final hands every wave of 8 x 8 pixels that reflects anything a run of 64 slots, so a level
// can hold the chunk's pixels padded to whole 32 x 8 tiles (and no more: deeper levels never grow)
size_t level_capacity(int w, size_t rows) {
    return (size_t)((w + 31) / 32) * 32 * ((rows + 7) / 8) * 8;
}
size_t setup_blocks_bound(int w, size_t crow, bool levels) {   // k_setup blocks of a chunk: 32 x 16 pixel tiles at level 0, runs of slots above
    const size_t slots = levels ? level_capacity(w, crow) : crow * (size_t)w;
    return std::max((size_t)((w + 31) / 32) * ((crow + RT_SETUP_THREADS / 32 - 1) / (RT_SETUP_THREADS / 32)),
                    (slots + RT_SETUP_THREADS - 1) / RT_SETUP_THREADS);
}
WsNeeds ws_needs(int w, int nrows, size_t crow, int npoint, int levels, int spp, bool own_hit, bool shade, bool callbacks) {
    WsNeeds n;
    const size_t npix = (size_t)w * (size_t)nrows, np = (size_t)(npoint > 0 ? std::min(npoint, 32) : 1);
    const size_t groups = (size_t)std::max(1, (npoint + 31) / 32);
    const size_t cpix = levels > 1 ? level_capacity(w, crow) : crow * (size_t)w;      // slots of a chunk (bounce levels: padded)
    n.levels = levels;
    if (own_hit || spp > 1) n.hit_id = npix;
    if (!shade) return n;
    if (spp > 1) n.accum = 3 * npix;
    n.hitrec = cpix * sizeof(HitRec);
    n.lit = cpix * groups;
    n.queue = cpix * np;
    const size_t ub = setup_blocks_bound(w, crow, levels > 1);
    n.qsparse = ub * RT_SETUP_THREADS * np;
    n.tcount = (ub + 1) * np;
    if (levels > 1) n.lvl = cpix;
    if (callbacks) n.cb = cpix;
    return n;
}
// Carves the buffers of `n` out of [base, base + cap); returns the bytes used, or 0 if they do not fit.  base == nullptr:
// sizing only (W is left empty).
size_t ws_carve(unsigned char *base, size_t cap, const WsNeeds &n, Workspace &W) {
    size_t used = 0;
    bool fits = true;
    auto take = [&](auto *&ptr, size_t count) {
        using T = std::remove_reference_t<decltype(*ptr)>;
        used = (used + 255) & ~(size_t)255;
        if (used + count * sizeof(T) > cap) { fits = false; return; }
        if (base) ptr = reinterpret_cast<T *>(base + used);
        used += count * sizeof(T);
    };
    W = Workspace{};
    // the queue counters sit right behind the statistics counters, at the start: a render clears both with one fill, and
    // the part of them that lives across renders (the lights' cost history) keeps its place whatever else is carved
    take(W.counters, (size_t)RT_STAT_SLOTS * 16 + ((size_t)RT_QCOUNT_WORDS + 1) / 2);
    if (fits && base) W.qcount = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(W.counters) + RT_COUNTER_BYTES);
    if (n.hit_id) take(W.hit_id, n.hit_id);
    if (n.accum) take(W.accum, n.accum);
    if (n.hitrec) take(W.hitrec, n.hitrec);
    if (n.lit) take(W.lit, n.lit);
    if (n.queue) take(W.queue, n.queue);
    if (n.qsparse) take(W.qsparse, n.qsparse);
    if (n.tcount) take(W.tcount, n.tcount);
    if (n.lvl) {
        take(W.lvl_col[0], n.lvl);
        for (int k = 1; k < n.levels && k < RT_MAX_LEVELS; ++k) { take(W.lvl_rays[k], n.lvl); take(W.lvl_hit[k], n.lvl); take(W.lvl_col[k], n.lvl); }
    }
    if (n.cb) { take(W.cb_slot, n.cb); take(W.cb_xy, 2 * n.cb); take(W.cb_group, n.cb); take(W.cb_normal, 3 * n.cb); }
    return fits ? used : 0;
}

// rt_render / rt_render_device: the work buffers of the (scene, device) pair are views of ONE allocation
// (DeviceScene::ws_arena), grown when a render needs more than it holds: a first render is one hipMalloc, not a dozen.
int ws_grow(DeviceScene &D, const WsNeeds &n) {
    Workspace sizing;
    const size_t total = ws_carve(nullptr, ~(size_t)0 >> 1, n, sizing);
    if (!D.ws_arena.p || D.ws_arena.cap < total) {
        if (D.ws_arena.p) { HIP_TRY(hipFree(D.ws_arena.p)); D.ws_arena.p = nullptr; D.ws_arena.cap = 0; }     // (hipFree waits for what is in flight)
        HIP_TRY(hipMalloc(&D.ws_arena.p, total));
        D.ws_arena.cap = total;
        // the counters that live across renders start at zero
        HIP_TRY(hipMemsetAsync((unsigned char *)D.ws_arena.p + RT_COUNTER_BYTES, 0, RT_QCOUNT_WORDS * sizeof(unsigned), (hipStream_t)D.stream));
        HIP_TRY(hipStreamSynchronize((hipStream_t)D.stream));
    }
    if (!ws_carve((unsigned char *)D.ws_arena.p, D.ws_arena.cap, n, D.ws)) { set_error("ws_grow: internal sizing error"); return RT_ERR_NOMEM; }
    return RT_OK;
}

// RT_NORMAL_CALLBACK: between the nearest-hit kernel and k_setup, the hit points that lie on callback leaves go to the
// host, the caller's fnorm is evaluated there (per callback group), and the normals come back.
int run_normal_callbacks(rt_scene &s, const RenderParams &P, hipStream_t st, unsigned pblocks) {
    HIP_TRY(hipMemsetAsync(P.cb_count, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL(k_cb_gather, dim3(pblocks), dim3(256), 0, st, P);
    unsigned count = 0;
    HIP_TRY(hipMemcpyAsync(&count, P.cb_count, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (count == 0) return RT_OK;
    std::vector<double> xy(2 * (size_t)count), nrm(3 * (size_t)count);
    std::vector<int32_t> grp(count);
    HIP_TRY(hipMemcpy(xy.data(), P.cb_xy, xy.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(grp.data(), P.cb_group, grp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    // the hit points sorted by callback group in one counting pass (a scene may register one group per interpolation cell:
    // a scan of all points per group was O(groups x points)), then one call per group that has any
    const size_t ngroups = s.normal_fn.size();
    std::vector<unsigned> start(ngroups + 1, 0u), idx(count);
    for (unsigned k = 0; k < count; ++k) if (grp[k] >= 0 && (size_t)grp[k] < ngroups) ++start[(size_t)grp[k] + 1];
    for (size_t g = 0; g < ngroups; ++g) start[g + 1] += start[g];
    {
        std::vector<unsigned> fill(start.begin(), start.end() - 1);
        for (unsigned k = 0; k < count; ++k) if (grp[k] >= 0 && (size_t)grp[k] < ngroups) idx[fill[(size_t)grp[k]]++] = k;
    }
    std::vector<double> gxy, gn;
    for (size_t g = 0; g < ngroups; ++g) {
        const size_t n = start[g + 1] - start[g];
        if (n == 0) continue;
        gxy.resize(2 * n); gn.assign(3 * n, 0.0);
        for (size_t j = 0; j < n; ++j) { const size_t k = idx[start[g] + j]; gxy[2 * j] = xy[2 * k]; gxy[2 * j + 1] = xy[2 * k + 1]; }
        s.normal_fn[g]((int)n, gxy.data(), gn.data(), s.normal_user[g]);
        for (size_t j = 0; j < n; ++j)
            for (int c = 0; c < 3; ++c) nrm[3 * (size_t)idx[start[g] + j] + c] = gn[3 * j + c];
    }
    HIP_TRY(hipMemcpy(const_cast<double *>(P.cb_normal), nrm.data(), nrm.size() * sizeof(double), hipMemcpyHostToDevice));
    return RT_OK;
}

// Enqueues the kernels of one row band on `st`.  `more` = this band continues a frame whose earlier bands were
// enqueued by the previous call(s): events, counters and ray counts then accumulate, so rt_get_stats describes the
// whole frame.  `W` holds buffers of at least ws_needs(...) elements for chunks of `crow` rows; `timed` records an
// event per kernel boundary (creates events on first use).
int render_rows(rt_scene *s, DeviceScene &D, const Workspace &W, size_t crow, int w, int h, int depth, int spp, uint64_t seed,
                int row0, int nrows, void *d_rgba, void *d_hit_id, hipStream_t st, bool more, bool timed, hipEvent_t hits_done = nullptr) {
    // hits_done: recorded on `st` as soon as the caller's hit-ID buffer is complete (after the nearest-hit kernel of
    // sample 0's last chunk: everything later only reads it), so that a copy of it can run beside the shading kernels
    int rc;
    const size_t npix = (size_t)w * (size_t)nrows;
    if (!more) { D.last_primary = 0; D.ev_used = 0; }
    D.last_primary += (uint64_t)npix * (uint64_t)spp;
    D.stats_pending = false;
    D.last_counters = W.counters;
    s->last_device = D.device;
    s->multi_stats = false;
    if (npix == 0) return RT_OK;
    (void)hipGetLastError();       // kernel launches report through the sticky error: start from a clean one
    // bounce levels: trace n reflects at most n - 1 times (trace 0 = black, Main.hs:542); only scenes with a
    // reflective material ever spawn a reflected ray (Main.hs:553)
    const int levels = render_levels(s, depth);
    int32_t *hit = (int32_t *)d_hit_id;
    if (!hit) hit = W.hit_id;
    if (spp > 1 && d_rgba) HIP_TRY(hipMemsetAsync(W.accum, 0, 3 * npix * sizeof(uint32_t), st));
    // (a render's first chunk finds its queue counters cleared by the same fill as the statistics counters)
    if (!more) HIP_TRY(hipMemsetAsync(W.counters, 0, RT_COUNTER_BYTES + RT_QCOUNT_CLEAR * sizeof(unsigned), st));
    bool queues_clear = !more;

#ifdef RT_VALU_FILTER
    // the vector-ALU kernels are compiled for Main.hs only: a Talk.hs scene always takes the matrix-core kernels
    const bool use_mfma = s->opt_filter == 0 || s->talk;
#else
    const bool use_mfma = true;
#endif
    RenderParams P{};
    P.leaves = (const LeafD *)D.leaves.p; P.aux = (const AuxD *)D.aux.p; P.guards = (const GuardD *)D.guards.p;
    P.mats = (const MaterialD *)D.mats.p; P.lights = (const LightD *)D.lights.p;
    P.tri_rec = (const FilterRec *)D.tri_rec.p; P.rect_rec = (const FilterRec *)D.rect_rec.p;
    P.tri_gid = (const int32_t *)D.tri_gid.p; P.rect_gid = (const int32_t *)D.rect_gid.p;
    P.counters = W.counters;
    P.hitrec = reinterpret_cast<HitRec *>(W.hitrec); P.queue = W.queue; P.qcount = W.qcount; P.lit = W.lit;
    P.qsparse = W.qsparse; P.tcount = W.tcount;
    P.cb_slot = W.cb_slot; P.cb_xy = W.cb_xy; P.cb_group = W.cb_group; P.cb_normal = W.cb_normal; P.cb_count = W.qcount + RT_QCOUNT_CB;
    P.gen_tri = (const GenRec *)D.gen_tri.p; P.gen_rect = (const GenRec *)D.gen_rect.p;
    P.gen_tri_gid = (const int32_t *)D.gen_tri_gid.p; P.gen_rect_gid = (const int32_t *)D.gen_rect_gid.p;
    P.gen_tri_sph = (const float *)D.gen_tri_sph.p; P.gen_rect_sph = (const float *)D.gen_rect_sph.p;
    P.gen_tri_rows = (const uint16_t *)D.gen_tri_rows.p; P.gen_rect_rows = (const uint16_t *)D.gen_rect_rows.p;
    P.gen_tri_ssph = (const float *)D.gen_tri_ssph.p; P.gen_rect_ssph = (const float *)D.gen_rect_ssph.p;
    P.sph_rec = (const FilterRec *)D.sph_rec.p; P.gen_sph = (const GenRec *)D.gen_sph.p; P.sph_gid = (const int32_t *)D.sph_gid.p;
    P.nsph = (int)s->sph_gid.size();
    P.mf_rows = (const uint16_t *)D.mf_rows.p; P.mf_gid = (const int32_t *)D.mf_gid.p; P.mf_rec = (const FilterRec *)D.mf_rec.p;
    P.mf_groups = s->mf_groups;
    P.mf_cone = (const float *)D.mf_cone.p; P.cull = s->opt_cull ? 1 : 0;
    P.nguard_cached = (s->opt_cull && !s->talk) ? s->guard_cached : 0;
    // point lights are worked off 32 at a time (one bit each in a word of `lit`); the reference's Light monoid has no limit
    std::vector<uint16_t> point_lights;
    for (size_t li = 0; li < s->lights.size(); ++li) if (s->lights[li].kind == 0) point_lights.push_back((uint16_t)li);
    const int ngroups = std::max(1, (int)((point_lights.size() + 31) / 32));
    P.cam = fixed_camera(w, h);
    P.eps = ray_eps();
    P.ntri = (int)s->tri_gid.size(); P.nrect = (int)s->rect_gid.size();
    P.nlights = (int)s->lights.size(); P.nleaves = (int)s->leaves.size(); P.npoint = std::min(32, s->npoint);
    P.w = w; P.h = h;
    P.talk = s->talk ? 1 : 0;
    P.spp = spp; P.seed_lo = (uint32_t)seed; P.seed_hi = (uint32_t)(seed >> 32);
    const bool callbacks = !s->normal_fn.empty();
#ifdef RT_WAVE_TIMES
    {
        static unsigned long long *dbg = nullptr;
        if (!dbg) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dbg), (8 + 8 * (1ull << 20)) * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(dbg, 0, 64, st));
        P.dbg = dbg;
        g_wave_dbg = dbg;
    }
#endif

    if (timed && (rc = next_event(D, st, 0))) return rc;
    for (int sample = 0; sample < spp; ++sample) {
        P.sample = sample;
        // sample 0 writes the caller's hit-ID buffer; later samples use the scratch one
        P.hit_id = (sample == 0) ? hit : W.hit_id;
        P.rgba = (spp == 1) ? (uint8_t *)d_rgba : nullptr;
        P.accum = (spp == 1) ? nullptr : W.accum;
        for (size_t r = 0; r < (size_t)nrows; r += crow) {
            const int rows = (int)((size_t)nrows - r < crow ? (size_t)nrows - r : crow);
            P.row0 = row0 + (int)r; P.nrows = rows;
            P.out_off = (long long)r * w;
            P.qcap = rows * w;
            P.lcap = (int)level_capacity(w, (size_t)rows);
            P.tiles_x = (w + RT_TILE_W - 1) / RT_TILE_W;
            P.tiles_y = (rows + RT_TILE_H - 1) / RT_TILE_H;
            const long long tiles = (long long)P.tiles_x * P.tiles_y;
            const unsigned tblocks = (unsigned)((tiles + RT_WAVES_PER_BLOCK - 1) / RT_WAVES_PER_BLOCK);
            const unsigned pblocks = (unsigned)(((size_t)P.qcap + 255) / 256);
            if (depth == 0) {      // trace 0 = black (Main.hs:542): every pixel misses
                HIP_TRY(hipMemsetAsync(P.hit_id + P.out_off, 0xff, (size_t)P.qcap * sizeof(int32_t), st));
            } else {
                if (use_mfma) hipLaunchKernelGGL(k_primary_mfma, dim3(tblocks), dim3(RT_WAVES_PER_BLOCK * 64), 0, st, P);
#ifdef RT_VALU_FILTER
                else if (spp == 1) hipLaunchKernelGGL(k_primary<true>, dim3(tblocks), dim3(RT_WAVES_PER_BLOCK * 64), 0, st, P);
                else hipLaunchKernelGGL(k_primary<false>, dim3(tblocks), dim3(RT_WAVES_PER_BLOCK * 64), 0, st, P);
#endif
            }
            if (timed && (rc = next_event(D, st, 1))) return rc;
            if (hits_done && sample == 0 && r + crow >= (size_t)nrows) HIP_TRY(hipEventRecord(hits_done, st));
            if (!d_rgba) continue;
            // grids: shadow waves of 256 jobs, one thread per slot; bounce levels have up to lcap slots
            const size_t slots_max = levels > 1 ? (size_t)P.lcap : (size_t)P.qcap;
            const unsigned sblocks = (unsigned)((slots_max + 256 * RT_WAVES_PER_BLOCK - 1) / (256 * RT_WAVES_PER_BLOCK));
            const unsigned lblocks = (unsigned)((slots_max + 255) / 256);
            int32_t *const hit0 = P.hit_id;
            const long long off0 = P.out_off;
            uint8_t *const rgba0 = P.rgba;
            uint32_t *const accum0 = P.accum;
            unsigned *const lvl_count = W.qcount + 32;
            if (!queues_clear) HIP_TRY(hipMemsetAsync(W.qcount, 0, RT_QCOUNT_CLEAR * sizeof(unsigned), st));
            queues_clear = false;
            for (int lvl = 0; lvl < levels; ++lvl) {
                if (lvl > 0) {      // trace (n-1) lights world reflected, Main.hs:554
                    HIP_TRY(hipMemsetAsync(W.qcount, 0, 32 * sizeof(unsigned), st));
                    P.rays = W.lvl_rays[lvl]; P.count_ptr = lvl_count + lvl;
                    P.hit_id = W.lvl_hit[lvl]; P.out_off = 0;
                    hipLaunchKernelGGL(k_general, dim3((unsigned)((slots_max + 63) / 64)), dim3(64), 0, st, P);
                    if (timed && (rc = next_event(D, st, 1))) return rc;
                }
                P.next_rays = (lvl + 1 < levels) ? W.lvl_rays[lvl + 1] : nullptr;
                P.next_count = lvl_count + lvl + 1;
                P.colour = (levels > 1) ? W.lvl_col[lvl] : nullptr;
                if (callbacks && (rc = run_normal_callbacks(*s, P, st, lvl == 0 ? pblocks : lblocks))) return rc;
                const unsigned ublocks = lvl == 0 ? (unsigned)P.tiles_x * (unsigned)((rows + RT_SETUP_THREADS / 32 - 1) / (RT_SETUP_THREADS / 32))
                                                  : (unsigned)((slots_max + RT_SETUP_THREADS - 1) / RT_SETUP_THREADS);
                P.setup_blocks = (int)ublocks;
                P.lit_stride = (long long)slots_max;
                for (int grp = 0; grp < ngroups; ++grp) {
                    const int first = 32 * grp, np = std::max(0, std::min(32, (int)point_lights.size() - first));
                    P.npoint = np;
                    for (int k = 0; k < np; ++k) P.point_light[k] = point_lights[(size_t)first + k];
                    P.lit_w = W.lit + (size_t)grp * slots_max;
                    if (grp > 0) HIP_TRY(hipMemsetAsync(W.qcount, 0, 32 * sizeof(unsigned), st));
                    hipLaunchKernelGGL(k_setup, dim3(ublocks), dim3(RT_SETUP_THREADS), 0, st, P);
                    if (np > 0) {
                        hipLaunchKernelGGL(k_queue_offsets, dim3((unsigned)np), dim3(1024), 0, st, P);
                        hipLaunchKernelGGL(k_queue_compact, dim3(ublocks, (unsigned)np), dim3(RT_COMPACT_THREADS), 0, st, P);
                    }
                    if (timed && (rc = next_event(D, st, 2))) return rc;
                    if (np > 0) {
                        if (use_mfma) hipLaunchKernelGGL(k_shadow_mfma, dim3(sblocks, (unsigned)np), dim3(RT_WAVES_PER_BLOCK * 64), 0, st, P);
#ifdef RT_VALU_FILTER
                        else hipLaunchKernelGGL(k_shadow, dim3(sblocks, (unsigned)np), dim3(RT_WAVES_PER_BLOCK * 64), 0, st, P);
#endif
                    }
                    if (timed && (rc = next_event(D, st, 3))) return rc;
                }
                if (P.next_rays)
                    hipLaunchKernelGGL(k_final, dim3(lvl == 0 ? (unsigned)P.tiles_x * (unsigned)((rows + 31) / 32) : (lblocks + 3) / 4), dim3(1024), 0, st, P);
                else
                    hipLaunchKernelGGL(k_final, dim3(lvl == 0 ? (unsigned)P.tiles_x * (unsigned)((rows + 7) / 8) : lblocks), dim3(256), 0, st, P);
                if (timed && (rc = next_event(D, st, 4))) return rc;
            }
            P.rays = nullptr; P.count_ptr = nullptr; P.next_rays = nullptr; P.hit_id = hit0; P.out_off = off0;
            if (levels > 1) {       // fold the levels back: direct `addPixelRGB8` scaled reflection, deepest first
                for (int lvl = levels - 1; lvl >= 1; --lvl)
                    hipLaunchKernelGGL(k_combine, dim3(lblocks), dim3(256), 0, st, W.lvl_rays[lvl], W.lvl_col[lvl],
                                       lvl_count + lvl, (unsigned)P.lcap, W.lvl_col[lvl - 1]);
                P.colour = W.lvl_col[0]; P.rgba = rgba0; P.accum = accum0;
                hipLaunchKernelGGL(k_store, dim3(pblocks), dim3(256), 0, st, P);
                P.colour = nullptr;
                if (timed && (rc = next_event(D, st, 4))) return rc;
            }
        }
    }
    if (d_rgba && spp > 1) {
        hipLaunchKernelGGL(k_resolve, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st,
                           W.accum, (uint8_t *)d_rgba, npix, (uint32_t)spp);
        if (timed && (rc = next_event(D, st, 4))) return rc;
    }
    HIP_TRY(hipGetLastError());
    D.stats_pending = true;
    return RT_OK;
}

int check_render_args(const rt_scene *s, int w, int h, int depth, int spp, int row0, int nrows) {
    if (!s) { set_error("rt_render: scene is NULL"); return RT_ERR_ARG; }
    if (w <= 0 || h <= 0 || spp < 1 || row0 < 0 || nrows < 0 || row0 + nrows > h) {
        set_error("rt_render: bad image size or row band"); return RT_ERR_ARG;
    }
    if (depth < 0) { set_error("rt_render: depth must be >= 0"); return RT_ERR_ARG; }
    if (s->lights.size() > 65535) { set_error("rt_render: more than 65535 lights are not supported"); return RT_ERR_UNSUPPORTED; }
    return RT_OK;
}
// chunk rows of the growing-workspace renders: a multiple of the tile height
size_t chunk_rows(int w, int nrows, int levels) {
    // a bounce level keeps 72 bytes per slot: a quarter of the pixels for up to 8 levels, fewer beyond
    const size_t pixels = levels > 1 ? std::min<size_t>(RT_CHUNK_PIXELS / 4, 2 * RT_CHUNK_PIXELS / (size_t)levels) : RT_CHUNK_PIXELS;
    size_t crow = pixels / (size_t)w;
    crow = crow < RT_TILE_H ? RT_TILE_H : crow - crow % RT_TILE_H;
    if (crow > (size_t)nrows) crow = (size_t)nrows;
    return crow ? crow : 1;
}

// rt_render_device on the growing workspace of the (scene, device) pair
int render_band(rt_scene *s, int device, int w, int h, int depth, int spp, uint64_t seed, int row0, int nrows,
                void *d_rgba, void *d_hit_id, hipStream_t st, bool more, hipEvent_t hits_done = nullptr) {
    int rc;
    DeviceScene &D = s->dev[device];
    if ((rc = device_prepare(*s, D, device, w, h))) return rc;
    if (!levels_supported(s, depth)) return RT_ERR_UNSUPPORTED;
    const int levels = render_levels(s, depth);
    const size_t crow = chunk_rows(w, nrows, levels);
    const WsNeeds need = ws_needs(w, nrows, crow, s->npoint, levels, spp, d_hit_id == nullptr, d_rgba != nullptr, !s->normal_fn.empty());
    if ((rc = ws_grow(D, need))) return rc;
    return render_rows(s, D, D.ws, crow, w, h, depth, spp, seed, row0, nrows, d_rgba, d_hit_id, st ? st : (hipStream_t)D.stream, more, true, hits_done);
}

} // namespace

extern "C" {

int rt_init(int device) { return device_init(device); }

int rt_scene_prepare_device(rt_scene *s, int w, int h) {
    if (!s || w <= 0 || h <= 0) { set_error("rt_scene_prepare_device: bad argument"); return RT_ERR_ARG; }
    int rc;
    if ((rc = enter())) return rc;
    return device_prepare(*s, s->dev[t_device], t_device, w, h);
}

void rt_scene_destroy(rt_scene *s) {
    if (!s) return;
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (DeviceScene &d : s->dev) {
        helper_join(d);
        if (d.device < 0) continue;
        if (hipSetDevice(d.device) != hipSuccess) continue;
        if (d.stream) (void)hipStreamSynchronize((hipStream_t)d.stream);
        if (d.copy_stream) (void)hipStreamSynchronize((hipStream_t)d.copy_stream);
        if (d.copy_stream2) (void)hipStreamSynchronize((hipStream_t)d.copy_stream2);
        // the scene's arrays and the work buffers are views of the two arenas
        void *ptrs[] = {d.arena.p, d.ws_arena.p, d.rgba, d.hit_out, d.stage_rgba, d.stage_hit};
        for (void *p : ptrs) if (p) (void)hipFree(p);
        for (void *e : d.evpool) if (e) (void)hipEventDestroy((hipEvent_t)e);
        for (void *e : d.band_done) if (e) (void)hipEventDestroy((hipEvent_t)e);
        for (void *e : d.hits_done) if (e) (void)hipEventDestroy((hipEvent_t)e);
        if (d.stream) (void)hipStreamDestroy((hipStream_t)d.stream);
        if (d.copy_stream) (void)hipStreamDestroy((hipStream_t)d.copy_stream);
        if (d.copy_stream2) (void)hipStreamDestroy((hipStream_t)d.copy_stream2);
    }
    if (cur >= 0) (void)hipSetDevice(cur);
    delete s;
}

int rt_render_device(rt_scene *s, int w, int h, int depth, int spp, uint64_t seed, int row0, int nrows,
                     void *d_rgba, void *d_hit_id, void *stream) {
    int rc;
    if ((rc = check_render_args(s, w, h, depth, spp, row0, nrows))) return rc;
    if ((rc = enter())) return rc;
    return render_band(s, t_device, w, h, depth, spp, seed, row0, nrows, d_rgba, d_hit_id, (hipStream_t)stream, false);
}

int rt_render(rt_scene *s, int w, int h, int depth, int spp, uint64_t seed, int row0, int nrows,
              uint8_t *rgba_out, int32_t *hit_id_out) {
    int rc;
    if ((rc = check_render_args(s, w, h, depth, spp, row0, nrows))) return rc;
    if ((rc = enter())) return rc;
    DeviceScene &D = s->dev[t_device];
    const size_t npix = (size_t)w * (size_t)nrows;
    if (npix == 0) return RT_OK;
    if ((rc = device_prepare(*s, D, t_device, w, h))) return rc;
    if ((rc = copy_tools(D))) return rc;
    // Colours into page-locked memory that this device can address (rt_host_alloc): k_final stores its coalesced 256-byte
    // lines straight through PCIe into the caller's buffer, so no copy of the colours is left exposed behind the last
    // kernel (33 MB at 4K: 0.6 ms).  Hit IDs still go through device memory: the shading kernels read them back.
    uint8_t *direct = nullptr;
#ifndef RT_NO_DIRECT_RGBA
    if (rgba_out && s->opt_host_bands <= 1) {
        hipPointerAttribute_t attr{};
        if (hipPointerGetAttributes(&attr, rgba_out) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer) {
            hipPointerAttribute_t last{};
            uint8_t *end = rgba_out + 4 * npix - 1;
            if (hipPointerGetAttributes(&last, end) == hipSuccess && last.type == hipMemoryTypeHost && last.devicePointer == (uint8_t *)attr.devicePointer + (4 * npix - 1))
                direct = (uint8_t *)attr.devicePointer;
        } else (void)hipGetLastError();                     // (an ordinary pointer: not an error)
    }
#endif
    if (!direct && (rc = ensure(D.rgba, D.rgba_cap, 4 * npix))) return rc;
    if ((rc = ensure(D.hit_out, D.hit_out_cap, npix))) return rc;
    hipStream_t st = (hipStream_t)D.stream, cst = (hipStream_t)D.copy_stream, hst = (hipStream_t)D.copy_stream2;
    // The frame is enqueued as row bands; band k's copies to the caller's memory run on their own streams beside the
    // kernels that follow: the hit IDs as soon as the band's nearest-hit kernel has ended (the shading kernels only read
    // them), the colours behind the band's last kernel, i.e. beside band k + 1.  Only the last band's colours are
    // exposed.  Bands cost kernel time (every band has its own tails), so there are few: one per ~4 Mi pixels.
    // Into rt_host_alloc'ed (page-locked) memory the copies are DMA at PCIe rate; into pageable memory they stage
    // through the runtime's bounce buffers and block this thread, which is why every band is enqueued first.
    int nb = s->opt_host_bands > 0 ? s->opt_host_bands : (int)(npix >> 22);
    nb = nb < 1 ? 1 : (nb > RT_HOST_BANDS_MAX ? RT_HOST_BANDS_MAX : nb);
    if (direct) nb = 1;
    int brows = (nrows + nb - 1) / nb;
    brows = (brows + RT_TILE_H - 1) / RT_TILE_H * RT_TILE_H;
    int used = 0;
    for (int r = 0; r < nrows && rc == RT_OK; r += brows, ++used) {
        const int rows = nrows - r < brows ? nrows - r : brows;
        const size_t off = (size_t)r * (size_t)w;
        rc = render_band(s, t_device, w, h, depth, spp, seed, row0 + r, rows, rgba_out ? (direct ? direct : D.rgba) + 4 * off : nullptr,
                         D.hit_out + off, st, r > 0, hit_id_out ? (hipEvent_t)D.hits_done[used] : nullptr);
        if (rc == RT_OK && !hip_ok(hipEventRecord((hipEvent_t)D.band_done[used], st), "hipEventRecord")) rc = RT_ERR_HIP;
    }
    int band = 0;
    for (int r = 0; r < nrows && rc == RT_OK; r += brows, ++band) {
        const int rows = nrows - r < brows ? nrows - r : brows;
        const size_t off = (size_t)r * (size_t)w, n = (size_t)rows * (size_t)w;
        if (hit_id_out) {
            if (!hip_ok(hipStreamWaitEvent(hst, (hipEvent_t)D.hits_done[band], 0), "hipStreamWaitEvent")) { rc = RT_ERR_HIP; break; }
            if (!hip_ok(hipMemcpyAsync(hit_id_out + off, D.hit_out + off, n * sizeof(int32_t), hipMemcpyDeviceToHost, hst), "hipMemcpyAsync")) { rc = RT_ERR_HIP; break; }
        }
        if (rgba_out && !direct) {
            if (!hip_ok(hipStreamWaitEvent(cst, (hipEvent_t)D.band_done[band], 0), "hipStreamWaitEvent")) { rc = RT_ERR_HIP; break; }
            if (!hip_ok(hipMemcpyAsync(rgba_out + 4 * off, D.rgba + 4 * off, 4 * n, hipMemcpyDeviceToHost, cst), "hipMemcpyAsync")) rc = RT_ERR_HIP;
        }
    }
    // always drain both streams, also after an error: nothing may still be writing the caller's buffers on return
    if (!hip_ok(hipStreamSynchronize(cst), "hipStreamSynchronize") && rc == RT_OK) rc = RT_ERR_HIP;
    if (!hip_ok(hipStreamSynchronize(hst), "hipStreamSynchronize") && rc == RT_OK) rc = RT_ERR_HIP;
    if (!hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize") && rc == RT_OK) rc = RT_ERR_HIP;
    return rc;
}

void *rt_host_alloc(size_t bytes) {
    if (enter() != RT_OK) return nullptr;
    void *p = nullptr;
    if (!hip_ok(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable), "hipHostMalloc")) return nullptr;
    return p;
}

void rt_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int rt_synchronize(void) {
    if (t_device < 0) return RT_OK;
    int rc;
    if ((rc = enter())) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return RT_OK;
}

} // extern "C"

namespace {
// statistics of the last render of the (scene, device) pair; synchronises with it
int device_stats(rt_scene *s, DeviceScene &D, rt_stats *out) {
    if (D.stats_pending) {
        static thread_local std::vector<unsigned long long> all(RT_STAT_SLOTS * 16);
        if (D.ev_used > 0) HIP_TRY(hipEventSynchronize((hipEvent_t)D.evpool[D.ev_used - 1]));
        else HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(all.data(), D.last_counters, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long c[16] = {0};
        for (int slot = 0; slot < RT_STAT_SLOTS; ++slot) for (int k = 0; k < 16; ++k) c[k] += all[(size_t)slot * 16 + k];
        double ms[5] = {0, 0, 0, 0, 0};
        for (int i = 1; i < D.ev_used; ++i) {
            float t = 0;
            HIP_TRY(hipEventElapsedTime(&t, (hipEvent_t)D.evpool[i - 1], (hipEvent_t)D.evpool[i]));
            ms[D.evkind[i]] += t;
        }
        float tot = 0;
        if (D.ev_used > 1) HIP_TRY(hipEventElapsedTime(&tot, (hipEvent_t)D.evpool[0], (hipEvent_t)D.evpool[D.ev_used - 1]));
        rt_stats st{};
        st.primary_rays = D.last_primary;
        st.shadow_rays = c[0];
        st.bounce_rays = c[5];
        st.bounce_blocks = c[8];
        const uint64_t rays = st.primary_rays + st.shadow_rays + st.bounce_rays;
        st.tri_pairs = rays * (uint64_t)s->tri_gid.size();
        st.rect_pairs = rays * (uint64_t)s->rect_gid.size();
        st.exact_evals = c[1] + c[2];
        st.record_tests = c[4];
        st.mfma_blocks_primary = c[6];
        st.mfma_blocks_shadow = c[7];
#ifdef RT_COUNT_EVENTS
        fprintf(stderr, "[rt] primary: true hits among the exact evaluations %llu; exact evals primary %llu shadow %llu\n", c[3], c[1], c[2]);
#endif
        st.ms_primary = ms[1]; st.ms_setup = ms[2]; st.ms_shadow = ms[3]; st.ms_final = ms[4]; st.ms_total = tot;
        st.upload_ms = D.stats.upload_ms;
        D.stats = st;
        D.stats_pending = false;
#ifdef RT_WAVE_TIMES
        if (const char *path = getenv("RT_WAVE_TIMES_FILE")) {
            std::vector<unsigned long long> rec(8 + 8 * (1ull << 20));
            HIP_TRY(hipMemcpy(rec.data(), g_wave_dbg, rec.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            if (FILE *f = fopen(path, "wb")) { fwrite(rec.data(), 8, 8 + 8 * (size_t)std::min<unsigned long long>(rec[0], 1ull << 20), f); fclose(f); }
        }
#endif
    }
    *out = D.stats;
    out->prepare_ms = s->prepare_ms;
    return RT_OK;
}
} // namespace

extern "C" int rt_get_stats(rt_scene *s, rt_stats *out) {
    if (!s || !out) { set_error("rt_get_stats: NULL argument"); return RT_ERR_ARG; }
    if (s->multi_stats) { *out = s->stats_multi; return RT_OK; }
    int rc;
    if ((rc = enter())) return rc;
    const int dev = s->last_device >= 0 ? s->last_device : t_device;
    if (dev != t_device) HIP_TRY(hipSetDevice(dev));
    rc = device_stats(s, s->dev[dev], out);
    if (dev != t_device) (void)hipSetDevice(t_device);
    return rc;
}

namespace {

// Relative cost of `ncost` equal horizontal strips of the w x h frame, measured: the strips are rendered one by one at
// a reduced scale (a quarter or half the width and height while a strip keeps >= 256 Ki pixels, else full scale) and
// timed with HIP events, the better of two passes.  The bands of a multi-GPU render are cut from this (split_rows).
int row_costs(rt_scene *s, int device, int w, int h, int depth, int spp, int ncost, double *cost) {
    int q = 4;
    while (q > 1 && ((size_t)(w / q) * (size_t)(h / q)) / (size_t)ncost < (256u << 10)) q >>= 1;
    const int sw = std::max(1, w / q), sh = std::max(ncost, h / q);
    int rc;
    DeviceScene &D = s->dev[device];
    if ((rc = device_prepare(*s, D, device, sw, sh))) return rc;
    const int maxrows = (sh + ncost - 1) / ncost + 1;
    if ((rc = ensure(D.stage_rgba, D.stage_rgba_cap, 4 * (size_t)sw * (size_t)maxrows))) return rc;
    for (int k = 0; k < ncost; ++k) cost[k] = 1e300;
    for (int pass = 0; pass < 2; ++pass)
        for (int k = 0; k < ncost; ++k) {
            const int r0 = (int)((long long)sh * k / ncost), r1 = (int)((long long)sh * (k + 1) / ncost);
            if (r1 <= r0) { cost[k] = 0; continue; }
            if ((rc = render_band(s, device, sw, sh, depth, spp, 0, r0, r1 - r0, D.stage_rgba, nullptr, (hipStream_t)D.stream, false))) return rc;
            rt_stats st;
            if ((rc = device_stats(s, D, &st))) return rc;
            cost[k] = std::min(cost[k], st.ms_total);
        }
    return RT_OK;
}

// port of raytrace_amd/distributed.py: sub_bands with the piece sizes of piece_fractions (1 : 2 : ... : 2 : 1 — a small
// first piece puts data on the link early, a small last one leaves little travelling when the render has ended)
void sub_bands(int r0, int nr, int parts, int align, int *p0, int *pn) {
    const int units = (nr + align - 1) / align;
    double total = 0, run = 0;
    for (int k = 0; k < parts; ++k) total += (parts <= 2 || (k > 0 && k + 1 < parts)) ? 2.0 : 1.0;
    int done = 0;
    for (int k = 0; k < parts; ++k) {
        run += (parts <= 2 || (k > 0 && k + 1 < parts)) ? 2.0 : 1.0;
        const int upto = k + 1 < parts ? std::min(nr, std::max(done, (int)((double)units * (run / total) + 0.5) * align)) : nr;
        p0[k] = r0 + done; pn[k] = upto - done;
        done = upto;
    }
}

} // namespace

extern "C" int rt_scene_row_costs(rt_scene *s, int w, int h, int depth, int spp, int ncost, double *cost) {
    int rc;
    if ((rc = check_render_args(s, w, h, depth, spp, 0, h))) return rc;
    if (!cost || ncost < 1 || ncost > h) { set_error("rt_scene_row_costs: bad argument"); return RT_ERR_ARG; }
    if ((rc = enter())) return rc;
    return row_costs(s, t_device, w, h, depth, spp, ncost, cost);
}

extern "C" int rt_render_multi(rt_scene *s, int w, int h, int depth, int spp, uint64_t seed, int ngpus, const int *devices,
                               uint8_t *rgba_out, int32_t *hit_id_out) {
    int rc;
    if ((rc = check_render_args(s, w, h, depth, spp, 0, h))) return rc;
    if (ngpus < 1 || ngpus > RT_MAX_DEVICES) { set_error("rt_render_multi: ngpus must be between 1 and 16"); return RT_ERR_ARG; }
    int ndev = 0;
    if (!hip_ok(hipGetDeviceCount(&ndev), "hipGetDeviceCount") || ndev <= 0) {
        set_error("rt_render_multi: no HIP device available (this library has no CPU fallback)"); return RT_ERR_HIP;
    }
    int devs[RT_MAX_DEVICES];
    for (int k = 0; k < ngpus; ++k) {
        devs[k] = devices ? devices[k] : k;
        if (devs[k] < 0 || devs[k] >= ndev || devs[k] >= RT_MAX_DEVICES) { set_error("rt_render_multi: device index out of range"); return RT_ERR_ARG; }
    }
    const int caller = t_device, home = devs[0];
    const size_t npix = (size_t)w * (size_t)h;
    constexpr int PIECES = 4;
    struct Band { int dev, row0, nrows; size_t stage_off; bool staged; int p0[PIECES], pn[PIECES]; };
    Band band[RT_MAX_DEVICES];
    std::vector<hipEvent_t> ev;            // per (band, piece): rendered, arrived on the home device
    bool first_on[RT_MAX_DEVICES], peer_ok[RT_MAX_DEVICES];
    for (bool &f : first_on) f = true;
    for (bool &f : peer_ok) f = false;
    auto finish = [&](int code) {          // drain every stream that was used, restore the caller's device
        for (int k = 0; k < ngpus; ++k) {
            DeviceScene &D = s->dev[devs[k]];
            if (hipSetDevice(devs[k]) != hipSuccess) continue;
            if (D.stream && !hip_ok(hipStreamSynchronize((hipStream_t)D.stream), "hipStreamSynchronize") && code == RT_OK) code = RT_ERR_HIP;
            if (D.copy_stream && !hip_ok(hipStreamSynchronize((hipStream_t)D.copy_stream), "hipStreamSynchronize") && code == RT_OK) code = RT_ERR_HIP;
            // (the home device's copies into the caller's buffers: nothing may still be writing them after an error return)
            if (D.copy_stream2 && !hip_ok(hipStreamSynchronize((hipStream_t)D.copy_stream2), "hipStreamSynchronize") && code == RT_OK) code = RT_ERR_HIP;
        }
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
        if (caller >= 0) (void)hipSetDevice(caller);
        t_device = caller;
        return code;
    };
    if ((rc = device_init(home))) return finish(rc);
    DeviceScene &H = s->dev[home];
    if ((rc = device_prepare(*s, H, home, w, h))) return finish(rc);
    if ((rc = copy_tools(H))) return finish(rc);
    // 1. bands of about equal measured cost
    int b0[RT_MAX_DEVICES] = {0}, bn[RT_MAX_DEVICES] = {h};
    if (ngpus > 1) {
        const int ncost = std::min(32, std::max(1, h / 8));
        if (s->multi_cost.size() != (size_t)ncost || s->multi_cost_key[0] != s->version || s->multi_cost_key[1] != (uint64_t)w ||
            s->multi_cost_key[2] != (uint64_t)h || s->multi_cost_key[3] != (uint64_t)depth || s->multi_cost_key[4] != (uint64_t)spp) {
            s->multi_cost.assign(ncost, 1.0);
            if ((rc = row_costs(s, home, w, h, depth, spp, ncost, s->multi_cost.data()))) return finish(rc);
            if ((rc = device_prepare(*s, H, home, w, h))) return finish(rc);
            const uint64_t key[5] = {s->version, (uint64_t)w, (uint64_t)h, (uint64_t)depth, (uint64_t)spp};
            std::copy(key, key + 5, s->multi_cost_key);
        }
        // devices[0] collects: it sends nothing, so it may render while the others' last pieces travel — one more of
        // the PIECES * 3/2 - 1 units a band's pieces are made of (distributed.py: plan_frame has the long form)
        double weight[RT_MAX_DEVICES];
        for (int k = 0; k < ngpus; ++k) weight[k] = 1.0;
        if (devs[0] != devs[ngpus - 1] || ngpus > 1) weight[0] = 1.0 + 1.0 / (2.0 * (PIECES - 1));
        split_rows(s->multi_cost.data(), ncost, h, ngpus, RT_TILE_H, b0, bn, weight);
    }
    // 2. buffers: the whole frame on the home device, a staging buffer per other device
    if (rgba_out && (rc = ensure(H.rgba, H.rgba_cap, 4 * npix))) return finish(rc);
    if ((rc = ensure(H.hit_out, H.hit_out_cap, npix))) return finish(rc);
    size_t stage_pix[RT_MAX_DEVICES] = {0};
    for (int k = 0; k < ngpus; ++k) {
        Band &B = band[k];
        B.dev = devs[k]; B.row0 = b0[k]; B.nrows = bn[k];
        B.staged = B.dev != home || s->opt_multi_stage;
        B.stage_off = stage_pix[B.dev];
        if (B.staged) stage_pix[B.dev] += (size_t)w * (size_t)B.nrows;
        sub_bands(B.row0, B.nrows, PIECES, RT_TILE_H, B.p0, B.pn);
    }
    for (int k = 0; k < ngpus; ++k) {
        const int d = devs[k];
        if (!stage_pix[d]) continue;
        if ((rc = device_init(d))) return finish(rc);
        DeviceScene &D = s->dev[d];
        if ((rc = device_prepare(*s, D, d, w, h))) return finish(rc);
        if ((rc = copy_tools(D))) return finish(rc);
        if (rgba_out && (rc = ensure(D.stage_rgba, D.stage_rgba_cap, 4 * stage_pix[d]))) return finish(rc);
        if ((rc = ensure(D.stage_hit, D.stage_hit_cap, stage_pix[d]))) return finish(rc);
        if (d != home) {                  // direct xGMI copies into the home device's memory
            const hipError_t e = hipDeviceEnablePeerAccess(home, 0);
            peer_ok[d] = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;   // else hipMemcpyPeerAsync stages through the host
            (void)hipGetLastError();
        }
    }
    // 3. every device is driven by a host thread of its own (enqueueing 6 kernels per piece costs the host ~0.1 ms: from
    // one thread, 8 devices x 4 pieces would take as long as a device needs for its whole band); a device's pieces are
    // enqueued in order, and a piece travels to the home device (peer-to-peer over xGMI, on the band's copy stream) while
    // the next one renders.  Scenes with fnorm callbacks stay on this thread (the caller's closure is not ours to run
    // concurrently).
    ev.assign((size_t)ngpus * PIECES * 2, nullptr);
    for (size_t i = 0; i < ev.size(); ++i)
        if (!hip_ok(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming), "hipEventCreate")) return finish(RT_ERR_HIP);
    auto drive = [&](int dev, std::string &err) -> int {        // all bands of one device, piece by piece
        int rc2;
        if ((rc2 = device_init(dev))) { err = rt_last_error(); return rc2; }
        DeviceScene &D = s->dev[dev];
        bool first = true;
        for (int p = 0; p < PIECES; ++p)
            for (int k = 0; k < ngpus; ++k) {
                Band &B = band[k];
                if (B.dev != dev || B.pn[p] <= 0) continue;
                const size_t foff = (size_t)B.p0[p] * (size_t)w, n = (size_t)B.pn[p] * (size_t)w;
                const size_t soff = B.stage_off + (size_t)(B.p0[p] - B.row0) * (size_t)w;
                uint8_t *dst = !rgba_out ? nullptr : (B.staged ? D.stage_rgba + 4 * soff : H.rgba + 4 * foff);
                int32_t *dhit = B.staged ? D.stage_hit + soff : H.hit_out + foff;
                if ((rc2 = render_band(s, dev, w, h, depth, spp, seed, B.p0[p], B.pn[p], dst, dhit, (hipStream_t)D.stream, !first))) { err = rt_last_error(); return rc2; }
                first = false;
                first_on[dev] = false;
                hipEvent_t er = ev[((size_t)k * PIECES + p) * 2], ea = ev[((size_t)k * PIECES + p) * 2 + 1];
                bool ok = hip_ok(hipEventRecord(er, (hipStream_t)D.stream), "hipEventRecord");
                if (ok && B.staged) {
                    hipStream_t cs = (hipStream_t)D.copy_stream;
                    ok = hip_ok(hipStreamWaitEvent(cs, er, 0), "hipStreamWaitEvent") &&
                         (!rgba_out || hip_ok(hipMemcpyPeerAsync(H.rgba + 4 * foff, home, dst, dev, 4 * n, cs), "hipMemcpyPeerAsync")) &&
                         hip_ok(hipMemcpyPeerAsync(H.hit_out + foff, home, dhit, dev, n * sizeof(int32_t), cs), "hipMemcpyPeerAsync") &&
                         hip_ok(hipEventRecord(ea, cs), "hipEventRecord");
                }
                if (!ok) { err = rt_last_error(); return RT_ERR_HIP; }
            }
        return RT_OK;
    };
    {
        int distinct[RT_MAX_DEVICES], nd = 0;
        for (int k = 0; k < ngpus; ++k) { bool seen = false; for (int j = 0; j < nd; ++j) seen |= distinct[j] == devs[k]; if (!seen) distinct[nd++] = devs[k]; }
        int rcs[RT_MAX_DEVICES] = {0};
        std::string errs[RT_MAX_DEVICES];
        if (nd > 1 && s->normal_fn.empty()) {
            std::vector<std::thread> workers;
            for (int j = 1; j < nd; ++j) workers.emplace_back([&, j] { rcs[j] = drive(distinct[j], errs[j]); });
            rcs[0] = drive(distinct[0], errs[0]);
            for (std::thread &t : workers) t.join();
        } else {
            for (int j = 0; j < nd && (j == 0 || rcs[j - 1] == RT_OK); ++j) rcs[j] = drive(distinct[j], errs[j]);
        }
        for (int j = 0; j < nd; ++j) if (rcs[j] != RT_OK) { set_error(errs[j]); return finish(rcs[j]); }
    }
    // 4. from the home device to the caller's memory, piece by piece as they arrive
    if ((rc = device_init(home))) return finish(rc);
    for (int p = 0; p < PIECES; ++p)
        for (int k = 0; k < ngpus; ++k) {
            const Band &B = band[k];
            if (B.pn[p] <= 0) continue;
            const size_t foff = (size_t)B.p0[p] * (size_t)w, n = (size_t)B.pn[p] * (size_t)w;
            hipEvent_t arrived = ev[((size_t)k * PIECES + p) * 2 + (B.staged ? 1 : 0)];
            hipStream_t cs = (hipStream_t)H.copy_stream2;
            if (!hip_ok(hipStreamWaitEvent(cs, arrived, 0), "hipStreamWaitEvent")) return finish(RT_ERR_HIP);
            if (rgba_out && !hip_ok(hipMemcpyAsync(rgba_out + 4 * foff, H.rgba + 4 * foff, 4 * n, hipMemcpyDeviceToHost, cs), "hipMemcpyAsync")) return finish(RT_ERR_HIP);
            if (hit_id_out && !hip_ok(hipMemcpyAsync(hit_id_out + foff, H.hit_out + foff, n * sizeof(int32_t), hipMemcpyDeviceToHost, cs), "hipMemcpyAsync")) return finish(RT_ERR_HIP);
        }
    if (!hip_ok(hipStreamSynchronize((hipStream_t)H.copy_stream2), "hipStreamSynchronize")) return finish(RT_ERR_HIP);
    if ((rc = finish(RT_OK))) return rc;
    // 5. statistics of the whole frame: counts summed over the devices, times = the slowest device
    rt_stats tot{};
    bool seen[RT_MAX_DEVICES] = {false};
    for (int k = 0; k < ngpus; ++k) {
        const int d = devs[k];
        if (seen[d] || first_on[d]) continue;
        seen[d] = true;
        rt_stats st;
        if (hipSetDevice(d) != hipSuccess || device_stats(s, s->dev[d], &st) != RT_OK) continue;
        tot.primary_rays += st.primary_rays; tot.shadow_rays += st.shadow_rays; tot.bounce_rays += st.bounce_rays;
        tot.tri_pairs += st.tri_pairs; tot.rect_pairs += st.rect_pairs; tot.exact_evals += st.exact_evals;
        tot.record_tests += st.record_tests; tot.mfma_blocks_primary += st.mfma_blocks_primary; tot.mfma_blocks_shadow += st.mfma_blocks_shadow;
        tot.ms_primary = std::max(tot.ms_primary, st.ms_primary); tot.ms_setup = std::max(tot.ms_setup, st.ms_setup);
        tot.ms_shadow = std::max(tot.ms_shadow, st.ms_shadow); tot.ms_final = std::max(tot.ms_final, st.ms_final);
        tot.ms_total = std::max(tot.ms_total, st.ms_total); tot.upload_ms = std::max(tot.upload_ms, st.upload_ms);
    }
    tot.prepare_ms = s->prepare_ms;
    for (int k = 0; k < ngpus; ++k) {          // how the bands reached devices[0]
        if (devs[k] == home) continue;
        if (peer_ok[devs[k]]) ++tot.multi_bands_direct; else ++tot.multi_bands_staged;
    }
    s->stats_multi = tot;
    s->multi_stats = true;
    if (caller >= 0) (void)hipSetDevice(caller);
    return RT_OK;
}

// rt_render_device without a single allocation: no hipMalloc / hipFree, no event or stream creation.  The scene must be
// resident already (rt_scene_prepare_device); work buffers come from the static device array `scratch` (64 MiB per GPU).
// A band that needs more than that is rendered as consecutive sub-bands that each fit; only a frame so wide that 16 rows
// do not fit is refused.  All calls on one GPU share the array: calls that overlap in time race.
extern "C" int rt_render_device_noalloc(rt_scene *s, int w, int h, int depth, int spp, uint64_t seed, int row0, int nrows,
                                        void *d_rgba, void *d_hit_id, void *stream) {
    int rc;
    if ((rc = check_render_args(s, w, h, depth, spp, row0, nrows))) return rc;
    if (t_device < 0) { set_error("rt_render_device_noalloc: call rt_init(device) on this thread first"); return RT_ERR_ARG; }
    HIP_TRY(hipSetDevice(t_device));
    DeviceScene &D = s->dev[t_device];
    if (!host_ready(*s, w, h) || !D.stream || D.version != s->version || D.eye_version != s->eye_version) {
        set_error("rt_render_device_noalloc: the scene is not resident on this device for this image size: call rt_scene_prepare_device(scene, w, h) first");
        return RT_ERR_ARG;
    }
    if (!s->normal_fn.empty()) { set_error("rt_render_device_noalloc: scenes with fnorm callbacks need host staging and are not supported here"); return RT_ERR_UNSUPPORTED; }
    if (!levels_supported(s, depth)) return RT_ERR_UNSUPPORTED;
    if (!d_rgba && !d_hit_id) { set_error("rt_render_device_noalloc: no output buffer"); return RT_ERR_ARG; }
    const int levels = render_levels(s, depth);
    unsigned char *base = g_ctx[t_device].scratch;       // the device address, never the array's name
    // the most rows (all of them, or a multiple of the 16-row set-up tiles) whose buffers fit
    Workspace W;
    auto fits = [&](int rows) {
        return ws_carve(base, RT_STATIC_SCRATCH_BYTES, ws_needs(w, rows, (size_t)rows, s->npoint, levels, spp, d_hit_id == nullptr, d_rgba != nullptr, false), W) != 0;
    };
    int rows_fit = nrows;
    if (nrows > 0 && !fits(nrows)) {
        int lo = 0, hi = nrows / 16;                       // in units of 16 rows: lo fits (0 = nothing), hi may not
        while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (fits(mid * 16)) lo = mid; else hi = mid - 1; }
        rows_fit = lo * 16;
        if (rows_fit == 0) {                               // not even 16 rows (or the band has fewer): single rows, if one fits
            if (!fits(1)) {
                set_error("rt_render_device_noalloc: the image is too wide: one row of it needs more than the 64 MiB of static scratch");
                return RT_ERR_UNSUPPORTED;
            }
            rows_fit = 1;
            while (rows_fit < 15 && rows_fit < nrows && fits(rows_fit + 1)) ++rows_fit;
        }
        (void)fits(rows_fit);                              // W = the layout that is used
    }
    hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)D.stream;
    // (the array keeps whatever the last call left: no history of the lights' costs here, they are taken in their own order)
    if (nrows > 0) HIP_TRY(hipMemsetAsync(W.qcount + RT_QCOUNT_HIST, 0, (RT_QCOUNT_WORDS - RT_QCOUNT_HIST) * sizeof(unsigned), st));
    for (int r = 0; r < nrows; r += rows_fit) {
        const int rows = nrows - r < rows_fit ? nrows - r : rows_fit;
        const size_t off = (size_t)r * (size_t)w;
        if ((rc = render_rows(s, D, W, (size_t)rows, w, h, depth, spp, seed, row0 + r, rows,
                              d_rgba ? (uint8_t *)d_rgba + 4 * off : nullptr, d_hit_id ? (int32_t *)d_hit_id + off : nullptr,
                              st, r > 0, false))) return rc;
    }
    if (nrows == 0) { D.last_primary = 0; D.ev_used = 0; D.stats_pending = false; }
    return RT_OK;
}
