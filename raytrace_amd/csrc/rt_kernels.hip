// rt_kernels.hip — the gfx950 (MI355X, CDNA4) kernels of the render path, and the functions that launch them.
//
// Replaces parallelTrace / computeP (reference src/Main.hs:623-637) and everything its element
// function calls: cast (506-533), trace (541-557), intersect world (196-217, 226-277, 363-385),
// light lights (436-492).
//
// Design (DESIGN.md has the long form):
//   * Every ray is decided against every leaf, through a chain of conservative filters in front of the exact test.  All rays
//     of a pass leave a common apex (the eye; a light for shadow rays), so a leaf is seen from the apex under a bounding
//     cone (stage 0, one linear form per pair) and its exact shape is three linear forms of the ray direction (stage 1,
//     rt_scene.cpp); blocks of 32 leaves are tested against the bound of a wave's 256 rays first (stage -1).
//   * Matrix-core kernels (k_primary_mfma, k_shadow_mfma, k_general): stage 0 for 32 leaves x 32 rays is one
//     v_mfma_f32_32x32x16_f16; survivors go through per-wave LDS queues to stage 1 (FP32) and to the exact stage, 64
//     entries at a time.
//   * (ray, leaf) pairs that pass the filters — a handful per ray — are evaluated in IEEE double with the reference's
//     own expression order (planeLineIsect etc., rt_device.h), which decides the hit, the tie-break (left-most leaf,
//     Main.hs:213-214) and the colour exactly.  Compiled with -ffp-contract=off; the FP32 code uses explicit fma builtins.
//   * The FP32 vector-ALU kernel family of the cross-check build lives in rt_kernels_valu.hip; the host runtime that
//     enqueues all this in rt_render.hip.
#include "rt_device.h"

namespace {

// ------------------------------------------------------------------------------------------
// Kernel 2: per hit pixel, the exact hit point and shading normal; shadow jobs for every point light
//   with L.N > 0, compacted into one queue per light (wave-aggregated atomics).
//   When L.N <= 0 both Phong terms are exactly 0 (Main.hs:480-484), so the shadow ray cannot change
//   the pixel and is not cast.

// Wave history (RenderParams::wcost_p ...): one work-group sorts n waves by the duration they had, longest first — a
// counting sort on the duration in units of 0.64 us (512 bins, one thread per bin; the order inside a bin is whatever
// the atomics give).  Every index 0 .. n - 1 is written to exactly one place: the result is a permutation whatever the
// durations were.  `have` = 0: no durations yet, the identity.  Durations are consumed (zeroed): a wave that does not run
// next time (fewer shadow jobs) then counts as free.  These are the FIRST work-groups of k_setup's grid (1 + lights of
// them, only when a history is kept): they run beside the thousands of work-groups that do k_setup's own work, off the
// critical path (as extra work-groups of k_queue_offsets they cost 0.038 ms per frame, mostly serialised LDS atomics).
// Measured against it and no better (profiles/r03_wave_history.txt): only the longest 5 - 30 % first and the shortest
// 10 - 30 % last with the middle left in launch order; two tiles per work-group, a long one and a short one.
constexpr int RT_WBINS = 512;
__device__ __forceinline__ unsigned wave_bin(unsigned ticks) { return (unsigned)(RT_WBINS - 1) - min(ticks >> 6, (unsigned)(RT_WBINS - 1)); }
__device__ void wave_order_block(uint32_t *cost, uint32_t *order, int n, bool have) {
    __shared__ unsigned s_bin[RT_WBINS], s_scan[RT_WBINS];
    const int t = (int)threadIdx.x;
    static_assert(RT_SETUP_THREADS == RT_WBINS, "one thread per bin");
    if (!have) {
        for (int i = t; i < n; i += RT_SETUP_THREADS) { order[i] = (uint32_t)i; cost[i] = 0u; }
        return;
    }
    s_bin[t] = 0u;
    __syncthreads();
    for (int i = t; i < n; i += RT_SETUP_THREADS) atomicAdd(&s_bin[wave_bin(cost[i])], 1u);
    __syncthreads();
    s_scan[t] = s_bin[t];
    __syncthreads();
    for (int o = 1; o < RT_WBINS; o <<= 1) {             // inclusive scan over the bins
        const unsigned v = t >= o ? s_scan[t - o] : 0u;
        __syncthreads();
        s_scan[t] += v;
        __syncthreads();
    }
    s_bin[t] = s_scan[t] - s_bin[t];                     // exclusive: where the bin starts
    __syncthreads();
    for (int i = t; i < n; i += RT_SETUP_THREADS) {
        order[atomicAdd(&s_bin[wave_bin(cost[i])], 1u)] = (uint32_t)i;
        cost[i] = 0u;
    }
}

#ifndef RT_WAVES_EU_SETUP
#define RT_WAVES_EU_SETUP 4
#endif
__global__ void __attribute__((amdgpu_waves_per_eu(RT_WAVES_EU_SETUP, 8))) __launch_bounds__(RT_SETUP_THREADS) k_setup(const RenderParams P) {
    // shadow jobs of the block per light: per-wave counts -> offsets inside the block's run
    __shared__ unsigned s_cnt[32][RT_SETUP_THREADS / 64];
    if ((int)blockIdx.x < P.wsort_blocks) {               // wave history: the sorting work-groups come first (0 of them: none kept)
        const int b = (int)blockIdx.x;
        if (b == 0) wave_order_block(P.wcost_p, P.worder_p_out, P.tiles_x * P.tiles_y, true);
        else wave_order_block(P.wcost_s + (size_t)(b - 1) * (size_t)P.wstride_s, P.worder_s + (size_t)(b - 1) * (size_t)P.wstride_s, P.wstride_s, P.whave != 0);
        return;
    }
    const unsigned block = blockIdx.x - (unsigned)P.wsort_blocks;
    // level 0: a block takes a 32 x 16 pixel tile, so that the shadow rays it queues (256 consecutive entries make one
    // wave of k_shadow_mfma) leave from neighbouring pixels and form a narrow bundle; bounce levels: slots in order
    int pix = (int)(block * RT_SETUP_THREADS + threadIdx.x);
    bool act = (unsigned)pix < num_slots(P);
    if (!P.rays) {
        // a wave takes 8 x 8 pixels of the tile (objects are blobs: fewer waves straddle an outline than with 32 x 2 strips)
        const int tx = (int)block % P.tiles_x, ty = (int)block / P.tiles_x;
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
        store_hit(P, (size_t)pix, H, normal_is_the_leafs(L));
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
        P.tcount[(size_t)threadIdx.x * (size_t)P.setup_blocks + block] = run;
    }
    __syncthreads();
    // the block's jobs go to its own slots; k_queue_offsets + k_queue_compact then line the blocks' runs up in block
    // order, so that the 256 consecutive jobs of a k_shadow_mfma wave come from the same or from neighbouring tiles
    for (pl = 0; pl < P.npoint; ++pl) {
        const bool need = (needmask >> pl) & 1u;
        const unsigned long long m = __ballot(need);
        if (need) P.qsparse[((size_t)pl * (size_t)P.setup_blocks + block) * RT_SETUP_THREADS + s_cnt[pl][wv] + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)pix;
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
#if RT_WAVES_PER_BLOCK == 1
    const unsigned wave = P.worder_s ? P.worder_s[(size_t)pl * (size_t)P.wstride_s + blockIdx.x] : blockIdx.x;   // longest wave of the frame before first
#else
    const unsigned wave = blockIdx.x * RT_WAVES_PER_BLOCK + (threadIdx.x >> 6);
#endif
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
            const HitRec H = load_hit_point(P, pix);
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
            const HitRec H = load_hit_point(P, s_pix[wv][e]);
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
    const unsigned ticks = (unsigned)(wall_clock64() - t_start);
    if (lane == 0 && (wave & 7u) == 3u) {       // every eighth wave reports how long it took (100 MHz ticks): the lights' order next time
        atomicAdd(&P.qcount[RT_QCOUNT_HIST + 2 * pl], ticks);
        atomicAdd(&P.qcount[RT_QCOUNT_HIST + 2 * pl + 1], 1u);
    }
#if RT_WAVES_PER_BLOCK == 1
    if (lane == 0 && P.wcost_s) P.wcost_s[(size_t)pl * (size_t)P.wstride_s + wave] = ticks;     // ... and every wave, for its own place in the order
#endif
#ifdef RT_WAVE_TIMES
    wave_ph[2] = (Q.t_s1 & 0xffff) | ((Q.t_ex & 0xffff) << 16) | ((Q.t_bits & 0xffff) << 32) | ((Q.t_loop & 0xffff) << 48);
    wave_record(P, 1u | ((unsigned)pl << 8), (unsigned long long)P.tiles_x * P.tiles_y + (unsigned long long)gridDim.x * blockIdx.y + blockIdx.x, wave, wave_t0, nblocks, Q.nrec, nexact, wave_ph);
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
#if RT_WAVES_PER_BLOCK == 1
    const int wave = P.worder_p ? (int)P.worder_p[blockIdx.x] : (int)blockIdx.x;                 // longest wave of the frame before first
    const unsigned long long t_start = wall_clock64();
#else
    const int wave = (int)(blockIdx.x * RT_WAVES_PER_BLOCK) + wv;
#endif
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
#if RT_WAVES_PER_BLOCK == 1
    if (lane == 0 && P.wcost_p) P.wcost_p[wave] = (unsigned)(wall_clock64() - t_start);
#endif
#ifdef RT_WAVE_TIMES
    wave_ph[2] = (Q.t_s1 & 0xffff) | ((Q.t_ex & 0xffff) << 16) | ((Q.t_bits & 0xffff) << 32) | ((Q.t_loop & 0xffff) << 48);
    wave_record(P, 0u, blockIdx.x, (unsigned)wave, wave_t0, nblocks, Q.nrec, nexact, wave_ph);
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
        const LeafD &Lf = P.leaves[id];
        const MaterialD M = P.mats[Lf.mat];
        const HitRec H = load_hit(P, (size_t)pix, Lf);
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
    const uint32_t rgb = (uint32_t)cr | ((uint32_t)cg << 8) | ((uint32_t)cb << 16);
    if (!P.colour && P.rgba && P.bpp == 3) { store_rgb24(P.rgba, P.out_off + pix, rgb, act, P.rgb_packed != 0); return; }    // (every lane)
    if (!act) return;
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
    wave_record(P, 2u, blockIdx.x, wave, wave_t0, nblocks, Q.nrec, nexact, wave_ph);
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
    const bool act = pix < P.qcap;
    const uint32_t rgb = act ? P.colour[pix] : 0u;
    if (P.rgba && P.bpp == 3) { store_rgb24(P.rgba, P.out_off + pix, rgb, act, P.rgb_packed != 0); return; }
    if (!act) return;
    if (P.rgba) reinterpret_cast<uint32_t *>(P.rgba)[P.out_off + pix] = rgb | 0xff000000u;
    else {
        uint32_t *a = P.accum + 3 * (P.out_off + pix);
        a[0] += rgb & 255u; a[1] += (rgb >> 8) & 255u; a[2] += (rgb >> 16) & 255u;
    }
}

// spp > 1: mean of the per-sample 8-bit colours, (sum + spp/2) / spp
__global__ void k_resolve(const uint32_t *__restrict__ accum, uint8_t *__restrict__ rgba, size_t npix, uint32_t spp, int bpp, int rgb_packed) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = i < npix;
    uint32_t rgb = 0u;
    if (act) {
        const uint32_t r = (accum[3 * i] + spp / 2) / spp, g = (accum[3 * i + 1] + spp / 2) / spp, b = (accum[3 * i + 2] + spp / 2) / spp;
        rgb = r | (g << 8) | (b << 16);
    }
    if (bpp == 3) { store_rgb24(rgba, (long long)i, rgb, act, rgb_packed != 0); return; }
    if (act) reinterpret_cast<uint32_t *>(rgba)[i] = rgb | 0xff000000u;
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

namespace rt {
void launch_primary_mfma(const RenderParams &P, unsigned tblocks, hipStream_t st) { hipLaunchKernelGGL(k_primary_mfma, dim3(tblocks), dim3(RT_WAVES_PER_BLOCK * 64), 0, st, P); }
void launch_shadow_mfma(const RenderParams &P, unsigned sblocks, unsigned nlights, hipStream_t st) { hipLaunchKernelGGL(k_shadow_mfma, dim3(sblocks, nlights), dim3(RT_WAVES_PER_BLOCK * 64), 0, st, P); }
void launch_setup(const RenderParams &P, unsigned ublocks, hipStream_t st) { hipLaunchKernelGGL(k_setup, dim3(ublocks + (unsigned)P.wsort_blocks), dim3(RT_SETUP_THREADS), 0, st, P); }
void launch_queue_offsets(const RenderParams &P, unsigned nlights, hipStream_t st) { hipLaunchKernelGGL(k_queue_offsets, dim3(nlights), dim3(1024), 0, st, P); }
void launch_queue_compact(const RenderParams &P, unsigned ublocks, unsigned nlights, hipStream_t st) { hipLaunchKernelGGL(k_queue_compact, dim3(ublocks, nlights), dim3(RT_COMPACT_THREADS), 0, st, P); }
void launch_final(const RenderParams &P, unsigned blocks, unsigned threads, hipStream_t st) { hipLaunchKernelGGL(k_final, dim3(blocks), dim3(threads), 0, st, P); }
void launch_general(const RenderParams &P, unsigned runs, hipStream_t st) { hipLaunchKernelGGL(k_general, dim3(runs), dim3(64), 0, st, P); }
void launch_combine(const RayRec *rays, const uint32_t *col, const unsigned *count, unsigned cap, uint32_t *parent_col, unsigned blocks, hipStream_t st) {
    hipLaunchKernelGGL(k_combine, dim3(blocks), dim3(256), 0, st, rays, col, count, cap, parent_col);
}
void launch_store(const RenderParams &P, unsigned blocks, hipStream_t st) { hipLaunchKernelGGL(k_store, dim3(blocks), dim3(256), 0, st, P); }
void launch_resolve(const uint32_t *accum, uint8_t *rgba, size_t npix, uint32_t spp, int bpp, int rgb_packed, hipStream_t st) {
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, accum, rgba, npix, spp, bpp, rgb_packed);
}
void launch_cb_gather(const RenderParams &P, unsigned blocks, hipStream_t st) { hipLaunchKernelGGL(k_cb_gather, dim3(blocks), dim3(256), 0, st, P); }

const char *kernel_build_flags() {
    return ""
#ifdef RT_MEASUREMENT_BUILD
        " MEASUREMENT BUILD (results may differ from the reference's):"
#ifdef RT_NO_GUARDS
        " RT_NO_GUARDS"
#endif
#ifdef RT_X_NOSCAN
        " RT_X_NOSCAN"
#endif
#ifdef RT_X_SCANONLY
        " RT_X_SCANONLY"
#endif
#ifdef RT_MF_NOSLOW
        " RT_MF_NOSLOW"
#endif
#ifdef RT_NO_DEPTH_SKIP
        " RT_NO_DEPTH_SKIP"
#endif
#ifdef RT_NO_TALK
        " RT_NO_TALK"
#endif
#endif
#ifdef RT_WAVE_TIMES
        " RT_WAVE_TIMES"
#endif
        ;
}
} // namespace rt
