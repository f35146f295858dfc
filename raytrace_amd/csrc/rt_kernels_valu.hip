// rt_kernels_valu.hip — the FP32 vector-ALU kernel family (k_primary, k_shadow): an independent implementation of the
// conservative filters, compiled ONLY into the cross-check build libraytrace_gpu_valu.so (-DRT_VALU_FILTER), which tests and
// tools compare whole frames of the product's matrix-core kernels against (rt_scene_set_option RT_OPT_FILTER 1).
//   One wavefront owns a 32 x 8 pixel tile, a lane 4 pixels as two float2 pairs (v_pk_fma_f32); leaf records are
//   wave-uniform and are read with scalar loads into SGPRs: no VGPRs, no LDS bandwidth, no bank conflicts.  Stage 1 (the FP32
//   record of the exact shape) is evaluated for every (ray, leaf) pair; survivors go through the exact FP64 path of rt_device.h.
#ifndef RT_VALU_FILTER
#error "rt_kernels_valu.hip belongs to the cross-check build only: compile with -DRT_VALU_FILTER"
#endif
#include "rt_device.h"

// occupancy (waves per SIMD) the register allocator must reach for the two brute-force kernels
#ifndef RT_WAVES_EU_PRIMARY
#define RT_WAVES_EU_PRIMARY 8
#endif
#ifndef RT_WAVES_EU_SHADOW
#define RT_WAVES_EU_SHADOW 6
#endif

namespace {

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

// ------------------------------------------------------------------------------------------
// Kernel 3: shadow rays of one point light (blockIdx.y), 4 jobs per lane, brute force over all leaves.
//   replaces `intersect world ray_to_light` + the t <= distance test of pointLight (Main.hs:472-474).
//   Apex = the light, D = shadow-ray origin - light; the filter's clip keeps leaves between them.

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
                        const HitRec H = load_hit_point(P, pr);
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
            const HitRec H = load_hit_point(P, pix[r]);
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

} // namespace

namespace rt {
void launch_primary_valu(const RenderParams &P, bool shared_y, unsigned tblocks, hipStream_t st) {
    if (shared_y) hipLaunchKernelGGL(k_primary<true>, dim3(tblocks), dim3(RT_WAVES_PER_BLOCK * 64), 0, st, P);
    else hipLaunchKernelGGL(k_primary<false>, dim3(tblocks), dim3(RT_WAVES_PER_BLOCK * 64), 0, st, P);
}
void launch_shadow_valu(const RenderParams &P, unsigned sblocks, unsigned nlights, hipStream_t st) {
    hipLaunchKernelGGL(k_shadow, dim3(sblocks, nlights), dim3(RT_WAVES_PER_BLOCK * 64), 0, st, P);
}
} // namespace rt
