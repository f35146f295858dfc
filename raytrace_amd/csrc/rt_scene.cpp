// rt_scene.cpp — host scene builder behind the C ABI (include/raytrace_gpu.h).
//
// Every rt_add_* reifies one constructor of the reference as leaves: the constants the reference
// closes over (normal, d, point', ww, hh / u, v, uv, uu, vv, n) are computed here once, in IEEE
// double and in the reference's own expression order (no FMA contraction: build with
// -ffp-contract=off), because the kernels' exact path must reproduce the reference bit for bit.
#include "rt_scene.h"

#ifndef RT_SHADOW_CLIP
#define RT_SHADOW_CLIP 0      // must match rt_kernels.hip
#endif

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>
#include <limits>

namespace rt {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

struct D3 { double x, y, z; };
static inline D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline D3 scale(double s, D3 v) { return {s * v.x, s * v.y, s * v.z}; }
static inline double dot3(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          // Main.hs:63
static inline D3 cross3(D3 a, D3 b) {                                                         // Main.hs:68-69
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - b.x * a.y};
}
static inline D3 unit(D3 v) { double n = std::sqrt(dot3(v, v)); return {v.x / n, v.y / n, v.z / n}; } // :79
static inline D3 from(const double p[3]) { return {p[0], p[1], p[2]}; }

double ray_eps() { return 2.2 * std::pow(10.0, -308.0); }                                     // Main.hs:114

CameraD fixed_camera(int width, int height) {                                                 // Main.hs:506-527
    CameraD c;
    const double w = (double)width, h = (double)height;
    const double fov = M_PI / 2;
    c.scaleX = 1.0 / w;
    c.scaleY = c.scaleX * (-h / w);
    c.dX = -(c.scaleX * w / 2);
    c.dY = -(c.scaleY * h / 2);
    c.d = std::tan(fov / 2) * c.dX;
    return c;
}

static int add_material(rt_scene &s, const rt_material &m) {
    MaterialD d{};
    d.specularity = m.specularity;
    d.reflectivity = m.reflectivity;
    std::memcpy(d.diffuse, m.diffuse, 3);
    std::memcpy(d.specular, m.specular, 3);
    s.mats.push_back(d);
    return (int)s.mats.size() - 1;
}

static inline double hmin(double x, double y) { return x <= y ? x : y; }   // Haskell min / max on Double
static inline double hmax(double x, double y) { return x <= y ? y : x; }

static void attach(rt_scene &s, int node) {
    if (s.open.empty()) (s.cur_world ? s.shadow_top[(size_t)s.cur_world - 1] : s.top).push_back(node);
    else s.nodes[s.open.back()].kids.push_back(node);
}
// Shapes of the traced world come first: a hit ID is the leaf's place in the traced world's depth-first order.
static bool world_accepts_shapes(const rt_scene &s, const char *who) {
    if (s.cur_world == 0 && !s.shadow_top.empty()) {
        set_error(std::string(who) + ": the traced world must be complete before the first rt_begin_shadow_world");
        return false;
    }
    return true;
}
static int open_group(rt_scene &s, int kind) {
    if (!world_accepts_shapes(s, kind == 1 ? "rt_begin_mconcat" : "rt_begin_mappend")) return RT_ERR_ARG;
    s.nodes.push_back(GNode{kind, -1, {}});
    const int id = (int)s.nodes.size() - 1;
    attach(s, id);
    s.open.push_back(id);
    s.dirty = true;
    return RT_OK;
}
static int close_group(rt_scene &s, int kind, const char *who) {
    if (s.open.empty() || s.nodes[s.open.back()].kind != kind) { set_error(std::string(who) + ": no matching open bracket"); return RT_ERR_ARG; }
    if (kind == 2 && s.nodes[s.open.back()].kids.size() != 2) { set_error(std::string(who) + ": mappend takes exactly two children"); return RT_ERR_ARG; }
    s.open.pop_back();
    return RT_OK;
}

static int push_leaf(rt_scene &s, LeafD &L, const Box &bb) {
    s.leaf_bb.push_back(bb);
    s.leaf_world.push_back(s.cur_world);
    s.nodes.push_back(GNode{0, (int)s.leaves.size(), {}});
    attach(s, (int)s.nodes.size() - 1);
    L.guard = 0;
    L.aux = -1;
    L.nmode = RT_NORMAL_GEOMETRIC;
    if (L.kind == LEAF_TRI) { L.slot = (int)s.tri_gid.size(); s.tri_gid.push_back((int)s.leaves.size()); }
    else if (L.kind == LEAF_SPHERE) { L.slot = (int)s.sph_gid.size(); s.sph_gid.push_back((int)s.leaves.size()); }
    else { L.slot = (int)s.rect_gid.size(); s.rect_gid.push_back((int)s.leaves.size()); }
    s.leaves.push_back(L);
    s.dirty = true;
    return (int)s.leaves.size() - 1;
}

// rectangle material point width height — Main.hs:226-232
static int add_rect(rt_scene &s, const rt_material &m, D3 point, D3 width, D3 height) {
    LeafD L{};
    L.kind = LEAF_RECT;
    L.mat = add_material(s, m);
    D3 normal = unit(cross3(width, height));                                          // Main.hs:228
    if (s.talk) {                                                                     // Talk.hs:66-67, 114: scalar (recip (norm v)) v
        const D3 c = cross3(width, height);
        const double rn = 1.0 / std::sqrt(dot3(c, c));
        normal = D3{rn * c.x, rn * c.y, rn * c.z};
    }
    const D3 corner = (point - scale(0.5, width)) - scale(0.5, height);
    L.nx = normal.x; L.ny = normal.y; L.nz = normal.z;
    L.d = -(dot3(point, normal));
    L.p0x = corner.x; L.p0y = corner.y; L.p0z = corner.z;
    L.e1x = width.x; L.e1y = width.y; L.e1z = width.z;
    L.e2x = height.x; L.e2y = height.y; L.e2z = height.z;
    L.c0 = dot3(width, width);
    L.c1 = dot3(height, height);
    const D3 far = (corner + width) + height;                                         // Main.hs:244
    Box bb = {{hmin(corner.x, far.x), hmin(corner.y, far.y), hmin(corner.z, far.z)},   // :245 (two corners only)
              {hmax(corner.x, far.x), hmax(corner.y, far.y), hmax(corner.z, far.z)}};
    return push_leaf(s, L, bb);
}

// triangle material pa pb pc — Main.hs:363-372
static int add_tri(rt_scene &s, const rt_material &m, D3 pa, D3 pb, D3 pc) {
    LeafD L{};
    L.kind = LEAF_TRI;
    L.mat = add_material(s, m);
    const D3 u = pb - pa, v = pc - pa;
    const double uv = dot3(u, v), uu = dot3(u, u), vv = dot3(v, v);
    const D3 normal = unit(cross3(u, v));
    L.nx = normal.x; L.ny = normal.y; L.nz = normal.z;
    L.d = dot3(scale(-1, pa), normal);
    L.p0x = pa.x; L.p0y = pa.y; L.p0z = pa.z;
    L.e1x = u.x; L.e1y = u.y; L.e1z = u.z;
    L.e2x = v.x; L.e2y = v.y; L.e2z = v.z;
    L.c0 = uv; L.c1 = uu; L.c2 = vv;
    L.c3 = uv * uv - uu * vv;
    Box bb = {{hmin(pa.x, hmin(pb.x, pc.x)), hmin(pa.y, hmin(pb.y, pc.y)), hmin(pa.z, hmin(pb.z, pc.z))},   // :384
              {hmax(pa.x, hmax(pb.x, pc.x)), hmax(pa.y, hmax(pb.y, pc.y)), hmax(pa.z, hmax(pb.z, pc.z))}};
    return push_leaf(s, L, bb);
}

// sphere material center radius — Main.hs:392-415 (boundingBox :412-414)
static int add_sphere(rt_scene &s, const rt_material &m, D3 c, double radius) {
    LeafD L{};
    L.kind = LEAF_SPHERE;
    L.mat = add_material(s, m);
    L.p0x = c.x; L.p0y = c.y; L.p0z = c.z;
    L.c0 = radius;
    Box bb = {{c.x - radius, c.y - radius, c.z - radius}, {c.x + radius, c.y + radius, c.z + radius}};
    return push_leaf(s, L, bb);
}

static inline float f_down(double v);

// Sphere filter records.  The reference solves t^2 + b t + c = 0 assuming a unit direction (Main.hs:396-400):
// delta = (2 d.o)^2 - 4 (o.o - r^2) >= 0, i.e. the LINE of the ray passes within r of the centre; that does not
// depend on where on the line o lies, so it can be tested from the apex: (D.oc)^2 - |D|^2 (|oc|^2 - r^2) >= 0.
static FilterRec make_sphere_rec(const LeafD &L, const double A[3], const double dmax[3]) {
    FilterRec r{};
    const double oc[3] = {A[0] - L.p0x, A[1] - L.p0y, A[2] - L.p0z};
    const double oo = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2], cs = oo - L.c0 * L.c0;
    double B = 0, DD = 0;
    for (int i = 0; i < 3; ++i) { B += std::fabs(oc[i]) * dmax[i]; DD += dmax[i] * dmax[i]; }
    const double m = std::ldexp(B * B + DD * (oo + L.c0 * L.c0), -19) + 1e-30;
    r.c[9] = 1.0f;
    if (!std::isfinite(m)) { r.c[9] = -1.0f; return r; }      // all-zero forms, negm = -1: always passes
    for (int i = 0; i < 3; ++i) r.c[i] = (float)oc[i];
    r.c[3] = (float)cs;
    r.c[9] = f_down(-m);
    return r;
}
static GenRec make_sphere_gen_rec(const LeafD &L, const double omax[3]) {
    GenRec r{};
    double O = 0;
    const double c[3] = {L.p0x, L.p0y, L.p0z};
    for (int i = 0; i < 3; ++i) O += (omax[i] + std::fabs(c[i])) * (omax[i] + std::fabs(c[i]));
    // reflected directions are unit vectors only up to rounding (and not at all after a mantle normal): |d| <= 4 is ample
    const double m = std::ldexp(16.0 * O + O + L.c0 * L.c0, -18) + 1e-30;
    for (int i = 0; i < 3; ++i) r.c[i] = (float)c[i];
    r.c[3] = (float)(L.c0 * L.c0);
    r.c[12] = std::isfinite(m) ? f_down(-m) : -INFINITY;
    return r;
}

// ------------------------------------------------------------------------------------------
// FP32 filter records.
//
// For rays p(u) = A + u*D leaving a common apex A, the plane of a leaf is met at u* = K / N with
// K = n.(p0 - A) and N = n.D, and every inside-test quantity q of the reference (the barycentrics
// r, s of Main.hs:379-380, the normalised rectangle coordinates of Main.hs:238-241) is affine in
// the hit point, so q*N is LINEAR in D:  q*N = D . (q(A) n + K grad q).  The filter evaluates those
// linear forms in FP32 and accepts when they are within a per-leaf error margin of the accepting
// side — no division, so no NaN/inf can hide a hit.  Records are sign-normalised to K >= 0, so a
// hit in front of the apex has N > 0.
static inline float f_down(double v) {
    float f = (float)v;
    if ((double)f > v) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}

static FilterRec never_rec() {   // degenerate leaf (NaN normal): the reference never hits it
    FilterRec r{};
    r.c[10] = 0.0f;
    r.c[9] = 1.0f;     // z <= 0 < 1: never passes
    return r;
}
static FilterRec always_rec() {  // apex (numerically) in the leaf's plane: let the exact path decide
    FilterRec r{};
    r.c[10] = -1.0f;
    r.c[9] = -1.0f;
    return r;
}

// The linear forms of one leaf seen from one apex, in double: c1, c2 (the two inside coordinates times N) and cn (N),
// sign-normalised to K >= 0.  status: 0 usable, 1 the leaf can never be hit, 2 let the exact path decide every ray.
struct Forms { double c1[3], c2[3], cn[3], K, m, mK; int status; };

static Forms leaf_forms(const LeafD &L, const double A[3], const double dmax[3]) {
    Forms F{};
    const double n[3] = {L.nx, L.ny, L.nz};
    const double e1[3] = {L.e1x, L.e1y, L.e1z}, e2[3] = {L.e2x, L.e2y, L.e2z};
    double g1[3], g2[3], q1A, q2A;   // gradients of the two inside coordinates and their value at A
    if (L.kind == LEAF_TRI) {
        // r = w.gr, s = w.gs with w = p - pa (Main.hs:376-380)
        for (int i = 0; i < 3; ++i) {
            g1[i] = (L.c0 * e2[i] - L.c2 * e1[i]) / L.c3;
            g2[i] = (L.c0 * e1[i] - L.c1 * e2[i]) / L.c3;
        }
        const double w[3] = {A[0] - L.p0x, A[1] - L.p0y, A[2] - L.p0z};
        q1A = g1[0] * w[0] + g1[1] * w[1] + g1[2] * w[2];
        q2A = g2[0] * w[0] + g2[1] * w[1] + g2[2] * w[2];
    } else {
        // a' = 2*dw/ww - 1, b' = 2*dh/hh - 1 in [-1, 1] with dV = p - point' (Main.hs:238-241).  Measured from point',
        // not from the centre: when width and height are not orthogonal, a' is not 0 at the centre.
        for (int i = 0; i < 3; ++i) { g1[i] = 2.0 * e1[i] / L.c0; g2[i] = 2.0 * e2[i] / L.c1; }
        const double w[3] = {A[0] - L.p0x, A[1] - L.p0y, A[2] - L.p0z};
        q1A = g1[0] * w[0] + g1[1] * w[1] + g1[2] * w[2] - 1.0;
        q2A = g2[0] * w[0] + g2[1] * w[1] + g2[2] * w[2] - 1.0;
    }
    double K = -(L.d + (n[0] * A[0] + n[1] * A[1] + n[2] * A[2]));
    double *c1 = F.c1, *c2 = F.c2, *cn = F.cn;
    for (int i = 0; i < 3; ++i) { c1[i] = q1A * n[i] + K * g1[i]; c2[i] = q2A * n[i] + K * g2[i]; cn[i] = n[i]; }
    if (K < 0) { K = -K; for (int i = 0; i < 3; ++i) { c1[i] = -c1[i]; c2[i] = -c2[i]; cn[i] = -cn[i]; } }
    double S = 0, SN = 0;
    for (int i = 0; i < 3; ++i) {
        S += (std::fabs(c1[i]) + std::fabs(c2[i]) + std::fabs(cn[i])) * dmax[i];
        SN += std::fabs(cn[i]) * dmax[i];
    }
    if (!std::isfinite(S) || !std::isfinite(K)) { F.status = 1; return F; }
    // margins: every input is rounded to FP32 once (2^-24 relative) and each linear form takes
    // three FP32 operations, W two more; 2^-20 of the magnitude sum leaves a factor > 3 of slack.
    F.K = K;
    F.m = std::ldexp(S, -20) + 1e-30;
    F.mK = std::ldexp(SN + K, -20) + 1e-30;
    if (K <= 4 * F.mK) F.status = 2;
    return F;
}

// eye_d: 0 for a light apex; the camera's d for the eye apex, whose records are affine in (posX, posY).
static FilterRec make_rec(const LeafD &L, const double A[3], const double dmax[3], double eye_d = 0.0) {
    const Forms F = leaf_forms(L, A, dmax);
    if (F.status == 1) return never_rec();
    if (F.status == 2) return always_rec();
    const double *c1 = F.c1, *c2 = F.c2, *cn = F.cn;
    const double K = F.K, m = F.m, mK = F.mK;
    FilterRec r{};
    if (eye_d != 0.0) {
        // Eye apex: D = (posX, posY, d) with d constant, so every form is  f.x*posX + f.y*posY + f.z*d.
        // Triangles store W = N - U - V instead of N (no clip test for primary rays: a hit between the eye
        // and the viewport plane is rejected by the exact path, Main.hs:257).
        double f3[3] = {cn[0], cn[1], cn[2]};
        if (L.kind == LEAF_TRI) for (int i = 0; i < 3; ++i) f3[i] = cn[i] - c1[i] - c2[i];
        const double *f[3] = {c1, c2, f3};
        for (int k = 0; k < 3; ++k) {
            r.c[3 * k] = (float)f[k][0]; r.c[3 * k + 1] = (float)f[k][1]; r.c[3 * k + 2] = (float)(f[k][2] * eye_d);
        }
        r.c[9] = f_down(-m);
        r.c[10] = 0.0f;
        for (int i = 0; i < 11; ++i) if (!std::isfinite(r.c[i])) return always_rec();
        return r;
    }
#if RT_SHADOW_CLIP
    for (int i = 0; i < 3; ++i) { r.c[i] = (float)c1[i]; r.c[3 + i] = (float)c2[i]; r.c[6 + i] = (float)cn[i]; }
#else
    // Light apex without the clip form: a triangle stores W = N - U - V, like the eye apex.  Leaves beyond the
    // shadow-ray origin then pass the filter too; k_shadow drops them with a plane-side test before the exact path.
    for (int i = 0; i < 3; ++i) {
        r.c[i] = (float)c1[i]; r.c[3 + i] = (float)c2[i];
        r.c[6 + i] = (float)(L.kind == LEAF_TRI ? cn[i] - c1[i] - c2[i] : cn[i]);
    }
#endif
    r.c[9] = f_down(-m);
#if RT_SHADOW_CLIP
    r.c[10] = f_down((K - mK) - (double)r.c[9]);   // Kq: N - Kq >= -m  <=>  N >= K - mK
#else
    r.c[10] = f_down(K - mK);                       // candidate path: N.D >= Kq, N in c[11..13]
#endif
    for (int i = 0; i < 3; ++i) r.c[11 + i] = (float)cn[i];
    for (int i = 0; i < 14; ++i) if (!std::isfinite(r.c[i])) return always_rec();   // FP32 overflow
    return r;
}

// ------------------------------------------------------------------------------------------
// MFMA rows (v_mfma_f32_32x32x16_f16): the first stage of the matrix-core kernels tests every ray against a
// bounding cone of every planar leaf — ONE linear form per (ray, leaf), so a 32 x 32 tile of the matrix product
// decides 32 leaves x 32 rays.
//
// Seen from the apex A (the eye, or a light), the directions to the points of a convex planar leaf form a convex
// spherical polygon.  If it lies inside a cap  { u : a.u >= cos(theta) }  around a unit axis a (theta < 90 deg:
// the angular distance to a is then largest at a vertex), a ray  A + u D (u > 0)  can only meet the leaf if
//     a . D^ - cos(theta) >= 0          with D^ = D / |D|.
// The kernels normalise D in FP32 and split 2^10 D^ into FP16 hi + mid (hi = fp16(v), mid = fp16(v - hi): 22
// significant bits).  The factor 2^10 is there in case the matrix core flushes FP16 denormals: a mid part below
// 2^-14 belongs to a component below 2^-3, i.e. |D^_i| < 2^-13, and losing it costs at most 2^-11 2^-3 2^10 = 2^-4
// of 2^20 per product: 2^-24 in unscaled units.  A row holds 2^10 (a_x, a_y, a_z, M - cos(theta)) split the same
// way, and the 16 products of one row x column are exactly (a_hi + a_mid)(D_hi + D_mid) over x, y, z plus the
// constant times w = 2^10 (|values| <= 2^10 << 65504; products are exact in FP32, sums stay below 2^22).
// Error budget (unscaled units, |a| = 1, |D^| <= 1 + 2^-22): splits 2 * 2^-22, constant 2^-22, D^ 2^-21,
// FP32 accumulation of 16 products of magnitude <= 2: 2^-19.  The margin M = 2^-17 is more than 4 times their sum.
// The exact shape (the FP32 record of the leaf, then the FP64 evaluation) is tested on the candidates only.
static inline uint16_t f16_rne(float f) {             // float -> IEEE half, round to nearest even; |f| < 65504
    uint32_t u; std::memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const int32_t e = (int32_t)((u >> 23) & 0xffu) - 127 + 15;
    uint32_t m = u & 0x7fffffu;
    if (((u >> 23) & 0xffu) == 0) return (uint16_t)sign;                  // zero / float denormal
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);                        // overflow -> inf (callers check the range)
    if (e <= 0) {                                                          // half denormal
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        const int shift = 14 - e;                                          // 14 .. 24
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;               // may carry into the exponent: still correct
    return (uint16_t)(sign | r);
}
static inline float f16_val(uint16_t hbits) {
    const uint32_t sign = (uint32_t)(hbits & 0x8000u) << 16, e = (hbits >> 10) & 0x1fu, m = hbits & 0x3ffu;
    float f;
    if (e == 0) { f = std::ldexp((float)m, -24); uint32_t u; std::memcpy(&u, &f, 4); u |= sign; std::memcpy(&f, &u, 4); return f; }
    const uint32_t u = sign | ((e - 15 + 127) << 23) | (m << 13);
    std::memcpy(&f, &u, 4);
    return f;
}
static void split_f16(double v, uint16_t &hi, uint16_t &mid) {             // |v| <= 2^11
    hi = f16_rne((float)v);
    mid = f16_rne((float)(v - (double)f16_val(hi)));
}
static constexpr double MF_SCALE = 1024.0;                                 // 2^10, on both operands
static constexpr double MF_MARGIN = 1.0 / 131072.0;                        // 2^-17
static void const_row(uint16_t row[8], float w) {                          // form = w * 2^20: never (< 0) or always (> 0) a candidate
    for (int j = 0; j < 8; ++j) row[j] = 0;
    row[6] = f16_rne(w * (float)MF_SCALE);
}
// Smallest cap around the unit vectors u[0..n-1] (n <= 4) among the caps through two or three of them; returns
// cos(theta) (or -2: none found) and the axis.
static double bounding_cap(const double u[4][3], int n, double axis[3]) {
    double best = -2.0;
    auto consider = [&](const double a[3]) {
        const double len = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        if (!(len > 1e-300)) return;
        const double c[3] = {a[0] / len, a[1] / len, a[2] / len};
        double mn = 2.0;
        for (int i = 0; i < n; ++i) mn = std::fmin(mn, c[0] * u[i][0] + c[1] * u[i][1] + c[2] * u[i][2]);
        if (mn > best) { best = mn; for (int k = 0; k < 3; ++k) axis[k] = c[k]; }
    };
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            const double a[3] = {u[i][0] + u[j][0], u[i][1] + u[j][1], u[i][2] + u[j][2]};
            consider(a);
            for (int k = j + 1; k < n; ++k) {       // the circle through three of them: axis = normal of their plane
                const double p[3] = {u[j][0] - u[i][0], u[j][1] - u[i][1], u[j][2] - u[i][2]};
                const double q[3] = {u[k][0] - u[i][0], u[k][1] - u[i][1], u[k][2] - u[i][2]};
                double a3[3] = {p[1] * q[2] - p[2] * q[1], p[2] * q[0] - p[0] * q[2], p[0] * q[1] - p[1] * q[0]};
                if (a3[0] * u[i][0] + a3[1] * u[i][1] + a3[2] * u[i][2] < 0) for (int t = 0; t < 3; ++t) a3[t] = -a3[t];
                consider(a3);
            }
        }
    double mean[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) mean[k] += u[i][k];
    consider(mean);
    return best;
}
// The points of the plane that the reference's inside test accepts: the triangle pa pb pc (Main.hs:382), or the
// parallelogram 0 <= dV.w <= ww, 0 <= dV.h <= hh (Main.hs:238-241; w and h need not be orthogonal).
static int leaf_polygon(const LeafD &L, double v[4][3]) {
    const double p0[3] = {L.p0x, L.p0y, L.p0z}, e1[3] = {L.e1x, L.e1y, L.e1z}, e2[3] = {L.e2x, L.e2y, L.e2z};
    if (L.kind == LEAF_TRI) {
        for (int i = 0; i < 3; ++i) { v[0][i] = p0[i]; v[1][i] = p0[i] + e1[i]; v[2][i] = p0[i] + e2[i]; }
        return 3;
    }
    const double ww = L.c0, hh = L.c1, wh = e1[0] * e2[0] + e1[1] * e2[1] + e1[2] * e2[2];
    const double det = ww * hh - wh * wh;
    if (!(std::fabs(det) > 1e-12 * std::fabs(ww * hh)) || !std::isfinite(det)) return 0;
    const double st[4][2] = {{0, 0}, {ww, 0}, {ww, hh}, {0, hh}};
    for (int k = 0; k < 4; ++k) {
        const double al = (st[k][0] * hh - st[k][1] * wh) / det, be = (st[k][1] * ww - st[k][0] * wh) / det;
        for (int i = 0; i < 3; ++i) v[k][i] = p0[i] + al * e1[i] + be * e2[i];
    }
    return 4;
}
// Bounding cone of a planar leaf seen from the apex: status 0 = axis and cs = cos(theta) hold, 1 = the reference never
// hits this leaf, 2 = wide or degenerate (every ray is a candidate).
struct LeafCone { double axis[3]; double cs; int status; };
static LeafCone leaf_cone(const LeafD &L, const double A[3], const double dmax[3]) {
    LeafCone C = {{0, 0, 0}, -2.0, 2};
    const Forms F = leaf_forms(L, A, dmax);
    if (F.status == 1) { C.status = 1; return C; }               // NaN constants
    double v[4][3], u[4][3];
    const int n = leaf_polygon(L, v);
    bool ok = n > 0;
    for (int k = 0; k < n && ok; ++k) {
        double d[3], len2 = 0, scale = 0;
        for (int i = 0; i < 3; ++i) { d[i] = v[k][i] - A[i]; len2 += d[i] * d[i]; scale += std::fabs(v[k][i]) + std::fabs(A[i]); }
        const double len = std::sqrt(len2);
        ok = std::isfinite(len) && len > 1e-9 * (1.0 + scale);
        for (int i = 0; i < 3 && ok; ++i) u[k][i] = d[i] / len;
    }
    C.cs = ok ? bounding_cap(u, n, C.axis) : -2.0;
    C.status = (C.cs > 0.05) ? 0 : 2;
    return C;
}
static void make_cone_row(const LeafCone &C, uint16_t row[8]) {
    if (C.status == 1) { const_row(row, -1.0f); return; }
    if (C.status == 2) { const_row(row, 1.0f); return; }
    for (int i = 0; i < 3; ++i) split_f16(MF_SCALE * C.axis[i], row[2 * i], row[2 * i + 1]);
    split_f16(MF_SCALE * (MF_MARGIN - C.cs * (1.0 - 1e-12)), row[6], row[7]);
}

// Stream order of one apex.  The kernels skip a block of 32 rows when the bounding cone of the block misses the
// bounding cone of a wave's 256 rays, so leaves that lie in the same direction from the apex should share a block:
// the cone axes are mapped to the unit square (octahedral map) and split at medians, always at a multiple of 32
// leaves, along the wider of the two coordinates.  Any order gives the same image: the nearest hit is the minimum
// over (t, leaf index) and a shadow ray only asks whether some leaf occludes.
static void direction_split(std::vector<int> &idx, const std::vector<double> &u, const std::vector<double> &v, size_t lo, size_t hi) {
    const size_t n = hi - lo;
    if (n <= (size_t)MF_LEAVES) return;
    const size_t half = ((n / 2 + MF_LEAVES / 2) / MF_LEAVES) * MF_LEAVES;
    if (half == 0 || half >= n) return;
    double ulo = 2, uhi = -2, vlo = 2, vhi = -2;
    for (size_t k = lo; k < hi; ++k) {
        ulo = std::fmin(ulo, u[idx[k]]); uhi = std::fmax(uhi, u[idx[k]]);
        vlo = std::fmin(vlo, v[idx[k]]); vhi = std::fmax(vhi, v[idx[k]]);
    }
    const std::vector<double> &key = (uhi - ulo >= vhi - vlo) ? u : v;
    std::nth_element(idx.begin() + lo, idx.begin() + lo + half, idx.begin() + hi,
                     [&](int a, int b) { return key[a] < key[b] || (key[a] == key[b] && a < b); });
    direction_split(idx, u, v, lo, lo + half);
    direction_split(idx, u, v, lo + half, hi);
}
// Bounding cone of a block of leaf cones, as the kernels read it: {axis xyz, cos T, sin T, code, 0, 0}; code 0: a wave
// scans the block when the axis lies within T + rho of the bound of its ray directions (rt_kernels.hip: culled_scan);
// code 1 "never" (only padding or never-hit leaves) and 2 "always" (some leaf is wide or degenerate) come with a
// zero axis and cos T = +-1e30.
static void block_cone(const LeafCone *C, int n, float out[8]) {
    for (int k = 0; k < 8; ++k) out[k] = 0.f;
    double ax[3] = {0, 0, 0};
    int live = 0;
    for (int i = 0; i < n; ++i) {
        if (C[i].status == 2) { out[3] = -1e30f; out[5] = 2.f; return; }
        if (C[i].status == 0) { ++live; for (int k = 0; k < 3; ++k) ax[k] += C[i].axis[k]; }
    }
    if (live == 0) { out[3] = 1e30f; out[5] = 1.f; return; }
    const double len = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    if (!(len > 1e-6 * live)) { out[3] = -1e30f; out[5] = 2.f; return; }
    for (int k = 0; k < 3; ++k) ax[k] /= len;
    double T = 0;
    for (int i = 0; i < n; ++i) {
        if (C[i].status != 0) continue;
        const double dt = ax[0] * C[i].axis[0] + ax[1] * C[i].axis[1] + ax[2] * C[i].axis[2];
        T = std::fmax(T, std::acos(std::fmax(-1.0, std::fmin(1.0, dt))) + std::acos(std::fmax(-1.0, std::fmin(1.0, C[i].cs))));
    }
    T = T * (1.0 + 1e-9) + 2e-6;                                  // slack for the FP32 axis and the FP32 test
    if (!(T < 1.4)) { out[3] = -1e30f; out[5] = 2.f; return; }    // wider than 80 degrees: not worth a test
    out[0] = (float)ax[0]; out[1] = (float)ax[1]; out[2] = (float)ax[2];
    out[3] = (float)std::cos(T); out[4] = (float)std::sin(T);
}

// ------------------------------------------------------------------------------------------
// Guard boxes.  Monoid BoundingBox (Main.hs:142-150): MkBB 0 0 is the identity.  Every mappend node on
// the path from the root to a leaf tests its box (Main.hs:209); the distinct boxes of that path form
// the leaf's guard chain.
static const Box BOX_EMPTY = {{0, 0, 0}, {0, 0, 0}};
static bool box_is_empty(const Box &b) {
    for (int i = 0; i < 3; ++i) if (b.lo[i] != 0 || b.hi[i] != 0) return false;
    return true;
}
static Box box_append(const Box &a, const Box &b) {
    if (box_is_empty(a)) return b;
    if (box_is_empty(b)) return a;
    Box r;
    for (int i = 0; i < 3; ++i) { r.lo[i] = hmin(a.lo[i], b.lo[i]); r.hi[i] = hmax(a.hi[i], b.hi[i]); }
    return r;
}
static Box node_box(const rt_scene &s, int id) {
    const GNode &n = s.nodes[id];
    if (n.kind == 0) return s.leaf_bb[n.leaf];
    if (n.kind == 2) return box_append(node_box(s, n.kids[0]), node_box(s, n.kids[1]));
    Box acc = BOX_EMPTY;                                   // foldr mappend mempty
    for (size_t k = n.kids.size(); k-- > 0;) acc = box_append(node_box(s, n.kids[k]), acc);
    return acc;
}
static bool box_eq(const Box &a, const GuardD &g) {
    return a.lo[0] == g.x0 && a.lo[1] == g.y0 && a.lo[2] == g.z0 && a.hi[0] == g.x1 && a.hi[1] == g.y1 && a.hi[2] == g.z1;
}
// A new link of the guard chain below `parent`; an identical box directly above is reused (the
// test is a pure function of box and ray, so evaluating it once is exact).
static int new_guard(rt_scene &s, const Box &b, int parent) {
    if (parent >= 0 && box_eq(b, s.guards[parent])) return parent;
    s.guards.push_back(GuardD{b.lo[0], b.lo[1], b.lo[2], b.hi[0], b.hi[1], b.hi[2], parent, 0});
    return (int)s.guards.size() - 1;
}
static void assign_guards(rt_scene &s, int id, int parent) {
    const GNode &n = s.nodes[id];
    if (n.kind == 0) { s.leaves[n.leaf].guard = parent; return; }
    if (n.kind == 2) {                                      // shape1 <> shape2: bb = bb1 <> bb2, Main.hs:215
        const int g = new_guard(s, node_box(s, id), parent);
        assign_guards(s, n.kids[0], g);
        assign_guards(s, n.kids[1], g);
        return;
    }
    // mconcat: x_0 <> (x_1 <> ( ... (x_n-1 <> mempty))): child k sits below the nodes of suffixes 0..k
    std::vector<Box> sfx(n.kids.size() + 1, BOX_EMPTY);
    for (size_t k = n.kids.size(); k-- > 0;) sfx[k] = box_append(node_box(s, n.kids[k]), sfx[k + 1]);
    int cur = parent;
    for (size_t k = 0; k < n.kids.size(); ++k) {
        cur = new_guard(s, sfx[k], cur);
        assign_guards(s, n.kids[k], cur);
    }
}
// Renumbers the guard boxes so that the (at most) 64 boxes with the most leaves beneath them come first, parents
// before children: a wave of k_primary_mfma decides once, for its whole pixel tile, which of those every ray of the
// tile certainly passes (rt_kernels.hip: tile_guard_mask), and a leaf's chain walk stops at the first such box whose
// ancestors are all of that kind.  A box has at least as many leaves beneath it as any box below it, so the set is
// closed under "parent of".
static void order_guards(rt_scene &s) {
    const size_t n = s.guards.size();
    s.guard_cached = 0;
    if (n == 0) return;
    std::vector<long long> count(n, 0);
    std::vector<int> depth(n, 0);
    for (size_t g = 0; g < n; ++g) depth[g] = s.guards[g].parent < 0 ? 0 : depth[s.guards[g].parent] + 1;   // parents are created first
    for (const LeafD &L : s.leaves) for (int g = L.guard; g >= 0; g = s.guards[g].parent) ++count[g];
    std::vector<int> order(n);
    for (size_t g = 0; g < n; ++g) order[g] = (int)g;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        return count[a] != count[b] ? count[a] > count[b] : depth[a] < depth[b];
    });
    const size_t k = std::min<size_t>(64, n);
    std::vector<char> chosen(n, 0);
    size_t taken = 0;
    for (size_t i = 0; i < n && taken < k; ++i) {           // (a box whose parent was not taken is left out: keeps the set closed)
        const int g = order[i], p = s.guards[g].parent;
        if (p < 0 || chosen[p]) { chosen[g] = 1; ++taken; }
    }
    std::vector<int> renum(n, -1);
    int next = 0;
    for (size_t i = 0; i < n; ++i) if (chosen[order[i]]) renum[order[i]] = next++;
    s.guard_cached = next;
    for (size_t g = 0; g < n; ++g) if (!chosen[g]) renum[g] = next++;
    std::vector<GuardD> out(n);
    for (size_t g = 0; g < n; ++g) {
        GuardD B = s.guards[g];
        if (B.parent >= 0) B.parent = renum[B.parent];
        out[renum[g]] = B;
    }
    s.guards.swap(out);
    for (LeafD &L : s.leaves) if (L.guard >= 0) L.guard = renum[L.guard];
}
bool compute_guards(rt_scene &s) {
    if (!s.open.empty()) { set_error("rt_render: an rt_begin_mconcat/rt_begin_mappend bracket is still open"); return false; }
    if (s.cur_world != 0) { set_error("rt_render: an rt_begin_shadow_world is still open"); return false; }
    s.guards.clear();
    s.guard_cached = 0;
    if (s.talk) {                                         // Talk.hs:185-194: the monoid has no bounding boxes
        for (LeafD &L : s.leaves) L.guard = -1;
        return true;
    }
    auto one_world = [&](const std::vector<int> &top) {
        if (top.empty()) return;
        if (top.size() == 1) assign_guards(s, top[0], -1);
        else {
            s.nodes.push_back(GNode{1, -1, top});              // implicit top-level mconcat
            assign_guards(s, (int)s.nodes.size() - 1, -1);
            s.nodes.pop_back();
        }
    };
    one_world(s.top);
    for (const std::vector<int> &top : s.shadow_top) one_world(top);     // (every world is a tree of its own)
    if (s.guards.empty()) return true;
    order_guards(s);
    return true;
}

// General-ray record (reflected rays have no common apex).  With a = n.d, b = n.o + dpl, e_k = g_k.o + c_k,
// f_k = g_k.d the hit distance is t = -b/a and the inside coordinates are q_k = e_k + t f_k, so
// q_k * a = e_k a - b f_k and t * a = -b: the ray hits iff those all have the sign of a.
static GenRec make_gen_rec(const LeafD &L, const double omax[3]) {
    GenRec r{};
    const double n[3] = {L.nx, L.ny, L.nz};
    const double e1[3] = {L.e1x, L.e1y, L.e1z}, e2[3] = {L.e2x, L.e2y, L.e2z};
    double g1[3], g2[3], ref[3];
    if (L.kind == LEAF_TRI) {
        for (int i = 0; i < 3; ++i) {
            g1[i] = (L.c0 * e2[i] - L.c2 * e1[i]) / L.c3;
            g2[i] = (L.c0 * e1[i] - L.c1 * e2[i]) / L.c3;
        }
        ref[0] = L.p0x; ref[1] = L.p0y; ref[2] = L.p0z;
    } else {
        for (int i = 0; i < 3; ++i) { g1[i] = 2.0 * e1[i] / L.c0; g2[i] = 2.0 * e2[i] / L.c1; }
        ref[0] = L.p0x; ref[1] = L.p0y; ref[2] = L.p0z;
    }
    // triangle: q_k(p) = g_k.(p - pa); rectangle: q_k(p) = g_k.(p - point') - 1 in [-1, 1] (also for non-orthogonal edges)
    const double shift = L.kind == LEAF_TRI ? 0.0 : 1.0;
    const double c1 = -(g1[0] * ref[0] + g1[1] * ref[1] + g1[2] * ref[2]) - shift;
    const double c2 = -(g2[0] * ref[0] + g2[1] * ref[1] + g2[2] * ref[2]) - shift;
    double An = 0, Bn = std::fabs(L.d), E1 = std::fabs(c1), E2 = std::fabs(c2), F1 = 0, F2 = 0;
    for (int i = 0; i < 3; ++i) {
        An += std::fabs(n[i]); Bn += std::fabs(n[i]) * omax[i];
        E1 += std::fabs(g1[i]) * omax[i]; E2 += std::fabs(g2[i]) * omax[i];
        F1 += std::fabs(g1[i]); F2 += std::fabs(g2[i]);
    }
    const double m = std::ldexp(E1 * An + Bn * F1 + E2 * An + Bn * F2 + An + Bn * An + Bn, -19) + 1e-30;
    r.c[12] = 1.0f;                                        // never passes (z <= 0 for all-zero coefficients)
    if (!std::isfinite(m)) return r;
    for (int i = 0; i < 3; ++i) { r.c[i] = (float)n[i]; r.c[4 + i] = (float)g1[i]; r.c[8 + i] = (float)g2[i]; }
    r.c[3] = (float)L.d; r.c[7] = (float)c1; r.c[11] = (float)c2;
    r.c[12] = f_down(-m);
    // bounds of the FP32 error of a (for a unit direction) and of b: k_general's "behind the nearest hit so far" skip.
    // a = n . u^: the coefficients are rounded (2^-24 each), so are the components of u^ (the direction as a float, times
    // a rounded 1 / |d| that is itself 1 ulp off: 3 x 2^-24), and three fused steps add 3 x 2^-24 of the partial sums:
    // < 8 x 2^-24 An = 2^-21 An in all; b likewise with 6 x 2^-24 Bn.  Twice that is stored.
    r.c[13] = -f_down(-std::ldexp(An, -20));
    r.c[14] = -f_down(-(std::ldexp(Bn, -20) + 1e-30));
    for (int i = 0; i < 15; ++i)
        if (!std::isfinite(r.c[i])) { GenRec a{}; a.c[12] = -1.0f; return a; }   // overflow: always pass, exact path decides
    return r;
}

// Reflected rays have no common apex, so their leaf records are ordered in space instead: the leaves' centroids are cut
// top down at medians (after a first pass along a Morton curve) into blocks of 32 with a bounding sphere each (k_general skips the blocks no ray of a
// 64-ray run can reach).  Any order gives the same image: the nearest hit is the minimum over (t, leaf index).
//
// Stage 0 of the bounce levels, on the matrix cores (k_general): the line of a ray (o, unit d) meets the bounding sphere
// (c, r) of a leaf iff |c - q|^2 - (c.d)^2 <= r^2, where q = o - (o.d) d is the point of the line nearest the origin of
// the coordinates.  Taken relative to the centre C and in units of the radius R of a FRAME that holds the leaf (the
// bounding sphere of the leaf's super-block: 32 blocks = 1024 leaves; c~ = (c - C)/R, r~ = r/R, q~ likewise, all of size
// <= 1 for a ray that meets the leaf), the test is ONE linear form between a vector of the leaf and a vector of the ray,
//   F = (r~^2 - |c~|^2 + M) * 1 + 2 c~ . q~ - |q~|^2 + sum_ij c~_i c~_j d_i d_j  > 0,
// 11 products.  Every operand is split into an FP16 hi and mid part (22 significant bits) and the products hi*hi,
// hi*mid, mid*hi are kept: 31 FP16 products = two steps of v_mfma_f32_32x32x16_f16 (K = 32), which decide 32 leaves x
// 32 rays.  The ray's vector depends only on the frame, so it is built once per super-block, not per block.
// Slots (A = leaf value, B = ray value; X = 2 c~x .., Q = {c~x^2, c~y^2, c~z^2, 2 c~x c~y, 2 c~x c~z, 2 c~y c~z},
// D = {dx^2, dy^2, dz^2, dx dy, dx dz, dy dz}, k = r~^2 - |c~|^2 + M; every value times 2^10 so that no mid part is an FP16
// denormal the matrix core might flush):
//   step 0, K-half 0: A {k_h, k_m, -1, -1, X_h, X_h, X_m, Q5_h}   B {1, 1, qq_h, qq_m, qx_h, qx_m, qx_h, D5_m}
//   step 0, K-half 1: A {Y_h, Y_h, Y_m, Z_h, Z_h, Z_m, Q5_m, 0}   B {qy_h, qy_m, qy_h, qz_h, qz_m, qz_h, D5_h, 0}
//   step 1, K-half 0: A {Q0_h, Q0_h, Q0_m, Q1_h, Q1_h, Q1_m, Q2_h, Q2_h}   B {D0_h, D0_m, D0_h, D1_h, D1_m, D1_h, D2_h, D2_m}
//   step 1, K-half 1: A {Q2_m, Q3_h, Q3_h, Q3_m, Q4_h, Q4_h, Q4_m, Q5_h}   B {D2_h, D3_h, D3_m, D3_h, D4_h, D4_m, D4_h, D5_h}
// Margin M: for a ray that meets the leaf |q~| <= 1, so the absolute values of the 11 terms sum to <= 5; a split operand
// is off by <= 2^-22 of its value and the dropped mid*mid products by 2^-22: <= 5 * 3 * 2^-22 = 3.6e-6; FP32 accumulation
// of 32 products of a sum <= 5: <= 32 * 2^-24 * 5 = 9.6e-6; the ray's q is computed in FP32 from coordinates of size
// <= dall, an error of <= 2^-20 dall / R in q~, which moves F by <= 4 times that.  M = 2.5e-5 + 4 * 2^-20 dall / R.
// A far ray (|q~| large) has F ~ -|q~|^2, far more negative than any rounding error; q~ is clamped to +-4 in the kernel
// (a ray that meets the leaf has |q~| <= |c~| + r~ <= 2: leaves that do not fit that get the constant row "always").
constexpr double GEN_ROW_MARGIN = 2.5e-5, GEN_SCALE = 1024.0;
static void sphere_row(const double c[3], double r, const float S[4], double dall, bool real, uint16_t out[32]) {
    for (int k = 0; k < 32; ++k) out[k] = 0;
    const double R = (double)S[3];
    bool ok = real && std::isfinite(R) && R > 0 && std::isfinite(r) && std::isfinite(c[0]) && std::isfinite(c[1]) && std::isfinite(c[2]);
    double ct[3] = {0, 0, 0}, rt = 0, M = 0;
    if (ok) {
        for (int i = 0; i < 3; ++i) ct[i] = (c[i] - (double)S[i]) / R;
        rt = (r * (1.0 + 1e-6) + 1e-7 * dall) / R;
        M = GEN_ROW_MARGIN + 4.0 * std::ldexp(dall, -20) / R;
        ok = M < 0.25 && std::sqrt(ct[0] * ct[0] + ct[1] * ct[1] + ct[2] * ct[2]) + rt <= 2.0;
    }
    if (!ok) { out[0] = f16_rne((float)(real ? GEN_SCALE : -GEN_SCALE)); return; }     // a constant form: every ray (a real leaf) or none (padding)
    uint16_t kh, km, Xh[3], Xm[3], Qh[6], Qm[6];
    split_f16(GEN_SCALE * (rt * rt - (ct[0] * ct[0] + ct[1] * ct[1] + ct[2] * ct[2]) + M), kh, km);
    for (int i = 0; i < 3; ++i) split_f16(GEN_SCALE * 2 * ct[i], Xh[i], Xm[i]);
    const double Q[6] = {ct[0] * ct[0], ct[1] * ct[1], ct[2] * ct[2], 2 * ct[0] * ct[1], 2 * ct[0] * ct[2], 2 * ct[1] * ct[2]};
    for (int t = 0; t < 6; ++t) split_f16(GEN_SCALE * Q[t], Qh[t], Qm[t]);
    const uint16_t neg1 = f16_rne((float)-GEN_SCALE);
    const uint16_t s0h0[8] = {kh, km, neg1, neg1, Xh[0], Xh[0], Xm[0], Qh[5]};
    const uint16_t s0h1[8] = {Xh[1], Xh[1], Xm[1], Xh[2], Xh[2], Xm[2], Qm[5], 0};
    const uint16_t s1h0[8] = {Qh[0], Qh[0], Qm[0], Qh[1], Qh[1], Qm[1], Qh[2], Qh[2]};
    const uint16_t s1h1[8] = {Qm[2], Qh[3], Qh[3], Qm[3], Qh[4], Qh[4], Qm[4], Qh[5]};
    for (int k = 0; k < 8; ++k) { out[k] = s0h0[k]; out[8 + k] = s0h1[k]; out[16 + k] = s1h0[k]; out[24 + k] = s1h1[k]; }
}

static void spatial_order(const rt_scene &s, const std::vector<int32_t> &gids, const double lo[3], const double hi[3], double dall,
                          std::vector<int32_t> &order, std::vector<float> &spheres, std::vector<uint16_t> &rows,
                          std::vector<float> &sup_spheres) {
    const size_t n = gids.size();
    std::vector<std::pair<uint32_t, int32_t>> keyed(n);
    std::vector<int> nv(n);
    std::vector<double> verts(n * 12);
    for (size_t k = 0; k < n; ++k) {
        const LeafD &L = s.leaves[gids[k]];
        double v[4][3];
        nv[k] = leaf_polygon(L, v);
        double c[3] = {0, 0, 0};
        bool ok = nv[k] > 0;
        for (int j = 0; j < nv[k]; ++j) for (int i = 0; i < 3; ++i) { verts[k * 12 + j * 3 + i] = v[j][i]; c[i] += v[j][i] / nv[k]; ok = ok && std::isfinite(v[j][i]); }
        if (!ok) nv[k] = 0;
        uint32_t key = 0xffffffffu;                             // degenerate leaves last
        if (ok) {
            key = 0;
            for (int i = 0; i < 3; ++i) {
                const double ext = hi[i] - lo[i];
                double q = ext > 0 ? (c[i] - lo[i]) / ext : 0.0;
                q = q < 0 ? 0 : (q > 1 ? 1 : q);
                uint32_t u = (uint32_t)(q * 1023.0);
                u = (u | (u << 16)) & 0x030000ffu; u = (u | (u << 8)) & 0x0300f00fu; u = (u | (u << 4)) & 0x030c30c3u; u = (u | (u << 2)) & 0x09249249u;
                key |= u << i;
            }
        }
        keyed[k] = {key, (int32_t)k};
    }
    // leaves much larger than the typical one (walls, floors) would blow up the sphere of whatever block they fall into
    // and drag its 31 neighbours into every scan: they come first, in blocks of their own (but for the last of them)
    std::vector<double> rad(n, 0.0), cen(n * 3, 0.0);          // bounding sphere of every leaf: centroid of its vertices, farthest vertex
    {
        for (size_t k = 0; k < n; ++k) {
            double c[3] = {0, 0, 0}, r2 = 0;
            for (int j = 0; j < nv[k]; ++j) for (int i = 0; i < 3; ++i) c[i] += verts[k * 12 + j * 3 + i] / nv[k];
            for (int j = 0; j < nv[k]; ++j) { double d2 = 0; for (int i = 0; i < 3; ++i) { const double d = verts[k * 12 + j * 3 + i] - c[i]; d2 += d * d; } r2 = std::fmax(r2, d2); }
            rad[k] = std::sqrt(r2);
            for (int i = 0; i < 3; ++i) cen[k * 3 + i] = c[i];
        }
        std::vector<double> sorted(rad);
        std::sort(sorted.begin(), sorted.end());
        const double big = n ? 8.0 * sorted[n / 2] : 0.0;
        for (size_t k = 0; k < n; ++k)
            if (!(nv[k] > 0 && rad[k] > big) && keyed[k].first != 0xffffffffu) keyed[k].first |= 0x40000000u;   // (Morton keys use 30 bits)
    }
    std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<uint32_t, int32_t> &a, const std::pair<uint32_t, int32_t> &b) { return a.first < b.first; });
#ifndef RT_GEN_ORDER_MORTON
    // The ordinary leaves (after the big ones, before the degenerate ones) are then re-cut top down: a range is split at
    // the block boundary nearest its middle, along the longest axis of its centroids' box, until it is one block.  A
    // block of 32 consecutive Morton keys straddles cell boundaries of every level; the cuts follow the leaves instead,
    // and the blocks' spheres come out a third smaller in cross-section (what a run's line of sight meets is pi r^2).
    {
        size_t nb0 = 0, ne0 = n;
        while (nb0 < n && keyed[nb0].first < 0x40000000u) ++nb0;
        while (ne0 > nb0 && keyed[ne0 - 1].first == 0xffffffffu) --ne0;
        std::vector<std::pair<size_t, size_t>> todo(1, {nb0, ne0});
        while (!todo.empty()) {
            const size_t b0 = todo.back().first, e0 = todo.back().second;
            todo.pop_back();
            const size_t first = (b0 / 32 + 1) * 32;                 // block boundaries strictly inside (b0, e0)
            if (first >= e0) continue;
            const size_t inside = (e0 - 1 - first) / 32 + 1;
            const size_t mid = first + ((inside - 1) / 2) * 32;
            double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
            for (size_t k = b0; k < e0; ++k)
                for (int i = 0; i < 3; ++i) { const double v = cen[(size_t)keyed[k].second * 3 + i]; blo[i] = std::fmin(blo[i], v); bhi[i] = std::fmax(bhi[i], v); }
            int ax = 0;
            for (int i = 1; i < 3; ++i) if (bhi[i] - blo[i] > bhi[ax] - blo[ax]) ax = i;
            std::nth_element(keyed.begin() + (std::ptrdiff_t)b0, keyed.begin() + (std::ptrdiff_t)mid, keyed.begin() + (std::ptrdiff_t)e0,
                             [&](const std::pair<uint32_t, int32_t> &a, const std::pair<uint32_t, int32_t> &b) {
                                 const double va = cen[(size_t)a.second * 3 + ax], vb = cen[(size_t)b.second * 3 + ax];
                                 return va < vb || (va == vb && a.second < b.second);
                             });
            todo.push_back({b0, mid}); todo.push_back({mid, e0});
        }
    }
#endif
    order.resize(n);
    const size_t nblk = (n + 31) / 32;
    spheres.assign((nblk + 1) * 4, 0.f);
    for (size_t b = 0; b < nblk; ++b) {
        double c[3] = {0, 0, 0};
        size_t cnt = 0;
        bool always = false;
        for (size_t k = b * 32; k < std::min(n, b * 32 + 32); ++k) {
            const size_t src = (size_t)keyed[k].second;
            order[k] = gids[src];
            if (nv[src] == 0) { always = true; continue; }
            for (int j = 0; j < nv[src]; ++j) { for (int i = 0; i < 3; ++i) c[i] += verts[src * 12 + j * 3 + i]; ++cnt; }
        }
        float *S = &spheres[b * 4];
        if (always || cnt == 0) { S[3] = INFINITY; continue; }
        for (int i = 0; i < 3; ++i) c[i] /= (double)cnt;
        double r2 = 0;
        auto farthest = [&](const double *from, double *far) {
            double best = -1;
            for (size_t k = b * 32; k < std::min(n, b * 32 + 32); ++k) {
                const size_t src = (size_t)keyed[k].second;
                for (int j = 0; j < nv[src]; ++j) {
                    double d2 = 0;
                    for (int i = 0; i < 3; ++i) { const double d = verts[src * 12 + j * 3 + i] - from[i]; d2 += d * d; }
                    if (d2 > best) { best = d2; if (far) for (int i = 0; i < 3; ++i) far[i] = verts[src * 12 + j * 3 + i]; }
                }
            }
            return best;
        };
#ifndef RT_GEN_ORDER_MORTON
        // the centre walks towards the farthest vertex in ever smaller steps (Badoiu & Clarkson): close to the smallest
        // enclosing sphere; the radius below is measured from wherever it ends up, so the bound holds whatever it found
        {
            double best_c[3] = {c[0], c[1], c[2]}, best_r2 = farthest(c, nullptr);
            for (int it = 1; it <= 96; ++it) {
                double far[3];
                const double d2 = farthest(c, far);
                if (d2 < best_r2) { best_r2 = d2; for (int i = 0; i < 3; ++i) best_c[i] = c[i]; }
                for (int i = 0; i < 3; ++i) c[i] += (far[i] - c[i]) / (double)(it + 1);
            }
            if (farthest(c, nullptr) > best_r2) for (int i = 0; i < 3; ++i) c[i] = best_c[i];
            if (!(std::isfinite(c[0]) && std::isfinite(c[1]) && std::isfinite(c[2]))) for (int i = 0; i < 3; ++i) c[i] = best_c[i];
        }
#endif
        r2 = farthest(c, nullptr);
        // (a convex polygon lies inside any sphere that holds its vertices)  centre rounded to float: the radius pays for it
        const double r = std::sqrt(r2) * (1.0 + 1e-6) + 1e-6 * (std::fabs(c[0]) + std::fabs(c[1]) + std::fabs(c[2]) + 1.0);
        S[0] = (float)c[0]; S[1] = (float)c[1]; S[2] = (float)c[2];
        S[3] = std::nextafterf((float)r, INFINITY);
        if (!std::isfinite(S[0]) || !std::isfinite(S[1]) || !std::isfinite(S[2]) || !std::isfinite(S[3])) { S[0] = S[1] = S[2] = 0.f; S[3] = INFINITY; }
    }
    spheres[nblk * 4 + 3] = INFINITY;                           // (the padding block: never looked at)
    // 32 consecutive blocks make a super-block; its bounding sphere is the frame its leaves' rows are written in
    const size_t nsup = (nblk + 31) / 32;
    sup_spheres.assign((nsup + 1) * 4, 0.f);
    for (size_t sb = 0; sb <= nsup; ++sb) {
        float *S = &sup_spheres[sb * 4];
        S[3] = INFINITY;
        double c[3] = {0, 0, 0};
        size_t cnt = 0;
        for (size_t b = sb * 32; b < std::min(nblk, sb * 32 + 32); ++b)
            if (std::isfinite(spheres[b * 4 + 3])) { for (int i = 0; i < 3; ++i) c[i] += (double)spheres[b * 4 + i]; ++cnt; }
        if (!cnt) continue;
        for (int i = 0; i < 3; ++i) c[i] /= (double)cnt;
        double r = 0;
        for (size_t b = sb * 32; b < std::min(nblk, sb * 32 + 32); ++b) {
            if (!std::isfinite(spheres[b * 4 + 3])) continue;
            double d2 = 0;
            for (int i = 0; i < 3; ++i) { const double d = (double)spheres[b * 4 + i] - c[i]; d2 += d * d; }
            r = std::fmax(r, std::sqrt(d2) + (double)spheres[b * 4 + 3]);
        }
        r = r * (1.0 + 1e-6) + 1e-6 * (std::fabs(c[0]) + std::fabs(c[1]) + std::fabs(c[2]) + 1.0);
        S[0] = (float)c[0]; S[1] = (float)c[1]; S[2] = (float)c[2]; S[3] = std::nextafterf((float)r, INFINITY);
        if (!std::isfinite(S[0]) || !std::isfinite(S[1]) || !std::isfinite(S[2]) || !std::isfinite(S[3])) { S[0] = S[1] = S[2] = 0.f; S[3] = INFINITY; }
    }
    // the matrix-core rows of every block, in the frame of its super-block: lane (m, h) of a wave reads 16 bytes at
    // rows[(((b * 2 + step) * 2 + h) * 32 + m) * 8] for each of the two K steps
    rows.assign((nblk + 1) * 128 * 8, 0);
    for (size_t b = 0; b <= nblk; ++b)
        for (size_t m = 0; m < 32; ++m) {
            const size_t k = b * 32 + m;
            const bool real = b < nblk && k < n;
            const size_t src = real ? (size_t)keyed[k].second : 0;
            const double zero[3] = {0, 0, 0};
            uint16_t row[32];
            sphere_row(real && nv[src] > 0 ? &cen[src * 3] : zero, real && nv[src] > 0 ? rad[src] : INFINITY, &sup_spheres[(b / 32) * 4], dall, real, row);
            for (int st = 0; st < 2; ++st)
                for (int h = 0; h < 2; ++h)
                    for (int q = 0; q < 8; ++q) rows[(((b * 2 + st) * 2 + h) * 32 + m) * 8 + q] = row[st * 16 + h * 8 + q];
        }
}

void build_filter_records(rt_scene &s, int w, int h, bool eye_only) {
    (void)w; (void)h;
    const size_t ntri = s.tri_gid.size(), nrect = s.rect_gid.size(), nsph = s.sph_gid.size();
    const int napex = 1 + s.npoint;
    // scene extent (leaf corners), for the per-light bound on |D| = |hit - light|
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (const LeafD &L : s.leaves) {
        const double p[4][3] = {
            {L.p0x, L.p0y, L.p0z},
            {L.p0x + L.e1x, L.p0y + L.e1y, L.p0z + L.e1z},
            {L.p0x + L.e2x, L.p0y + L.e2y, L.p0z + L.e2z},
            {L.p0x + L.e1x + L.e2x, L.p0y + L.e1y + L.e2y, L.p0z + L.e1z + L.e2z}};
        if (L.kind == LEAF_SPHERE) {
            for (int i = 0; i < 3; ++i) {
                const double c = i == 0 ? L.p0x : (i == 1 ? L.p0y : L.p0z);
                if (std::isfinite(c) && std::isfinite(L.c0)) { lo[i] = std::fmin(lo[i], c - std::fabs(L.c0)); hi[i] = std::fmax(hi[i], c + std::fabs(L.c0)); }
            }
            continue;
        }
        const int np = L.kind == LEAF_TRI ? 3 : 4;
        for (int k = 0; k < np; ++k)
            for (int i = 0; i < 3; ++i)
                if (std::isfinite(p[k][i])) { lo[i] = std::fmin(lo[i], p[k][i]); hi[i] = std::fmax(hi[i], p[k][i]); }
        double v[4][3];                           // the parallelogram the inside test really accepts (non-orthogonal edges)
        const int nv = L.kind == LEAF_RECT ? leaf_polygon(L, v) : 0;
        for (int k = 0; k < nv; ++k)
            for (int i = 0; i < 3; ++i)
                if (std::isfinite(v[k][i])) { lo[i] = std::fmin(lo[i], v[k][i]); hi[i] = std::fmax(hi[i], v[k][i]); }
    }
    // MFMA stream: one row per planar leaf, 32 per block (see rt_types.h), in a per-apex order (direction_split), two
    // blocks of padding; live leaves first (padded to a whole block), then the "always" leaves, then the "never" ones
    std::vector<int32_t> planar;
    for (size_t g = 0; g < s.leaves.size(); ++g) if (s.leaves[g].kind != LEAF_SPHERE) planar.push_back((int32_t)g);
    const size_t nplanar = planar.size();
    s.mf_groups = (int)((nplanar + MF_LEAVES - 1) / MF_LEAVES) + 1;          // + 1: the padding behind the live leaves
    s.mf_groups = (s.mf_groups + 2) / 3 * 3;               // the scan handles three blocks per loop iteration (padding: never rows)
    const size_t gstride = (size_t)s.mf_groups + 2;
    if (!eye_only) {
        // four records of padding: the kernels prefetch up to record i + 2 while testing record i
        s.sph_rec.assign(nsph * (size_t)napex + 4, never_rec());
#ifdef RT_VALU_FILTER   // the per-type record blocks feed the vector-ALU kernels of the cross-check build only
        s.tri_rec.assign(ntri * (size_t)napex + 4, never_rec());
        s.rect_rec.assign(nrect * (size_t)napex + 4, never_rec());
#else
        s.tri_rec.assign(4, never_rec());
        s.rect_rec.assign(4, never_rec());
#endif
        s.mf_gid.assign(gstride * (size_t)napex * MF_LEAVES, -1);
        s.mf_rows.assign(gstride * (size_t)napex * 32 * 8, 0);
        s.mf_rec.assign(gstride * (size_t)napex * MF_LEAVES, never_rec());
        s.mf_cone.assign(gstride * (size_t)napex * 8, 0.f);
    }
    // an apex's slices start out as padding: never rows, never records, no leaf
    auto reset_apex = [&](int apex) {
        uint16_t never[8];
        const_row(never, -1.0f);
        FilterRec none = never_rec();
        const int32_t minus1 = -1;
        std::memcpy(&none.c[15], &minus1, 4);
        const size_t base = (size_t)apex * gstride * MF_LEAVES;
        for (size_t k = 0; k < gstride * MF_LEAVES; ++k) {
            s.mf_gid[base + k] = -1;
            s.mf_rec[base + k] = none;
            std::memcpy(&s.mf_rows[(base + k) * 8], never, 16);
        }
        for (size_t k = 0; k < gstride * 8; ++k) s.mf_cone[(size_t)apex * gstride * 8 + k] = 0.f;
    };
    // the apexes are independent of one another (disjoint slices of every array): one host thread each
    struct ApexJob { double A[3], dmax[3], eye_d; int apex, world; };
    std::vector<ApexJob> jobs;
    auto fill = [&](const double A[3], const double dmax[3], double eye_d, int apex, int world) {
        reset_apex(apex);
        std::vector<LeafCone> cone(nplanar);
        std::vector<double> cu(nplanar), cv(nplanar);
        std::vector<int> live, always, never;
        for (size_t k = 0; k < nplanar; ++k) {
            if (s.leaf_world[(size_t)planar[k]] != world) continue;      // not in this apex's world: stays padding (never a candidate)
            cone[k] = leaf_cone(s.leaves[planar[k]], A, dmax);
            if (cone[k].status == 0) {
                const double *a = cone[k].axis;
                const double l1 = std::fabs(a[0]) + std::fabs(a[1]) + std::fabs(a[2]);
                double ox = a[0] / l1, oy = a[1] / l1;
                if (a[2] > 0) {                             // the hemisphere behind the eye's view folds outwards
                    const double fx = (1.0 - std::fabs(oy)) * (ox >= 0 ? 1.0 : -1.0), fy = (1.0 - std::fabs(ox)) * (oy >= 0 ? 1.0 : -1.0);
                    ox = fx; oy = fy;
                }
                cu[k] = ox; cv[k] = oy;
                live.push_back((int)k);
            } else (cone[k].status == 2 ? always : never).push_back((int)k);
        }
        direction_split(live, cu, cv, 0, live.size());
        std::vector<int> order(live);
        order.resize((live.size() + MF_LEAVES - 1) / MF_LEAVES * MF_LEAVES, -1);
        order.insert(order.end(), always.begin(), always.end());
        order.insert(order.end(), never.begin(), never.end());
        const size_t base = (size_t)apex * gstride * MF_LEAVES;
        std::vector<LeafCone> ordered(gstride * MF_LEAVES, LeafCone{{0, 0, 0}, -2.0, 1});
        for (size_t k = 0; k < order.size(); ++k) {
            if (order[k] < 0) continue;
            const int32_t gid = planar[order[k]];
            const LeafD &L = s.leaves[gid];
            ordered[k] = cone[order[k]];
            s.mf_gid[base + k] = gid;
            make_cone_row(ordered[k], &s.mf_rows[(base + k) * 8]);
            // second stage of the matrix-core kernels: the leaf's FP32 record, found by stream position; the two
            // spare floats carry the leaf kind and index (bit patterns)
            FilterRec &r2 = s.mf_rec[base + k];
            r2 = make_rec(L, A, dmax, eye_d);
            std::memcpy(&r2.c[14], &L.kind, 4);
            std::memcpy(&r2.c[15], &gid, 4);
        }
        for (size_t g = 0; g < gstride; ++g) block_cone(&ordered[g * MF_LEAVES], MF_LEAVES, &s.mf_cone[((size_t)apex * gstride + g) * 8]);
#ifdef RT_VALU_FILTER
        for (size_t k = 0; k < ntri; ++k) s.tri_rec[(size_t)apex * ntri + k] = s.leaf_world[(size_t)s.tri_gid[k]] == world ? make_rec(s.leaves[s.tri_gid[k]], A, dmax, eye_d) : never_rec();
        for (size_t k = 0; k < nrect; ++k) s.rect_rec[(size_t)apex * nrect + k] = s.leaf_world[(size_t)s.rect_gid[k]] == world ? make_rec(s.leaves[s.rect_gid[k]], A, dmax, eye_d) : never_rec();
#endif
        for (size_t k = 0; k < nsph; ++k)
            s.sph_rec[(size_t)apex * nsph + k] = s.leaf_world[(size_t)s.sph_gid[k]] == world ? make_sphere_rec(s.leaves[s.sph_gid[k]], A, dmax) : never_rec();
    };
    auto add_job = [&](const double A[3], const double dmax[3], double eye_d, int world) {
        ApexJob j{};
        for (int i = 0; i < 3; ++i) { j.A[i] = A[i]; j.dmax[i] = dmax[i]; }
        j.eye_d = eye_d; j.apex = (int)jobs.size(); j.world = world;
        jobs.push_back(j);
        return j.apex;
    };
    {   // the eye: every primary ray o = (posX, posY, 0), D = (posX, posY, d) passes through o - D.  The apex does not
        // depend on the image size (d = -tan(fov/2)/2); only the bound on |posX|, |posY| does, through the margins.
        const CameraD cam = fixed_camera(1, 1);
        const double A[3] = {0.0, 0.0, -cam.d};
        add_job(A, s.eye_dmax, cam.d, 0);
    }
    if (!eye_only) {
        for (size_t li = 0; li < s.lights.size(); ++li) {
            LightD &Lt = s.lights[li];
            if (Lt.kind != 0) continue;
            const double A[3] = {Lt.px, Lt.py, Lt.pz};
            double dmax[3];
            for (int i = 0; i < 3; ++i) {
                const double a = std::fabs(lo[i] - A[i]), b = std::fabs(hi[i] - A[i]);
                dmax[i] = (s.leaves.empty() ? 1.0 : std::fmax(a, b)) * 1.001 + 1e-3;
            }
            Lt.rec_offset = add_job(A, dmax, 0.0, s.light_world[li]);          // the light's own world (pointLight world ..., Main.hs:467)
        }
    }
    {
        const size_t width = std::max<size_t>(1, std::min<size_t>(8, std::thread::hardware_concurrency()));
        for (size_t j0 = 0; j0 < jobs.size(); j0 += width) {
            std::vector<std::thread> pool;
            for (size_t j = j0 + 1; j < std::min(jobs.size(), j0 + width); ++j)
                pool.emplace_back([&, j] { fill(jobs[j].A, jobs[j].dmax, jobs[j].eye_d, jobs[j].apex, jobs[j].world); });
            fill(jobs[j0].A, jobs[j0].dmax, jobs[j0].eye_d, jobs[j0].apex, jobs[j0].world);
            for (std::thread &t : pool) t.join();
        }
    }
    if (!eye_only) {
        s.reflective = false;
        for (const MaterialD &m : s.mats) if (m.reflectivity > 0) s.reflective = true;
        s.gen_tri.clear(); s.gen_rect.clear(); s.gen_sph.clear();
        if (s.reflective) {
            double omax[3];
            for (int i = 0; i < 3; ++i) omax[i] = (s.leaves.empty() ? 1.0 : std::fmax(std::fabs(lo[i]), std::fabs(hi[i]))) * 1.001 + 1e-3;
            GenRec never{}; never.c[12] = 1.0f;
            const double dall = 2.0 * std::sqrt(omax[0] * omax[0] + omax[1] * omax[1] + omax[2] * omax[2]);   // |o - C| <= dall for points of the scene
            // records padded to whole blocks (+ one never block): k_general stages the 32 records of a block with one load
            const size_t tpad = ((ntri + 31) / 32 + 1) * 32, rpad = ((nrect + 31) / 32 + 1) * 32;
            s.gen_tri.assign(tpad, never); s.gen_rect.assign(rpad, never);
            spatial_order(s, s.tri_gid, lo, hi, dall, s.gen_tri_gid, s.gen_tri_sph, s.gen_tri_rows, s.gen_tri_ssph);
            spatial_order(s, s.rect_gid, lo, hi, dall, s.gen_rect_gid, s.gen_rect_sph, s.gen_rect_rows, s.gen_rect_ssph);
            auto with_gid = [](GenRec r, int32_t gid) { std::memcpy(&r.c[15], &gid, 4); return r; };       // the spare float carries the leaf index
            // (reflected rays meet the traced world only: leaves of shadow worlds keep the never record)
            for (size_t k = 0; k < ntri; ++k) if (s.leaf_world[(size_t)s.gen_tri_gid[k]] == 0) s.gen_tri[k] = with_gid(make_gen_rec(s.leaves[s.gen_tri_gid[k]], omax), s.gen_tri_gid[k]);
            for (size_t k = 0; k < nrect; ++k) if (s.leaf_world[(size_t)s.gen_rect_gid[k]] == 0) s.gen_rect[k] = with_gid(make_gen_rec(s.leaves[s.gen_rect_gid[k]], omax), s.gen_rect_gid[k]);
            s.gen_sph.assign(nsph + 1, never);
            for (size_t k = 0; k < nsph; ++k) if (s.leaf_world[(size_t)s.sph_gid[k]] == 0) s.gen_sph[k] = make_sphere_gen_rec(s.leaves[s.sph_gid[k]], omax);
        }
    }
}

// The eye-apex records hold for every camera whose |posX|, |posY| stay inside eye_dmax (half a pixel of jitter and a
// pixel of slack included), so most changes of the image size rebuild nothing; a scene change rebuilds everything.
bool prepare_host(rt_scene &s, int w, int h) {
    const CameraD cam = fixed_camera(w, h);
    const double need[3] = {std::fabs(cam.dX) + 2 * std::fabs(cam.scaleX), std::fabs(cam.dY) + 2 * std::fabs(cam.scaleY), std::fabs(cam.d)};
    bool grow = false;
    for (int i = 0; i < 3; ++i) if (!(need[i] <= s.eye_dmax[i])) grow = true;
    s.rec_w = w; s.rec_h = h;
    if (!s.dirty && !grow) return true;
    const auto t0 = std::chrono::steady_clock::now();
    // built for every image at least 16 pixels wide and no taller than wide; narrower or taller ones widen the bound
    const double floor_[3] = {0.625, 0.625, 0.5};
    if (grow) for (int i = 0; i < 3; ++i) s.eye_dmax[i] = std::fmax(std::fmax(need[i], floor_[i]), s.eye_dmax[i]);
    if (s.dirty) {
        if (!compute_guards(s)) return false;
        build_filter_records(s, w, h, false);
        ++s.version; ++s.eye_version;
        s.dirty = false;
    } else {
        build_filter_records(s, w, h, true);
        ++s.eye_version;
    }
    s.prepare_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return true;
}

// Are the records on the host valid for a w x h render as they stand (nothing to rebuild)?
bool host_ready(const rt_scene &s, int w, int h) {
    if (s.dirty || s.version == 0) return false;
    const CameraD cam = fixed_camera(w, h);
    const double need[3] = {std::fabs(cam.dX) + 2 * std::fabs(cam.scaleX), std::fabs(cam.dY) + 2 * std::fabs(cam.scaleY), std::fabs(cam.d)};
    for (int i = 0; i < 3; ++i) if (!(need[i] <= s.eye_dmax[i])) return false;
    return true;
}

// C port of raytrace_amd/distributed.py: split_rows (the two must agree; tests/test_host_scene.py compares them).
void split_rows(const double *cost_in, int ncost, int h, int n, int align, int *row0, int *nrows, const double *weights) {
    std::vector<double> cost(cost_in, cost_in + ncost);
    double tot = 0;
    for (double c : cost) tot += c;
    if (!(tot > 0)) { cost.assign(ncost, 1.0); }
    std::vector<double> cum(ncost + 1, 0.0);
    for (int i = 0; i < ncost; ++i) cum[i + 1] = cum[i] + cost[i];
    const double total = cum[ncost];
    for (double &c : cum) c /= total;
    std::vector<double> target(n + 1, 0.0);               // cumulative share of the cost at every cut
    if (weights) {
        double wsum = 0, run = 0;
        for (int k = 0; k < n; ++k) wsum += weights[k];
        for (int k = 0; k < n; ++k) { run += weights[k]; target[k + 1] = run / wsum; }
    } else for (int k = 0; k <= n; ++k) target[k] = (double)k / (double)n;
    std::vector<int> cuts(1, 0);
    for (int k = 1; k < n; ++k) {
        const double x = target[k];
        // numpy.interp(x, cum, 0..ncost): the last j with cum[j] <= x, linear inside [cum[j], cum[j+1]]
        double pos;
        if (x >= cum[ncost]) pos = (double)ncost;
        else {
            int j = (int)(std::upper_bound(cum.begin(), cum.end(), x) - cum.begin()) - 1;
            if (j < 0) j = 0;
            const double slope = 1.0 / (cum[j + 1] - cum[j]);
            pos = slope * (x - cum[j]) + (double)j;
        }
        long long r = (long long)std::nearbyint(pos * (double)h / (double)ncost);      // Python's round: half to even
        r = (r / align) * align;
        const long long lo = std::min<long long>(h, (long long)cuts.back() + align);
        const long long hi = std::max<long long>(lo, (long long)h - (long long)(n - k) * align);
        cuts.push_back((int)std::min(std::max(r, lo), hi));
    }
    cuts.push_back(h);
    for (int i = 0; i < n; ++i) { row0[i] = cuts[i]; nrows[i] = cuts[i + 1] - cuts[i]; }
}

} // namespace rt

using namespace rt;

// ------------------------------------------------------------------------------------------
// C ABI: construction and introspection (no device needed)

extern "C" {

const char *rt_last_error(void) { return g_err.c_str(); }
const char *rt_version(void) {
    static const std::string v = std::string("raytrace_amd 0.3 (gfx950)")
#ifdef RT_VALU_FILTER
        + " cross-check build (also holds the FP32 vector-ALU kernels)"
#endif
        + rt::kernel_build_flags();
    return v.c_str();
}

int rt_scene_create(rt_scene **out) {
    if (!out) { set_error("rt_scene_create: out is NULL"); return RT_ERR_ARG; }
    *out = new (std::nothrow) rt_scene();
    if (!*out) { set_error("rt_scene_create: out of memory"); return RT_ERR_NOMEM; }
    return RT_OK;
}

int rt_scene_set_option(rt_scene *s, int option, int value) {
    if (!s) { set_error("rt_scene_set_option: NULL scene"); return RT_ERR_ARG; }
    switch (option) {
    case RT_OPT_CULL: s->opt_cull = value != 0; return RT_OK;
    case RT_OPT_FILTER:
        if (value != 0 && value != 1) { set_error("rt_scene_set_option: RT_OPT_FILTER takes 0 (matrix cores) or 1 (vector ALU)"); return RT_ERR_ARG; }
#ifndef RT_VALU_FILTER
        if (value == 1) { set_error("rt_scene_set_option: the FP32 vector-ALU kernels exist only in the cross-check build libraytrace_gpu_valu.so"); return RT_ERR_UNSUPPORTED; }
#endif
        s->opt_filter = value; return RT_OK;
    case RT_OPT_HOST_BANDS:
        if (value < 0) { set_error("rt_scene_set_option: RT_OPT_HOST_BANDS must be >= 0"); return RT_ERR_ARG; }
        s->opt_host_bands = value; return RT_OK;
    case RT_OPT_MULTI_STAGE: s->opt_multi_stage = value != 0; return RT_OK;
    case RT_OPT_LANE:
        if (value < -1 || value >= rt::RT_LANES) { set_error("rt_scene_set_option: RT_OPT_LANE takes -1 .. " + std::to_string(rt::RT_LANES - 1)); return RT_ERR_ARG; }
        s->opt_lane = value; return RT_OK;
    case RT_OPT_CONTINUE: s->opt_continue = value != 0; return RT_OK;
    case RT_OPT_HISTORY: s->opt_history = value != 0; return RT_OK;
    case RT_OPT_RGB24: s->opt_rgb24 = value != 0; return RT_OK;
    }
    set_error("rt_scene_set_option: unknown option");
    return RT_ERR_ARG;
}

int rt_split_rows(const double *cost, int ncost, int h, int n, int align, int *row0, int *nrows) {
    return rt_split_rows_weighted(cost, ncost, h, n, align, nullptr, row0, nrows);
}

int rt_split_rows_weighted(const double *cost, int ncost, int h, int n, int align, const double *weights, int *row0, int *nrows) {
    if (!cost || !row0 || !nrows || ncost < 1 || h < 0 || n < 1 || align < 1) { set_error("rt_split_rows: bad argument"); return RT_ERR_ARG; }
    for (int i = 0; i < ncost; ++i) if (!(cost[i] >= 0)) { set_error("rt_split_rows: cost must be non-negative"); return RT_ERR_ARG; }
    if (weights) for (int k = 0; k < n; ++k) if (!(weights[k] > 0)) { set_error("rt_split_rows: weights must be positive"); return RT_ERR_ARG; }
    split_rows(cost, ncost, h, n, align, row0, nrows, weights);
    return RT_OK;
}

// Talk.hs dialect (before the first shape).  Talk.hs has rectangles (and `cube`, six of them) only.
int rt_scene_set_dialect(rt_scene *s, int dialect) {
    if (!s || (dialect != RT_DIALECT_MAIN && dialect != RT_DIALECT_TALK)) { set_error("rt_scene_set_dialect: bad argument"); return RT_ERR_ARG; }
    if (!s->leaves.empty() || !s->nodes.empty()) { set_error("rt_scene_set_dialect: must be called before the first shape"); return RT_ERR_ARG; }
    s->talk = dialect == RT_DIALECT_TALK;
    s->dirty = true;
    return RT_OK;
}
#define RT_TALK_ONLY_RECT(s, what) do { if ((s)->talk) { set_error(what ": the Talk.hs dialect has rectangles only"); return RT_ERR_UNSUPPORTED; } } while (0)

int rt_add_rectangle(rt_scene *s, const rt_material *m, const double point[3],
                     const double width[3], const double height[3]) {
    if (!s || !m || !point || !width || !height) { set_error("rt_add_rectangle: NULL argument"); return RT_ERR_ARG; }
    if (!world_accepts_shapes(*s, "rt_add_rectangle")) return RT_ERR_ARG;
    return add_rect(*s, *m, from(point), from(width), from(height));
}

int rt_add_triangle(rt_scene *s, const rt_material *m, const double pa[3], const double pb[3],
                    const double pc[3]) {
    if (!s || !m || !pa || !pb || !pc) { set_error("rt_add_triangle: NULL argument"); return RT_ERR_ARG; }
    if (!world_accepts_shapes(*s, "rt_add_triangle")) return RT_ERR_ARG;
    RT_TALK_ONLY_RECT(s, "rt_add_triangle");
    return add_tri(*s, *m, from(pa), from(pb), from(pc));
}

int rt_add_sphere(rt_scene *s, const rt_material *m, const double center[3], double radius) {
    if (!s || !m || !center) { set_error("rt_add_sphere: NULL argument"); return RT_ERR_ARG; }
    if (!world_accepts_shapes(*s, "rt_add_sphere")) return RT_ERR_ARG;
    RT_TALK_ONLY_RECT(s, "rt_add_sphere");
    return add_sphere(*s, *m, from(center), radius);
}

// colorcuboid materials point l h d — Main.hs:301-320: six rectangles in the order
// top, bottom, front, back, left(+x), right(-x), materials cycled over that order.
int rt_add_cuboid(rt_scene *s, const rt_material *mats, int nmats, const double point[3],
                  double l, double h, double d) {
    if (!s || !mats || !point) { set_error("rt_add_cuboid: NULL argument"); return RT_ERR_ARG; }
    if (!world_accepts_shapes(*s, "rt_add_cuboid")) return RT_ERR_ARG;
    if (nmats <= 0) { set_error("colorcuboid: list of materials must not be empty."); return RT_ERR_ARG; } // :302
    RT_TALK_ONLY_RECT(s, "rt_add_cuboid");      // Talk.hs's cube has other faces (Talk.hs:197-218): build it from rectangles
    const D3 p = from(point);
    const double l2 = l / 2, h2 = h / 2, d2 = d / 2;
    open_group(*s, 1);                                                                // mconcat [...], :308
    const int first = add_rect(*s, mats[0 % nmats], p + D3{0, h2, 0}, D3{l, 0, 0}, D3{0, 0, -d});
    add_rect(*s, mats[1 % nmats], p - D3{0, h2, 0}, D3{l, 0, 0}, D3{0, 0, d});
    add_rect(*s, mats[2 % nmats], p + D3{0, 0, d2}, D3{l, 0, 0}, D3{0, h, 0});
    add_rect(*s, mats[3 % nmats], p - D3{0, 0, d2}, D3{l, 0, 0}, D3{0, -h, 0});
    add_rect(*s, mats[4 % nmats], p + D3{l2, 0, 0}, D3{0, h, 0}, D3{0, 0, d});
    add_rect(*s, mats[5 % nmats], p - D3{l2, 0, 0}, D3{0, h, 0}, D3{0, 0, -d});
    close_group(*s, 1, "rt_add_cuboid");
    return first;
}

// cylinder topM botM mantleM point n h r — Main.hs:725-774.  Leaves: n bottom triangles, n top
// triangles, n mantle rectangles (`bot <> top <> mantle`).  Reproduced quirks: botPoint equals
// topPoint (:735), mantle normals are lerped and not renormalised (:769-771).
int rt_add_cylinder(rt_scene *s, const rt_material *top, const rt_material *bot,
                    const rt_material *mantle, const double point[3], int n, double h, double r) {
    if (!s || !top || !bot || !mantle || !point) { set_error("rt_add_cylinder: NULL argument"); return RT_ERR_ARG; }
    if (!world_accepts_shapes(*s, "rt_add_cylinder")) return RT_ERR_ARG;
    if (n < 1) { set_error("rt_add_cylinder: n must be >= 1"); return RT_ERR_ARG; }
    RT_TALK_ONLY_RECT(s, "rt_add_cylinder");
    const D3 c = from(point);
    const double nn = (double)n, alpha = 2 * M_PI / nn, h2 = h / 2;
    std::vector<double> rx((size_t)n + 1), rz((size_t)n + 1);
    for (int i = 0; i < n; ++i) { const double fi = (double)i; rx[i] = r * std::cos(fi * alpha); rz[i] = r * std::sin(fi * alpha); }
    rx[n] = r; rz[n] = 0;                                                             // :729
    const D3 topPoint = c + D3{0, h2, 0};                                             // :734
    const D3 botPoint = c - D3{0, -h2, 0};                                            // :735 (sic)
    const int first = (int)s->leaves.size();
    open_group(*s, 2); open_group(*s, 2);                                             // (bot <> top) <> mantle, :774
    open_group(*s, 1);
    for (int i = 0; i < n; ++i)                                                       // bot, :747-751
        add_tri(*s, *bot, botPoint, c + D3{rx[i], -h2, rz[i]}, c + D3{rx[i + 1], -h2, rz[i + 1]});
    close_group(*s, 1, "rt_add_cylinder");
    open_group(*s, 1);
    for (int i = 0; i < n; ++i)                                                       // top, :742-746
        add_tri(*s, *top, topPoint, c + D3{rx[i + 1], h2, rz[i + 1]}, c + D3{rx[i], h2, rz[i]});
    close_group(*s, 1, "rt_add_cylinder");
    close_group(*s, 2, "rt_add_cylinder");
    open_group(*s, 1);
    for (int i = 0; i < n; ++i) {                                                     // mantle, :752-773
        const D3 p1 = c + D3{rx[i], 0, rz[i]}, p2 = c + D3{rx[i + 1], 0, rz[i + 1]};
        const D3 n1 = {rx[i] / r, 0, rz[i] / r}, n2 = {rx[i + 1] / r, 0, rz[i + 1] / r};
        const D3 mid = scale(0.5, p1 + p2), dP = p1 - p2, dN = n1 - n2;
        const int leaf = add_rect(*s, *mantle, mid, dP, D3{0, h, 0});
        AuxD a{};
        a.a[0] = n2.x; a.a[1] = n2.y; a.a[2] = n2.z;
        a.a[3] = p2.x; a.a[4] = p2.y; a.a[5] = p2.z;
        a.a[6] = dN.x / dP.x; a.a[7] = dN.y / dP.y; a.a[8] = dN.z / dP.z;             // :764
        s->aux.push_back(a);
        s->leaves[leaf].nmode = RT_NORMAL_MANTLE;
        s->leaves[leaf].aux = (int)s->aux.size() - 1;
    }
    close_group(*s, 1, "rt_add_cylinder");
    close_group(*s, 2, "rt_add_cylinder");
    return first;
}

int rt_begin_mconcat(rt_scene *s) { if (!s) { set_error("rt_begin_mconcat: NULL scene"); return RT_ERR_ARG; } return open_group(*s, 1); }
int rt_end_mconcat(rt_scene *s) { if (!s) { set_error("rt_end_mconcat: NULL scene"); return RT_ERR_ARG; } return close_group(*s, 1, "rt_end_mconcat"); }
int rt_begin_mappend(rt_scene *s) { if (!s) { set_error("rt_begin_mappend: NULL scene"); return RT_ERR_ARG; } return open_group(*s, 2); }
int rt_end_mappend(rt_scene *s) { if (!s) { set_error("rt_end_mappend: NULL scene"); return RT_ERR_ARG; } return close_group(*s, 2, "rt_end_mappend"); }

int rt_set_normal_mode(rt_scene *s, int first_leaf, int nleaves, int mode, const double *params, int nparams) {
    if (!s || first_leaf < 0 || nleaves < 0 || (size_t)first_leaf + (size_t)nleaves > s->leaves.size()) {
        set_error("rt_set_normal_mode: leaf range out of bounds"); return RT_ERR_ARG;
    }
    int aux = -1;
    if (mode == RT_NORMAL_GAUSSIAN) {
        if (!params || nparams != 4) { set_error("rt_set_normal_mode: gaussian mode takes {scaleInv, ox, oy, oz}"); return RT_ERR_ARG; }
        AuxD a{};
        for (int i = 0; i < 4; ++i) a.a[i] = params[i];
        s->aux.push_back(a);
        aux = (int)s->aux.size() - 1;
    } else if (mode != RT_NORMAL_GEOMETRIC) {
        set_error("rt_set_normal_mode: mode must be RT_NORMAL_GEOMETRIC or RT_NORMAL_GAUSSIAN"); return RT_ERR_ARG;
    }
    for (int i = first_leaf; i < first_leaf + nleaves; ++i) { s->leaves[i].nmode = mode; s->leaves[i].aux = aux; }
    s->dirty = true;
    return RT_OK;
}

int rt_set_normal_callback(rt_scene *s, int first_leaf, int nleaves, const double params[4], rt_normal_fn fn, void *user) {
    if (!s || first_leaf < 0 || nleaves < 0 || (size_t)first_leaf + (size_t)nleaves > s->leaves.size()) {
        set_error("rt_set_normal_callback: leaf range out of bounds"); return RT_ERR_ARG;
    }
    if (!params || !fn) { set_error("rt_set_normal_callback: NULL argument"); return RT_ERR_ARG; }
    AuxD a{};
    for (int i = 0; i < 4; ++i) a.a[i] = params[i];
    a.a[4] = (double)s->normal_fn.size();                 // which callback
    s->normal_fn.push_back(fn);
    s->normal_user.push_back(user);
    s->aux.push_back(a);
    const int aux = (int)s->aux.size() - 1;
    for (int i = first_leaf; i < first_leaf + nleaves; ++i) { s->leaves[i].nmode = RT_NORMAL_CALLBACK; s->leaves[i].aux = aux; }
    s->dirty = true;
    return RT_OK;
}

int rt_add_point_light(rt_scene *s, double diffuse, double specular, const double pos[3]) {
    if (!s || !pos) { set_error("rt_add_point_light: NULL argument"); return RT_ERR_ARG; }
    LightD L{};
    L.kind = 0; L.px = pos[0]; L.py = pos[1]; L.pz = pos[2];
    L.diffuse = diffuse; L.specular = specular; L.rec_offset = -1;
    s->lights.push_back(L);
    s->light_world.push_back(0);
    s->npoint++;
    s->dirty = true;
    return (int)s->lights.size() - 1;
}

// pointLight world' kd ks pos with a world' of its own (Main.hs:467-474: `intersect world shadowRay` is the light's closure)
int rt_add_point_light_world(rt_scene *s, int world, double diffuse, double specular, const double pos[3]) {
    if (!s || !pos) { set_error("rt_add_point_light_world: NULL argument"); return RT_ERR_ARG; }
    if (world < 0 || (size_t)world > s->shadow_top.size()) { set_error("rt_add_point_light_world: no such world"); return RT_ERR_ARG; }
    const int id = rt_add_point_light(s, diffuse, specular, pos);
    if (id >= 0) s->light_world[(size_t)id] = world;
    return id;
}

int rt_begin_shadow_world(rt_scene *s) {
    if (!s) { set_error("rt_begin_shadow_world: NULL scene"); return RT_ERR_ARG; }
    if (!s->open.empty() || s->cur_world != 0) { set_error("rt_begin_shadow_world: inside an open bracket or another shadow world"); return RT_ERR_ARG; }
    s->shadow_top.emplace_back();
    s->cur_world = (int)s->shadow_top.size();
    s->dirty = true;
    return s->cur_world;
}
int rt_end_shadow_world(rt_scene *s) {
    if (!s) { set_error("rt_end_shadow_world: NULL scene"); return RT_ERR_ARG; }
    if (s->cur_world == 0 || !s->open.empty()) { set_error("rt_end_shadow_world: no shadow world is open, or a bracket inside it still is"); return RT_ERR_ARG; }
    s->cur_world = 0;
    return RT_OK;
}

int rt_add_ambient(rt_scene *s, double f) {
    if (!s) { set_error("rt_add_ambient: NULL scene"); return RT_ERR_ARG; }
    LightD L{};
    L.kind = 1; L.f = f; L.rec_offset = -1;
    s->lights.push_back(L);
    s->light_world.push_back(0);
    s->dirty = true;
    return (int)s->lights.size() - 1;
}

int rt_scene_num_leaves(const rt_scene *s) { return s ? (int)s->leaves.size() : RT_ERR_ARG; }
int rt_scene_num_lights(const rt_scene *s) { return s ? (int)s->lights.size() : RT_ERR_ARG; }

int rt_scene_get_leaf(const rt_scene *s, int leaf, double out[20]) {
    if (!s || !out || leaf < 0 || (size_t)leaf >= s->leaves.size()) { set_error("rt_scene_get_leaf: bad argument"); return RT_ERR_ARG; }
    const LeafD &L = s->leaves[leaf];
    const double v[20] = {(double)L.kind, (double)L.nmode, L.nx, L.ny, L.nz, L.d, L.p0x, L.p0y, L.p0z,
                          L.e1x, L.e1y, L.e1z, L.e2x, L.e2y, L.e2z, L.c0, L.c1, L.c2, L.c3, (double)L.mat};
    std::memcpy(out, v, sizeof v);
    return RT_OK;
}

int rt_scene_prepare_host(rt_scene *s, int w, int h) {
    if (!s || w <= 0 || h <= 0) { set_error("rt_scene_prepare_host: bad argument"); return RT_ERR_ARG; }
    if (!prepare_host(*s, w, h)) return RT_ERR_ARG;
    return (int)s->guards.size();
}

int rt_scene_get_filter_record(const rt_scene *s, int apex, int leaf, float out[16]) {
    if (!s || !out || leaf < 0 || (size_t)leaf >= s->leaves.size() || apex < 0 || apex > s->npoint || s->rec_w < 0) {
        set_error("rt_scene_get_filter_record: bad argument (call rt_scene_prepare_host first)"); return RT_ERR_ARG;
    }
    const LeafD &L = s->leaves[leaf];
    if (L.kind == LEAF_SPHERE) {
        std::memcpy(out, s->sph_rec[(size_t)apex * s->sph_gid.size() + (size_t)L.slot].c, sizeof(float) * 16);
        return RT_OK;
    }
    // a planar leaf's record sits in the apex's stream, at the position of its matrix-core row
    const size_t gstride = (size_t)s->mf_groups + 2, base = (size_t)apex * gstride * MF_LEAVES;
    for (size_t k = 0; k < gstride * MF_LEAVES; ++k)
        if (s->mf_gid[base + k] == leaf) {
            std::memcpy(out, s->mf_rec[base + k].c, sizeof(float) * 16);
            out[14] = out[15] = 0.f;                               // (the stream's copy carries kind and index there)
            return RT_OK;
        }
    set_error("rt_scene_get_filter_record: the leaf has no record");
    return RT_ERR_ARG;
}

int rt_scene_get_mfma_row(const rt_scene *s, int apex, int leaf, uint16_t out[8]) {
    if (!s || !out || leaf < 0 || (size_t)leaf >= s->leaves.size() || apex < 0 || apex > s->npoint || s->rec_w < 0) {
        set_error("rt_scene_get_mfma_row: bad argument (call rt_scene_prepare_host first)"); return RT_ERR_ARG;
    }
    const size_t gstride = (size_t)s->mf_groups + 2, base = (size_t)apex * gstride * MF_LEAVES;
    for (size_t k = 0; k < gstride * MF_LEAVES; ++k)
        if (s->mf_gid[base + k] == leaf) {
            std::memcpy(out, &s->mf_rows[(base + k) * 8], 16);
            return (int)k;
        }
    set_error("rt_scene_get_mfma_row: the leaf has no row (spheres are tested by their own loop)");
    return RT_ERR_ARG;
}

int rt_scene_get_block_cone(const rt_scene *s, int apex, int block, float out[8]) {
    if (!s || !out || apex < 0 || apex > s->npoint || s->rec_w < 0 || block < 0 || block >= s->mf_groups + 2) {
        set_error("rt_scene_get_block_cone: bad argument (call rt_scene_prepare_host first)"); return RT_ERR_ARG;
    }
    std::memcpy(out, &s->mf_cone[((size_t)apex * ((size_t)s->mf_groups + 2) + (size_t)block) * 8], 32);
    return s->mf_groups;
}

int rt_scene_get_general_block(const rt_scene *s, int kind, int block, float sphere[4], int32_t leaves[32]) {
    if (!s || !sphere || !leaves || (kind != LEAF_RECT && kind != LEAF_TRI) || s->rec_w < 0) {
        set_error("rt_scene_get_general_block: bad argument (call rt_scene_prepare_host first)"); return RT_ERR_ARG;
    }
    const std::vector<int32_t> &gid = kind == LEAF_TRI ? s->gen_tri_gid : s->gen_rect_gid;
    const std::vector<float> &sph = kind == LEAF_TRI ? s->gen_tri_sph : s->gen_rect_sph;
    const int nblk = (int)((gid.size() + 31) / 32);
    if (!s->reflective) return 0;
    if (block < 0 || block >= nblk) { set_error("rt_scene_get_general_block: block out of range"); return RT_ERR_ARG; }
    for (int k = 0; k < 4; ++k) sphere[k] = sph[(size_t)block * 4 + k];
    for (int k = 0; k < 32; ++k) { const size_t i = (size_t)block * 32 + k; leaves[k] = i < gid.size() ? gid[i] : -1; }
    return nblk;
}

int rt_scene_get_general_rows(const rt_scene *s, int kind, int block, float frame[4], uint16_t out[1024]) {
    if (!s || !out || !frame || (kind != LEAF_RECT && kind != LEAF_TRI) || s->rec_w < 0) {
        set_error("rt_scene_get_general_rows: bad argument (call rt_scene_prepare_host first)"); return RT_ERR_ARG;
    }
    const std::vector<int32_t> &gid = kind == LEAF_TRI ? s->gen_tri_gid : s->gen_rect_gid;
    const std::vector<uint16_t> &rows = kind == LEAF_TRI ? s->gen_tri_rows : s->gen_rect_rows;
    const std::vector<float> &ssph = kind == LEAF_TRI ? s->gen_tri_ssph : s->gen_rect_ssph;
    const int nblk = (int)((gid.size() + 31) / 32);
    if (!s->reflective) return 0;
    if (block < 0 || block >= nblk) { set_error("rt_scene_get_general_rows: block out of range"); return RT_ERR_ARG; }
    for (int k = 0; k < 4; ++k) frame[k] = ssph[(size_t)(block / 32) * 4 + k];
    for (int m = 0; m < 32; ++m)
        for (int st = 0; st < 2; ++st)
            for (int h = 0; h < 2; ++h)
                for (int k = 0; k < 8; ++k) out[m * 32 + st * 16 + h * 8 + k] = rows[((((size_t)block * 2 + st) * 2 + h) * 32 + m) * 8 + k];
    return nblk;
}

int rt_scene_get_guards(const rt_scene *s, int leaf, double *out, int cap) {
    if (!s || leaf < 0 || (size_t)leaf >= s->leaves.size()) { set_error("rt_scene_get_guards: bad argument"); return RT_ERR_ARG; }
    int n = 0;
    for (int g = s->leaves[leaf].guard; g >= 0 && (size_t)g < s->guards.size(); g = s->guards[g].parent, ++n)
        if (out && n < cap) {
            const GuardD &G = s->guards[g];
            const double b[6] = {G.x0, G.y0, G.z0, G.x1, G.y1, G.z1};
            std::memcpy(out + 6 * n, b, sizeof b);
        }
    return n;
}

int rt_camera_params(int w, int h, double out[5]) {
    if (!out || w <= 0 || h <= 0) { set_error("rt_camera_params: bad argument"); return RT_ERR_ARG; }
    const CameraD c = fixed_camera(w, h);
    out[0] = c.scaleX; out[1] = c.scaleY; out[2] = c.dX; out[3] = c.dY; out[4] = c.d;
    return RT_OK;
}

// saveBmpImage path (ImageRGB8 img) — Main.hs:621.  24-bit, bottom-up, BGR, rows padded to 4 bytes.
int rt_write_bmp(const char *path, const uint8_t *rgba, int w, int h) {
    if (!path || !rgba || w <= 0 || h <= 0) { set_error("rt_write_bmp: bad argument"); return RT_ERR_ARG; }
    FILE *f = std::fopen(path, "wb");
    if (!f) { set_error(std::string("rt_write_bmp: cannot open ") + path); return RT_ERR_ARG; }
    const uint32_t row = ((uint32_t)w * 3u + 3u) & ~3u, size = 54u + row * (uint32_t)h;
    uint8_t hdr[54] = {'B', 'M'};
    auto put32 = [&](int o, uint32_t v) { hdr[o] = v & 255; hdr[o + 1] = (v >> 8) & 255; hdr[o + 2] = (v >> 16) & 255; hdr[o + 3] = (v >> 24) & 255; };
    put32(2, size); put32(10, 54); put32(14, 40); put32(18, (uint32_t)w); put32(22, (uint32_t)h);
    hdr[26] = 1; hdr[28] = 24; put32(34, row * (uint32_t)h); put32(38, 2835); put32(42, 2835);
    std::fwrite(hdr, 1, 54, f);
    std::vector<uint8_t> line(row, 0);
    for (int y = h - 1; y >= 0; --y) {
        const uint8_t *src = rgba + (size_t)y * (size_t)w * 4;
        for (int x = 0; x < w; ++x) { line[3 * x] = src[4 * x + 2]; line[3 * x + 1] = src[4 * x + 1]; line[3 * x + 2] = src[4 * x]; }
        std::fwrite(line.data(), 1, row, f);
    }
    std::fclose(f);
    return RT_OK;
}

} // extern "C"
