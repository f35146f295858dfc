// rt_types.h — data layout shared by the host scene builder and the gfx950 kernels.
//
// Two representations of every leaf live in HBM:
//   * LeafD  — the exact one (IEEE double), with the constants the reference precomputes in its
//              constructors (rectangle Main.hs:228-232, triangle Main.hs:365-372).  Only the few
//              candidate (ray, leaf) pairs that survive the FP32 filter are evaluated on it, with
//              the reference's own expression order, so hit IDs and colours agree bit for bit.
//   * FilterRec — 16 floats per (apex, leaf): linear forms of the ray direction D from a common
//              apex (the eye for primary rays, a light for shadow rays).  This is what the
//              brute-force O(rays x leaves) loops read, as wave-uniform scalar operands.
#pragma once
#include <stdint.h>

namespace rt {

enum : int32_t { LEAF_RECT = 0, LEAF_TRI = 1, LEAF_SPHERE = 2 };

struct LeafD {                 // 176 bytes
    double nx, ny, nz, d;      // unit normal, plane offset (normal . p + d = 0)
    double p0x, p0y, p0z;      // rect: point' (corner, Main.hs:231); tri: pa
    double e1x, e1y, e1z;      // rect: width;  tri: u = pb - pa
    double e2x, e2y, e2z;      // rect: height; tri: v = pc - pa
    double c0, c1, c2, c3;     // rect: ww, hh, 0, 0;  tri: uv, uu, vv, n;  sphere: radius (p0 = centre)
    int32_t kind;              // LEAF_RECT / LEAF_TRI / LEAF_SPHERE
    int32_t nmode;             // rt_normal_mode
    int32_t mat;               // material index
    int32_t aux;               // index into AuxD (mantle / gaussian parameters), -1 if none
    int32_t slot;              // index of this leaf inside its per-type filter array
    int32_t guard;             // innermost GuardD above this leaf (-1: none); a hit counts only if the ray
                               // meets every box on the chain guard -> parent -> ... (Main.hs:209)
};

// Bounding box of one mappend node of the reference's Shape tree (Main.hs:215) and the next
// different box further up.  Runs of equal boxes along a path are stored once.
struct GuardD {
    double x0, y0, z0, x1, y1, z1;
    int32_t parent;            // -1 at the root
    int32_t _pad;
};

struct AuxD { double a[10]; }; // mantle: n2 xyz, p2 xyz, dNdP xyz ; gaussian: scaleInv, offset xyz

struct MaterialD {             // Main.hs:178-185
    double specularity, reflectivity;
    uint8_t diffuse[3], specular[3], _pad[2];
};

struct LightD {                // kind 0: pointLight (Main.hs:467), kind 1: ambient (Main.hs:490)
    double px, py, pz;
    double diffuse, specular, f;
    int32_t kind;
    int32_t rec_offset;        // index of this light's filter records block (point lights)
};

// 16 floats (one 64-byte line).  c[0..8]: three linear forms of the ray direction D from the apex —
// triangle U, V, W (= N - U - V); rectangle A, B, N.  c[9] = negm (= -margin): the loops keep c[0..9] in SGPRs.
// c[10] = Kq, c[11..13] = N xyz (the sign-normalised plane normal: N.D >= Kq  <=>  the plane is met between the
// apex and A + D): read from memory by the shadow kernel's candidate path only.  c[14..15]: spare in the per-type
// arrays of the vector-ALU kernels; leaf kind and leaf index (bit patterns) in the stream-ordered copy (mf_rec).
struct FilterRec { float c[16]; };
// Sphere, apex form: oc xyz (apex - centre), cs = |oc|^2 - r^2, ..., c[9] = negm: the line A + u D meets the
// sphere iff (D.oc)^2 - |D|^2 cs >= 0.  Sphere, general form (GenRec): centre xyz, r^2, ..., c[12] = negm.

// MFMA stream (v_mfma_f32_32x32x16_f16): blocks of 32 rows x 8 halves (512 bytes), one row per planar leaf in tree
// order (leaf at stream position k = row k % 32 of block k / 32).  Row r of a block is the A-operand fragment of
// lanes r and r + 32; the D tile puts row (q & 3) + 8 (q >> 2) + 4 hh into accumulator register q of lane half hh.
// Row = 2^10 (a_x, a_y, a_z, margin - cos theta) as FP16 hi/mid pairs: the bounding cone of the leaf seen from the
// apex (rt_scene.cpp: make_cone_row).
constexpr int MF_LEAVES = 32;

// 16 floats per leaf for rays with arbitrary origins (reflected rays, Main.hs:549-552):
//   n xyz, dpl, g1 xyz, c1, g2 xyz, c2, negm, 0, 0, 0     with q_k(p) = g_k . p + c_k the two inside coordinates
struct GenRec { float c[16]; };

// A reflected ray waiting to be traced: the arguments of `mkray` (Main.hs:552), the reflectivity that
// scales what it returns (Main.hs:554) and the slot of the ray that spawned it.
struct RayRec {
    double ox, oy, oz, dx, dy, dz, rho;
    uint32_t parent, _pad;
};

struct CameraD {               // fixedCamera, Main.hs:506-527
    double scaleX, scaleY, dX, dY, d;
};

} // namespace rt
