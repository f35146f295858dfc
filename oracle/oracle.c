/*
 * oracle/oracle.c — CPU oracle: a restatement in IEEE double of the reference ray tracer
 * /root/reference/src/Main.hs (Haskell).  TEST INFRASTRUCTURE ONLY — see oracle.h.
 *
 * PARITY UNPINNED by the reference's own tests (it has none); pinned by the
 * known-answer table of SURVEY.md §4 and by following the cited lines expression by expression:
 * same association order, same <= / >= closedness, Haskell's max/min NaN behaviour
 * (`max x y = if x <= y then y else x`), `round` = round-half-even, Word8 wrap-around.
 *
 * Build: gcc -O2 -ffp-contract=off (no fused multiply-add: GHC emits none), see oracle/Makefile.
 */
#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* Vector3D — Main.hs:24-86                                                                    */

typedef struct { double x, y, z; } v3;

static inline v3 V(double x, double y, double z) { v3 r = { x, y, z }; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }       /* :32 */
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }       /* :34 */
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }       /* :36 */
static inline v3 vdiv(v3 a, v3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }       /* :46 */
static inline v3 vrecip(v3 a) { return V(1.0 / a.x, 1.0 / a.y, 1.0 / a.z); }           /* :47 */
static inline double hs_min(double x, double y) { return x <= y ? x : y; }             /* Ord Double */
static inline double hs_max(double x, double y) { return x <= y ? y : x; }
static inline v3 minV(v3 a, v3 b) { return V(hs_min(a.x, b.x), hs_min(a.y, b.y), hs_min(a.z, b.z)); } /* :51 */
static inline v3 maxV(v3 a, v3 b) { return V(hs_max(a.x, b.x), hs_max(a.y, b.y), hs_max(a.z, b.z)); } /* :56 */
static inline double dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }     /* :63 */
static inline v3 cross(v3 a, v3 b) {                                                   /* :68-69 */
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - b.x * a.y);
}
static inline double norm(v3 v) { return sqrt(dot(v, v)); }                            /* :74 */
static inline v3 normalize(v3 v) { double n = norm(v); return V(v.x / n, v.y / n, v.z / n); } /* :79 */
static inline v3 scalar(double d, v3 v) { return V(d * v.x, d * v.y, d * v.z); }       /* :85 */
static inline double hs_signum(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : x); }  /* GHC.Float */
static inline int v3eq(v3 a, v3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

/* ------------------------------------------------------------------------------------------ */
/* Ray — Main.hs:94-127                                                                        */

typedef struct { v3 pos, dir, recip; } ray;

static double g_eps; /* 2.2*(10**(-308)), Main.hs:114 */
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void init_consts(void) { g_eps = 2.2 * pow(10.0, -308.0); }

static inline double clampd(double d) { return fabs(d) <= g_eps ? g_eps : d; }         /* :112 */
static inline ray mkray(v3 position, v3 direction) {                                   /* :107-115 */
    ray r;
    r.pos = position;
    r.dir = V(clampd(direction.x), clampd(direction.y), clampd(direction.z));
    r.recip = vrecip(r.dir);
    return r;
}
static inline double rayDistance(const ray *r, v3 p) { return (p.z - r->pos.z) * r->recip.z; } /* :122-126 */

/* ------------------------------------------------------------------------------------------ */
/* BoundingBox — Main.hs:134-174                                                               */

typedef struct { v3 lo, hi; } bbox;
static const bbox BB_EMPTY = { { 0, 0, 0 }, { 0, 0, 0 } };                             /* :145 */
static inline int bb_is_mempty(bbox b) { return v3eq(b.lo, BB_EMPTY.lo) && v3eq(b.hi, BB_EMPTY.hi); }
static inline bbox bb_append(bbox a, bbox b) {                                         /* :146-150 */
    if (bb_is_mempty(a)) return b;
    if (bb_is_mempty(b)) return a;
    bbox r = { minV(a.lo, b.lo), maxV(a.hi, b.hi) };
    return r;
}
static int intersectBB(bbox b, const ray *r) {                                         /* :152-174 */
    const double x0 = b.lo.x, y0 = b.lo.y, z0 = b.lo.z, x1 = b.hi.x, y1 = b.hi.y, z1 = b.hi.z;
    const double px = r->pos.x, py = r->pos.y, pz = r->pos.z;
    const double dx = r->dir.x, dy = r->dir.y, dz = r->dir.z;
    const double rx = r->recip.x, ry = r->recip.y, rz = r->recip.z;
    double t, a, c;
#define XY(zv) (t = ((zv) - pz) * rz, a = px + t * dx, c = py + t * dy, \
                (t >= 0 && x0 <= a && a <= x1 && y0 <= c && c <= y1))
#define YZ(xv) (t = ((xv) - px) * rx, a = py + t * dy, c = pz + t * dz, \
                (t >= 0 && y0 <= a && a <= y1 && z0 <= c && c <= z1))
#define XZ(yv) (t = ((yv) - py) * ry, a = px + t * dx, c = pz + t * dz, \
                (t >= 0 && x0 <= a && a <= x1 && z0 <= c && c <= z1))
    return XY(z0) || XY(z1) || XZ(y0) || XZ(y1) || YZ(x0) || YZ(x1);
#undef XY
#undef YZ
#undef XZ
}

/* Is the outcome of intersectBB numerically fragile?  tol-widened and tol-narrowed versions of the
 * six face tests disagree.  Only used for the ORC_FLAG_BBOX_EDGE diagnostic. */
static int bb_face(double t, double a, double alo, double ahi, double c, double clo, double chi, double s) {
    double ta = s * 1e-9 * (1 + fabs(a) + fabs(alo) + fabs(ahi)), tc = s * 1e-9 * (1 + fabs(c) + fabs(clo) + fabs(chi));
    return t >= -s * 1e-9 * (1 + fabs(t)) && alo - ta <= a && a <= ahi + ta && clo - tc <= c && c <= chi + tc;
}
static int bb_any(bbox b, const ray *r, double s) {
    const double px = r->pos.x, py = r->pos.y, pz = r->pos.z, dx = r->dir.x, dy = r->dir.y, dz = r->dir.z;
    const double zs[2] = { b.lo.z, b.hi.z }, ys[2] = { b.lo.y, b.hi.y }, xs[2] = { b.lo.x, b.hi.x };
    for (int k = 0; k < 2; ++k) {
        double t = (zs[k] - pz) * r->recip.z;
        if (bb_face(t, px + t * dx, b.lo.x, b.hi.x, py + t * dy, b.lo.y, b.hi.y, s)) return 1;
        t = (ys[k] - py) * r->recip.y;
        if (bb_face(t, px + t * dx, b.lo.x, b.hi.x, pz + t * dz, b.lo.z, b.hi.z, s)) return 1;
        t = (xs[k] - px) * r->recip.x;
        if (bb_face(t, py + t * dy, b.lo.y, b.hi.y, pz + t * dz, b.lo.z, b.hi.z, s)) return 1;
    }
    return 0;
}
static int bb_fragile(bbox b, const ray *r) { return bb_any(b, r, 1.0) != bb_any(b, r, -1.0); }

/* ------------------------------------------------------------------------------------------ */
/* Scene tree                                                                                  */

enum { N_EMPTY, N_RECT, N_TRI, N_SPHERE, N_MANTLE, N_APPEND, N_LIST, N_GAUSS, N_FNORM };

typedef struct {
    int kind;
    bbox bb;
    int mat;       /* leaves: material index */
    int leaf_id;   /* assigned per render */
    /* plane leaves */
    v3 normal; double d;
    /* rectangle: point', width, height, ww, hh   (Main.hs:228-232) */
    v3 p0, e1, e2; double ww, hh;
    /* triangle: pa, u, v, uv, uu, vv, n          (Main.hs:365-372) */
    double uv, uu, vv, n;
    /* mantle rectangle (Main.hs:760-773): n2, p2, dNdP */
    v3 n2, p2, dNdP;
    /* sphere */
    v3 center; double radius;
    /* interior */
    int left, right;          /* N_APPEND; N_GAUSS uses left */
    int *kids; bbox *sfx; int nkids;  /* N_LIST: children and suffix boxes */
    /* normals wrapper: N_GAUSS (main's fnorm) and N_FNORM (any fnorm, given as a C callback) */
    double scale_inv; v3 offset;
    orc_fnorm fnorm; void *fnorm_user;
} node;

typedef struct { int kind; double diffuse, specular, f; v3 pos; int world; } lightsrc; /* kind 0 point, 1 ambient; world: the
   Shape a point light closes over (`pointLight world ...`, Main.hs:467), a node of this scene, or -1 = the traced one */

struct orc_scene {
    int talk;      /* 1: the rules of src/Talk.hs (orc_scene_set_talk) */
    int literal_mconcat; /* 1: orc_mconcat builds the right-nested mappend chain itself (orc_scene_set_literal_mconcat) */
    node *nodes; int nnodes, capnodes;
    orc_material *mats; int nmats, capmats;
    lightsrc *lights; int nlights, caplights;
};

typedef struct { v3 pos, normal; double t; int mat; int leaf; } hit;

orc_scene *orc_scene_new(void) {
    pthread_once(&g_once, init_consts);
    return (orc_scene *)calloc(1, sizeof(orc_scene));
}
void orc_scene_free(orc_scene *s) {
    if (!s) return;
    for (int i = 0; i < s->nnodes; ++i) { free(s->nodes[i].kids); free(s->nodes[i].sfx); }
    free(s->nodes); free(s->mats); free(s->lights); free(s);
}
static int new_node(orc_scene *s, int kind) {
    if (s->nnodes == s->capnodes) {
        s->capnodes = s->capnodes ? 2 * s->capnodes : 64;
        s->nodes = (node *)realloc(s->nodes, sizeof(node) * (size_t)s->capnodes);
    }
    node *n = &s->nodes[s->nnodes];
    memset(n, 0, sizeof *n);
    n->kind = kind; n->bb = BB_EMPTY; n->leaf_id = -1; n->left = n->right = -1;
    return s->nnodes++;
}
static int add_mat(orc_scene *s, const orc_material *m) {
    if (s->nmats == s->capmats) {
        s->capmats = s->capmats ? 2 * s->capmats : 64;
        s->mats = (orc_material *)realloc(s->mats, sizeof(orc_material) * (size_t)s->capmats);
    }
    s->mats[s->nmats] = *m;
    return s->nmats++;
}
static inline v3 P(const double p[3]) { return V(p[0], p[1], p[2]); }

/* Talk.hs dialect: its own normalize (:66-67), rectangle test (:119-128), nearest-by-squared-distance monoid
 * without bounding boxes (:185-194), single-term point light (:237-248), no reflection (:255-258). */
int orc_scene_set_talk(orc_scene *s, int on) {
    if (!s || s->nnodes > 0) return -1;        /* before the first shape: constructors precompute normals */
    s->talk = on ? 1 : 0;
    return 0;
}
/* Talk.hs:66-67: normalize v = scalar (recip (norm v)) v */
static inline v3 normalize_talk(v3 v) { double rn = 1.0 / norm(v); return V(rn * v.x, rn * v.y, rn * v.z); }

/* Test switch: every later orc_mconcat (also the ones inside orc_colorcuboid, orc_cylinder and
 * orc_linear_interpolation_gaussian) becomes the literal chain x0 `mappend` (x1 `mappend` (... `mappend` mempty)) of
 * N_APPEND nodes, evaluated by the plain mappend rule, instead of the list node with suffix boxes.  The two must render
 * the same image (tests/test_oracle_known_answers.py): the list node is the oracle's one non-literal rewriting. */
int orc_scene_set_literal_mconcat(orc_scene *s, int on) { s->literal_mconcat = on ? 1 : 0; return 0; }

int orc_mempty(orc_scene *s) { return new_node(s, N_EMPTY); }                          /* :204 */

int orc_mappend(orc_scene *s, int l, int r) {                                          /* :205-216 */
    if (l < 0 || r < 0 || l >= s->nnodes || r >= s->nnodes) return -1;
    int i = new_node(s, N_APPEND);
    s->nodes[i].left = l; s->nodes[i].right = r;
    s->nodes[i].bb = bb_append(s->nodes[l].bb, s->nodes[r].bb);                        /* :215 */
    return i;
}

/* mconcat xs = foldr mappend mempty xs: x0 <> (x1 <> (... (xn-1 <> mempty))).  Stored as a list
 * node with the bounding box of every suffix, so that evaluation needs no deep recursion. */
int orc_mconcat(orc_scene *s, const int *kids, int n) {
    for (int k = 0; k < n; ++k) if (kids[k] < 0 || kids[k] >= s->nnodes) return -1;
    if (s->literal_mconcat) {       /* foldr mappend mempty, spelled out: Main.hs:196-217 with the Monoid default */
        int acc = orc_mempty(s);
        for (int k = n - 1; k >= 0; --k) acc = orc_mappend(s, kids[k], acc);
        return acc;
    }
    int i = new_node(s, N_LIST);
    node *nd = &s->nodes[i];
    nd->nkids = n;
    nd->kids = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    nd->sfx = (bbox *)malloc(sizeof(bbox) * (size_t)(n + 1));
    memcpy(nd->kids, kids, sizeof(int) * (size_t)n);
    nd->sfx[n] = BB_EMPTY;
    for (int k = n - 1; k >= 0; --k) nd->sfx[k] = bb_append(s->nodes[kids[k]].bb, nd->sfx[k + 1]);
    nd->bb = nd->sfx[0];
    return i;
}

int orc_rectangle(orc_scene *s, const orc_material *m, const double point_[3],
                  const double width_[3], const double height_[3]) {                   /* :226-246 */
    v3 point = P(point_), width = P(width_), height = P(height_);
    int i = new_node(s, N_RECT);
    node *n = &s->nodes[i];
    n->mat = add_mat(s, m);
    n->normal = s->talk ? normalize_talk(cross(width, height))                         /* Talk.hs:114 */
                        : normalize(cross(width, height));                             /* :228 */
    n->ww = dot(width, width);                                                         /* :229 */
    n->hh = dot(height, height);                                                       /* :230 */
    n->p0 = vsub(vsub(point, scalar(0.5, width)), scalar(0.5, height));                /* :231 */
    n->d = -(dot(point, n->normal));                                                   /* :232 */
    n->e1 = width; n->e2 = height;
    v3 p2 = vadd(vadd(n->p0, width), height);                                          /* :244 */
    n->bb.lo = minV(n->p0, p2); n->bb.hi = maxV(n->p0, p2);                            /* :245 */
    return i;
}

int orc_triangle(orc_scene *s, const orc_material *m, const double pa_[3],
                 const double pb_[3], const double pc_[3]) {                           /* :363-385 */
    v3 pa = P(pa_), pb = P(pb_), pc = P(pc_);
    int i = new_node(s, N_TRI);
    node *n = &s->nodes[i];
    n->mat = add_mat(s, m);
    v3 u = vsub(pb, pa), v = vsub(pc, pa);                                             /* :365-366 */
    n->uv = dot(u, v); n->uu = dot(u, u); n->vv = dot(v, v);                           /* :367-369 */
    n->n = n->uv * n->uv - n->uu * n->vv;                                              /* :370 */
    n->normal = normalize(cross(u, v));                                                /* :371 */
    n->d = dot(scalar(-1, pa), n->normal);                                             /* :372 */
    n->p0 = pa; n->e1 = u; n->e2 = v;
    n->bb.lo = minV(pa, minV(pb, pc)); n->bb.hi = maxV(pa, maxV(pb, pc));              /* :384 */
    return i;
}

int orc_sphere(orc_scene *s, const orc_material *m, const double c_[3], double radius) { /* :392-415 */
    int i = new_node(s, N_SPHERE);
    node *n = &s->nodes[i];
    n->mat = add_mat(s, m);
    n->center = P(c_); n->radius = radius;
    v3 rv = V(radius, radius, radius);
    n->bb.lo = vsub(n->center, rv); n->bb.hi = vadd(n->center, rv);                    /* :413-414 */
    return i;
}

int orc_colorcuboid(orc_scene *s, const orc_material *mats, int nmats,
                    const double point_[3], double l, double h, double d) {            /* :301-320 */
    if (nmats <= 0) return -2;                                                         /* :302 */
    v3 point = P(point_);
    double l2 = l / 2, h2 = h / 2, d2 = d / 2;
    const orc_material *m[6];
    for (int k = 0; k < 6; ++k) m[k] = &mats[k % nmats];                               /* :304 */
    struct { v3 p, w, hgt; } f[6] = {
        { vadd(point, V(0, h2, 0)), V(l, 0, 0), V(0, 0, -d) },   /* top    :310 */
        { vsub(point, V(0, h2, 0)), V(l, 0, 0), V(0, 0, d) },    /* bottom :312 */
        { vadd(point, V(0, 0, d2)), V(l, 0, 0), V(0, h, 0) },    /* front  :314 */
        { vsub(point, V(0, 0, d2)), V(l, 0, 0), V(0, -h, 0) },   /* back   :316 */
        { vadd(point, V(l2, 0, 0)), V(0, h, 0), V(0, 0, d) },    /* left   :318 */
        { vsub(point, V(l2, 0, 0)), V(0, h, 0), V(0, 0, -d) },   /* right  :320 */
    };
    int kids[6];
    for (int k = 0; k < 6; ++k) {
        double p[3] = { f[k].p.x, f[k].p.y, f[k].p.z };
        double w[3] = { f[k].w.x, f[k].w.y, f[k].w.z };
        double hh[3] = { f[k].hgt.x, f[k].hgt.y, f[k].hgt.z };
        kids[k] = orc_rectangle(s, m[k], p, w, hh);
    }
    return orc_mconcat(s, kids, 6);
}

static int mantle_rect(orc_scene *s, const orc_material *m, v3 p1, v3 p2, v3 n1, v3 n2, double h) { /* :760-773 */
    v3 p = scalar(0.5, vadd(p1, p2));                                                  /* :761 */
    v3 dP = vsub(p1, p2);                                                              /* :762 */
    v3 dN = vsub(n1, n2);                                                              /* :763 */
    v3 dNdP = vdiv(dN, dP);                                                            /* :764 */
    double pp[3] = { p.x, p.y, p.z }, w[3] = { dP.x, dP.y, dP.z }, hh[3] = { 0, h, 0 };
    int i = orc_rectangle(s, m, pp, w, hh);                                            /* :765 */
    node *n = &s->nodes[i];
    n->kind = N_MANTLE; n->n2 = n2; n->p2 = p2; n->dNdP = dNdP;
    return i;
}

int orc_cylinder(orc_scene *s, const orc_material *topM, const orc_material *botM,
                 const orc_material *mantleM, const double point_[3], int n, double h, double r) { /* :725-774 */
    if (n < 1) return -3;
    v3 point = P(point_);
    double nn = (double)n;                                                             /* :738 */
    double alpha = 2 * M_PI / nn;                                                      /* :736 */
    double h2 = h / 2;                                                                 /* :737 */
    int np = n + 1;
    double *px = (double *)malloc(sizeof(double) * (size_t)np), *pz = (double *)malloc(sizeof(double) * (size_t)np);
    for (int i = 0; i < n; ++i) { double fi = (double)i; px[i] = r * cos(fi * alpha); pz[i] = r * sin(fi * alpha); } /* :729 */
    px[n] = r; pz[n] = 0;
    v3 topPoint = vadd(point, V(0, h2, 0));                                            /* :734 */
    v3 botPoint = vsub(point, V(0, -h2, 0));                                           /* :735 (sic) */
    int *kb = (int *)malloc(sizeof(int) * (size_t)n), *kt = (int *)malloc(sizeof(int) * (size_t)n),
        *km = (int *)malloc(sizeof(int) * (size_t)n);
#define ARR(v) { (v).x, (v).y, (v).z }
    for (int i = 0; i < n; ++i) {                                                      /* bot :747-751 */
        v3 b0 = vadd(point, V(px[i], -h2, pz[i])), b1 = vadd(point, V(px[i + 1], -h2, pz[i + 1]));
        double a[3] = ARR(botPoint), b[3] = ARR(b0), c[3] = ARR(b1);
        kb[i] = orc_triangle(s, botM, a, b, c);
    }
    for (int i = 0; i < n; ++i) {                                                      /* top :742-746 */
        v3 t0 = vadd(point, V(px[i], h2, pz[i])), t1 = vadd(point, V(px[i + 1], h2, pz[i + 1]));
        double a[3] = ARR(topPoint), b[3] = ARR(t1), c[3] = ARR(t0);
        kt[i] = orc_triangle(s, topM, a, b, c);
    }
    for (int i = 0; i < n; ++i) {                                                      /* mantle :752-758 */
        v3 m0 = vadd(point, V(px[i], 0, pz[i])), m1 = vadd(point, V(px[i + 1], 0, pz[i + 1]));
        v3 n0 = V(px[i] / r, 0, pz[i] / r), n1 = V(px[i + 1] / r, 0, pz[i + 1] / r);   /* :733 */
        km[i] = mantle_rect(s, mantleM, m0, m1, n0, n1, h);
    }
#undef ARR
    int bot = orc_mconcat(s, kb, n), top = orc_mconcat(s, kt, n), mantle = orc_mconcat(s, km, n);
    free(px); free(pz); free(kb); free(kt); free(km);
    return orc_mappend(s, orc_mappend(s, bot, top), mantle);                           /* :774, infixl */
}

int orc_normals_gaussian(orc_scene *s, int child, double scale_inv, const double offset[3]) { /* :956-965 */
    if (child < 0 || child >= s->nnodes) return -1;
    int i = new_node(s, N_GAUSS);
    s->nodes[i].left = child;
    s->nodes[i].bb = s->nodes[child].bb;                                               /* :964 */
    s->nodes[i].scale_inv = scale_inv; s->nodes[i].offset = P(offset);
    return i;
}

/* the `normals` wrapper with an arbitrary fnorm (Main.hs:956-965 takes any Double -> Double -> Vector3D) */
int orc_normals_callback(orc_scene *s, int child, double scale_inv, const double offset[3], orc_fnorm fn, void *user) {
    if (child < 0 || child >= s->nnodes || !fn) return -1;
    int i = new_node(s, N_FNORM);
    s->nodes[i].left = child;
    s->nodes[i].bb = s->nodes[child].bb;                                               /* :964 */
    s->nodes[i].scale_inv = scale_inv; s->nodes[i].offset = P(offset);
    s->nodes[i].fnorm = fn; s->nodes[i].fnorm_user = user;
    return i;
}

static int add_light(orc_scene *s, lightsrc l) {
    if (s->nlights == s->caplights) {
        s->caplights = s->caplights ? 2 * s->caplights : 8;
        s->lights = (lightsrc *)realloc(s->lights, sizeof(lightsrc) * (size_t)s->caplights);
    }
    s->lights[s->nlights] = l;
    return s->nlights++;
}
int orc_point_light(orc_scene *s, double diffuse, double specular, const double pos[3]) {
    lightsrc l = { 0, diffuse, specular, 0, P(pos), -1 };
    return add_light(s, l);
}
/* pointLight world' kd ks pos with a world' that is not the traced one: Main.hs:467-474 tests `intersect world' shadowRay` */
int orc_point_light_world(orc_scene *s, int world, double diffuse, double specular, const double pos[3]) {
    lightsrc l = { 0, diffuse, specular, 0, P(pos), world };
    return add_light(s, l);
}
int orc_ambient(orc_scene *s, double f) {
    lightsrc l = { 1, 0, 0, f, V(0, 0, 0), -1 };
    return add_light(s, l);
}

/* gaussian of main, Main.hs:580-584 */
static inline double gauss_f(double x, double y) { return exp(-(x * x + y * y) * 4); }
static inline v3 gauss_fnorm(double x, double y) {
    double dfdx = -8 * x * gauss_f(x, y);
    double dfdy = -8 * y * gauss_f(x, y);
    return normalize(cross(V(1, dfdx, 0), V(0, dfdy, 1)));
}

int orc_grid_steps(double a, double b, double step, double *out, int cap) {            /* :948 */
    int n = 0;
    for (double v = a; v < b; v = v + step) { if (out && n < cap) out[n] = v; ++n; }
    return n;
}

int orc_linear_interpolation_gaussian(orc_scene *s, double x1, double y1, double x2, double y2,
                                      double step, const double origin_[3], double scale) { /* :946-988 */
    v3 origin = P(origin_);
    double cx = (x2 + x1) / 2, cy = (y2 + y1) / 2;                                     /* :983-984 */
    double w = (x2 - x1), h = (y2 - y1);                                               /* :985-986 */
    double scaleInv = 1.0 / scale;                                                     /* :987 */
    v3 offset = vsub(origin, scalar(scale, V(cx, 0, cy)));                             /* :982 */
    orc_material white = { { 255, 255, 255 }, { 255, 255, 255 }, { 0, 0 }, 1.0, 0.0 }; /* :979 */
    orc_material aquamarine = { { 69, 139, 116 }, { 69, 139, 116 }, { 0, 0 }, 1.0, 0.0 };
    int nx = orc_grid_steps(x1, x2, step, NULL, 0), ny = orc_grid_steps(y1, y2, step, NULL, 0);
    double *xs = (double *)malloc(sizeof(double) * (size_t)(nx + 1)), *ys = (double *)malloc(sizeof(double) * (size_t)(ny + 1));
    orc_grid_steps(x1, x2, step, xs, nx); orc_grid_steps(y1, y2, step, ys, ny);
    int *cells = (int *)malloc(sizeof(int) * (size_t)(nx * ny + 1));
    double off[3] = { offset.x, offset.y, offset.z };
    int nc = 0;
#define FP(a, b, out) do { v3 q_ = vadd(offset, scalar(scale, V((a), gauss_f((a), (b)), (b)))); \
                           (out)[0] = q_.x; (out)[1] = q_.y; (out)[2] = q_.z; } while (0)  /* :981 */
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j) {                                                 /* :947, :950-954 */
            double x = xs[i], y = ys[j], xp = x + step, yp = y + step;
            double a[3], b[3], c[3], d[3];
            FP(x, yp, a); FP(xp, yp, b); FP(x, y, c); FP(xp, y, d);
            int two[2];
            two[0] = orc_triangle(s, &white, a, b, c);                                 /* :952 */
            two[1] = orc_triangle(s, &white, c, b, d);                                 /* :953 */
            cells[nc++] = orc_normals_gaussian(s, orc_mconcat(s, two, 2), scaleInv, off);
        }
#undef FP
    int top[2];
    top[0] = orc_mconcat(s, cells, nc);                                                /* :968 */
    double rw[3] = { 1.5 * scale * w, 0, 0 }, rh[3] = { 0, 0, -1.5 * scale * h };      /* :972-973 */
    top[1] = orc_rectangle(s, &aquamarine, origin_, rw, rh);
    int shape = orc_mconcat(s, top, 2);                                                /* :967 */
    v3 lp = vadd(origin, V(scale * w / 4, scale * 10, scale * w / 4));                 /* :976 */
    double lpos[3] = { lp.x, lp.y, lp.z };
    orc_point_light(s, 0.3, 0.6, lpos);
    orc_ambient(s, 0.3);                                                               /* :977 */
    free(xs); free(ys); free(cells);
    return shape;
}

/* ------------------------------------------------------------------------------------------ */
/* Intersection                                                                                */

/* planeLineIsect, Main.hs:264-277 */
static inline int planeLineIsect(v3 normal, double d, v3 lp, v3 ld, v3 *out, double *frac_out) {
    double a = normal.x, b = normal.y, c = normal.z;
    double lx = lp.x, ly = lp.y, lz = lp.z;
    double a_ = ld.x, b_ = ld.y, c_ = ld.z;
    double frac = dot(normal, ld) / c_;                                                /* :269 */
    double z = (-d - a * lx - b * ly + (frac - c) * lz) / frac;                        /* :270 */
    double x = (a_ * (z - lz) / c_) + lx;                                              /* :271 */
    double y = (b_ * (z - lz) / c_) + ly;                                              /* :272 */
    if (frac_out) *frac_out = frac;
    if (fabs(frac) <= 0.00001) return 0;                                               /* :273 */
    *out = V(x, y, z);
    return 1;
}
/* planeRayIsect, Main.hs:252-258 */
static inline int planeRayIsect(v3 normal, double d, const ray *r, v3 *isect, double *t) {
    if (!planeLineIsect(normal, d, r->pos, r->dir, isect, NULL)) return 0;
    *t = rayDistance(r, *isect);                                                       /* :256 */
    return *t >= 0;                                                                    /* :257 */
}

static int isect_leaf(const node *n, const ray *r, hit *out, int talk) {
    v3 isect; double t;
    if (talk) {                                                                        /* Talk.hs:119-128 */
        if (n->kind != N_RECT) return 0;                                               /* Talk.hs has rectangles only */
        if (!planeLineIsect(n->normal, n->d, r->pos, r->dir, &isect, NULL)) return 0;  /* :120 */
        v3 v = vsub(isect, r->pos);
        if (!(dot(r->dir, v) >= 0)) return 0;                                          /* :121-122 */
        v3 dV = vsub(isect, n->p0);
        double dw = dot(dV, n->e1), dh = dot(dV, n->e2);
        if (!(0 <= dw && dw <= n->ww && 0 <= dh && dh <= n->hh)) return 0;             /* :123-127 */
        out->pos = isect; out->normal = n->normal; out->mat = n->mat; out->leaf = n->leaf_id;
        out->t = dot(v, v);                  /* the monoid's key: squared distance from the ray position, :192-193 */
        return 1;
    }
    switch (n->kind) {
    case N_RECT: case N_MANTLE: {                                                      /* :234-242 */
        if (!planeRayIsect(n->normal, n->d, r, &isect, &t)) return 0;
        v3 dV = vsub(isect, n->p0);
        double dw = dot(dV, n->e1), dh = dot(dV, n->e2);
        if (!(0 <= dw && dw <= n->ww && 0 <= dh && dh <= n->hh)) return 0;             /* :241 */
        out->pos = isect; out->t = t; out->mat = n->mat; out->leaf = n->leaf_id;
        if (n->kind == N_RECT) out->normal = n->normal;
        else {                                                                         /* :769-771 */
            v3 nn = vadd(n->n2, vmul(vsub(isect, n->p2), n->dNdP));
            out->normal = V(nn.x, 0, nn.z);
        }
        return 1;
    }
    case N_TRI: {                                                                      /* :374-383 */
        if (!planeRayIsect(n->normal, n->d, r, &isect, &t)) return 0;
        v3 w = vsub(isect, n->p0);
        double wv = dot(w, n->e2), wu = dot(w, n->e1);
        double rr = (n->uv * wv - n->vv * wu) / n->n;                                  /* :379 */
        double ss = (n->uv * wu - n->uu * wv) / n->n;                                  /* :380 */
        if (!(rr >= 0 && ss >= 0 && rr + ss <= 1)) return 0;                           /* :382 */
        out->pos = isect; out->normal = n->normal; out->t = t; out->mat = n->mat; out->leaf = n->leaf_id;
        return 1;
    }
    case N_SPHERE: {                                                                   /* :395-411 */
        v3 o = vsub(r->pos, n->center), d = r->dir;
        double b = 2 * dot(d, o);
        double c = dot(o, o) - n->radius * n->radius;
        double delta = b * b - 4 * c;
        if (!(delta >= 0)) return 0;
        if (delta > 0) {
            double t1 = (-b + sqrt(delta)) / 2, t2 = (-b - sqrt(delta)) / 2;
            t = hs_min(hs_max(t1, 0), hs_max(t2, 0));
        } else t = -b / 2;
        if (!(t > 0)) return 0;
        isect = vadd(r->pos, scalar(t, d));
        out->pos = isect; out->normal = scalar(1.0 / n->radius, vsub(isect, n->center));
        out->t = t; out->mat = n->mat; out->leaf = n->leaf_id;
        return 1;
    }
    default: return 0;
    }
}

static int isect_node(const orc_scene *s, int idx, const ray *r, hit *out, int use_bbox) {
    const node *n = &s->nodes[idx];
    if (s->talk) use_bbox = 0;                     /* Talk.hs:185-194: the monoid has no bounding boxes */
    switch (n->kind) {
    case N_EMPTY: return 0;
    case N_RECT: case N_TRI: case N_SPHERE: case N_MANTLE: return isect_leaf(n, r, out, s->talk);
    case N_GAUSS: {                                                                    /* :959-963 */
        if (!isect_node(s, n->left, r, out, use_bbox)) return 0;
        v3 q = scalar(n->scale_inv, vsub(out->pos, n->offset));
        v3 np = gauss_fnorm(q.x, q.z);
        out->normal = scalar(hs_signum(dot(out->normal, np)), np);
        return 1;
    }
    case N_FNORM: {                                                                    /* :959-963, fnorm' = the callback */
        if (!isect_node(s, n->left, r, out, use_bbox)) return 0;
        v3 q = scalar(n->scale_inv, vsub(out->pos, n->offset));
        double nv[3] = { 0, 0, 0 };
        n->fnorm(q.x, q.z, nv, n->fnorm_user);
        v3 np = V(nv[0], nv[1], nv[2]);
        out->normal = scalar(hs_signum(dot(out->normal, np)), np);
        return 1;
    }
    case N_APPEND: {                                                                   /* :206-214 */
        if (use_bbox && !intersectBB(n->bb, r)) return 0;
        hit h1, h2;
        int a = isect_node(s, n->left, r, &h1, use_bbox), b = isect_node(s, n->right, r, &h2, use_bbox);
        if (!a) { if (b) *out = h2; return b; }
        if (!b) { *out = h1; return 1; }
        *out = (h1.t <= h2.t) ? h1 : h2;
        return 1;
    }
    case N_LIST: {
        /* x0 <> (x1 <> ( ... <> mempty)): the suffix starting at the first failed guard is Nothing;
         * among the children before it the left-most minimal t wins (t1 <= t2 keeps the left). */
        int found = 0; hit h;
        for (int k = 0; k < n->nkids; ++k) {
            if (use_bbox && !intersectBB(n->sfx[k], r)) break;
            if (isect_node(s, n->kids[k], r, &h, use_bbox) && (!found || h.t < out->t)) { *out = h; found = 1; }
        }
        return found;
    }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Pixels and lights — Main.hs:436-492                                                         */

typedef struct { uint8_t r, g, b; } px8;
static inline uint8_t sat_add8(uint8_t x, uint8_t y) { uint8_t s = (uint8_t)(x + y); return s < x ? 255 : s; } /* :447 */
static inline px8 addPixel(px8 a, px8 b) { px8 r = { sat_add8(a.r, b.r), sat_add8(a.g, b.g), sat_add8(a.b, b.b) }; return r; }
/* Haskell `round` (half-even) to Integer, then fromInteger to Word8 (wraps mod 256), :451-456 */
static inline uint8_t round8(double v) {
    double r = nearbyint(v);      /* default rounding mode: to nearest, ties to even */
    if (!(fabs(r) < 9.0e18)) return 0;   /* NaN / inf / huge: outside what any scene here produces */
    long long i = (long long)r;
    return (uint8_t)(i & 0xff);
}
static inline px8 scalePixel(double f, const uint8_t c[3]) {
    px8 r = { round8(f * (double)c[0]), round8(f * (double)c[1]), round8(f * (double)c[2]) };
    return r;
}

static px8 shade(const orc_scene *s, int world, const hit *h, const ray *r, int use_bbox) { /* :436-441 */
    px8 acc = { 0, 0, 0 };
    const orc_material *m = &s->mats[h->mat];
    for (int i = 0; i < s->nlights; ++i) {
        const lightsrc *L = &s->lights[i];
        const int lworld = L->world >= 0 ? L->world : world;      /* the Shape the light closed over */
        px8 c = { 0, 0, 0 };
        if (L->kind == 1) c = scalePixel(L->f, m->diffuse);                            /* :490-492, Talk.hs:230-235 */
        else if (s->talk) {                                                            /* Talk.hs:237-248 */
            v3 to_light = normalize_talk(vsub(L->pos, h->pos));                        /* :239 */
            ray rl = mkray(vadd(h->pos, scalar(0.00001, to_light)), to_light);         /* :240 */
            hit sh;
            if (isect_node(s, lworld, &rl, &sh, 0) && dot(vsub(L->pos, sh.pos), to_light) >= 0) { /* :241-242 */
                /* black */
            } else {
                double f = hs_min(1, hs_max(0, dot(to_light, h->normal)) * L->diffuse); /* :244 */
                c = scalePixel(f, m->diffuse);                                         /* :245-248 */
            }
        } else {                                                                         /* :467-487 */
            v3 to_light = normalize(vsub(L->pos, h->pos));
            double sp = m->specularity;
            ray rl = mkray(vadd(h->pos, scalar(0.00001, to_light)), to_light);
            hit sh;
            if (isect_node(s, lworld, &rl, &sh, use_bbox) && sh.t <= rayDistance(&rl, L->pos)) { /* :474 */
                /* black */
            } else {
                double lndot = dot(to_light, h->normal);
                v3 reflection = vsub(to_light, scalar(2 * lndot, h->normal));
                double fDiffuse = L->diffuse * hs_max(0, lndot);
                double fSpecular = (lndot <= 0) ? 0 : L->specular * pow(hs_max(0, dot(reflection, r->dir)), sp);
                c = addPixel(scalePixel(fDiffuse, m->diffuse), scalePixel(fSpecular, m->specular));
            }
        }
        acc = addPixel(acc, c);
    }
    return acc;
}

/* trace, Main.hs:541-557 */
static px8 trace(const orc_scene *s, int world, int n, const ray *r, int use_bbox, hit *first, int *first_ok) {
    px8 black = { 0, 0, 0 };
    if (n == 0) { if (first_ok) *first_ok = 0; return black; }
    hit h;
    int ok = isect_node(s, world, r, &h, use_bbox);
    if (first_ok) { *first_ok = ok; if (ok) *first = h; }
    if (!ok) return black;
    px8 direct = shade(s, world, &h, r, use_bbox);
    if (s->talk) return direct;                                                        /* Talk.hs:255-258 */
    double reflectivity = s->mats[h.mat].reflectivity;
    double rr = dot(h.normal, r->dir);                                                 /* :549 */
    px8 reflection = black;
    if (reflectivity > 0 && rr < 0) {                                                  /* :553 */
        v3 rdir = vsub(r->dir, scalar(2 * rr, h.normal));                              /* :550 */
        ray refl = mkray(vadd(h.pos, scalar(0.00001, rdir)), rdir);                    /* :552 */
        px8 sub = trace(s, world, n - 1, &refl, use_bbox, NULL, NULL);
        uint8_t c[3] = { sub.r, sub.g, sub.b };
        reflection = scalePixel(reflectivity, c);                                      /* :554 */
    }
    return addPixel(direct, reflection);                                               /* :545 */
}

/* ------------------------------------------------------------------------------------------ */
/* Camera — Main.hs:506-533                                                                    */

typedef struct { double scaleX, scaleY, dX, dY, d; } camera;
static camera fixedCamera(int width, int height) {
    camera c;
    double w = (double)width, h = (double)height;
    double fov = M_PI / 2;                                                             /* :510 */
    c.scaleX = 1.0 / w;                                                                /* :516 */
    c.scaleY = c.scaleX * (-h / w);                                                    /* :518 */
    c.dX = -(c.scaleX * w / 2);                                                        /* :525 */
    c.dY = -(c.scaleY * h / 2);                                                        /* :526 */
    c.d = tan(fov / 2) * c.dX;                                                         /* :527 */
    return c;
}
static inline ray cast(const camera *c, double sx, double sy, int talk) {              /* :528-533, Talk.hs:148-153 */
    double posX = c->scaleX * sx + c->dX;
    double posY = c->scaleY * sy + c->dY;
    return mkray(V(posX, posY, 0), talk ? normalize_talk(V(posX, posY, c->d)) : normalize(V(posX, posY, c->d)));
}

double orc_eps(void) { pthread_once(&g_once, init_consts); return g_eps; }
void orc_camera_params(int w, int h, double out[5]) {
    camera c = fixedCamera(w, h);
    out[0] = c.scaleX; out[1] = c.scaleY; out[2] = c.dX; out[3] = c.dY; out[4] = c.d;
}
void orc_camera_ray(int w, int h, int x, int y, double out[9]) {
    pthread_once(&g_once, init_consts);
    camera c = fixedCamera(w, h);
    ray r = cast(&c, (double)x, (double)y, 0);
    out[0] = r.pos.x; out[1] = r.pos.y; out[2] = r.pos.z;
    out[3] = r.dir.x; out[4] = r.dir.y; out[5] = r.dir.z;
    out[6] = r.recip.x; out[7] = r.recip.y; out[8] = r.recip.z;
}

/* ------------------------------------------------------------------------------------------ */
/* Jittered supersampling — this repo's extension (spp > 1), specified in DESIGN.md            */

static inline uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}
static void sample_offset(int x, int y, int s, int spp, uint64_t seed, double *ox, double *oy) {
    if (spp <= 1) { *ox = 0; *oy = 0; return; }
    uint32_t h = mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) + 0x9e3779b9u));
    h = mix32(h ^ (uint32_t)x);
    h = mix32(h ^ ((uint32_t)y + 0x9e3779b9u));
    h = mix32(h ^ ((uint32_t)s + 0x7f4a7c15u));
    uint32_t h2 = mix32(h ^ 0x68bc21ebu);
    double u1 = (double)(h >> 8) * (1.0 / 16777216.0), u2 = (double)(h2 >> 8) * (1.0 / 16777216.0);
    int k = (int)floor(sqrt((double)spp) + 0.5);
    if (k * k == spp) {
        int sx = s % k, sy = s / k;
        *ox = ((double)sx + u1) / (double)k - 0.5;
        *oy = ((double)sy + u2) / (double)k - 0.5;
    } else { *ox = u1 - 0.5; *oy = u2 - 0.5; }
}

/* ------------------------------------------------------------------------------------------ */
/* Leaf numbering and the fragile-pixel analysis                                               */

static void number_leaves(orc_scene *s, int idx, int *next, int *list) {
    node *n = &s->nodes[idx];
    switch (n->kind) {
    case N_RECT: case N_TRI: case N_SPHERE: case N_MANTLE:
        if (list) list[*next] = idx;
        n->leaf_id = (*next)++;
        break;
    case N_GAUSS: case N_FNORM: number_leaves(s, n->left, next, list); break;
    case N_APPEND: number_leaves(s, n->left, next, list); number_leaves(s, n->right, next, list); break;
    case N_LIST: for (int k = 0; k < n->nkids; ++k) number_leaves(s, n->kids[k], next, list); break;
    default: break;
    }
}
int orc_num_leaves(orc_scene *s, int root) {
    if (root < 0 || root >= s->nnodes) return -1;
    int n = 0;
    number_leaves(s, root, &n, NULL);
    return n;
}

/* signed inside-margin of a plane leaf at the plane hit (normalised units); >= 0 means accepted */
static int leaf_margin(const node *n, const ray *r, double *t_out, double *margin, double *frac_out) {
    if (n->kind == N_SPHERE) return 0;
    v3 isect; double frac;
    int ok = planeLineIsect(n->normal, n->d, r->pos, r->dir, &isect, &frac);
    *frac_out = frac;
    if (!ok) return 0;
    *t_out = rayDistance(r, isect);
    if (n->kind == N_TRI) {
        v3 w = vsub(isect, n->p0);
        double wv = dot(w, n->e2), wu = dot(w, n->e1);
        double rr = (n->uv * wv - n->vv * wu) / n->n, ss = (n->uv * wu - n->uu * wv) / n->n;
        *margin = fmin(fmin(rr, ss), 1 - (rr + ss));
    } else {
        v3 dV = vsub(isect, n->p0);
        double a = dot(dV, n->e1) / n->ww, b = dot(dV, n->e2) / n->hh;
        *margin = fmin(fmin(a, 1 - a), fmin(b, 1 - b));
    }
    return 1;
}

static int guards_fragile(const orc_scene *s, int idx, const ray *r) {
    const node *n = &s->nodes[idx];
    switch (n->kind) {
    case N_GAUSS: case N_FNORM: return guards_fragile(s, n->left, r);
    case N_APPEND:
        return bb_fragile(n->bb, r) || guards_fragile(s, n->left, r) || guards_fragile(s, n->right, r);
    case N_LIST:
        for (int k = 0; k < n->nkids; ++k)
            if (bb_fragile(n->sfx[k], r) || guards_fragile(s, n->kids[k], r)) return 1;
        return 0;
    default: return 0;
    }
}

static uint8_t analyze(const orc_scene *s, int root, const int *leaves, int nleaves, const ray *r,
                       int guarded_ok, int guarded_leaf) {
    uint8_t flags = 0;
    if (guards_fragile(s, root, r)) flags |= ORC_FLAG_BBOX_EDGE;
    double t1 = INFINITY, t2 = INFINITY; int id1 = -1;
    for (int k = 0; k < nleaves; ++k) {
        hit h;
        if (isect_leaf(&s->nodes[leaves[k]], r, &h, s->talk)) {
            if (h.t < t1) { t2 = t1; t1 = h.t; id1 = k; }
            else if (h.t < t2) t2 = h.t;
        }
    }
    if (id1 >= 0 && t2 - t1 <= 1e-5) flags |= ORC_FLAG_TIE;
    if ((guarded_ok ? guarded_leaf : -1) != id1) flags |= ORC_FLAG_BBOX;
    for (int k = 0; k < nleaves; ++k) {
        double t = 0, m = 0, frac = 0;
        int ok = leaf_margin(&s->nodes[leaves[k]], r, &t, &m, &frac);
        if (fabs(fabs(frac) - 0.00001) <= 1e-10) flags |= ORC_FLAG_EDGE;
        if (!ok) continue;
        if (!(t >= -1e-5 && t <= t1 + 1e-5)) continue;
        if (fabs(t) <= 1e-5 && m >= -1e-5) flags |= ORC_FLAG_EDGE;
        if (fabs(m) <= 1e-5) flags |= ORC_FLAG_EDGE;
    }
    return flags;
}

/* ------------------------------------------------------------------------------------------ */
/* parallelTrace — Main.hs:623-637: contiguous chunks of rows per thread, like Repa's computeP  */

typedef struct {
    orc_scene *s; int root, w, h, depth, spp; uint64_t seed; int row0, y0, y1, use_bbox;
    uint8_t *rgb; int32_t *hit_id; double *hit_t; uint8_t *flags;
    const int *leaves; int nleaves;
} job;

static void *render_rows(void *arg) {
    job *j = (job *)arg;
    camera cam = fixedCamera(j->w, j->h);
    for (int y = j->y0; y < j->y1; ++y)
        for (int x = 0; x < j->w; ++x) {
            size_t o = (size_t)(y - j->row0) * (size_t)j->w + (size_t)x;
            unsigned sum[3] = { 0, 0, 0 };
            for (int sidx = 0; sidx < j->spp; ++sidx) {
                double ox, oy;
                sample_offset(x, y, sidx, j->spp, j->seed, &ox, &oy);
                ray r = cast(&cam, (double)x + ox, (double)y + oy, j->s->talk);
                hit h; int ok;
                px8 c = trace(j->s, j->root, j->depth, &r, j->use_bbox, &h, &ok);
                sum[0] += c.r; sum[1] += c.g; sum[2] += c.b;
                if (sidx == 0) {
                    if (j->hit_id) j->hit_id[o] = ok ? h.leaf : -1;
                    if (j->hit_t) j->hit_t[o] = ok ? h.t : INFINITY;
                    if (j->flags) j->flags[o] = analyze(j->s, j->root, j->leaves, j->nleaves, &r, ok, ok ? h.leaf : -1);
                }
            }
            unsigned n = (unsigned)j->spp;
            j->rgb[3 * o + 0] = (uint8_t)((sum[0] + n / 2) / n);
            j->rgb[3 * o + 1] = (uint8_t)((sum[1] + n / 2) / n);
            j->rgb[3 * o + 2] = (uint8_t)((sum[2] + n / 2) / n);
        }
    return NULL;
}

int orc_render(orc_scene *s, int root, int w, int h, int depth, int spp, uint64_t seed,
               int row0, int nrows, int nthreads, int use_bbox,
               uint8_t *rgb, int32_t *hit_id, double *hit_t, uint8_t *flags) {
    if (!s || !rgb || root < 0 || root >= s->nnodes || w <= 0 || h <= 0 || spp < 1 ||
        row0 < 0 || nrows < 0 || row0 + nrows > h) return -1;
    pthread_once(&g_once, init_consts);
    int nleaves = orc_num_leaves(s, root);
    int *leaves = (int *)malloc(sizeof(int) * (size_t)(nleaves > 0 ? nleaves : 1));
    int cnt = 0;
    number_leaves(s, root, &cnt, leaves);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > nrows) nthreads = nrows > 0 ? nrows : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    job *jobs = (job *)malloc(sizeof(job) * (size_t)nthreads);
    for (int k = 0; k < nthreads; ++k) {
        job jb = { s, root, w, h, depth, spp, seed, row0,
                   row0 + (int)((long long)nrows * k / nthreads), row0 + (int)((long long)nrows * (k + 1) / nthreads),
                   use_bbox, rgb, hit_id, hit_t, flags, leaves, nleaves };
        jobs[k] = jb;
        if (nthreads == 1) render_rows(&jobs[k]);
        else pthread_create(&th[k], NULL, render_rows, &jobs[k]);
    }
    if (nthreads > 1) for (int k = 0; k < nthreads; ++k) pthread_join(th[k], NULL);
    free(th); free(jobs); free(leaves);
    return 0;
}
