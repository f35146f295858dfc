/*
 * raytrace_gpu.h — C ABI of libraytrace_gpu.so, the MI355X (gfx950) render path that replaces the
 * per-pixel Repa loop of alexandervandenbroucke/raytrace:
 *     parallelTrace w h camera (trace depth lights world)     reference src/Main.hs:623-637, 571
 *     parallel      w h trace                                 reference src/Talk.hs:171-182
 *
 * The reference has no FFI/plugin interface of its own: its seam is a Haskell closure
 * (Ray -> PixelRGB8), which cannot cross a C ABI.  The boundary therefore sits at the scene
 * constructors the closure is built from; every entry point below names the reference function
 * (file:line under /root/reference/) that it reifies as data.  INTEGRATION.md shows the Haskell
 * `foreign import ccall` stubs that bind these symbols and keep the reference's scene API.
 *
 * Conventions: plain C types only; every function returns 0 (RT_OK) or a negative rt_status and
 * stores a message retrievable with rt_last_error() (thread-local).  No exception or abort crosses
 * the boundary.  rt_add_* return the index (>= 0) of the first leaf they created: leaf indices are
 * the depth-first, left-to-right leaf order of the reference's mconcat/mappend tree, which is the
 * order in which the reference resolves equal-distance hits (Main.hs:213-214), and they are the
 * values written to the hit-ID buffer.  There is NO CPU fallback: rendering without a usable HIP
 * device fails with RT_ERR_HIP.
 */
#ifndef RAYTRACE_GPU_H
#define RAYTRACE_GPU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_scene rt_scene;   /* opaque; owned by the library */

typedef enum {
    RT_OK              =  0,
    RT_ERR_ARG         = -1,   /* null pointer, bad size, bad index            */
    RT_ERR_HIP         = -2,   /* HIP runtime error / no device                */
    RT_ERR_UNSUPPORTED = -3,   /* feature of the reference not built yet       */
    RT_ERR_NOMEM       = -4
} rt_status;

/* Material — Main.hs:178-185 (mat_diffuse, mat_specular, mat_specularity, mat_reflectivity) */
typedef struct {
    uint8_t diffuse[3];
    uint8_t specular[3];
    uint8_t _pad[2];
    double  specularity;
    double  reflectivity;
} rt_material;

/* How the shading normal of a leaf is produced */
typedef enum {
    RT_NORMAL_GEOMETRIC = 0,  /* rectangle Main.hs:228 / triangle Main.hs:371                        */
    RT_NORMAL_MANTLE    = 1,  /* cylinder mantle: lerped, y forced to 0, not renormalised (:769-771) */
    RT_NORMAL_GAUSSIAN  = 2,  /* linearInterpolation's `normals` wrapper with main's gaussian fnorm
                                 (Main.hs:956-965, 580-584); params = {scaleInv, offset x,y,z}      */
    RT_NORMAL_CALLBACK  = 3   /* the same wrapper with ANY fnorm (Main.hs:956-965 takes an arbitrary
                                 Double -> Double -> Vector3D): set with rt_set_normal_callback       */
} rt_normal_mode;

/* An `fnorm` closure of linearInterpolation (Main.hs:937, 956-965) as a C callback: for each of the n hit points,
 * xy[2i], xy[2i+1] are the arguments x, y the reference passes to fnorm ((x, _, y) = scaleInv * (i - offset),
 * Main.hs:958) and normals[3i..3i+2] receives fnorm x y.  Called on the host, from the thread inside rt_render. */
typedef void (*rt_normal_fn)(int n, const double *xy, double *normals, void *user);

/* Per-render statistics (filled by rt_get_stats after a render) */
typedef struct {
    uint64_t primary_rays;       /* w * nrows * spp                                           */
    uint64_t shadow_rays;        /* shadow rays actually cast, all lights: hit pixel, L.N > 0, and a light whose
                                    contribution to the pixel does not round to (0, 0, 0) anyway             */
    uint64_t tri_pairs;          /* ray x triangle pairs the render decided = rays x triangles (an identity, not a
                                    count: most pairs are decided 256 rays x 32 leaves at a time by a block's cone) */
    uint64_t rect_pairs;         /* ray x rectangle pairs decided, likewise                     */
    uint64_t exact_evals;        /* FP64 exact leaf evaluations (candidates of the filter)     */
    double   ms_primary;         /* HIP-event time of k_primary (nearest hit)                  */
    double   ms_setup;           /* ... of k_setup (exact hit records + shadow-job queues)     */
    double   ms_shadow;          /* ... of k_shadow_mfma / k_shadow (shadow rays)               */
    double   ms_final;           /* ... of k_final (Phong + RGBA8 store)                       */
    double   ms_total;           /* first launch to last kernel end                            */
    uint64_t mfma_blocks_primary; /* (wave of 256 rays) x (block of 32 leaves) pairs k_primary_mfma scanned, 8 MFMAs
                                     each; the others were skipped by the block-cone test         */
    uint64_t mfma_blocks_shadow;  /* the same for k_shadow_mfma, all lights                        */
    uint64_t record_tests;        /* FP32 record tests (second stage), counted in the kernels       */
    uint64_t bounce_rays;         /* reflected rays traced (all levels)                             */
    uint64_t bounce_blocks;       /* (run of 64 reflected rays) x (block of 32 leaves) pairs k_general looked at */
    double   prepare_ms;          /* host time of the last rebuild of guard chains and filter records */
    double   upload_ms;           /* host time of the last upload of the scene to this device       */
    uint64_t multi_bands_direct;  /* rt_render_multi: bands copied to devices[0] with peer access enabled (xGMI DMA)    */
    uint64_t multi_bands_staged;  /* ... without (hipMemcpyPeerAsync then stages through host memory); 0 + 0: not a multi render */
} rt_stats;

/* ---- library / device ------------------------------------------------------------------- */
/* Selects the HIP device the CALLING THREAD renders on (hipSetDevice) and sets the library up on it.  Optional: the
 * first render of a thread that never called it uses device 0.  Every (scene, device) pair has its own streams and
 * work buffers, so distinct scenes may be rendered from different threads, on the same or on different devices; one
 * scene must not be rendered from two threads at once. */
int  rt_init(int device);
const char *rt_last_error(void);
const char *rt_version(void);

/* Dialect of the scene: RT_DIALECT_MAIN = src/Main.hs (default); RT_DIALECT_TALK = the smaller program src/Talk.hs:
 * normalize = scalar (recip (norm v)) v (Talk.hs:66-67), rectangle accepts D.(i - o) >= 0 (:119-128), nearest by
 * squared distance and no bounding-box guards (:185-194), pointLight intensity lp = round (min 1 (max 0 (L.N) *
 * intensity) * colour), shadowed iff (lp - p').L >= 0 (:237-248; pass the intensity as `diffuse` of
 * rt_add_point_light), no reflection (traceAndShade, :255-258).  Rectangles only.  Call before the first shape. */
enum { RT_DIALECT_MAIN = 0, RT_DIALECT_TALK = 1 };
int  rt_scene_set_dialect(rt_scene *, int dialect);

/* Per-scene options (no counterpart in the reference; nothing is read from the environment).
 *   RT_OPT_CULL        1 (default): a wave skips the blocks of 32 leaves whose bounding cone misses the bound of its
 *                      256 rays; 0: every wave scans every block (the plain all-pairs scan).  Same pixels either way.
 *   RT_OPT_FILTER      0 (default): first stage on the matrix cores; 1: the FP32 vector-ALU kernels, which exist only
 *                      in the cross-check build libraytrace_gpu_valu.so (RT_ERR_UNSUPPORTED in the product library).
 *   RT_OPT_HOST_BANDS  bands of rt_render, 1..8 (0 = default: 2 or 3, one per 8 Mi pixels beyond 24 Mi).  Consecutive bands run on
 *                      the scene's two lanes, so band k + 1's kernels fill the tails of band k's; a band's hit IDs
 *                      travel to the host as soon as its nearest-hit kernel has ended, its colours beside the next band.
 *   RT_OPT_MULTI_STAGE 1: rt_render_multi sends the bands of devices[0] through a staging buffer and a peer copy too,
 *                      like those of the other devices (default 0: they are rendered in place; tests use 1).
 *   RT_OPT_LANE        A scene has two lanes per device: a set of work buffers and a stream each.  Kernels of two lanes run
 *                      beside each other and fill each other's tails.  -1 (default): rt_render_device renders a band of
 *                      12 Mi pixels or more — 3 Mi when reflected rays are traced — as two halves, one per lane (on the
 *                      lanes' own streams, forked from and joined to the stream it is given; same pixels, 7 - 17 % faster).  0 or 1: the following rt_render_device calls use that lane and are not
 *                      split: a caller who cuts a frame into pieces renders them alternately on lane 0 and lane 1
 *                      (stream = NULL: the lane's own stream, or two streams of the caller's) and they overlap on the
 *                      device.  Two calls in flight on the SAME lane must be on the same stream (they share work buffers).
 *   RT_OPT_CONTINUE    1: the following rt_render_device calls add to the frame in progress as far as rt_get_stats is
 *                      concerned (counts and times of all its pieces, both lanes); 0 (default): a call on lane 0 begins a
 *                      new frame, a call on lane 1 adds to it.
 *   RT_OPT_HISTORY     1 (default): every wave of the two big kernels leaves its duration behind, and the next render of
 *                      the same band of the same scene dispatches the waves longest first, which shortens the kernels'
 *                      tails (a few hundred KB of device memory per band; the no-alloc entry point keeps none).  The order
 *                      decides when a wave runs, never what it computes.  0: launch order.
 *   RT_OPT_RGB24       1: every colour output of the render calls (`rgba_out`, `d_rgba`) is 3 bytes per pixel, R G B —
 *                      the reference's PixelRGB8 (Main.hs:621), what rt_write_bmp and a 24-bit BMP take — instead of
 *                      4 (R G B 255, default 0): a quarter less to carry over PCIe and between devices.  Buffers are
 *                      then 3 * w * nrows bytes.  Same colours. */
enum { RT_OPT_CULL = 0, RT_OPT_FILTER = 1, RT_OPT_HOST_BANDS = 2, RT_OPT_MULTI_STAGE = 3, RT_OPT_LANE = 4, RT_OPT_CONTINUE = 5, RT_OPT_HISTORY = 6, RT_OPT_RGB24 = 7 };
int  rt_scene_set_option(rt_scene *, int option, int value);

/* ---- scene construction (replaces the closure-building constructors) -------------------- */
int  rt_scene_create(rt_scene **out);
void rt_scene_destroy(rt_scene *);
/* rectangle material point width height                                   Main.hs:226-246 */
int  rt_add_rectangle(rt_scene *, const rt_material *, const double point[3],
                      const double width[3], const double height[3]);
/* triangle material pa pb pc                                              Main.hs:363-385 */
int  rt_add_triangle(rt_scene *, const rt_material *, const double pa[3],
                     const double pb[3], const double pc[3]);
/* sphere material center radius                                           Main.hs:392-415 */
int  rt_add_sphere(rt_scene *, const rt_material *, const double center[3], double radius);
/* colorcuboid materials point l h d (cube/colorcube/cuboid are special cases) Main.hs:286-320 */
int  rt_add_cuboid(rt_scene *, const rt_material *mats, int nmats, const double point[3],
                   double l, double h, double d);
/* cylinder topM botM mantleM point n h r                                  Main.hs:725-774 */
int  rt_add_cylinder(rt_scene *, const rt_material *top, const rt_material *bot,
                     const rt_material *mantle, const double point[3], int n, double h, double r);
/* Tree structure: the reference's Shape is a tree of mappend nodes (Main.hs:205-216), and every
 * mappend node guards its subtree with a bounding-box test (Main.hs:209, 152-174).  The boxes of
 * rectangles whose two edge vectors share an axis plane are too small (Main.hs:243-245 takes only
 * two opposite corners), so the guards clip such rectangles — visible in the reference's output and
 * reproduced here.  Which box guards a leaf depends on the tree, so the tree is passed as brackets:
 *   mconcat [a, b, c]   =  rt_begin_mconcat; a; b; c; rt_end_mconcat      (foldr mappend mempty)
 *   a `mappend` b       =  rt_begin_mappend; a; b; rt_end_mappend         (exactly two children)
 * where a, b, c are rt_add_* calls or nested brackets.  rt_add_cuboid and rt_add_cylinder emit their
 * own inner structure.  Children added outside any bracket form an implicit top-level mconcat,
 * unless there is exactly one, which then is the world itself. */
int  rt_begin_mconcat(rt_scene *);
int  rt_end_mconcat(rt_scene *);
int  rt_begin_mappend(rt_scene *);
int  rt_end_mappend(rt_scene *);
/* the `normals` wrapper of linearInterpolation                            Main.hs:956-965 */
int  rt_set_normal_mode(rt_scene *, int first_leaf, int nleaves, int mode,
                        const double *params, int nparams);
/* the `normals` wrapper with an arbitrary fnorm (e.g. the sine-cosine surface, Main.hs:599-605): leaves
 * [first_leaf, first_leaf + nleaves) take their shading normal from `fn` (RT_NORMAL_CALLBACK); params =
 * {scaleInv, offset x,y,z} as for RT_NORMAL_GAUSSIAN.  The render pauses after the hit points are known, hands the
 * hit points of those leaves to `fn` on the host and continues with the returned normals, so the closure itself
 * is evaluated and parity stays exact. */
int  rt_set_normal_callback(rt_scene *, int first_leaf, int nleaves, const double params[4],
                            rt_normal_fn fn, void *user);
/* pointLight world diffuse specular lpos (world = this scene)             Main.hs:467-487 */
int  rt_add_point_light(rt_scene *, double diffuse, double specular, const double pos[3]);
/* ambient f                                                               Main.hs:490-492 */
int  rt_add_ambient(rt_scene *, double f);

/* ---- introspection (host logic tests; no device needed) --------------------------------- */
/* `pointLight world kd ks pos` (Main.hs:467-474) closes over the Shape its shadow rays are cast against, and that Shape need
 * not be the one that is traced (every scene of the reference passes the traced world; rt_add_point_light does that).
 * A shadow world is a second shape tree: between rt_begin_shadow_world (returns its number, >= 1) and rt_end_shadow_world
 * the rt_add_* / rt_begin_* calls build it instead of the traced world; rt_add_point_light_world names the world (0 = the
 * traced one) a light tests its shadow rays against.  The traced world must be complete before the first shadow world
 * begins: hit IDs stay its depth-first leaf order.  Shadow-world shapes are never seen by camera or reflected rays. */
int  rt_begin_shadow_world(rt_scene *);
int  rt_end_shadow_world(rt_scene *);
int  rt_add_point_light_world(rt_scene *, int world, double diffuse, double specular, const double pos[3]);

int  rt_scene_num_leaves(const rt_scene *);
int  rt_scene_num_lights(const rt_scene *);
/* out[0..19]: kind(0 rect,1 tri,2 sphere), normal mode, normal xyz, d, p0 xyz, e1 xyz, e2 xyz, c0..c3 */
int  rt_scene_get_leaf(const rt_scene *, int leaf, double out[20]);
/* Host-side preparation of everything a w x h render uploads: guard-box chains (fails on an open
 * bracket) and the FP32 filter records.  Returns the number of guard boxes.  No device needed. */
int  rt_scene_prepare_host(rt_scene *, int w, int h);
/* Filter record of `leaf` for apex 0 (the eye) or 1 + k (the k-th point light), after
 * rt_scene_prepare_host: out[0..15] as laid out in raytrace_amd/csrc/rt_types.h (FilterRec). */
int  rt_scene_get_filter_record(const rt_scene *, int apex, int leaf, float out[16]);
/* First-stage row of `leaf` (matrix-core kernels) for apex 0 / 1 + k, after rt_scene_prepare_host: 8 IEEE halves
 * = 2^10 (a_x, a_y, a_z, margin - cos theta) of the leaf's bounding cone as hi/mid pairs (rt_types.h).  Returns
 * the leaf's position in the stream. */
int  rt_scene_get_mfma_row(const rt_scene *, int apex, int leaf, uint16_t out[8]);
/* Bounding cone of block `block` (32 consecutive stream positions) of that apex's stream: {axis xyz, cos T, sin T,
 * code, 0, 0}; code 1 / 2 (zero axis, cos T = 1e30 / -1e30) = a block no ray / every ray has to look at.  A wave
 * skips the block when the cone misses the bound of its 256 ray directions.  Returns the number of blocks. */
int  rt_scene_get_block_cone(const rt_scene *, int apex, int block, float out[8]);
/* Reflected rays (trace depth > 1, Main.hs:546-556) have no common apex: their leaf records are kept in a space-filling
 * order per leaf kind (0 rectangles, 1 triangles), in blocks of 32 with a bounding sphere each, and a run of 64 reflected
 * rays skips the blocks its bound cannot reach.  Block `block` of that order, after rt_scene_prepare_host: sphere =
 * {centre xyz, radius} (radius inf: always looked at), leaves = its leaf indices (-1: none).  Returns the number of blocks
 * (0 for a scene without reflective materials: nothing is built). */
int  rt_scene_get_general_block(const rt_scene *, int kind, int block, float sphere[4], int32_t leaves[32]);
/* The matrix-core rows of the same block (stage 0 of the bounce levels): 32 rows x 32 halfs (IEEE binary16 bit patterns,
   K index = 16 step + 8 half + slot), row m = the bounding sphere of leaf m of the block as a linear form in the ray's
   vector, written in the frame {centre xyz, radius} of the block's super-block (rt_scene.cpp: sphere_row; DESIGN.md
   section 3).  Introspection for the CPU tests of the filter; returns the number of blocks. */
int  rt_scene_get_general_rows(const rt_scene *, int kind, int block, float frame[4], uint16_t out[1024]);
/* Guard chain of `leaf`: writes up to cap boxes {x0,y0,z0,x1,y1,z1}, innermost first; returns the count. */
int  rt_scene_get_guards(const rt_scene *, int leaf, double *out, int cap);
/* fixedCamera w h: out = {scaleX, scaleY, dX, dY, d}                      Main.hs:506-527 */
int  rt_camera_params(int w, int h, double out[5]);

/* ---- render: parallelTrace w h (fixedCamera w h) (trace depth lights world) -------------- */
/* Rows [row0, row0+nrows) of the w x h image (y = 0 is the top row, Main.hs:629-632).
 *   rgba_out   nrows*w*4 bytes, R,G,B,255 per pixel, row-major            (nullable)
 *   hit_id_out nrows*w int32, leaf index of the primary hit, -1 = miss    (nullable)
 * spp = 1 is the reference (one ray through the integer pixel coordinate, Main.hs:529-530);
 * spp > 1 (jittered supersampling, seed) is an extension documented in DESIGN.md.
 * rt_render takes HOST buffers (what a Haskell caller hands over); rt_render_device takes DEVICE
 * pointers on the calling thread's device and an optional hipStream_t, and does not synchronise.
 * stream = NULL means the scene's OWN stream on that device, which is created non-blocking: it does
 * NOT synchronise with the legacy default stream, so a caller that consumes the frame on another stream
 * must either pass that stream here or call rt_synchronize() first. */
int  rt_render(rt_scene *, int w, int h, int depth, int spp, uint64_t seed,
               int row0, int nrows, uint8_t *rgba_out, int32_t *hit_id_out);
int  rt_render_device(rt_scene *, int w, int h, int depth, int spp, uint64_t seed,
                      int row0, int nrows, void *d_rgba, void *d_hit_id, void *stream);
/* Makes the scene resident on the calling thread's device for w x h renders: builds the records on the host if the
 * scene changed (rt_get_stats reports that time as prepare_ms) and uploads what is stale.  Renders do this
 * themselves; call it to keep that work out of a timed region. */
int  rt_scene_prepare_device(rt_scene *, int w, int h);
/* parallelTrace for a whole image on `ngpus` devices of this process at once (Main.hs:623-637 in ONE call, for a
 * caller that is a single process, e.g. the Haskell program): the image is cut into contiguous row bands of about
 * equal cost (rt_split_rows over a 1/16-scale pre-render on devices[0]), band k is rendered on devices[k] (NULL:
 * devices 0..ngpus-1; a device may be named more than once), copied to devices[0] over xGMI (peer-to-peer DMA) as
 * its pieces finish, and from there to the caller's HOST buffers band by band.  Same pixels as rt_render. */
int  rt_render_multi(rt_scene *, int w, int h, int depth, int spp, uint64_t seed, int ngpus, const int *devices,
                     uint8_t *rgba_out, int32_t *hit_id_out);
/* Relative cost of `ncost` equal horizontal strips of the w x h frame on the calling thread's device, measured with
 * HIP events on a reduced-scale render of each strip (what rt_render_multi and bench.py cut their bands from). */
int  rt_scene_row_costs(rt_scene *, int w, int h, int depth, int spp, int ncost, double *cost);
/* The band splitter of rt_render_multi (and of raytrace_amd/distributed.py, which must agree with it): n contiguous
 * bands (row0[k], nrows[k]) of [0, h), cut at multiples of `align` rows, of about equal summed cost; `cost` is a
 * per-row estimate at any vertical resolution (ncost entries cover [0, h)).  No device needed. */
int  rt_split_rows(const double *cost, int ncost, int h, int n, int align, int *row0, int *nrows);
/* The same with unequal shares: band k gets weights[k] / sum(weights) of the cost (NULL: equal).  The device that collects
 * the bands sends nothing and so may render longer than the ones whose last piece still has to travel: rt_render_multi and
 * bench.py give it the weight their pipeline model works out (raytrace_amd/distributed.py: plan_frame). */
int  rt_split_rows_weighted(const double *cost, int ncost, int h, int n, int align, const double *weights, int *row0, int *nrows);
/* rt_render_device that ALLOCATES NOTHING.  No counterpart in the reference: it is for callers that cannot call
 * hipMalloc — code under HIP graph capture, tight serving loops, harnesses that forbid allocation.  Same arguments and
 * same pixels as rt_render_device; differences: the scene must already be resident on the calling thread's device for
 * this image size (rt_scene_prepare_device; RT_ERR_ARG otherwise), rt_init must have been called on this thread, and
 * every work buffer comes from a 64 MiB array in static device memory that each GPU the library runs on holds until
 * the process ends.  A band whose buffers exceed it is rendered as consecutive sub-bands that fit (multiples of 16 rows,
 * single rows if need be); an image so wide that ONE row of it does not fit is refused with RT_ERR_UNSUPPORTED, as are
 * scenes with fnorm callbacks.  No kernel times in rt_get_stats (no events are created).  All calls on one GPU share
 * the array: calls that overlap in time on one GPU race, whatever their streams. */
int  rt_render_device_noalloc(rt_scene *, int w, int h, int depth, int spp, uint64_t seed,
                              int row0, int nrows, void *d_rgba, void *d_hit_id, void *stream);
int  rt_synchronize(void);                      /* waits for everything enqueued on the calling thread's device */
/* Page-locked host memory for rt_render's output buffers, in place of the reference's storable-vector
 * allocation behind `generateImage` (Main.hs:621 builds the image rt_render now fills).  Optional: any
 * host pointer works, but into these buffers the hit IDs travel by DMA beside the shading kernels and the
 * last kernel stores the colours straight into them over PCIe, so no copy is left behind the render
 * (DESIGN.md section 8).  NULL on failure (rt_last_error).  Free with rt_host_free only. */
void *rt_host_alloc(size_t bytes);
void  rt_host_free(void *p);
int  rt_get_stats(rt_scene *, rt_stats *out);   /* synchronises; stats of the last render of this scene */

/* ---- saveBmpImage "trace.bmp" (ImageRGB8 ...)                           Main.hs:621 ------ */
/* 24-bit uncompressed BMP from an RGBA8 buffer, so that a harness runs without JuicyPixels. */
int  rt_write_bmp(const char *path, const uint8_t *rgba, int w, int h);

#ifdef __cplusplus
}
#endif
#endif
