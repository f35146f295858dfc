/*
 * oracle/oracle.h — C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a CPU restatement (IEEE double, one source file, plain C)
 * of the reference ray tracer /root/reference/src/Main.hs.  It is the checker the GPU path is
 * compared against; it is never linked into, imported by or called from the product library
 * (raytrace_amd/, libraytrace_gpu.so).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden images, and no Haskell
 * toolchain exists in this image, so the oracle is pinned only by the known-answer values of
 * SURVEY.md §4 (tests/test_oracle_known_answers.py) and by line-by-line restatement.
 *
 * Scene model: the reference's `Shape` is a closure tree built with mappend / mconcat
 * (Main.hs:196-217).  The oracle keeps that tree (nodes), because every mappend node guards its
 * subtree with a bounding-box test (Main.hs:209) that can reject edge hits.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* Main.hs:178-185 */
typedef struct {
    uint8_t diffuse[3];
    uint8_t specular[3];
    uint8_t _pad[2];
    double  specularity;
    double  reflectivity;
} orc_material;

/* flags written by orc_render (one byte per pixel) */
#define ORC_FLAG_TIE   1u  /* two accepted leaf hits (guards ignored) have |t1-t2| <= 1e-5          */
#define ORC_FLAG_BBOX  2u  /* a bounding-box guard (Main.hs:209) changed the primary nearest hit   */
#define ORC_FLAG_BBOX_EDGE 8u /* the outcome of some bounding-box guard hinges on a comparison that is
                                within 1e-9 (relative) of equality: not reproducible across formulations */
#define ORC_FLAG_EDGE  4u  /* an accept/reject predicate of a leaf at or in front of the hit is    */
                           /* within 1e-5 of its boundary                                           */

orc_scene *orc_scene_new(void);
void       orc_scene_free(orc_scene *);

/* Switches the scene to the rules of src/Talk.hs (before the first shape): normalize = scalar (recip (norm v)) v
 * (Talk.hs:66-67), rectangle accepts D.(i - o) >= 0 (:119-128), nearest by squared distance and no bounding boxes
 * (:185-194), pointLight intensity = min 1 (max 0 (L.N) * intensity), shadow iff (lp - p').L >= 0 (:237-248; the
 * intensity is passed as `diffuse`), no reflection (:255-258).  Rectangles only. */
int orc_scene_set_talk(orc_scene *, int on);

/* Test switch (before the first shape): orc_mconcat builds the literal right-nested chain of mappend nodes ending in
 * mempty (foldr mappend mempty, Main.hs:196-217) instead of its list node with suffix boxes; same image either way. */
int orc_scene_set_literal_mconcat(orc_scene *, int on);

/* Shape constructors; every call returns a node handle (>= 0) or a negative error. */
int orc_mempty(orc_scene *);                                                        /* Main.hs:204 */
int orc_mappend(orc_scene *, int left, int right);                                  /* Main.hs:205-216 */
int orc_mconcat(orc_scene *, const int *children, int n);                           /* foldr mappend mempty */
int orc_rectangle(orc_scene *, const orc_material *, const double point[3],
                  const double width[3], const double height[3]);                   /* Main.hs:226-246 */
int orc_triangle(orc_scene *, const orc_material *, const double pa[3],
                 const double pb[3], const double pc[3]);                           /* Main.hs:363-385 */
int orc_sphere(orc_scene *, const orc_material *, const double center[3], double radius); /* Main.hs:392-415 */
int orc_colorcuboid(orc_scene *, const orc_material *mats, int nmats,
                    const double point[3], double l, double h, double d);           /* Main.hs:301-320 */
int orc_cylinder(orc_scene *, const orc_material *top, const orc_material *bot,
                 const orc_material *mantle, const double point[3], int n,
                 double h, double r);                                               /* Main.hs:725-774 */
/* `normals` wrapper of linearInterpolation with the gaussian fnorm of main (Main.hs:580-584, 956-965) */
int orc_normals_gaussian(orc_scene *, int child, double scale_inv, const double offset[3]);
/* the same wrapper with any fnorm (Main.hs:956-965): out[0..2] = fnorm x y; called from the render threads */
typedef void (*orc_fnorm)(double x, double y, double out[3], void *user);
int orc_normals_callback(orc_scene *, int child, double scale_inv, const double offset[3], orc_fnorm fn, void *user);
/* linearInterpolation with f = gaussian, fnorm = Just gaussian normal (Main.hs:578-585, 937-988).
 * Returns the world node; adds its point light and ambient light to the scene's light list. */
int orc_linear_interpolation_gaussian(orc_scene *, double x1, double y1, double x2, double y2,
                                      double step, const double origin[3], double scale);

/* Lights (the scene's light list is `mconcat` of everything added, Main.hs:436-441). */
int orc_point_light(orc_scene *, double diffuse, double specular, const double pos[3]); /* Main.hs:467-487 */
int orc_point_light_world(orc_scene *, int world, double diffuse, double specular, const double pos[3]); /* ... closing over `world` (a node) */
int orc_ambient(orc_scene *, double f);                                                 /* Main.hs:490-492 */

/* Number of leaves below `root` and leaf order (depth-first, left to right) */
int orc_num_leaves(orc_scene *, int root);

/*
 * parallelTrace w h (fixedCamera w h) (trace depth lights world) for rows [row0, row0+nrows)
 * (Main.hs:623-637, 506-533, 541-557).  Point lights cast their shadow rays against `root`.
 *   rgb     nrows*w*3 bytes, row-major, y = row0 is first           (required)
 *   hit_id  nrows*w int32: depth-first leaf index of the primary hit, -1 = miss   (nullable)
 *   hit_t   nrows*w double: primary hit distance, +inf on a miss     (nullable)
 *   flags   nrows*w bytes of ORC_FLAG_*                              (nullable; costs ~3x)
 *   spp/seed: spp == 1 is the reference (one ray through the integer pixel coordinate,
 *   Main.hs:529-530).  spp > 1 is this repo's extension (jittered supersampling).
 *   use_bbox: 1 = reference behaviour; 0 = ignore the bounding-box guards.
 * Returns 0 or a negative error.
 */
int orc_render(orc_scene *, int root, int w, int h, int depth, int spp, uint64_t seed,
               int row0, int nrows, int nthreads, int use_bbox,
               uint8_t *rgb, int32_t *hit_id, double *hit_t, uint8_t *flags);

/* Known-answer probes */
double orc_eps(void);                                              /* Main.hs:114 */
void   orc_camera_params(int w, int h, double out[5]);             /* scaleX scaleY dX dY d, Main.hs:516-527 */
void   orc_camera_ray(int w, int h, int x, int y, double out[9]);  /* position, direction, recip */
int    orc_grid_steps(double a, double b, double step, double *out, int cap); /* Main.hs:948 */

#ifdef __cplusplus
}
#endif
#endif
