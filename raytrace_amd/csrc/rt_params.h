// rt_params.h — what the host runtime (rt_render.hip) and the kernels (rt_kernels.hip, rt_kernels_valu.hip) share: the
// launch geometry, the argument block of every kernel, the layout of the counters, and the launch functions through which
// the runtime starts kernels it does not see the definition of.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_scene.h"

#define RT_TILE_W 32
#define RT_TILE_H 8
// One wave per work-group: the wave slots of a 4-wave group only came free when its slowest wave had finished, and the
// waves of these kernels take between 10 and 200 us; measured on the benchmark frame, a third of the slots sat empty
// (tools/wave_times.py), 9 % of the frame time.
#ifndef RT_WAVES_PER_BLOCK
#define RT_WAVES_PER_BLOCK 1
#endif
#ifndef RT_SETUP_THREADS
#define RT_SETUP_THREADS 512
#endif
#ifndef RT_COMPACT_THREADS
#define RT_COMPACT_THREADS 128
#endif

namespace rt {

struct HitRec;
struct HitVec;
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
    HitVec *hitpos, *hitnrm;        // per chunk pixel: exact hit point; shading normal (two arrays: the shadow kernels gather points only)
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
    // Wave history of this chunk (Lane::history; nullptr: none kept, waves run in launch order).  Every wave of the two big
    // kernels leaves its duration (100 MHz ticks); the first work-groups of k_setup sort the waves by it, longest
    // first, and the next frame of the same band maps blockIdx through that order: the kernel's tail is then made of its
    // shortest waves instead of whichever came last (replayed through a list scheduler: 744 -> 587 us for the shadow kernel).
    // The order decides WHEN a wave runs, never what it computes; it is always a permutation of the launch's work-groups.
    uint32_t *wcost_p;              // per tile: duration of its k_primary_mfma wave in this frame
    const uint32_t *worder_p;       // work-group -> tile for k_primary_mfma (nullptr: identity; the frame before wrote it)
    uint32_t *worder_p_out;         // ... as sorted in this frame, for the next one
    uint32_t *wcost_s;              // per (light, wave of 256 jobs): duration of its k_shadow_mfma wave, wstride_s per light
    uint32_t *worder_s;             // work-group -> wave for k_shadow_mfma, per light (written by this frame's k_setup)
    int wstride_s;                  // waves per light = blockIdx.x range of k_shadow_mfma
    int whave;                      // 1: wcost_s holds the durations of the frame before (else the order is the identity)
    int wsort_blocks;               // k_setup's first work-groups sort the history: 1 + lights, or 0
    int bpp;                        // bytes per pixel of `rgba`: 4 (R, G, B, 255) or 3 (R, G, B: RT_OPT_RGB24, the reference's PixelRGB8)
    int rgb_packed;                 // bpp == 3 and w % 4 == 0: four pixels are stored as three whole dwords (store_rgb24)
};

struct HitRec { double px, py, pz, nx, ny, nz; };     // in registers; in memory: hitpos[pix], hitnrm[pix]
struct HitVec { double x, y, z; };

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
// Launches (defined beside the kernels).  Grids are in work-groups; every launch goes to `st`.
void launch_primary_mfma(const RenderParams &P, unsigned tblocks, hipStream_t st);
void launch_shadow_mfma(const RenderParams &P, unsigned sblocks, unsigned nlights, hipStream_t st);
void launch_setup(const RenderParams &P, unsigned ublocks, hipStream_t st);            // (+ P.wsort_blocks work-groups in front that sort the wave history)
void launch_queue_offsets(const RenderParams &P, unsigned nlights, hipStream_t st);
void launch_queue_compact(const RenderParams &P, unsigned ublocks, unsigned nlights, hipStream_t st);
void launch_final(const RenderParams &P, unsigned blocks, unsigned threads, hipStream_t st);
void launch_general(const RenderParams &P, unsigned runs, hipStream_t st);
void launch_combine(const RayRec *rays, const uint32_t *col, const unsigned *count, unsigned cap, uint32_t *parent_col, unsigned blocks, hipStream_t st);
void launch_store(const RenderParams &P, unsigned blocks, hipStream_t st);
void launch_resolve(const uint32_t *accum, uint8_t *rgba, size_t npix, uint32_t spp, int bpp, int rgb_packed, hipStream_t st);
void launch_cb_gather(const RenderParams &P, unsigned blocks, hipStream_t st);
#ifdef RT_VALU_FILTER       // the cross-check build's FP32 vector-ALU kernels (rt_kernels_valu.hip)
void launch_primary_valu(const RenderParams &P, bool shared_y, unsigned tblocks, hipStream_t st);
void launch_shadow_valu(const RenderParams &P, unsigned sblocks, unsigned nlights, hipStream_t st);
#endif
#ifdef RT_WAVE_TIMES
extern unsigned long long *g_wave_dbg;
#endif

} // namespace rt
