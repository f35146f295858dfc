// rt_scene.h — host-side scene: the reference's Shape/Light values reified as data.
//
// The reference builds `world :: Shape` as a tree of closures (Main.hs:189-217) and `lights ::
// Light` as a sum of closures (Main.hs:423-441).  Here a scene is the flat list of leaves in the
// tree's depth-first order (that order decides equal-distance hits, Main.hs:213-214), the list of
// lights, and the per-(apex, leaf) FP32 filter records derived from them.
#pragma once
#include <array>
#include <map>
#include <string>
#include <vector>

#include "../../include/raytrace_gpu.h"
#include "rt_types.h"

namespace rt {

constexpr int RT_MAX_DEVICES = 16;       // HIP devices one process may render on (rt_init / rt_render_multi)
constexpr int RT_HOST_BANDS_MAX = 8;     // copy bands of rt_render
constexpr int RT_MAX_LEVELS = 64;        // bounce levels kept for the fold of the colours (trace depth with mirrors)

// One uploaded array (a view of DeviceScene::arena) or one allocation with its capacity.
struct DevBuf { void *p = nullptr; size_t cap = 0; };

// Device memory of one render beyond its inputs and outputs.  rt_render / rt_render_device grow it on demand
// (hipMalloc, kept in the DeviceScene); rt_render_device_noalloc carves it out of the library's static device
// array instead (rt_kernels.hip: `scratch`).
struct Workspace {
    unsigned long long *counters = nullptr;  // RT_STAT_SLOTS x 16 x u64
    unsigned *qcount = nullptr;              // shadow jobs per point light [0..31]; rays per bounce level [32..96]; callback count; lights' cost history
    int32_t *hit_id = nullptr;               // per band pixel (only when the caller gives no buffer, or spp > 1)
    uint32_t *accum = nullptr;               // spp > 1: 3 x u32 per band pixel
    char *hitrec = nullptr;                  // per chunk pixel: 24-byte hit points, then (hit_slots on) 24-byte normals
    size_t hit_slots = 0;
    uint32_t *queue = nullptr;               // shadow-job queues, npoint x chunk pixels
    uint32_t *qsparse = nullptr;             // k_setup's per-block runs of shadow jobs, before compaction
    unsigned *tcount = nullptr;              // shadow jobs per (light, k_setup block) -> offsets
    uint32_t *lit = nullptr;                 // per chunk pixel: lit-by-light bit mask
    RayRec *lvl_rays[RT_MAX_LEVELS] = {};    // bounce levels 1.. (reflective scenes only): rays, hit ids, colours per slot
    int32_t *lvl_hit[RT_MAX_LEVELS] = {};
    uint32_t *lvl_col[RT_MAX_LEVELS] = {};   // [0] = colour of level 0
    uint32_t *cb_slot = nullptr;             // RT_NORMAL_CALLBACK: per chunk slot, index into the list of hit points
    double *cb_xy = nullptr;                 // per list entry: the arguments of fnorm
    int32_t *cb_group = nullptr;             // per list entry: which callback
    double *cb_normal = nullptr;             // per list entry: fnorm x y (from the host)
};

// One set of work buffers and the stream the renders that use it go to.  A (scene, device) pair has two: consecutive row
// bands of a frame (rt_render's copy bands, rt_render_multi's pieces, a caller's own pieces: RT_OPT_LANE) alternate
// between them, so that a band's kernels fill the tails of the other's and its copy to the host or to a peer device
// runs beside the other band's kernels.
// Wave history of one band (the same rows of the same frame of the same scene, rendered again and again): durations and
// dispatch order of the waves of k_primary_mfma and k_shadow_mfma, per chunk (RenderParams::wcost_p ...).
struct WaveHistory {
    void *buf = nullptr;             // ONE allocation: chunks x 2 x words
    size_t words = 0;                // per chunk and per array (cost, order): tiles + lights x shadow waves
    bool ready = false;              // a whole frame has been enqueued with it: its orders are permutations
};
struct Lane {
    std::map<std::array<uint64_t, 6>, WaveHistory> history;   // by (w, h, row0, nrows, chunk rows, lights); dropped when the scene changes
    uint64_t history_version = 0;        // rt_scene::version the histories belong to
    Workspace ws;                    // views of ws_arena (ws_grow)
    DevBuf ws_arena;                 // ONE allocation
    const unsigned long long *last_counters = nullptr;     // where the counters of this lane's part of the last frame live
    std::vector<void *> evpool;      // hipEvent_t, one per kernel boundary of this lane's part of the last frame
    std::vector<int> evkind;         // which kernel ended at that event (1 primary .. 4 final)
    int ev_used = 0;
    bool used = false;               // by the last frame
};
#ifndef RT_LANES_N
#define RT_LANES_N 2
#endif
constexpr int RT_LANES = RT_LANES_N;

struct DeviceScene {          // the scene on ONE device (HBM), owned by the rt_scene
    int device = -1;
    uint64_t version = 0;            // rt_scene::version of the uploaded copy (0: nothing uploaded)
    uint64_t eye_version = 0;        // rt_scene::eye_version of the uploaded eye-apex records
    DevBuf arena;                    // ONE allocation: every array below is a view of it (device_prepare)
    DevBuf leaves, aux, guards, mats, lights;
    DevBuf tri_rec, rect_rec;        // (1 + npoint) blocks of ntri / nrect records: eye, then each point light (VALU filter build)
    DevBuf tri_gid, rect_gid;        // per-type slot -> leaf index
    DevBuf sph_rec, gen_sph, sph_gid;
    DevBuf mf_rows, mf_gid, mf_cone, mf_rec;   // MFMA stream per apex (rt_types.h)
    DevBuf gen_tri, gen_rect;        // general-ray records (reflective scenes only), in a space-filling order
    DevBuf gen_tri_gid, gen_rect_gid, gen_tri_sph, gen_rect_sph;   // that order's leaf indices; bounding spheres of its blocks of 32
    DevBuf gen_tri_rows, gen_rect_rows;                            // the blocks' matrix-core rows (leaf bounding spheres as linear forms)
    DevBuf gen_tri_ssph, gen_rect_ssph;                            // super-blocks of 32 blocks: bounding sphere = the frame of their rows
    void *helper = nullptr;          // rt_kernels.hip: Helper — the thread that makes the streams beside the first host index build
    void *stream = nullptr;          // hipStream_t: renders of this scene on this device (when the caller passes none), lane 0
    void *fork_ev = nullptr, *join_evs[RT_LANES] = {};   // hipEvent_t: rt_render_device's own split of a big band over lanes 0 and 1
    void *xstream[RT_LANES] = {};    // hipStream_t: the same for lanes 1.. (entry 0 stays empty: lane 0 runs on `stream`)
    void *copy_stream = nullptr;     // hipStream_t: rt_render's copies to host memory, band by band; rt_render_multi's peer copies
    void *copy_stream2 = nullptr;    // hipStream_t: rt_render_multi's copies from the home device to host memory
    void *band_done[RT_HOST_BANDS_MAX] = {};   // hipEvent_t
    void *hits_done[RT_HOST_BANDS_MAX] = {};   // hipEvent_t: the band's hit-ID buffer is complete (rt_render copies it early)
    // work buffers of rt_render / rt_render_device, grown on demand
    Lane lane[RT_LANES];
    uint8_t *rgba = nullptr;    size_t rgba_cap = 0;       // rt_render (host buffers) staging
    int32_t *hit_out = nullptr; size_t hit_out_cap = 0;
    uint8_t *stage_rgba = nullptr; size_t stage_rgba_cap = 0;   // rt_render_multi: this device's bands before the peer copy
    int32_t *stage_hit = nullptr;  size_t stage_hit_cap = 0;
    // statistics of the last render on this device
    bool stats_pending = false;
    uint64_t last_primary = 0;
    rt_stats stats{};
};

} // namespace rt

namespace rt {
struct Box { double lo[3], hi[3]; };                     // BoundingBox, Main.hs:134-139
struct GNode { int kind; int leaf; std::vector<int> kids; };   // kind: 0 leaf, 1 mconcat, 2 mappend
}

struct rt_scene {
    std::vector<rt::LeafD> leaves;
    std::vector<rt::Box> leaf_bb;         // boundingBox of every leaf (Main.hs:243-245, 384)
    std::vector<rt::GNode> nodes;         // the mappend/mconcat tree
    std::vector<int> open;                // stack of open brackets (node indices)
    std::vector<int> top;                 // children outside any bracket (of the traced world)
    // `pointLight world ...` (Main.hs:467-474) closes over a Shape of its own: the world its shadow rays are cast against,
    // which need not be the traced one.  Shadow world k >= 1 is a second shape tree (rt_begin_shadow_world); its leaves live
    // in `leaves` behind the traced world's (hit IDs stay the traced world's depth-first order) and are seen only by the
    // lights that name it: every apex's streams are built from the leaves of ITS world.
    std::vector<std::vector<int>> shadow_top;   // children outside any bracket of shadow world k (index k - 1)
    std::vector<int32_t> leaf_world;      // per leaf: 0 = the traced world, k = shadow world k
    std::vector<int32_t> light_world;     // per light: the world its shadow rays meet (0: the traced world)
    int cur_world = 0;                    // the world shapes are being added to
    std::vector<rt::AuxD> aux;
    std::vector<rt::GuardD> guards;       // guard-box chains, rebuilt by compute_guards
    int guard_cached = 0;                 // guards [0, guard_cached): the boxes above the most leaves, parents first (<= 64)
    std::vector<rt::MaterialD> mats;
    std::vector<rt::LightD> lights;
    std::vector<int32_t> tri_gid, rect_gid, sph_gid;
    int npoint = 0;                       // number of point lights
    bool talk = false;                    // the rules of src/Talk.hs (rt_scene_set_dialect)

    // filter records, rebuilt when the camera (w, h) or the scene changes
    std::vector<rt::FilterRec> tri_rec, rect_rec, sph_rec;
    std::vector<rt::GenRec> gen_tri, gen_rect, gen_sph;   // only filled when a material reflects; gen_tri / gen_rect in spatial order
    std::vector<int32_t> gen_tri_gid, gen_rect_gid;       // position in that order -> leaf index
    std::vector<float> gen_tri_sph, gen_rect_sph;         // per block of 32 records: bounding sphere {centre xyz, radius}
    std::vector<uint16_t> gen_tri_rows, gen_rect_rows;    // per block: 2 steps x 2 halves x 32 x 8 halfs (rt_scene.cpp: sphere_row), + one padding block
    std::vector<float> gen_tri_ssph, gen_rect_ssph;       // per super-block of 32 blocks: bounding sphere
    std::vector<uint16_t> mf_rows;        // MFMA rows: (1 + npoint) x (mf_groups + 2) blocks of 32 rows x 8 halves
    std::vector<int32_t> mf_gid;          // per apex: MFMA stream position -> leaf index (-1: padding)
    std::vector<float> mf_cone;           // per apex and block: bounding cone of the block's leaves, 8 floats (rt_scene.cpp: block_cone)
    std::vector<rt::FilterRec> mf_rec;    // FP32 records in MFMA stream order, per apex; c[14] = kind, c[15] = leaf index (bits)
    int mf_groups = 0;
    double scene_lo[3] = {0, 0, 0}, scene_hi[3] = {0, 0, 0};   // extent of the leaves (build_filter_records)
    bool reflective = false;
    int rec_w = -1, rec_h = -1;           // image size of the last host preparation (-1: none yet)
    double eye_dmax[3] = {0, 0, 0};       // bound on |posX|, |posY|, |d| the eye-apex records were built for
    bool dirty = true;                    // host scene changed since the records were built
    uint64_t version = 0;                 // bumped whenever the records are rebuilt from a changed scene
    uint64_t eye_version = 0;             // bumped whenever the eye-apex records are rebuilt (scene or camera bound changed)
    double prepare_ms = 0;                // host time of the last rebuild of the records (guards, cones, sort)
    // per-scene options (rt_scene_set_option)
    int opt_cull = 1;                     // 0: every wave scans every block of the leaf stream
    int opt_filter = 0;                   // 0: matrix-core first stage; 1: FP32 vector-ALU kernels (test build only)
    int opt_host_bands = 0;               // copy bands of rt_render (0: 2, one per 8 Mi pixels beyond 24 Mi)
    int opt_multi_stage = 0;              // rt_render_multi: also the home device's bands go through a staging buffer + peer copy
    bool opt_continue = false;            // rt_render_device calls add to the statistics of the frame in progress
    bool opt_rgb24 = false;               // RT_OPT_RGB24: colour outputs are 3 bytes per pixel
    bool opt_history = true;              // RT_OPT_HISTORY: waves of a band rendered before are dispatched longest first
    int opt_lane = -1;                    // -1: rt_render_device splits big bands over both lanes itself; 0, 1: the lane its calls use
    // arbitrary `fnorm` closures (Main.hs:956-965): per callback group, the caller's function
    std::vector<rt_normal_fn> normal_fn;
    std::vector<void *> normal_user;
    rt::DeviceScene dev[rt::RT_MAX_DEVICES];
    int last_device = -1;                 // device of the last render (rt_get_stats)
    std::vector<double> multi_cost;       // rt_render_multi: measured strip costs, valid for multi_cost_key
    uint64_t multi_cost_key[5] = {0, 0, 0, 0, 0};   // version, w, h, depth, spp
    bool multi_stats = false;             // stats_multi holds the merged statistics of the last rt_render_multi
    rt_stats stats_multi{};
};

namespace rt {

void set_error(const std::string &msg);
const char *kernel_build_flags();                         // rt_kernels.hip: the measurement switches the kernels were compiled with ("" for a product build)
CameraD fixed_camera(int w, int h);                       // Main.hs:506-527
double ray_eps();                                         // Main.hs:114
// (Re)build what a w x h render uploads: guard chains and every apex's records when the scene changed, the eye
// apex's records alone when only the camera bound grew.  Returns false (and sets the error) on an open bracket.
bool prepare_host(rt_scene &s, int w, int h);
bool host_ready(const rt_scene &s, int w, int h);
// (Re)build the FP32 filter records for the eye of a w x h camera and for every point light.
void build_filter_records(rt_scene &s, int w, int h, bool eye_only = false);
// Contiguous row bands of about equal cost (C port of raytrace_amd/distributed.py: split_rows).
// weights (n positive numbers, or nullptr for equal ones): band k's share of the summed cost.
void split_rows(const double *cost, int ncost, int h, int n, int align, int *row0, int *nrows, const double *weights = nullptr);
// Compute every leaf's guard box from the bracket tree; false (and set_error) on an open bracket.
bool compute_guards(rt_scene &s);

} // namespace rt
