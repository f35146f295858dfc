// rt_scene.h — host-side scene: the reference's Shape/Light values reified as data.
//
// The reference builds `world :: Shape` as a tree of closures (Main.hs:189-217) and `lights ::
// Light` as a sum of closures (Main.hs:423-441).  Here a scene is the flat list of leaves in the
// tree's depth-first order (that order decides equal-distance hits, Main.hs:213-214), the list of
// lights, and the per-(apex, leaf) FP32 filter records derived from them.
#pragma once
#include <string>
#include <vector>

#include "../../include/raytrace_gpu.h"
#include "rt_types.h"

namespace rt {

struct DeviceScene {          // device copies (HBM), owned by the scene
    LeafD *leaves = nullptr;
    AuxD *aux = nullptr;
    GuardD *guards = nullptr;
    MaterialD *mats = nullptr;
    LightD *lights = nullptr;
    FilterRec *tri_rec = nullptr;    // (1 + npoint) blocks of ntri records: eye, then each point light
    FilterRec *rect_rec = nullptr;   // (1 + npoint) blocks of nrect records
    int32_t *tri_gid = nullptr;      // per-type slot -> leaf index
    int32_t *rect_gid = nullptr;
    FilterRec *sph_rec = nullptr;    // (1 + npoint) blocks of nsph sphere records
    GenRec *gen_sph = nullptr;
    int32_t *sph_gid = nullptr;
    uint16_t *mf_rows = nullptr;     // MFMA row blocks (rt_types.h), per apex
    int32_t *mf_gid = nullptr;
    float *mf_cone = nullptr;        // per apex and block: {axis xyz, cos T, sin T, 0, 0, 0}
    FilterRec *mf_rec = nullptr;
    unsigned long long *counters = nullptr;  // 8 x u64
    // work buffers, grown on demand
    int32_t *hit_id = nullptr;  size_t hit_id_cap = 0;     // per band pixel
    uint32_t *accum = nullptr;  size_t accum_cap = 0;      // spp > 1: 3 x u32 per pixel
    uint8_t *rgba = nullptr;    size_t rgba_cap = 0;       // rt_render (host buffers) staging
    int32_t *hit_out = nullptr; size_t hit_out_cap = 0;    // rt_render (host buffers) staging
    char *hitrec = nullptr;     size_t hitrec_cap = 0;     // per chunk pixel: 48-byte HitRec (capacity in bytes)
    uint32_t *queue = nullptr;  size_t queue_cap = 0;      // shadow-job queues, npoint x chunk pixels
    uint32_t *qsparse = nullptr; size_t qsparse_cap = 0;   // k_setup's per-block runs of shadow jobs, before compaction
    unsigned *tcount = nullptr; size_t tcount_cap = 0;     // shadow jobs per (light, k_setup block) -> offsets
    uint32_t *lit = nullptr;    size_t lit_cap = 0;        // per chunk pixel: lit-by-light bit mask
    unsigned *qcount = nullptr;                            // shadow jobs per point light; [32..39] rays per bounce level
    GenRec *gen_tri = nullptr, *gen_rect = nullptr;        // general-ray filter records (reflective scenes only)
    // bounce levels 1..7 (reflective scenes only): rays, hit ids and colours per slot; [0] = colour of level 0
    RayRec *lvl_rays[8] = {};   size_t lvl_rays_cap[8] = {};
    int32_t *lvl_hit[8] = {};   size_t lvl_hit_cap[8] = {};
    uint32_t *lvl_col[8] = {};  size_t lvl_col_cap[8] = {};
    int device = -1;
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
    std::vector<int> top;                 // children outside any bracket
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
    std::vector<rt::GenRec> gen_tri, gen_rect, gen_sph;   // only filled when a material reflects
    std::vector<uint16_t> mf_rows;        // MFMA rows: (1 + npoint) x (mf_groups + 2) blocks of 32 rows x 8 halves
    std::vector<int32_t> mf_gid;          // per apex: MFMA stream position -> leaf index (-1: padding)
    std::vector<float> mf_cone;           // per apex and block: bounding cone of the block's leaves, 8 floats (rt_scene.cpp: block_cone)
    std::vector<rt::FilterRec> mf_rec;    // FP32 records in MFMA stream order, per apex; c[14] = kind, c[15] = leaf index (bits)
    int mf_groups = 0;
    bool reflective = false;
    int rec_w = -1, rec_h = -1;
    bool dirty = true;                    // host scene changed since the last upload
    rt::DeviceScene dev;
    rt_stats stats{};
    std::vector<void *> evpool;           // hipEvent_t, one per kernel boundary of the last render
    std::vector<int> evkind;              // which kernel ended at that event (1 primary .. 4 final)
    int ev_used = 0;
    bool stats_pending = false;
    uint64_t last_primary = 0;
};

namespace rt {

void set_error(const std::string &msg);
CameraD fixed_camera(int w, int h);                       // Main.hs:506-527
double ray_eps();                                         // Main.hs:114
// (Re)build the FP32 filter records for the eye of a w x h camera and for every point light.
void build_filter_records(rt_scene &s, int w, int h);
// Compute every leaf's guard box from the bracket tree; false (and set_error) on an open bracket.
bool compute_guards(rt_scene &s);

} // namespace rt
