// rt_render.hip — the host runtime of the render path: devices and threads, the resident copy of a scene (one arena), the
// work buffers of a render (one arena, or the static array `scratch` for rt_render_device_noalloc), the order in which the
// kernels of a frame are enqueued (render_rows: parallelTrace / computeP, reference src/Main.hs:623-637, with trace's
// bounce levels :541-557), statistics, the multi-device render.  The kernels themselves are in rt_kernels.hip (and
// rt_kernels_valu.hip in the cross-check build) and are started through the launch functions of rt_params.h; the C ABI
// (rt_abi.hip) forwards to the *_impl functions below.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <type_traits>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rt_params.h"
#include "rt_render.h"

// Scratch of rt_render_device_noalloc (for callers that cannot allocate: HIP graph capture, tight serving loops):
// every buffer a render needs beyond its inputs and outputs is carved out of this array instead of hipMalloc'ed.
// Each GPU the library runs on holds its own copy, at its own address, until the process ends.  On the host the name
// is only a stand-in for HIP_SYMBOL(): the device address comes from hipGetSymbolAddress (device_init), once per device.
constexpr size_t RT_STATIC_SCRATCH_BYTES = 64u << 20;
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];

namespace rt {

#ifdef RT_WAVE_TIMES
unsigned long long *g_wave_dbg = nullptr;
#endif

namespace {

// ------------------------------------------------------------------------------------------
// Host side

// Library state per device and per thread.  Everything else lives in the (scene, device) pair: streams, events,
// uploaded records, work buffers — so distinct scenes can be rendered from different threads.
struct DeviceCtx {
    bool ready = false;
    unsigned char *scratch = nullptr;    // this device's copy of `scratch` (hipGetSymbolAddress, with the device current)
};
DeviceCtx g_ctx[RT_MAX_DEVICES];
std::mutex g_ctx_mutex;
thread_local int t_device = -1;          // the device this thread renders on (rt_init)

bool hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}
#define HIP_TRY(expr) do { if (!hip_ok((expr), #expr)) return RT_ERR_HIP; } while (0)


double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Sets the library up on `device` and makes it the calling thread's device.
int device_init(int device) {
    int n = 0;
    if (!hip_ok(hipGetDeviceCount(&n), "hipGetDeviceCount") || n <= 0) {
        set_error("rt_init: no HIP device available (this library has no CPU fallback)");
        return RT_ERR_HIP;
    }
    if (device < 0 || device >= n || device >= RT_MAX_DEVICES) { set_error("rt_init: device index out of range"); return RT_ERR_ARG; }
    HIP_TRY(hipSetDevice(device));
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        DeviceCtx &C = g_ctx[device];
        if (!C.ready) {
            unsigned char *ptr = nullptr;
            if (!hip_ok(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(scratch)") || !ptr) return RT_ERR_HIP;
            C.scratch = ptr;
            C.ready = true;
        }
    }
    t_device = device;
    return RT_OK;
}
// Every entry point that touches the device starts here: the thread's device is made current again (the caller may
// have switched it, e.g. through another library).
int enter() {
    if (t_device < 0) return device_init(0);
    HIP_TRY(hipSetDevice(t_device));
    return RT_OK;
}

// The resident copy of a scene is ONE allocation (DeviceScene::arena) that the arrays are views of: a first upload is one
// hipMalloc and a run of copies on the scene's stream instead of ~25 allocations and blocking copies (a one-shot caller
// pays for it: examples/cold_start.cpp).  plan(): size of the views, 256-byte aligned; place(): view + copy.
struct UploadPlan {
    size_t total = 0;
    template <typename T> void add(const std::vector<T> &src) { total = ((total + 255) & ~(size_t)255) + sizeof(T) * (src.empty() ? 1 : src.size()); }
};
template <typename T>
int place(DeviceScene &D, size_t &off, DevBuf &dst, const std::vector<T> &src, unsigned char *staging) {
    off = (off + 255) & ~(size_t)255;
    const size_t bytes = sizeof(T) * (src.empty() ? 1 : src.size());
    dst.p = (char *)D.arena.p + off; dst.cap = bytes;
    if (!src.empty()) std::memcpy(staging + off, src.data(), sizeof(T) * src.size());
    off += bytes;
    return RT_OK;
}
// elements [first, first + count) of an array that is already resident
template <typename T>
int upload_slice(DevBuf &dst, const std::vector<T> &src, size_t first, size_t count) {
    if (count == 0) return RT_OK;
    if (!dst.p || dst.cap < sizeof(T) * (first + count) || src.size() < first + count) { set_error("upload_slice: array not resident"); return RT_ERR_ARG; }
    HIP_TRY(hipMemcpy((char *)dst.p + sizeof(T) * first, src.data() + first, sizeof(T) * count, hipMemcpyHostToDevice));
    return RT_OK;
}

template <typename T>
int ensure(T *&buf, size_t &cap, size_t need) {
    if (cap >= need && buf) return RT_OK;
    if (buf) { HIP_TRY(hipFree(buf)); buf = nullptr; cap = 0; }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&buf), sizeof(T) * (need ? need : 1)));
    cap = need;
    return RT_OK;
}

// Makes the scene resident on the thread's device for w x h renders (rt_scene_prepare_device).
// What the runtime sets up when streams are first made costs a one-shot caller milliseconds (a stream = a hardware queue:
// 8 ms for the first of a process, 4.6 ms for the two copy streams and their events; examples/cold_start.cpp): a helper
// thread makes them while the calling thread builds the host index and uploads.  The render stream is waited for in
// device_prepare; the copy tools by the first call that needs them (copy_tools), or by rt_scene_destroy.
struct Helper { std::thread th; std::mutex m; std::condition_variable cv; bool stream_ready = false; hipError_t err = hipSuccess; };
void helper_join(DeviceScene &D) {
    Helper *H = static_cast<Helper *>(D.helper);
    if (!H) return;
    if (H->th.joinable()) H->th.join();
    delete H;
    D.helper = nullptr;
}
int make_copy_tools(DeviceScene &D) {
    if (!D.copy_stream) { hipStream_t st; HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); D.copy_stream = st; }
    if (!D.copy_stream2) { hipStream_t st; HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); D.copy_stream2 = st; }
    for (int k = 1; k < RT_LANES; ++k) if (!D.xstream[k]) { hipStream_t st; HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); D.xstream[k] = st; }
    for (void *&e : D.band_done) if (!e) { hipEvent_t ev; HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); e = ev; }
    for (void *&e : D.hits_done) if (!e) { hipEvent_t ev; HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); e = ev; }
    return RT_OK;
}

int device_prepare(rt_scene &s, DeviceScene &D, int device, int w, int h) {
    if (!D.stream && !D.helper && !host_ready(s, w, h)) {
        Helper *H = new Helper;
        D.helper = H;
        H->th = std::thread([&D, H, device] {
            hipStream_t st = nullptr;
            hipError_t e = hipSetDevice(device);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            {
                std::lock_guard<std::mutex> lock(H->m);
                if (e == hipSuccess) D.stream = st;
                H->err = e; H->stream_ready = true;
            }
            H->cv.notify_all();
            if (e == hipSuccess && make_copy_tools(D) != RT_OK) (void)hipGetLastError();     // (copy_tools() makes what is missing, and reports)
        });
    }
    D.device = device;              // (from here on rt_scene_destroy looks at this entry)
    const bool host_ok = prepare_host(s, w, h);
    if (Helper *H = static_cast<Helper *>(D.helper)) {
        std::unique_lock<std::mutex> lock(H->m);
        H->cv.wait(lock, [H] { return H->stream_ready; });
        if (H->err != hipSuccess) { const hipError_t e = H->err; lock.unlock(); helper_join(D); HIP_TRY(e); }
    }
    if (!host_ok) return RT_ERR_ARG;
    if (s.mf_groups + 2 > 65535) { set_error("rt_render: more than 2 million planar leaves are not supported (16-bit block indices)"); return RT_ERR_UNSUPPORTED; }
    if (!D.stream) { hipStream_t st; HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); D.stream = st; }
    if (D.version == s.version && D.eye_version == s.eye_version) return RT_OK;
    const double t0 = now_ms();
    int rc;
    // a render of the stale copy may still be in flight on some stream: the arrays are overwritten in place
    if (D.version != 0) HIP_TRY(hipDeviceSynchronize());
    const size_t nsph = s.sph_gid.size();
#ifdef RT_VALU_FILTER
    const size_t ntri = s.tri_gid.size(), nrect = s.rect_gid.size();
#endif
    const size_t gstride = (size_t)s.mf_groups + 2;
    if (D.version != s.version) {
        // every array of the scene, in one arena (grown with a quarter of slack so that a slightly bigger scene fits again)
        auto each = [&](auto &&f) -> int {
            int r;
            if ((r = f(D.leaves, s.leaves)) || (r = f(D.aux, s.aux)) || (r = f(D.guards, s.guards)) || (r = f(D.mats, s.mats)) ||
                (r = f(D.lights, s.lights)) || (r = f(D.tri_rec, s.tri_rec)) || (r = f(D.rect_rec, s.rect_rec)) ||
                (r = f(D.sph_rec, s.sph_rec)) || (r = f(D.sph_gid, s.sph_gid)) || (r = f(D.mf_rows, s.mf_rows)) ||
                (r = f(D.mf_gid, s.mf_gid)) || (r = f(D.mf_cone, s.mf_cone)) || (r = f(D.mf_rec, s.mf_rec)) ||
                (r = f(D.tri_gid, s.tri_gid)) || (r = f(D.rect_gid, s.rect_gid))) return r;
            if (s.reflective) {
                if ((r = f(D.gen_sph, s.gen_sph)) || (r = f(D.gen_tri, s.gen_tri)) || (r = f(D.gen_rect, s.gen_rect)) ||
                    (r = f(D.gen_tri_gid, s.gen_tri_gid)) || (r = f(D.gen_rect_gid, s.gen_rect_gid)) ||
                    (r = f(D.gen_tri_sph, s.gen_tri_sph)) || (r = f(D.gen_rect_sph, s.gen_rect_sph)) ||
                    (r = f(D.gen_tri_rows, s.gen_tri_rows)) || (r = f(D.gen_rect_rows, s.gen_rect_rows)) ||
                    (r = f(D.gen_tri_ssph, s.gen_tri_ssph)) || (r = f(D.gen_rect_ssph, s.gen_rect_ssph))) return r;
            }
            return RT_OK;
        };
        UploadPlan plan;
        (void)each([&](DevBuf &, const auto &src) { plan.add(src); return (int)RT_OK; });
        if (!D.arena.p || D.arena.cap < plan.total) {
            if (D.arena.p) { HIP_TRY(hipFree(D.arena.p)); D.arena.p = nullptr; D.arena.cap = 0; }
            const size_t cap = plan.total + plan.total / 4 + 4096;
            HIP_TRY(hipMalloc(&D.arena.p, cap));
            D.arena.cap = cap;
        }
        // the arrays are laid out in a host image of the arena and travel in ONE copy (25 copies out of pageable vectors
        // cost 8.7 ms for 10 MB: each one is staged and waited for on its own)
        size_t off = 0;
        unsigned char *image = nullptr;                   // page-locked: one DMA, no staging inside the runtime
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&image), plan.total, hipHostMallocDefault));
        rc = each([&](DevBuf &dst, const auto &src) { return place(D, off, dst, src, image); });
        if (rc == RT_OK && (!hip_ok(hipMemcpyAsync(D.arena.p, image, off, hipMemcpyHostToDevice, (hipStream_t)D.stream), "hipMemcpyAsync") ||
                            !hip_ok(hipStreamSynchronize((hipStream_t)D.stream), "hipStreamSynchronize"))) rc = RT_ERR_HIP;   // complete on return
        (void)hipHostFree(image);
        if (rc) return rc;
    } else {
        // only the camera bound grew: apex 0 (the eye) of every per-apex array
#ifdef RT_VALU_FILTER
        if ((rc = upload_slice(D.tri_rec, s.tri_rec, 0, ntri))) return rc;
        if ((rc = upload_slice(D.rect_rec, s.rect_rec, 0, nrect))) return rc;
#endif
        if ((rc = upload_slice(D.sph_rec, s.sph_rec, 0, nsph))) return rc;
        if ((rc = upload_slice(D.mf_rows, s.mf_rows, 0, gstride * 32 * 8))) return rc;
        if ((rc = upload_slice(D.mf_gid, s.mf_gid, 0, gstride * MF_LEAVES))) return rc;
        if ((rc = upload_slice(D.mf_cone, s.mf_cone, 0, gstride * 8))) return rc;
        if ((rc = upload_slice(D.mf_rec, s.mf_rec, 0, gstride * MF_LEAVES))) return rc;
    }
    D.version = s.version;
    D.eye_version = s.eye_version;
    D.stats.upload_ms = now_ms() - t0;
    return RT_OK;
}

// The streams and events of the copies to host memory (rt_render) and between devices (rt_render_multi).
int copy_tools(DeviceScene &D) {
    helper_join(D);                 // (it may be making them right now)
    return make_copy_tools(D);
}

// event pool: one event per kernel boundary, so that rt_get_stats can attribute time to each kernel
int next_event(Lane &D, hipStream_t st, int kind) {
    if (D.ev_used == (int)D.evpool.size()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        D.evpool.push_back(e);
        D.evkind.push_back(0);
    }
    D.evkind[D.ev_used] = kind;
    HIP_TRY(hipEventRecord((hipEvent_t)D.evpool[D.ev_used], st));
    ++D.ev_used;
    return RT_OK;
}
// the stream of a lane (lane 1's is made on first use)
int lane_stream(DeviceScene &D, int lane, hipStream_t &out) {
    if (lane == 0) { out = (hipStream_t)D.stream; return RT_OK; }
    helper_join(D);                 // (the helper thread of rt_scene_prepare_device may be making it right now)
    if (!D.xstream[lane]) { hipStream_t st; HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); D.xstream[lane] = st; }
    out = (hipStream_t)D.xstream[lane];
    return RT_OK;
}

inline size_t bpp(const rt_scene *s) { return s->opt_rgb24 ? 3 : 4; }      // bytes per pixel of the colour output (RT_OPT_RGB24)
#ifndef RT_SPLIT_PIECES_LEVELS
#define RT_SPLIT_PIECES_LEVELS 2
#endif
#ifndef RT_SPLIT_KI_LEVELS
#define RT_SPLIT_KI_LEVELS 3072
#endif
constexpr size_t RT_SPLIT_PIXELS = 12u << 20;                          // plain pipeline: two halves from 12 Mi pixels on
constexpr size_t RT_SPLIT_PIXELS_LEVELS = (size_t)RT_SPLIT_KI_LEVELS << 10;   // with bounce levels: from 3 Mi pixels on    // rt_render_device splits bands of this many pixels over the two lanes
constexpr size_t RT_CHUNK_PIXELS = 16u << 20;   // rows are processed in chunks of about 16 Mi pixels

int render_levels(const rt_scene *s, int depth) {
    return (s->reflective && depth > 1 && !s->talk) ? (depth < RT_MAX_LEVELS ? depth : RT_MAX_LEVELS) : 1;
}
bool levels_supported(const rt_scene *s, int depth) {
    if (s->reflective && !s->talk && depth > RT_MAX_LEVELS) {
        set_error("rt_render: trace depth > 64 with reflective materials is not supported"); return false;
    }
    return true;
}

// What one render needs beyond its inputs and outputs, in elements of each buffer.
struct WsNeeds {
    size_t hit_id = 0, accum = 0, hitrec = 0, queue = 0, qsparse = 0, tcount = 0, lit = 0, lvl = 0, cb = 0;
    int levels = 1;
};
// slots of a bounce level: k_final hands every wave of 8 x 8 pixels that reflects anything a run of 64 slots, so a level
// can hold the chunk's pixels padded to whole 32 x 8 tiles (and no more: deeper levels never grow)
size_t level_capacity(int w, size_t rows) {
    return (size_t)((w + 31) / 32) * 32 * ((rows + 7) / 8) * 8;
}
size_t setup_blocks_bound(int w, size_t crow, bool levels) {   // k_setup blocks of a chunk: 32 x 16 pixel tiles at level 0, runs of slots above
    const size_t slots = levels ? level_capacity(w, crow) : crow * (size_t)w;
    return std::max((size_t)((w + 31) / 32) * ((crow + RT_SETUP_THREADS / 32 - 1) / (RT_SETUP_THREADS / 32)),
                    (slots + RT_SETUP_THREADS - 1) / RT_SETUP_THREADS);
}
WsNeeds ws_needs(int w, int nrows, size_t crow, int npoint, int levels, int spp, bool own_hit, bool shade, bool callbacks) {
    WsNeeds n;
    const size_t npix = (size_t)w * (size_t)nrows, np = (size_t)(npoint > 0 ? std::min(npoint, 32) : 1);
    const size_t groups = (size_t)std::max(1, (npoint + 31) / 32);
    const size_t cpix = levels > 1 ? level_capacity(w, crow) : crow * (size_t)w;      // slots of a chunk (bounce levels: padded)
    n.levels = levels;
    if (own_hit || spp > 1) n.hit_id = npix;
    if (!shade) return n;
    if (spp > 1) n.accum = 3 * npix;
    n.hitrec = cpix * sizeof(HitRec);
    n.lit = cpix * groups;
    n.queue = cpix * np;
    const size_t ub = setup_blocks_bound(w, crow, levels > 1);
    n.qsparse = ub * RT_SETUP_THREADS * np;
    n.tcount = (ub + 1) * np;
    if (levels > 1) n.lvl = cpix;
    if (callbacks) n.cb = cpix;
    return n;
}
// Carves the buffers of `n` out of [base, base + cap); returns the bytes used, or 0 if they do not fit.  base == nullptr:
// sizing only (W is left empty).
size_t ws_carve(unsigned char *base, size_t cap, const WsNeeds &n, Workspace &W) {
    size_t used = 0;
    bool fits = true;
    auto take = [&](auto *&ptr, size_t count) {
        using T = std::remove_reference_t<decltype(*ptr)>;
        used = (used + 255) & ~(size_t)255;
        if (used + count * sizeof(T) > cap) { fits = false; return; }
        if (base) ptr = reinterpret_cast<T *>(base + used);
        used += count * sizeof(T);
    };
    W = Workspace{};
    // the queue counters sit right behind the statistics counters, at the start: a render clears both with one fill, and
    // the part of them that lives across renders (the lights' cost history) keeps its place whatever else is carved
    take(W.counters, (size_t)RT_STAT_SLOTS * 16 + ((size_t)RT_QCOUNT_WORDS + 1) / 2);
    if (fits && base) W.qcount = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(W.counters) + RT_COUNTER_BYTES);
    if (n.hit_id) take(W.hit_id, n.hit_id);
    if (n.accum) take(W.accum, n.accum);
    if (n.hitrec) { take(W.hitrec, n.hitrec); W.hit_slots = n.hitrec / sizeof(HitRec); }
    if (n.lit) take(W.lit, n.lit);
    if (n.queue) take(W.queue, n.queue);
    if (n.qsparse) take(W.qsparse, n.qsparse);
    if (n.tcount) take(W.tcount, n.tcount);
    if (n.lvl) {
        take(W.lvl_col[0], n.lvl);
        for (int k = 1; k < n.levels && k < RT_MAX_LEVELS; ++k) { take(W.lvl_rays[k], n.lvl); take(W.lvl_hit[k], n.lvl); take(W.lvl_col[k], n.lvl); }
    }
    if (n.cb) { take(W.cb_slot, n.cb); take(W.cb_xy, 2 * n.cb); take(W.cb_group, n.cb); take(W.cb_normal, 3 * n.cb); }
    return fits ? used : 0;
}

// rt_render / rt_render_device: the work buffers of a lane of the (scene, device) pair are views of ONE allocation
// (Lane::ws_arena), grown when a render needs more than it holds: a first render is one hipMalloc, not a dozen.
int ws_grow(DeviceScene &D, Lane &LN, const WsNeeds &n) {
    Workspace sizing;
    const size_t total = ws_carve(nullptr, ~(size_t)0 >> 1, n, sizing);
    if (!LN.ws_arena.p || LN.ws_arena.cap < total) {
        // The head of the arena (statistics counters, queue counters) lives across the bands of a frame and across
        // renders: a lane that grows in the middle of a frame (a later piece is bigger than the first) takes it along.
        const size_t head = RT_COUNTER_BYTES + RT_QCOUNT_WORDS * sizeof(unsigned);
        void *fresh = nullptr;
        HIP_TRY(hipDeviceSynchronize());                    // what is in flight still uses the old buffers
        HIP_TRY(hipMalloc(&fresh, total));
        hipError_t e = LN.ws_arena.p ? hipMemcpy(fresh, LN.ws_arena.p, head, hipMemcpyDeviceToDevice) : hipMemset(fresh, 0, head);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) { (void)hipFree(fresh); HIP_TRY(e); }
        if (LN.ws_arena.p) HIP_TRY(hipFree(LN.ws_arena.p));
        LN.ws_arena.p = fresh;
        LN.ws_arena.cap = total;
    }
    if (!ws_carve((unsigned char *)LN.ws_arena.p, LN.ws_arena.cap, n, LN.ws)) { set_error("ws_grow: internal sizing error"); return RT_ERR_NOMEM; }
    return RT_OK;
}

// RT_NORMAL_CALLBACK: between the nearest-hit kernel and k_setup, the hit points that lie on callback leaves go to the
// host, the caller's fnorm is evaluated there (per callback group), and the normals come back.
int run_normal_callbacks(rt_scene &s, const RenderParams &P, hipStream_t st, unsigned pblocks) {
    HIP_TRY(hipMemsetAsync(P.cb_count, 0, sizeof(unsigned), st));
    launch_cb_gather(P, pblocks, st);
    unsigned count = 0;
    HIP_TRY(hipMemcpyAsync(&count, P.cb_count, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (count == 0) return RT_OK;
    std::vector<double> xy(2 * (size_t)count), nrm(3 * (size_t)count);
    std::vector<int32_t> grp(count);
    HIP_TRY(hipMemcpy(xy.data(), P.cb_xy, xy.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(grp.data(), P.cb_group, grp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    // the hit points sorted by callback group in one counting pass (a scene may register one group per interpolation cell:
    // a scan of all points per group was O(groups x points)), then one call per group that has any
    const size_t ngroups = s.normal_fn.size();
    std::vector<unsigned> start(ngroups + 1, 0u), idx(count);
    for (unsigned k = 0; k < count; ++k) if (grp[k] >= 0 && (size_t)grp[k] < ngroups) ++start[(size_t)grp[k] + 1];
    for (size_t g = 0; g < ngroups; ++g) start[g + 1] += start[g];
    {
        std::vector<unsigned> fill(start.begin(), start.end() - 1);
        for (unsigned k = 0; k < count; ++k) if (grp[k] >= 0 && (size_t)grp[k] < ngroups) idx[fill[(size_t)grp[k]]++] = k;
    }
    std::vector<double> gxy, gn;
    for (size_t g = 0; g < ngroups; ++g) {
        const size_t n = start[g + 1] - start[g];
        if (n == 0) continue;
        gxy.resize(2 * n); gn.assign(3 * n, 0.0);
        for (size_t j = 0; j < n; ++j) { const size_t k = idx[start[g] + j]; gxy[2 * j] = xy[2 * k]; gxy[2 * j + 1] = xy[2 * k + 1]; }
        s.normal_fn[g]((int)n, gxy.data(), gn.data(), s.normal_user[g]);
        for (size_t j = 0; j < n; ++j)
            for (int c = 0; c < 3; ++c) nrm[3 * (size_t)idx[start[g] + j] + c] = gn[3 * j + c];
    }
    HIP_TRY(hipMemcpy(const_cast<double *>(P.cb_normal), nrm.data(), nrm.size() * sizeof(double), hipMemcpyHostToDevice));
    return RT_OK;
}

// Enqueues the kernels of one row band on `st`.  `more` = this band continues a frame whose earlier bands were
// enqueued by the previous call(s): events, counters and ray counts then accumulate, so rt_get_stats describes the
// whole frame (per lane: a lane's first band of a frame starts its events and counters afresh).  `W` holds buffers of at
// least ws_needs(...) elements for chunks of `crow` rows; `timed` records an event per kernel boundary (creates events on
// first use).
int render_rows(rt_scene *s, DeviceScene &D, Lane &LN, const Workspace &W, size_t crow, int w, int h, int depth, int spp, uint64_t seed,
                int row0, int nrows, void *d_rgba, void *d_hit_id, hipStream_t st, bool more, bool timed, hipEvent_t hits_done = nullptr) {
    // hits_done: recorded on `st` as soon as the caller's hit-ID buffer is complete (after the nearest-hit kernel of
    // sample 0's last chunk: everything later only reads it), so that a copy of it can run beside the shading kernels
    int rc;
    const size_t npix = (size_t)w * (size_t)nrows;
    if (!more) {                   // a new frame
        D.last_primary = 0;
        for (Lane &l : D.lane) { l.used = false; l.ev_used = 0; }
    }
    more = more && LN.used;        // ... and this lane's first band of it
    LN.used = true;
    D.last_primary += (uint64_t)npix * (uint64_t)spp;
    D.stats_pending = false;
    LN.last_counters = W.counters;
    s->last_device = D.device;
    s->multi_stats = false;
    if (npix == 0) return RT_OK;
    (void)hipGetLastError();       // kernel launches report through the sticky error: start from a clean one
    // bounce levels: trace n reflects at most n - 1 times (trace 0 = black, Main.hs:542); only scenes with a
    // reflective material ever spawn a reflected ray (Main.hs:553)
    const int levels = render_levels(s, depth);
    int32_t *hit = (int32_t *)d_hit_id;
    if (!hit) hit = W.hit_id;
    if (spp > 1 && d_rgba) HIP_TRY(hipMemsetAsync(W.accum, 0, 3 * npix * sizeof(uint32_t), st));
    // (a render's first chunk finds its queue counters cleared by the same fill as the statistics counters)
    if (!more) HIP_TRY(hipMemsetAsync(W.counters, 0, RT_COUNTER_BYTES + RT_QCOUNT_CLEAR * sizeof(unsigned), st));
    bool queues_clear = !more;

#ifdef RT_VALU_FILTER
    // the vector-ALU kernels are compiled for Main.hs only: a Talk.hs scene always takes the matrix-core kernels
    const bool use_mfma = s->opt_filter == 0 || s->talk;
#else
    const bool use_mfma = true;
#endif
    RenderParams P{};
    P.leaves = (const LeafD *)D.leaves.p; P.aux = (const AuxD *)D.aux.p; P.guards = (const GuardD *)D.guards.p;
    P.mats = (const MaterialD *)D.mats.p; P.lights = (const LightD *)D.lights.p;
    P.tri_rec = (const FilterRec *)D.tri_rec.p; P.rect_rec = (const FilterRec *)D.rect_rec.p;
    P.tri_gid = (const int32_t *)D.tri_gid.p; P.rect_gid = (const int32_t *)D.rect_gid.p;
    P.counters = W.counters;
    P.hitpos = reinterpret_cast<HitVec *>(W.hitrec); P.hitnrm = P.hitpos ? P.hitpos + W.hit_slots : nullptr; P.queue = W.queue; P.qcount = W.qcount; P.lit = W.lit;
    P.qsparse = W.qsparse; P.tcount = W.tcount;
    P.cb_slot = W.cb_slot; P.cb_xy = W.cb_xy; P.cb_group = W.cb_group; P.cb_normal = W.cb_normal; P.cb_count = W.qcount + RT_QCOUNT_CB;
    P.gen_tri = (const GenRec *)D.gen_tri.p; P.gen_rect = (const GenRec *)D.gen_rect.p;
    P.gen_tri_gid = (const int32_t *)D.gen_tri_gid.p; P.gen_rect_gid = (const int32_t *)D.gen_rect_gid.p;
    P.gen_tri_sph = (const float *)D.gen_tri_sph.p; P.gen_rect_sph = (const float *)D.gen_rect_sph.p;
    P.gen_tri_rows = (const uint16_t *)D.gen_tri_rows.p; P.gen_rect_rows = (const uint16_t *)D.gen_rect_rows.p;
    P.gen_tri_ssph = (const float *)D.gen_tri_ssph.p; P.gen_rect_ssph = (const float *)D.gen_rect_ssph.p;
    P.sph_rec = (const FilterRec *)D.sph_rec.p; P.gen_sph = (const GenRec *)D.gen_sph.p; P.sph_gid = (const int32_t *)D.sph_gid.p;
    P.nsph = (int)s->sph_gid.size();
    P.mf_rows = (const uint16_t *)D.mf_rows.p; P.mf_gid = (const int32_t *)D.mf_gid.p; P.mf_rec = (const FilterRec *)D.mf_rec.p;
    P.mf_groups = s->mf_groups;
    P.mf_cone = (const float *)D.mf_cone.p; P.cull = s->opt_cull ? 1 : 0;
    P.nguard_cached = (s->opt_cull && !s->talk) ? s->guard_cached : 0;
    // point lights are worked off 32 at a time (one bit each in a word of `lit`); the reference's Light monoid has no limit
    std::vector<uint16_t> point_lights;
    for (size_t li = 0; li < s->lights.size(); ++li) if (s->lights[li].kind == 0) point_lights.push_back((uint16_t)li);
    const int ngroups = std::max(1, (int)((point_lights.size() + 31) / 32));
    P.cam = fixed_camera(w, h);
    P.eps = ray_eps();
    P.ntri = (int)s->tri_gid.size(); P.nrect = (int)s->rect_gid.size();
    P.nlights = (int)s->lights.size(); P.nleaves = (int)s->leaves.size(); P.npoint = std::min(32, s->npoint);
    P.w = w; P.h = h;
    P.talk = s->talk ? 1 : 0;
    P.bpp = s->opt_rgb24 ? 3 : 4;
    P.rgb_packed = (P.bpp == 3 && w % 4 == 0 && (reinterpret_cast<uintptr_t>(d_rgba) & 3u) == 0) ? 1 : 0;   // (whole dwords need an aligned buffer)
    P.spp = spp; P.seed_lo = (uint32_t)seed; P.seed_hi = (uint32_t)(seed >> 32);
    const bool callbacks = !s->normal_fn.empty();
#ifdef RT_WAVE_TIMES
    {
        static unsigned long long *dbg = nullptr;
        if (!dbg) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dbg), (8 + 8 * (1ull << 20)) * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(dbg, 0, (8 + 8 * (1ull << 20)) * sizeof(unsigned long long), st));
        P.dbg = dbg;
        g_wave_dbg = dbg;
    }
#endif

    // Wave history (rt_params.h): kept per band for the plain pipeline — level 0 only, one group of lights, the growing
    // work buffers (`timed` renders: the no-alloc entry point allocates nothing, so it keeps none).
    WaveHistory *hist = nullptr;
    size_t hist_tiles = 0, hist_swaves = 0;
#if RT_WAVES_PER_BLOCK == 1
    if (timed && use_mfma && s->opt_history && levels == 1 && ngroups == 1 && !point_lights.empty() && d_rgba && depth > 0 && !callbacks) {
        if (LN.history_version != s->version || LN.history.size() > 64) {       // another scene (or too many bands): start over
            if (!LN.history.empty()) HIP_TRY(hipDeviceSynchronize());
            for (auto &kv : LN.history) if (kv.second.buf) (void)hipFree(kv.second.buf);
            LN.history.clear();
            LN.history_version = s->version;
        }
        hist_tiles = (size_t)((w + RT_TILE_W - 1) / RT_TILE_W) * ((crow + RT_TILE_H - 1) / RT_TILE_H);
        hist_swaves = (crow * (size_t)w + 255) / 256;
        const size_t words = hist_tiles + point_lights.size() * hist_swaves, chunks = ((size_t)nrows + crow - 1) / crow;
        const std::array<uint64_t, 6> key = {(uint64_t)w, (uint64_t)h, (uint64_t)row0, (uint64_t)nrows, (uint64_t)crow,
                                             (uint64_t)point_lights.size() | (uint64_t)s->eye_version << 16};
        WaveHistory &H = LN.history[key];
        if (!H.buf) {
            HIP_TRY(hipMalloc(&H.buf, chunks * 2 * words * sizeof(uint32_t)));
            HIP_TRY(hipMemsetAsync(H.buf, 0, chunks * 2 * words * sizeof(uint32_t), st));
            H.words = words;
        }
        hist = &H;
    }
#endif
    const bool hist_ready = hist && hist->ready;
    if (hist) hist->ready = false;         // (until this frame is enqueued whole)

    if (timed && (rc = next_event(LN, st, 0))) return rc;
    for (int sample = 0; sample < spp; ++sample) {
        P.sample = sample;
        // sample 0 writes the caller's hit-ID buffer; later samples use the scratch one
        P.hit_id = (sample == 0) ? hit : W.hit_id;
        P.rgba = (spp == 1) ? (uint8_t *)d_rgba : nullptr;
        P.accum = (spp == 1) ? nullptr : W.accum;
        for (size_t r = 0; r < (size_t)nrows; r += crow) {
            const int rows = (int)((size_t)nrows - r < crow ? (size_t)nrows - r : crow);
            P.row0 = row0 + (int)r; P.nrows = rows;
            P.out_off = (long long)r * w;
            P.qcap = rows * w;
            P.lcap = (int)level_capacity(w, (size_t)rows);
            P.tiles_x = (w + RT_TILE_W - 1) / RT_TILE_W;
            P.tiles_y = (rows + RT_TILE_H - 1) / RT_TILE_H;
            const long long tiles = (long long)P.tiles_x * P.tiles_y;
            const unsigned tblocks = (unsigned)((tiles + RT_WAVES_PER_BLOCK - 1) / RT_WAVES_PER_BLOCK);
            const unsigned pblocks = (unsigned)(((size_t)P.qcap + 255) / 256);
            if (hist) {                // this chunk's part of the history: costs, then orders
                uint32_t *base = (uint32_t *)hist->buf + (r / crow) * 2 * hist->words;
                P.wcost_p = base; P.wcost_s = base + hist_tiles;
                P.worder_p_out = base + hist->words; P.worder_s = P.worder_p_out + hist_tiles;
                // (a later sample of the same frame finds the orders of the sample before: as good as the last frame's)
                P.worder_p = (hist_ready || sample > 0) ? P.worder_p_out : nullptr;
                P.whave = (hist_ready || sample > 0) ? 1 : 0;
                P.wsort_blocks = 1 + (int)point_lights.size();
            }
            if (depth == 0) {      // trace 0 = black (Main.hs:542): every pixel misses
                HIP_TRY(hipMemsetAsync(P.hit_id + P.out_off, 0xff, (size_t)P.qcap * sizeof(int32_t), st));
            } else {
                if (use_mfma) launch_primary_mfma(P, tblocks, st);
#ifdef RT_VALU_FILTER
                else launch_primary_valu(P, spp == 1, tblocks, st);
#endif
            }
            if (timed && (rc = next_event(LN, st, 1))) return rc;
            if (hits_done && sample == 0 && r + crow >= (size_t)nrows) HIP_TRY(hipEventRecord(hits_done, st));
            if (!d_rgba) continue;
            // grids: shadow waves of 256 jobs, one thread per slot; bounce levels have up to lcap slots
            const size_t slots_max = levels > 1 ? (size_t)P.lcap : (size_t)P.qcap;
            const unsigned sblocks = (unsigned)((slots_max + 256 * RT_WAVES_PER_BLOCK - 1) / (256 * RT_WAVES_PER_BLOCK));
            P.wstride_s = (int)sblocks;     // (<= hist_swaves: a last, shorter chunk uses the front of its arrays)
            const unsigned lblocks = (unsigned)((slots_max + 255) / 256);
            int32_t *const hit0 = P.hit_id;
            const long long off0 = P.out_off;
            uint8_t *const rgba0 = P.rgba;
            uint32_t *const accum0 = P.accum;
            unsigned *const lvl_count = W.qcount + 32;
            if (!queues_clear) HIP_TRY(hipMemsetAsync(W.qcount, 0, RT_QCOUNT_CLEAR * sizeof(unsigned), st));
            queues_clear = false;
            for (int lvl = 0; lvl < levels; ++lvl) {
                if (lvl > 0) {      // trace (n-1) lights world reflected, Main.hs:554
                    HIP_TRY(hipMemsetAsync(W.qcount, 0, 32 * sizeof(unsigned), st));
                    P.rays = W.lvl_rays[lvl]; P.count_ptr = lvl_count + lvl;
                    P.hit_id = W.lvl_hit[lvl]; P.out_off = 0;
                    launch_general(P, (unsigned)((slots_max + 63) / 64), st);
                    if (timed && (rc = next_event(LN, st, 1))) return rc;
                }
                P.next_rays = (lvl + 1 < levels) ? W.lvl_rays[lvl + 1] : nullptr;
                P.next_count = lvl_count + lvl + 1;
                P.colour = (levels > 1) ? W.lvl_col[lvl] : nullptr;
                if (callbacks && (rc = run_normal_callbacks(*s, P, st, lvl == 0 ? pblocks : lblocks))) return rc;
                const unsigned ublocks = lvl == 0 ? (unsigned)P.tiles_x * (unsigned)((rows + RT_SETUP_THREADS / 32 - 1) / (RT_SETUP_THREADS / 32))
                                                  : (unsigned)((slots_max + RT_SETUP_THREADS - 1) / RT_SETUP_THREADS);
                P.setup_blocks = (int)ublocks;
                P.lit_stride = (long long)slots_max;
                for (int grp = 0; grp < ngroups; ++grp) {
                    const int first = 32 * grp, np = std::max(0, std::min(32, (int)point_lights.size() - first));
                    P.npoint = np;
                    for (int k = 0; k < np; ++k) P.point_light[k] = point_lights[(size_t)first + k];
                    P.lit_w = W.lit + (size_t)grp * slots_max;
                    if (grp > 0) HIP_TRY(hipMemsetAsync(W.qcount, 0, 32 * sizeof(unsigned), st));
                    launch_setup(P, ublocks, st);
                    if (np > 0) {
                        launch_queue_offsets(P, (unsigned)np, st);
                        launch_queue_compact(P, ublocks, (unsigned)np, st);
                    }
                    if (timed && (rc = next_event(LN, st, 2))) return rc;
                    if (np > 0) {
                        if (use_mfma) launch_shadow_mfma(P, sblocks, (unsigned)np, st);
#ifdef RT_VALU_FILTER
                        else launch_shadow_valu(P, sblocks, (unsigned)np, st);
#endif
                    }
                    if (timed && (rc = next_event(LN, st, 3))) return rc;
                }
                if (P.next_rays)
                    launch_final(P, lvl == 0 ? (unsigned)P.tiles_x * (unsigned)((rows + 31) / 32) : (lblocks + 3) / 4, 1024u, st);
                else
                    launch_final(P, lvl == 0 ? (unsigned)P.tiles_x * (unsigned)((rows + 7) / 8) : lblocks, 256u, st);
                if (timed && (rc = next_event(LN, st, 4))) return rc;
            }
            P.rays = nullptr; P.count_ptr = nullptr; P.next_rays = nullptr; P.hit_id = hit0; P.out_off = off0;
            if (levels > 1) {       // fold the levels back: direct `addPixelRGB8` scaled reflection, deepest first
                for (int lvl = levels - 1; lvl >= 1; --lvl)
                    launch_combine(W.lvl_rays[lvl], W.lvl_col[lvl], lvl_count + lvl, (unsigned)P.lcap, W.lvl_col[lvl - 1], lblocks, st);
                P.colour = W.lvl_col[0]; P.rgba = rgba0; P.accum = accum0;
                launch_store(P, pblocks, st);
                P.colour = nullptr;
                if (timed && (rc = next_event(LN, st, 4))) return rc;
            }
        }
    }
    if (d_rgba && spp > 1) {
        launch_resolve(W.accum, (uint8_t *)d_rgba, npix, (uint32_t)spp, P.bpp, P.rgb_packed, st);
        if (timed && (rc = next_event(LN, st, 4))) return rc;
    }
    HIP_TRY(hipGetLastError());
    if (hist) hist->ready = true;
    D.stats_pending = true;
    return RT_OK;
}

int check_render_args(const rt_scene *s, int w, int h, int depth, int spp, int row0, int nrows) {
    if (!s) { set_error("rt_render: scene is NULL"); return RT_ERR_ARG; }
    if (w <= 0 || h <= 0 || spp < 1 || row0 < 0 || nrows < 0 || row0 + nrows > h) {
        set_error("rt_render: bad image size or row band"); return RT_ERR_ARG;
    }
    if (depth < 0) { set_error("rt_render: depth must be >= 0"); return RT_ERR_ARG; }
    if (s->lights.size() > 65535) { set_error("rt_render: more than 65535 lights are not supported"); return RT_ERR_UNSUPPORTED; }
    return RT_OK;
}
// chunk rows of the growing-workspace renders: a multiple of the tile height
size_t chunk_rows(int w, int nrows, int levels) {
    // a bounce level keeps 72 bytes per slot: a quarter of the pixels for up to 8 levels, fewer beyond
    const size_t pixels = levels > 1 ? std::min<size_t>(RT_CHUNK_PIXELS / 4, 2 * RT_CHUNK_PIXELS / (size_t)levels) : RT_CHUNK_PIXELS;
    size_t crow = pixels / (size_t)w;
    crow = crow < RT_TILE_H ? RT_TILE_H : crow - crow % RT_TILE_H;
    if (crow > (size_t)nrows) crow = (size_t)nrows;
    return crow ? crow : 1;
}

// rt_render_device on the growing workspace of a lane of the (scene, device) pair; st == nullptr: the lane's own stream
int render_band(rt_scene *s, int device, int w, int h, int depth, int spp, uint64_t seed, int row0, int nrows,
                void *d_rgba, void *d_hit_id, hipStream_t st, bool more, hipEvent_t hits_done = nullptr, int lane = 0) {
    int rc;
    DeviceScene &D = s->dev[device];
    if ((rc = device_prepare(*s, D, device, w, h))) return rc;
    Lane &LN = D.lane[lane];
    if (!st && (rc = lane_stream(D, lane, st))) return rc;
    if (!levels_supported(s, depth)) return RT_ERR_UNSUPPORTED;
    const int levels = render_levels(s, depth);
    const size_t crow = chunk_rows(w, nrows, levels);
    const WsNeeds need = ws_needs(w, nrows, crow, s->npoint, levels, spp, d_hit_id == nullptr, d_rgba != nullptr, !s->normal_fn.empty());
    if ((rc = ws_grow(D, LN, need))) return rc;
    return render_rows(s, D, LN, LN.ws, crow, w, h, depth, spp, seed, row0, nrows, d_rgba, d_hit_id, st, more, true, hits_done);
}

} // namespace


int init_impl(int device) { return device_init(device); }

int scene_prepare_device_impl(rt_scene *s, int w, int h) {
    if (!s || w <= 0 || h <= 0) { set_error("rt_scene_prepare_device: bad argument"); return RT_ERR_ARG; }
    int rc;
    if ((rc = enter())) return rc;
    return device_prepare(*s, s->dev[t_device], t_device, w, h);
}

void scene_destroy_impl(rt_scene *s) {
    if (!s) return;
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (DeviceScene &d : s->dev) {
        helper_join(d);
        if (d.device < 0) continue;
        if (hipSetDevice(d.device) != hipSuccess) continue;
        if (d.stream) (void)hipStreamSynchronize((hipStream_t)d.stream);
        if (d.copy_stream) (void)hipStreamSynchronize((hipStream_t)d.copy_stream);
        if (d.copy_stream2) (void)hipStreamSynchronize((hipStream_t)d.copy_stream2);
        // the scene's arrays and the work buffers are views of the two arenas
        for (void *x : d.xstream) if (x) (void)hipStreamSynchronize((hipStream_t)x);
        void *ptrs[] = {d.arena.p, d.rgba, d.hit_out, d.stage_rgba, d.stage_hit};
        for (void *p : ptrs) if (p) (void)hipFree(p);
        for (Lane &l : d.lane) if (l.ws_arena.p) (void)hipFree(l.ws_arena.p);
        for (Lane &l : d.lane) for (auto &kv : l.history) if (kv.second.buf) (void)hipFree(kv.second.buf);
        for (Lane &l : d.lane) for (void *e : l.evpool) if (e) (void)hipEventDestroy((hipEvent_t)e);
        for (void *x : d.xstream) if (x) (void)hipStreamDestroy((hipStream_t)x);
        for (void *e : d.band_done) if (e) (void)hipEventDestroy((hipEvent_t)e);
        if (d.fork_ev) (void)hipEventDestroy((hipEvent_t)d.fork_ev);
        for (void *e : d.join_evs) if (e) (void)hipEventDestroy((hipEvent_t)e);
        for (void *e : d.hits_done) if (e) (void)hipEventDestroy((hipEvent_t)e);
        if (d.stream) (void)hipStreamDestroy((hipStream_t)d.stream);
        if (d.copy_stream) (void)hipStreamDestroy((hipStream_t)d.copy_stream);
        if (d.copy_stream2) (void)hipStreamDestroy((hipStream_t)d.copy_stream2);
    }
    if (cur >= 0) (void)hipSetDevice(cur);
    delete s;
}

int render_device_impl(rt_scene *s, int w, int h, int depth, int spp, uint64_t seed, int row0, int nrows,
                     void *d_rgba, void *d_hit_id, void *stream) {
    int rc;
    if ((rc = check_render_args(s, w, h, depth, spp, row0, nrows))) return rc;
    if ((rc = enter())) return rc;
    // A big band is rendered as pieces that alternate between the two lanes, forked from and joined to the caller's
    // stream: the kernels of the pieces run beside each other and fill each other's tails (tools/lane_split_timing.py).
    //   * the plain pipeline (no bounce levels): two halves from RT_SPLIT_PIXELS on — a 16384 x 2048 band 4.44 -> 4.12 ms,
    //     the 8K frame 4.80 -> 4.48; the 4K frame gains 0.03 ms from the overlap and loses 0.05 to the fork and joins, so
    //     it stays whole;
    //   * with bounce levels (reflective scene, trace depth > 1): RT_SPLIT_PIECES_LEVELS pieces from RT_SPLIT_PIXELS_LEVELS
    //     on — the deeper levels are a string of short kernels over few rays (level 3 of the mirror workload: 3 % of the
    //     rays, 10 % of the frame), which find the other lane's big kernels to run beside: the mirror workload at 4K 6.04 ->
    //     5.17 ms with two halves (4 pieces 5.77, 8 pieces 7.23; 3 or 4 lanes 5.74 / 5.97), at 3200 x 1800 4.61 -> 3.98, at
    //     2560 x 1440 3.16 -> 3.03; at 1920 x 1080 and below the frame is one chain of launch latencies (1.9 ms whatever
    //     the size) and the halves only add to it (2.37 -> 2.47).
    // Not for a caller who drives the lanes (RT_OPT_LANE 0 or 1, RT_OPT_CONTINUE), nor for fnorm callbacks (a band with
    // callbacks waits for the host in the middle: nothing would overlap).
    const size_t npix = (size_t)w * (size_t)nrows;
    const bool bounces = render_levels(s, depth) > 1;
    int pieces = 1;
    if (s->opt_lane < 0 && !s->opt_continue && s->normal_fn.empty()) {
        if (bounces && npix >= RT_SPLIT_PIXELS_LEVELS) pieces = RT_SPLIT_PIECES_LEVELS;
        else if (!bounces && npix >= RT_SPLIT_PIXELS) pieces = 2;
        while (pieces > 1 && nrows < pieces * 2 * RT_TILE_H) pieces /= 2;
    }
    if (pieces > 1) {
        DeviceScene &D = s->dev[t_device];
        if ((rc = device_prepare(*s, D, t_device, w, h))) return rc;
        // The pieces run on the LANES' OWN streams; the caller's stream carries only the fork and the joins.  (Streams share
        // few hardware queues, 4 by default: kernels of a caller's stream that happens to sit on lane 1's queue would
        // run behind lane 1's, not beside them — measured 1.65 ms instead of 1.32 for the 4K frame.)
        hipStream_t cs = (hipStream_t)stream, lst[RT_LANES];
        for (int l = 0; l < RT_LANES; ++l) if ((rc = lane_stream(D, l, lst[l]))) return rc;
        if (!cs) cs = lst[0];
        if (!D.fork_ev) { hipEvent_t ev; HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); D.fork_ev = ev; }
        for (void *&e : D.join_evs) if (!e) { hipEvent_t ev; HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); e = ev; }
        HIP_TRY(hipEventRecord((hipEvent_t)D.fork_ev, cs));
        for (int l = 0; l < RT_LANES; ++l) if (lst[l] != cs) HIP_TRY(hipStreamWaitEvent(lst[l], (hipEvent_t)D.fork_ev, 0));
        int prow = (nrows + pieces - 1) / pieces;
        prow = (prow + RT_TILE_H - 1) / RT_TILE_H * RT_TILE_H;
        int k = 0;
        for (int r = 0; r < nrows && rc == RT_OK; r += prow, ++k) {
            const int rows = nrows - r < prow ? nrows - r : prow;
            const size_t off = (size_t)r * (size_t)w;
            rc = render_band(s, t_device, w, h, depth, spp, seed, row0 + r, rows, d_rgba ? (uint8_t *)d_rgba + bpp(s) * off : nullptr,
                             d_hit_id ? (int32_t *)d_hit_id + off : nullptr, lst[k % RT_LANES], k > 0, nullptr, k % RT_LANES);
        }
        // always join: whatever the lanes were given is ordered before what the caller enqueues next
        bool ok = true;
        for (int l = 0; l < RT_LANES; ++l)
            if (lst[l] != cs)
                ok = hip_ok(hipEventRecord((hipEvent_t)D.join_evs[l], lst[l]), "hipEventRecord") && hip_ok(hipStreamWaitEvent(cs, (hipEvent_t)D.join_evs[l], 0), "hipStreamWaitEvent") && ok;
        return rc ? rc : (ok ? (int)RT_OK : (int)RT_ERR_HIP);
    }
    // RT_OPT_LANE picks the work buffers (and, for stream == NULL, the stream): lane 1 continues the frame lane 0 began, RT_OPT_CONTINUE makes lane 0 continue too
    return render_band(s, t_device, w, h, depth, spp, seed, row0, nrows, d_rgba, d_hit_id, (hipStream_t)stream, s->opt_lane > 0 || s->opt_continue,
                       nullptr, s->opt_lane > 0 ? s->opt_lane : 0);
}

int render_impl(rt_scene *s, int w, int h, int depth, int spp, uint64_t seed, int row0, int nrows,
              uint8_t *rgba_out, int32_t *hit_id_out) {
    int rc;
    if ((rc = check_render_args(s, w, h, depth, spp, row0, nrows))) return rc;
    if ((rc = enter())) return rc;
    DeviceScene &D = s->dev[t_device];
    const size_t npix = (size_t)w * (size_t)nrows;
    if (npix == 0) return RT_OK;
    if ((rc = device_prepare(*s, D, t_device, w, h))) return rc;
    if ((rc = copy_tools(D))) return rc;
    // Colours AND hit IDs into page-locked memory that this device can address (rt_host_alloc): k_final stores its
    // coalesced lines straight through PCIe into the caller's buffer, so no copy of the colours is left behind the last
    // kernel while the copy engine is busy with the hit IDs (33 + 33 MB at 4K; hit IDs always go through device memory:
    // the shading kernels read them back).  Colours alone: a copy per band on the copy engine, which then has the link to
    // itself and runs beside the next band's kernels, is faster (4K frame, tools/host_render_timing.py: 1.70 against 1.82
    // ms; 3 bytes per pixel 1.63 against 1.71) — k_final's waves wait on the link and hold their slots while they do.
    uint8_t *direct = nullptr;
#ifndef RT_NO_DIRECT_RGBA
    if (rgba_out && hit_id_out) {
        hipPointerAttribute_t attr{};
        if (hipPointerGetAttributes(&attr, rgba_out) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer) {
            hipPointerAttribute_t last{};
            uint8_t *end = rgba_out + bpp(s) * npix - 1;
            if (hipPointerGetAttributes(&last, end) == hipSuccess && last.type == hipMemoryTypeHost && last.devicePointer == (uint8_t *)attr.devicePointer + (bpp(s) * npix - 1))
                direct = (uint8_t *)attr.devicePointer;
        } else (void)hipGetLastError();                     // (an ordinary pointer: not an error)
    }
#endif
    if (!direct && (rc = ensure(D.rgba, D.rgba_cap, bpp(s) * npix))) return rc;
    if ((rc = ensure(D.hit_out, D.hit_out_cap, npix))) return rc;
    hipStream_t lst[RT_LANES], cst = (hipStream_t)D.copy_stream, hst = (hipStream_t)D.copy_stream2;
    for (int k = 0; k < RT_LANES; ++k) if ((rc = lane_stream(D, k, lst[k]))) return rc;
    // The frame is enqueued as row bands that ALTERNATE between the two lanes (work buffers + stream each): band k + 1's
    // kernels run beside band k's, filling the tails of its kernels, and band k's colours travel while band k + 1 still
    // computes — as k_final's own stores over PCIe when the destination is page-locked memory this device can address, as a
    // copy on a stream of its own otherwise.  The hit IDs of a band are copied as soon as its nearest-hit kernel has ended
    // (the shading kernels only read them).  Default: 2 bands (one per lane) when k_final stores the colours itself, 3
    // when the copy engine carries them (bands 1 and 2 run side by side and travel beside band 3), up to 24 Mi pixels; one
    // per 8 Mi pixels beyond.  Measured on the 4K benchmark frame into page-locked memory (tools/host_render_timing.py,
    // colours + hit IDs): 1 band 1.85 ms, 2 bands 1.80, 3 bands 1.99, 4 bands 2.03 — the bands are short enough for launch
    // gaps to show; on ONE lane a band costs 0.15 - 0.27 ms.  Into pageable memory the copies stage through the runtime's
    // bounce buffers and block this thread, which is why every band is enqueued first.
    int nb = s->opt_host_bands > 0 ? s->opt_host_bands : std::max(direct || !rgba_out ? 2 : 3, (int)(npix >> 23));
    nb = nb < 1 ? 1 : (nb > RT_HOST_BANDS_MAX ? RT_HOST_BANDS_MAX : nb);
    if (nrows < 2 * RT_TILE_H) nb = 1;
    int bstart[RT_HOST_BANDS_MAX + 1];                   // band k = rows [bstart[k], bstart[k + 1]) of the call's rows
    {
        int brows = (nrows + nb - 1) / nb;
        brows = (brows + RT_TILE_H - 1) / RT_TILE_H * RT_TILE_H;
        int k = 0;
        for (int r = 0; r < nrows; r += brows) bstart[k++] = r;
        nb = k;
        bstart[nb] = nrows;
    }
    for (int used = 0; used < nb && rc == RT_OK; ++used) {
        const int r = bstart[used], rows = bstart[used + 1] - r;
        const size_t off = (size_t)r * (size_t)w;
        const int lane = used % RT_LANES;
        rc = render_band(s, t_device, w, h, depth, spp, seed, row0 + r, rows, rgba_out ? (direct ? direct : D.rgba) + bpp(s) * off : nullptr,
                         D.hit_out + off, lst[lane], r > 0, hit_id_out ? (hipEvent_t)D.hits_done[used] : nullptr, lane);
        if (rc == RT_OK && !hip_ok(hipEventRecord((hipEvent_t)D.band_done[used], lst[lane]), "hipEventRecord")) rc = RT_ERR_HIP;
    }
    for (int band = 0; band < nb && rc == RT_OK; ++band) {
        const int r = bstart[band], rows = bstart[band + 1] - r;
        const size_t off = (size_t)r * (size_t)w, n = (size_t)rows * (size_t)w;
        if (hit_id_out) {
            if (!hip_ok(hipStreamWaitEvent(hst, (hipEvent_t)D.hits_done[band], 0), "hipStreamWaitEvent")) { rc = RT_ERR_HIP; break; }
            if (!hip_ok(hipMemcpyAsync(hit_id_out + off, D.hit_out + off, n * sizeof(int32_t), hipMemcpyDeviceToHost, hst), "hipMemcpyAsync")) { rc = RT_ERR_HIP; break; }
        }
        if (rgba_out && !direct) {
            if (!hip_ok(hipStreamWaitEvent(cst, (hipEvent_t)D.band_done[band], 0), "hipStreamWaitEvent")) { rc = RT_ERR_HIP; break; }
            if (!hip_ok(hipMemcpyAsync(rgba_out + bpp(s) * off, D.rgba + bpp(s) * off, bpp(s) * n, hipMemcpyDeviceToHost, cst), "hipMemcpyAsync")) rc = RT_ERR_HIP;
        }
    }
    // always drain both streams, also after an error: nothing may still be writing the caller's buffers on return
    if (!hip_ok(hipStreamSynchronize(cst), "hipStreamSynchronize") && rc == RT_OK) rc = RT_ERR_HIP;
    if (!hip_ok(hipStreamSynchronize(hst), "hipStreamSynchronize") && rc == RT_OK) rc = RT_ERR_HIP;
    for (int k = 0; k < RT_LANES; ++k) if (!hip_ok(hipStreamSynchronize(lst[k]), "hipStreamSynchronize") && rc == RT_OK) rc = RT_ERR_HIP;
    return rc;
}

void *host_alloc_impl(size_t bytes) {
    if (enter() != RT_OK) return nullptr;
    void *p = nullptr;
    if (!hip_ok(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable), "hipHostMalloc")) return nullptr;
    return p;
}

void host_free_impl(void *p) {
    if (p) (void)hipHostFree(p);
}

int synchronize_impl(void) {
    if (t_device < 0) return RT_OK;
    int rc;
    if ((rc = enter())) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return RT_OK;
}


namespace {
// statistics of the last render of the (scene, device) pair; synchronises with it
int device_stats(rt_scene *s, DeviceScene &D, rt_stats *out) {
    if (D.stats_pending) {
        static thread_local std::vector<unsigned long long> all(RT_STAT_SLOTS * 16);
        // the lanes the frame used: wait for each, sum their counters, add their kernel times up; ms_total runs from the
        // frame's first event to its last, whichever lanes they are on
        unsigned long long c[16] = {0};
        double ms[5] = {0, 0, 0, 0, 0};
        float tot = 0;
        bool any_event = false;
        for (Lane &l : D.lane) if (l.used && l.ev_used > 0) { HIP_TRY(hipEventSynchronize((hipEvent_t)l.evpool[l.ev_used - 1])); any_event = true; }
        if (!any_event) HIP_TRY(hipDeviceSynchronize());
        const Lane *first = nullptr;
        for (Lane &l : D.lane) {
            if (!l.used || !l.last_counters) continue;
            HIP_TRY(hipMemcpy(all.data(), l.last_counters, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            for (int slot = 0; slot < RT_STAT_SLOTS; ++slot) for (int k = 0; k < 16; ++k) c[k] += all[(size_t)slot * 16 + k];
            for (int i = 1; i < l.ev_used; ++i) {
                float t = 0;
                HIP_TRY(hipEventElapsedTime(&t, (hipEvent_t)l.evpool[i - 1], (hipEvent_t)l.evpool[i]));
                ms[l.evkind[i]] += t;
            }
            if (l.ev_used > 0 && !first) first = &l;
        }
        if (first)
            for (Lane &l : D.lane) {
                if (!l.used || l.ev_used == 0) continue;
                float t = 0;
                HIP_TRY(hipEventElapsedTime(&t, (hipEvent_t)first->evpool[0], (hipEvent_t)l.evpool[l.ev_used - 1]));
                tot = std::max(tot, t);
            }
        rt_stats st{};
        st.primary_rays = D.last_primary;
        st.shadow_rays = c[0];
        st.bounce_rays = c[5];
        st.bounce_blocks = c[8];
        const uint64_t rays = st.primary_rays + st.shadow_rays + st.bounce_rays;
        st.tri_pairs = rays * (uint64_t)s->tri_gid.size();
        st.rect_pairs = rays * (uint64_t)s->rect_gid.size();
        st.exact_evals = c[1] + c[2];
        st.record_tests = c[4];
        st.mfma_blocks_primary = c[6];
        st.mfma_blocks_shadow = c[7];
#ifdef RT_COUNT_EVENTS
        fprintf(stderr, "[rt] primary: true hits among the exact evaluations %llu; exact evals primary %llu shadow %llu\n", c[3], c[1], c[2]);
#endif
        st.ms_primary = ms[1]; st.ms_setup = ms[2]; st.ms_shadow = ms[3]; st.ms_final = ms[4]; st.ms_total = tot;
        st.upload_ms = D.stats.upload_ms;
        D.stats = st;
        D.stats_pending = false;
#ifdef RT_WAVE_TIMES
        if (const char *path = getenv("RT_WAVE_TIMES_FILE")) {
            std::vector<unsigned long long> rec(8 + 8 * (1ull << 20));
            HIP_TRY(hipMemcpy(rec.data(), g_wave_dbg, rec.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            if (FILE *f = fopen(path, "wb")) { fwrite(rec.data(), 8, rec.size(), f); fclose(f); }
        }
#endif
    }
    *out = D.stats;
    out->prepare_ms = s->prepare_ms;
    return RT_OK;
}
} // namespace

int get_stats_impl(rt_scene *s, rt_stats *out) {
    if (!s || !out) { set_error("rt_get_stats: NULL argument"); return RT_ERR_ARG; }
    if (s->multi_stats) { *out = s->stats_multi; return RT_OK; }
    int rc;
    if ((rc = enter())) return rc;
    const int dev = s->last_device >= 0 ? s->last_device : t_device;
    if (dev != t_device) HIP_TRY(hipSetDevice(dev));
    rc = device_stats(s, s->dev[dev], out);
    if (dev != t_device) (void)hipSetDevice(t_device);
    return rc;
}

namespace {

// Relative cost of `ncost` equal horizontal strips of the w x h frame, measured: the strips are rendered one by one at
// a reduced scale (a quarter or half the width and height while a strip keeps >= 256 Ki pixels, else full scale) and
// timed with HIP events, the better of two passes.  The bands of a multi-GPU render are cut from this (split_rows).
int row_costs(rt_scene *s, int device, int w, int h, int depth, int spp, int ncost, double *cost) {
    int q = 4;
    while (q > 1 && ((size_t)(w / q) * (size_t)(h / q)) / (size_t)ncost < (256u << 10)) q >>= 1;
    const int sw = std::max(1, w / q), sh = std::max(ncost, h / q);
    int rc;
    DeviceScene &D = s->dev[device];
    if ((rc = device_prepare(*s, D, device, sw, sh))) return rc;
    const int maxrows = (sh + ncost - 1) / ncost + 1;
    if ((rc = ensure(D.stage_rgba, D.stage_rgba_cap, 4 * (size_t)sw * (size_t)maxrows))) return rc;
    for (int k = 0; k < ncost; ++k) cost[k] = 1e300;
    for (int pass = 0; pass < 2; ++pass)
        for (int k = 0; k < ncost; ++k) {
            const int r0 = (int)((long long)sh * k / ncost), r1 = (int)((long long)sh * (k + 1) / ncost);
            if (r1 <= r0) { cost[k] = 0; continue; }
            if ((rc = render_band(s, device, sw, sh, depth, spp, 0, r0, r1 - r0, D.stage_rgba, nullptr, (hipStream_t)D.stream, false))) return rc;
            rt_stats st;
            if ((rc = device_stats(s, D, &st))) return rc;
            cost[k] = std::min(cost[k], st.ms_total);
        }
    return RT_OK;
}

// port of raytrace_amd/distributed.py: sub_bands with the piece sizes of piece_fractions (1 : 2 : ... : 2 : 1 — a small
// first piece puts data on the link early, a small last one leaves little travelling when the render has ended)
void sub_bands(int r0, int nr, int parts, int align, int *p0, int *pn) {
    const int units = (nr + align - 1) / align;
    double total = 0, run = 0;
    for (int k = 0; k < parts; ++k) total += (parts <= 2 || (k > 0 && k + 1 < parts)) ? 2.0 : 1.0;
    int done = 0;
    for (int k = 0; k < parts; ++k) {
        run += (parts <= 2 || (k > 0 && k + 1 < parts)) ? 2.0 : 1.0;
        const int upto = k + 1 < parts ? std::min(nr, std::max(done, (int)((double)units * (run / total) + 0.5) * align)) : nr;
        p0[k] = r0 + done; pn[k] = upto - done;
        done = upto;
    }
}

} // namespace

int scene_row_costs_impl(rt_scene *s, int w, int h, int depth, int spp, int ncost, double *cost) {
    int rc;
    if ((rc = check_render_args(s, w, h, depth, spp, 0, h))) return rc;
    if (!cost || ncost < 1 || ncost > h) { set_error("rt_scene_row_costs: bad argument"); return RT_ERR_ARG; }
    if ((rc = enter())) return rc;
    return row_costs(s, t_device, w, h, depth, spp, ncost, cost);
}

int render_multi_impl(rt_scene *s, int w, int h, int depth, int spp, uint64_t seed, int ngpus, const int *devices,
                               uint8_t *rgba_out, int32_t *hit_id_out) {
    int rc;
    if ((rc = check_render_args(s, w, h, depth, spp, 0, h))) return rc;
    if (ngpus < 1 || ngpus > RT_MAX_DEVICES) { set_error("rt_render_multi: ngpus must be between 1 and 16"); return RT_ERR_ARG; }
    int ndev = 0;
    if (!hip_ok(hipGetDeviceCount(&ndev), "hipGetDeviceCount") || ndev <= 0) {
        set_error("rt_render_multi: no HIP device available (this library has no CPU fallback)"); return RT_ERR_HIP;
    }
    int devs[RT_MAX_DEVICES];
    for (int k = 0; k < ngpus; ++k) {
        devs[k] = devices ? devices[k] : k;
        if (devs[k] < 0 || devs[k] >= ndev || devs[k] >= RT_MAX_DEVICES) { set_error("rt_render_multi: device index out of range"); return RT_ERR_ARG; }
    }
    const int caller = t_device, home = devs[0];
    const size_t npix = (size_t)w * (size_t)h;
    constexpr int PIECES = 4;
    struct Band { int dev, row0, nrows; size_t stage_off; bool staged; int p0[PIECES], pn[PIECES]; };
    Band band[RT_MAX_DEVICES];
    std::vector<hipEvent_t> ev;            // per (band, piece): rendered, arrived on the home device
    bool first_on[RT_MAX_DEVICES], peer_ok[RT_MAX_DEVICES];
    for (bool &f : first_on) f = true;
    for (bool &f : peer_ok) f = false;
    auto finish = [&](int code) {          // drain every stream that was used, restore the caller's device
        for (int k = 0; k < ngpus; ++k) {
            DeviceScene &D = s->dev[devs[k]];
            if (hipSetDevice(devs[k]) != hipSuccess) continue;
            if (D.stream && !hip_ok(hipStreamSynchronize((hipStream_t)D.stream), "hipStreamSynchronize") && code == RT_OK) code = RT_ERR_HIP;
            for (void *x : D.xstream) if (x && !hip_ok(hipStreamSynchronize((hipStream_t)x), "hipStreamSynchronize") && code == RT_OK) code = RT_ERR_HIP;
            if (D.copy_stream && !hip_ok(hipStreamSynchronize((hipStream_t)D.copy_stream), "hipStreamSynchronize") && code == RT_OK) code = RT_ERR_HIP;
            // (the home device's copies into the caller's buffers: nothing may still be writing them after an error return)
            if (D.copy_stream2 && !hip_ok(hipStreamSynchronize((hipStream_t)D.copy_stream2), "hipStreamSynchronize") && code == RT_OK) code = RT_ERR_HIP;
        }
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
        if (caller >= 0) (void)hipSetDevice(caller);
        t_device = caller;
        return code;
    };
    if ((rc = device_init(home))) return finish(rc);
    DeviceScene &H = s->dev[home];
    if ((rc = device_prepare(*s, H, home, w, h))) return finish(rc);
    if ((rc = copy_tools(H))) return finish(rc);
    // 1. bands of about equal measured cost
    int b0[RT_MAX_DEVICES] = {0}, bn[RT_MAX_DEVICES] = {h};
    if (ngpus > 1) {
        const int ncost = std::min(32, std::max(1, h / 8));
        if (s->multi_cost.size() != (size_t)ncost || s->multi_cost_key[0] != s->version || s->multi_cost_key[1] != (uint64_t)w ||
            s->multi_cost_key[2] != (uint64_t)h || s->multi_cost_key[3] != (uint64_t)depth || s->multi_cost_key[4] != (uint64_t)spp) {
            s->multi_cost.assign(ncost, 1.0);
            if ((rc = row_costs(s, home, w, h, depth, spp, ncost, s->multi_cost.data()))) return finish(rc);
            if ((rc = device_prepare(*s, H, home, w, h))) return finish(rc);
            const uint64_t key[5] = {s->version, (uint64_t)w, (uint64_t)h, (uint64_t)depth, (uint64_t)spp};
            std::copy(key, key + 5, s->multi_cost_key);
        }
        // devices[0] collects: it sends nothing, so it may render while the others' last pieces travel — one more of
        // the PIECES * 3/2 - 1 units a band's pieces are made of (distributed.py: plan_frame has the long form)
        double weight[RT_MAX_DEVICES];
        for (int k = 0; k < ngpus; ++k) weight[k] = 1.0;
        if (devs[0] != devs[ngpus - 1] || ngpus > 1) weight[0] = 1.0 + 1.0 / (2.0 * (PIECES - 1));
        split_rows(s->multi_cost.data(), ncost, h, ngpus, RT_TILE_H, b0, bn, weight);
    }
    // 2. buffers: the whole frame on the home device, a staging buffer per other device
    if (rgba_out && (rc = ensure(H.rgba, H.rgba_cap, bpp(s) * npix))) return finish(rc);
    if ((rc = ensure(H.hit_out, H.hit_out_cap, npix))) return finish(rc);
    size_t stage_pix[RT_MAX_DEVICES] = {0};
    for (int k = 0; k < ngpus; ++k) {
        Band &B = band[k];
        B.dev = devs[k]; B.row0 = b0[k]; B.nrows = bn[k];
        B.staged = B.dev != home || s->opt_multi_stage;
        B.stage_off = stage_pix[B.dev];
        if (B.staged) stage_pix[B.dev] += (size_t)w * (size_t)B.nrows;
        sub_bands(B.row0, B.nrows, PIECES, RT_TILE_H, B.p0, B.pn);
    }
    for (int k = 0; k < ngpus; ++k) {
        const int d = devs[k];
        if (!stage_pix[d]) continue;
        if ((rc = device_init(d))) return finish(rc);
        DeviceScene &D = s->dev[d];
        if ((rc = device_prepare(*s, D, d, w, h))) return finish(rc);
        if ((rc = copy_tools(D))) return finish(rc);
        if (rgba_out && (rc = ensure(D.stage_rgba, D.stage_rgba_cap, bpp(s) * stage_pix[d]))) return finish(rc);
        if ((rc = ensure(D.stage_hit, D.stage_hit_cap, stage_pix[d]))) return finish(rc);
        if (d != home) {                  // direct xGMI copies into the home device's memory
            const hipError_t e = hipDeviceEnablePeerAccess(home, 0);
            peer_ok[d] = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;   // else hipMemcpyPeerAsync stages through the host
            (void)hipGetLastError();
        }
    }
    // 3. every device is driven by a host thread of its own (enqueueing 6 kernels per piece costs the host ~0.1 ms: from
    // one thread, 8 devices x 4 pieces would take as long as a device needs for its whole band); a device's pieces are
    // enqueued in order, and a piece travels to the home device (peer-to-peer over xGMI, on the band's copy stream) while
    // the next one renders.  Scenes with fnorm callbacks stay on this thread (the caller's closure is not ours to run
    // concurrently).
    ev.assign((size_t)ngpus * PIECES * 2, nullptr);
    for (size_t i = 0; i < ev.size(); ++i)
        if (!hip_ok(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming), "hipEventCreate")) return finish(RT_ERR_HIP);
    auto drive = [&](int dev, std::string &err) -> int {        // all bands of one device, piece by piece
        int rc2;
        if ((rc2 = device_init(dev))) { err = rt_last_error(); return rc2; }
        DeviceScene &D = s->dev[dev];
        bool first = true;
        int npiece = 0;
        for (int p = 0; p < PIECES; ++p)
            for (int k = 0; k < ngpus; ++k) {
                Band &B = band[k];
                if (B.dev != dev || B.pn[p] <= 0) continue;
                const size_t foff = (size_t)B.p0[p] * (size_t)w, n = (size_t)B.pn[p] * (size_t)w;
                const size_t soff = B.stage_off + (size_t)(B.p0[p] - B.row0) * (size_t)w;
                uint8_t *dst = !rgba_out ? nullptr : (B.staged ? D.stage_rgba + bpp(s) * soff : H.rgba + bpp(s) * foff);
                int32_t *dhit = B.staged ? D.stage_hit + soff : H.hit_out + foff;
                // consecutive pieces alternate between the device's two lanes: the next piece's kernels fill the tails of this one's
                const int lane = npiece++ % RT_LANES;
                hipStream_t ls = nullptr;
                if ((rc2 = lane_stream(D, lane, ls))) { err = rt_last_error(); return rc2; }
                if ((rc2 = render_band(s, dev, w, h, depth, spp, seed, B.p0[p], B.pn[p], dst, dhit, ls, !first, nullptr, lane))) { err = rt_last_error(); return rc2; }
                first = false;
                first_on[dev] = false;
                hipEvent_t er = ev[((size_t)k * PIECES + p) * 2], ea = ev[((size_t)k * PIECES + p) * 2 + 1];
                bool ok = hip_ok(hipEventRecord(er, ls), "hipEventRecord");
                if (ok && B.staged) {
                    hipStream_t cs = (hipStream_t)D.copy_stream;
                    ok = hip_ok(hipStreamWaitEvent(cs, er, 0), "hipStreamWaitEvent") &&
                         (!rgba_out || hip_ok(hipMemcpyPeerAsync(H.rgba + bpp(s) * foff, home, dst, dev, bpp(s) * n, cs), "hipMemcpyPeerAsync")) &&
                         hip_ok(hipMemcpyPeerAsync(H.hit_out + foff, home, dhit, dev, n * sizeof(int32_t), cs), "hipMemcpyPeerAsync") &&
                         hip_ok(hipEventRecord(ea, cs), "hipEventRecord");
                }
                if (!ok) { err = rt_last_error(); return RT_ERR_HIP; }
            }
        return RT_OK;
    };
    {
        int distinct[RT_MAX_DEVICES], nd = 0;
        for (int k = 0; k < ngpus; ++k) { bool seen = false; for (int j = 0; j < nd; ++j) seen |= distinct[j] == devs[k]; if (!seen) distinct[nd++] = devs[k]; }
        int rcs[RT_MAX_DEVICES] = {0};
        std::string errs[RT_MAX_DEVICES];
        if (nd > 1 && s->normal_fn.empty()) {
            std::vector<std::thread> workers;
            for (int j = 1; j < nd; ++j) workers.emplace_back([&, j] { rcs[j] = drive(distinct[j], errs[j]); });
            rcs[0] = drive(distinct[0], errs[0]);
            for (std::thread &t : workers) t.join();
        } else {
            for (int j = 0; j < nd && (j == 0 || rcs[j - 1] == RT_OK); ++j) rcs[j] = drive(distinct[j], errs[j]);
        }
        for (int j = 0; j < nd; ++j) if (rcs[j] != RT_OK) { set_error(errs[j]); return finish(rcs[j]); }
    }
    // 4. from the home device to the caller's memory, piece by piece as they arrive
    if ((rc = device_init(home))) return finish(rc);
    for (int p = 0; p < PIECES; ++p)
        for (int k = 0; k < ngpus; ++k) {
            const Band &B = band[k];
            if (B.pn[p] <= 0) continue;
            const size_t foff = (size_t)B.p0[p] * (size_t)w, n = (size_t)B.pn[p] * (size_t)w;
            hipEvent_t arrived = ev[((size_t)k * PIECES + p) * 2 + (B.staged ? 1 : 0)];
            hipStream_t cs = (hipStream_t)H.copy_stream2;
            if (!hip_ok(hipStreamWaitEvent(cs, arrived, 0), "hipStreamWaitEvent")) return finish(RT_ERR_HIP);
            if (rgba_out && !hip_ok(hipMemcpyAsync(rgba_out + bpp(s) * foff, H.rgba + bpp(s) * foff, bpp(s) * n, hipMemcpyDeviceToHost, cs), "hipMemcpyAsync")) return finish(RT_ERR_HIP);
            if (hit_id_out && !hip_ok(hipMemcpyAsync(hit_id_out + foff, H.hit_out + foff, n * sizeof(int32_t), hipMemcpyDeviceToHost, cs), "hipMemcpyAsync")) return finish(RT_ERR_HIP);
        }
    if (!hip_ok(hipStreamSynchronize((hipStream_t)H.copy_stream2), "hipStreamSynchronize")) return finish(RT_ERR_HIP);
    if ((rc = finish(RT_OK))) return rc;
    // 5. statistics of the whole frame: counts summed over the devices, times = the slowest device
    rt_stats tot{};
    bool seen[RT_MAX_DEVICES] = {false};
    for (int k = 0; k < ngpus; ++k) {
        const int d = devs[k];
        if (seen[d] || first_on[d]) continue;
        seen[d] = true;
        rt_stats st;
        if (hipSetDevice(d) != hipSuccess || device_stats(s, s->dev[d], &st) != RT_OK) continue;
        tot.primary_rays += st.primary_rays; tot.shadow_rays += st.shadow_rays; tot.bounce_rays += st.bounce_rays;
        tot.tri_pairs += st.tri_pairs; tot.rect_pairs += st.rect_pairs; tot.exact_evals += st.exact_evals;
        tot.record_tests += st.record_tests; tot.mfma_blocks_primary += st.mfma_blocks_primary; tot.mfma_blocks_shadow += st.mfma_blocks_shadow;
        tot.ms_primary = std::max(tot.ms_primary, st.ms_primary); tot.ms_setup = std::max(tot.ms_setup, st.ms_setup);
        tot.ms_shadow = std::max(tot.ms_shadow, st.ms_shadow); tot.ms_final = std::max(tot.ms_final, st.ms_final);
        tot.ms_total = std::max(tot.ms_total, st.ms_total); tot.upload_ms = std::max(tot.upload_ms, st.upload_ms);
    }
    tot.prepare_ms = s->prepare_ms;
    for (int k = 0; k < ngpus; ++k) {          // how the bands reached devices[0]
        if (devs[k] == home) continue;
        if (peer_ok[devs[k]]) ++tot.multi_bands_direct; else ++tot.multi_bands_staged;
    }
    s->stats_multi = tot;
    s->multi_stats = true;
    if (caller >= 0) (void)hipSetDevice(caller);
    return RT_OK;
}

// rt_render_device without a single allocation: no hipMalloc / hipFree, no event or stream creation.  The scene must be
// resident already (rt_scene_prepare_device); work buffers come from the static device array `scratch` (64 MiB per GPU).
// A band that needs more than that is rendered as consecutive sub-bands that each fit; only a frame so wide that 16 rows
// do not fit is refused.  All calls on one GPU share the array: calls that overlap in time race.
int render_device_noalloc_impl(rt_scene *s, int w, int h, int depth, int spp, uint64_t seed, int row0, int nrows,
                                        void *d_rgba, void *d_hit_id, void *stream) {
    int rc;
    if ((rc = check_render_args(s, w, h, depth, spp, row0, nrows))) return rc;
    if (t_device < 0) { set_error("rt_render_device_noalloc: call rt_init(device) on this thread first"); return RT_ERR_ARG; }
    HIP_TRY(hipSetDevice(t_device));
    DeviceScene &D = s->dev[t_device];
    if (!host_ready(*s, w, h) || !D.stream || D.version != s->version || D.eye_version != s->eye_version) {
        set_error("rt_render_device_noalloc: the scene is not resident on this device for this image size: call rt_scene_prepare_device(scene, w, h) first");
        return RT_ERR_ARG;
    }
    if (!s->normal_fn.empty()) { set_error("rt_render_device_noalloc: scenes with fnorm callbacks need host staging and are not supported here"); return RT_ERR_UNSUPPORTED; }
    if (!levels_supported(s, depth)) return RT_ERR_UNSUPPORTED;
    if (!d_rgba && !d_hit_id) { set_error("rt_render_device_noalloc: no output buffer"); return RT_ERR_ARG; }
    const int levels = render_levels(s, depth);
    unsigned char *base = g_ctx[t_device].scratch;       // the device address, never the array's name
    // the most rows (all of them, or a multiple of the 16-row set-up tiles) whose buffers fit
    Workspace W;
    auto fits = [&](int rows) {
        return ws_carve(base, RT_STATIC_SCRATCH_BYTES, ws_needs(w, rows, (size_t)rows, s->npoint, levels, spp, d_hit_id == nullptr, d_rgba != nullptr, false), W) != 0;
    };
    int rows_fit = nrows;
    if (nrows > 0 && !fits(nrows)) {
        int lo = 0, hi = nrows / 16;                       // in units of 16 rows: lo fits (0 = nothing), hi may not
        while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (fits(mid * 16)) lo = mid; else hi = mid - 1; }
        rows_fit = lo * 16;
        if (rows_fit == 0) {                               // not even 16 rows (or the band has fewer): single rows, if one fits
            if (!fits(1)) {
                set_error("rt_render_device_noalloc: the image is too wide: one row of it needs more than the 64 MiB of static scratch");
                return RT_ERR_UNSUPPORTED;
            }
            rows_fit = 1;
            while (rows_fit < 15 && rows_fit < nrows && fits(rows_fit + 1)) ++rows_fit;
        }
        (void)fits(rows_fit);                              // W = the layout that is used
    }
    hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)D.stream;
    // (the array keeps whatever the last call left: no history of the lights' costs here, they are taken in their own order)
    if (nrows > 0) HIP_TRY(hipMemsetAsync(W.qcount + RT_QCOUNT_HIST, 0, (RT_QCOUNT_WORDS - RT_QCOUNT_HIST) * sizeof(unsigned), st));
    for (int r = 0; r < nrows; r += rows_fit) {
        const int rows = nrows - r < rows_fit ? nrows - r : rows_fit;
        const size_t off = (size_t)r * (size_t)w;
        if ((rc = render_rows(s, D, D.lane[0], W, (size_t)rows, w, h, depth, spp, seed, row0 + r, rows,
                              d_rgba ? (uint8_t *)d_rgba + bpp(s) * off : nullptr, d_hit_id ? (int32_t *)d_hit_id + off : nullptr,
                              st, r > 0, false))) return rc;
    }
    if (nrows == 0) { D.last_primary = 0; for (Lane &l : D.lane) { l.used = false; l.ev_used = 0; } D.stats_pending = false; }
    return RT_OK;
}

} // namespace rt
