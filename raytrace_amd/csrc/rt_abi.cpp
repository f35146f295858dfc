// rt_abi.hip — the C ABI of the render path (include/raytrace_gpu.h; construction and introspection are in rt_scene.cpp):
// plain C entry points that forward to the host runtime (rt_render.hip).  No exception or abort crosses the boundary: the
// runtime reports through return codes and rt_last_error().
#include "rt_render.h"

extern "C" {

int rt_init(int device) { return rt::init_impl(device); }
int rt_scene_prepare_device(rt_scene *s, int w, int h) { return rt::scene_prepare_device_impl(s, w, h); }
void rt_scene_destroy(rt_scene *s) { rt::scene_destroy_impl(s); }
int rt_render_device_noalloc(rt_scene *s, int w, int h, int depth, int spp, uint64_t seed, int row0, int nrows, void *d_rgba, void *d_hit_id, void *stream) { return rt::render_device_noalloc_impl(s, w, h, depth, spp, seed, row0, nrows, d_rgba, d_hit_id, stream); }
int rt_render_device(rt_scene *s, int w, int h, int depth, int spp, uint64_t seed, int row0, int nrows, void *d_rgba, void *d_hit_id, void *stream) { return rt::render_device_impl(s, w, h, depth, spp, seed, row0, nrows, d_rgba, d_hit_id, stream); }
int rt_render_multi(rt_scene *s, int w, int h, int depth, int spp, uint64_t seed, int ngpus, const int *devices, uint8_t *rgba_out, int32_t *hit_id_out) { return rt::render_multi_impl(s, w, h, depth, spp, seed, ngpus, devices, rgba_out, hit_id_out); }
int rt_render(rt_scene *s, int w, int h, int depth, int spp, uint64_t seed, int row0, int nrows, uint8_t *rgba_out, int32_t *hit_id_out) { return rt::render_impl(s, w, h, depth, spp, seed, row0, nrows, rgba_out, hit_id_out); }
void * rt_host_alloc(size_t bytes) { return rt::host_alloc_impl(bytes); }
void rt_host_free(void *p) { rt::host_free_impl(p); }
int rt_synchronize(void) { return rt::synchronize_impl(); }
int rt_get_stats(rt_scene *s, rt_stats *out) { return rt::get_stats_impl(s, out); }
int rt_scene_row_costs(rt_scene *s, int w, int h, int depth, int spp, int ncost, double *cost) { return rt::scene_row_costs_impl(s, w, h, depth, spp, ncost, cost); }

} // extern "C"
