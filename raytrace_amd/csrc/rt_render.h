// rt_render.h — what the C ABI (rt_abi.hip) forwards to: one function of the host runtime (rt_render.hip) per entry point
// of include/raytrace_gpu.h that touches a device.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "rt_scene.h"

namespace rt {

int init_impl(int device);
int scene_prepare_device_impl(rt_scene *s, int w, int h);
void scene_destroy_impl(rt_scene *s);
int render_device_noalloc_impl(rt_scene *s, int w, int h, int depth, int spp, uint64_t seed, int row0, int nrows, void *d_rgba, void *d_hit_id, void *stream);
int render_device_impl(rt_scene *s, int w, int h, int depth, int spp, uint64_t seed, int row0, int nrows, void *d_rgba, void *d_hit_id, void *stream);
int render_multi_impl(rt_scene *s, int w, int h, int depth, int spp, uint64_t seed, int ngpus, const int *devices, uint8_t *rgba_out, int32_t *hit_id_out);
int render_impl(rt_scene *s, int w, int h, int depth, int spp, uint64_t seed, int row0, int nrows, uint8_t *rgba_out, int32_t *hit_id_out);
void *host_alloc_impl(size_t bytes);
void host_free_impl(void *p);
int synchronize_impl(void);
int get_stats_impl(rt_scene *s, rt_stats *out);
int scene_row_costs_impl(rt_scene *s, int w, int h, int depth, int spp, int ncost, double *cost);

} // namespace rt
