// examples/cold_start.cpp — what a ONE-SHOT caller pays (the reference's main renders one image and exits,
// src/Main.hs:569-621): every constructor call of a scene through the C ABI, then the first rt_render into host memory,
// timed part by part with the steady-state render beside it.  The scene comes as the flat script tools/dump_scene.py
// writes (parsed before the clock starts).
//   usage: cold_start scene.txt [width height [depth]]      prints one JSON object
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "raytrace_gpu.h"

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
struct Op { char kind; int n = 0; std::vector<rt_material> m; double v[12] = {0}; };
#define CHECK(expr) do { if ((expr) < 0) { std::fprintf(stderr, "%s: %s\n", #expr, rt_last_error()); return 1; } } while (0)

static bool read_mat(char *&p, rt_material &m) {
    int c[6];
    for (int k = 0; k < 6; ++k) c[k] = (int)std::strtol(p, &p, 10);
    for (int k = 0; k < 3; ++k) { m.diffuse[k] = (uint8_t)c[k]; m.specular[k] = (uint8_t)c[3 + k]; }
    m._pad[0] = m._pad[1] = 0;
    m.specularity = std::strtod(p, &p); m.reflectivity = std::strtod(p, &p);
    return true;
}

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: cold_start scene.txt [width height [depth]]\n"); return 2; }
    const int w = argc > 3 ? std::atoi(argv[2]) : 3840, h = argc > 3 ? std::atoi(argv[3]) : 2160, depth = argc > 4 ? std::atoi(argv[4]) : 4;
    std::vector<Op> ops;
    {
        FILE *f = std::fopen(argv[1], "r");
        if (!f) { std::perror(argv[1]); return 2; }
        static char line[1 << 16];
        while (std::fgets(line, sizeof line, f)) {
            Op op; op.kind = line[0];
            char *p = line + 1;
            auto nums = [&](int n, int at) { for (int k = 0; k < n; ++k) op.v[at + k] = std::strtod(p, &p); };
            switch (op.kind) {
            case 'T': case 'R': op.m.resize(1); read_mat(p, op.m[0]); nums(9, 0); break;
            case 'C': op.n = (int)std::strtol(p, &p, 10); op.m.resize(op.n); for (auto &m : op.m) read_mat(p, m); nums(6, 0); break;
            case 'Y': op.m.resize(3); for (auto &m : op.m) read_mat(p, m); nums(3, 0); op.n = (int)std::strtol(p, &p, 10); nums(2, 3); break;
            case 'L': nums(5, 0); break;
            case 'A': nums(1, 0); break;
            default: break;
            }
            ops.push_back(std::move(op));
        }
        std::fclose(f);
    }
    const size_t npix = (size_t)w * h;
    const double t0 = now_ms();
    CHECK(rt_init(0));                                       // HIP runtime + device set-up: not this library's to shorten
    const double t_init = now_ms();
    rt_scene *s = nullptr;
    CHECK(rt_scene_create(&s));
    for (const Op &op : ops) {
        switch (op.kind) {
        case '[': CHECK(rt_begin_mconcat(s)); break;
        case ']': CHECK(rt_end_mconcat(s)); break;
        case '(': CHECK(rt_begin_mappend(s)); break;
        case ')': CHECK(rt_end_mappend(s)); break;
        case 'T': CHECK(rt_add_triangle(s, &op.m[0], op.v, op.v + 3, op.v + 6)); break;
        case 'R': CHECK(rt_add_rectangle(s, &op.m[0], op.v, op.v + 3, op.v + 6)); break;
        case 'C': CHECK(rt_add_cuboid(s, op.m.data(), op.n, op.v, op.v[3], op.v[4], op.v[5])); break;
        case 'Y': CHECK(rt_add_cylinder(s, &op.m[0], &op.m[1], &op.m[2], op.v, op.n, op.v[3], op.v[4])); break;
        case 'L': CHECK(rt_add_point_light(s, op.v[0], op.v[1], op.v + 2)); break;
        case 'A': CHECK(rt_add_ambient(s, op.v[0])); break;
        default: break;
        }
    }
    const double t_built = now_ms();
    uint8_t *rgba = (uint8_t *)rt_host_alloc(4 * npix);
    int32_t *hit = (int32_t *)rt_host_alloc(sizeof(int32_t) * npix);
    if (!rgba || !hit) { std::fprintf(stderr, "rt_host_alloc: %s\n", rt_last_error()); return 1; }
    const double t_alloc = now_ms();
    CHECK(rt_scene_prepare_device(s, w, h));                 // host index (guards, cones, records) + upload
    const double t_prep = now_ms();
    CHECK(rt_render(s, w, h, depth, 1, 0, 0, h, rgba, hit));
    const double t_first = now_ms();
    rt_stats st;
    CHECK(rt_get_stats(s, &st));
    double best = 1e300;
    for (int k = 0; k < 5; ++k) {
        const double a = now_ms();
        CHECK(rt_render(s, w, h, depth, 1, 0, 0, h, rgba, hit));
        best = std::min(best, now_ms() - a);
    }
    unsigned long long sum = 0;
    for (size_t i = 0; i < npix; i += 97) sum += rgba[4 * i] + (unsigned)(hit[i] & 0xff);
    std::printf("{\"leaves\": %d, \"width\": %d, \"height\": %d, \"rt_init_ms\": %.3f, \"scene_calls\": %zu, \"scene_build_ms\": %.3f, "
                "\"host_alloc_ms\": %.3f, \"prepare_device_ms\": %.3f, \"host_index_build_ms\": %.3f, \"upload_ms\": %.3f, "
                "\"first_render_ms\": %.3f, \"steady_render_ms\": %.3f, \"cold_start_ms\": %.3f, \"cold_start_with_rt_init_ms\": %.3f, \"checksum\": %llu}\n",
                rt_scene_num_leaves(s), w, h, t_init - t0, ops.size(), t_built - t_init, t_alloc - t_built, t_prep - t_alloc, st.prepare_ms, st.upload_ms,
                t_first - t_prep, best, (t_built - t_init) + (t_first - t_alloc), t_first - t0, sum);
    rt_host_free(rgba); rt_host_free(hit);
    rt_scene_destroy(s);
    return 0;
}
