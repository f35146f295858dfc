// raytrace.hpp — C++ host-side mirror of the reference's scene API (reference src/Main.hs), header-only,
// over the C ABI of libraytrace_gpu.so (raytrace_gpu.h).
//
// The reference is compiled code (Haskell); its toolchain is absent from the build image, so the host
// side a Haskell maintainer would write is mirrored here in C++ with the reference's
// names, argument order and error behaviour:
//
//   Vector3D, Material, colours                     Main.hs:24-86, 178-185, 326-355
//   Shape: rectangle triangle sphere cube colorcube cuboid colorcuboid cylinder, mempty, <> (operator+), mconcat
//                                                   Main.hs:189-217, 226, 363, 392, 286-320, 725
//   Light: pointLight ambient, <> (operator+), mconcat   Main.hs:423-441, 467, 490
//   linearInterpolation (gaussian fnorm of main)    Main.hs:937-988, 580-584
//   fixedCamera, trace, parallelTrace, saveBmpImage Main.hs:506, 541, 623, 621
//
// A Shape is a tree of constructor calls (the reference's closure tree as data); parallelTrace flattens it
// into rt_add_* / rt_begin_* calls in depth-first order and renders the whole image on the GPU.
#pragma once
#include <cmath>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "raytrace_gpu.h"

namespace raytrace {

struct Vector3D {                                             // Main.hs:24-29
    double x, y, z;
};
inline Vector3D operator+(Vector3D a, Vector3D b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vector3D operator-(Vector3D a, Vector3D b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vector3D scalar(double d, Vector3D v) { return {d * v.x, d * v.y, d * v.z}; }          // Main.hs:84

struct PixelRGB8 { uint8_t r, g, b; };

struct Material {                                             // Main.hs:178-185
    PixelRGB8 mat_diffuse, mat_specular;
    double mat_specularity = 1.0, mat_reflectivity = 0.0;
};
inline Material colour(uint8_t r, uint8_t g, uint8_t b) { return {{r, g, b}, {r, g, b}, 1.0, 0.0}; }   // Main.hs:343-354
inline const Material black = colour(0, 0, 0), white = colour(255, 255, 255), red = colour(255, 0, 0),
                      green = colour(0, 255, 0), blue = colour(0, 0, 255), magenta = colour(255, 0, 255),
                      cyan = colour(0, 255, 255), yellow = colour(255, 255, 0), orange = colour(255, 134, 0),
                      orchid = colour(153, 50, 204), aquamarine = colour(69, 139, 116);

// ---- Shape ------------------------------------------------------------------------------------
struct ShapeNode;
using Shape = std::shared_ptr<const ShapeNode>;
struct ShapeNode {
    enum Kind { Empty, Rect, Tri, Sph, Cuboid, Cyl, Append, Concat, GaussNormals, FnNormals } kind = Empty;
    std::function<Vector3D(double, double)> fnorm;   // FnNormals: the closure of Main.hs:956-965
    std::vector<Material> mats;
    Vector3D a{}, b{}, c{};          // points / vectors of the constructor
    double p[4] = {0, 0, 0, 0};      // scalars of the constructor
    int n = 0;
    std::vector<Shape> kids;
};
inline Shape node(ShapeNode n) { return std::make_shared<const ShapeNode>(std::move(n)); }

inline const Shape mempty = node({});                                                          // Main.hs:204
inline Shape mappend(Shape l, Shape r) { ShapeNode n; n.kind = ShapeNode::Append; n.kids = {std::move(l), std::move(r)}; return node(n); }
inline Shape operator+(Shape l, Shape r) { return mappend(std::move(l), std::move(r)); }      // <>
inline Shape mconcat(std::vector<Shape> xs) { ShapeNode n; n.kind = ShapeNode::Concat; n.kids = std::move(xs); return node(n); }

inline Shape rectangle(Material m, Vector3D point, Vector3D width, Vector3D height) {          // Main.hs:226
    ShapeNode n; n.kind = ShapeNode::Rect; n.mats = {m}; n.a = point; n.b = width; n.c = height; return node(n);
}
inline Shape triangle(Material m, Vector3D pa, Vector3D pb, Vector3D pc) {                     // Main.hs:363
    ShapeNode n; n.kind = ShapeNode::Tri; n.mats = {m}; n.a = pa; n.b = pb; n.c = pc; return node(n);
}
inline Shape sphere(Material m, Vector3D center, double radius) {                              // Main.hs:392
    ShapeNode n; n.kind = ShapeNode::Sph; n.mats = {m}; n.a = center; n.p[0] = radius; return node(n);
}
inline Shape colorcuboid(std::vector<Material> ms, Vector3D point, double l, double h, double d) {   // Main.hs:301
    if (ms.empty()) throw std::invalid_argument("colorcuboid: list of materials must not be empty.");   // :302
    ShapeNode n; n.kind = ShapeNode::Cuboid; n.mats = std::move(ms); n.a = point; n.p[0] = l; n.p[1] = h; n.p[2] = d; return node(n);
}
inline Shape cuboid(Material m, Vector3D point, double l, double h, double d) { return colorcuboid({m}, point, l, h, d); }   // :296
inline Shape colorcube(std::vector<Material> ms, Vector3D point, double l) { return colorcuboid(std::move(ms), point, l, l, l); }   // :291
inline Shape cube(Material m, Vector3D point, double l) { return colorcube({m}, point, l); }  // :286
inline Shape cylinder(Material topM, Material botM, Material mantleM, Vector3D point, int n_, double h, double r) {   // Main.hs:725
    ShapeNode n; n.kind = ShapeNode::Cyl; n.mats = {topM, botM, mantleM}; n.a = point; n.n = n_; n.p[0] = h; n.p[1] = r; return node(n);
}

// ---- Light ------------------------------------------------------------------------------------
struct LightSrc { bool point; double diffuse, specular, f; Vector3D pos; Shape world; };   // world: the Shape a point light closed over (null: the traced one)
using Light = std::vector<LightSrc>;                                                           // a sum of sources, Main.hs:436-441
inline Light pointLight(double diffuse, double specular, Vector3D lpos) { return {{true, diffuse, specular, 0, lpos, nullptr}}; }   // Main.hs:467 (world bound at trace)
// pointLight world diffuse specular lpos, as Main.hs:467 has it: the light tests its shadow rays against `world`, whatever is traced
inline Light pointLight(Shape world, double diffuse, double specular, Vector3D lpos) { return {{true, diffuse, specular, 0, lpos, std::move(world)}}; }
inline Light ambient(double f) { return {{false, 0, 0, f, {0, 0, 0}, nullptr}}; }              // Main.hs:490
inline Light operator+(Light a, const Light &b) { a.insert(a.end(), b.begin(), b.end()); return a; }
inline Light mconcat(const std::vector<Light> &ls) { Light out; for (const Light &l : ls) out.insert(out.end(), l.begin(), l.end()); return out; }

// ---- linearInterpolation, Main.hs:937-988 -----------------------------------------------------------
inline double gaussian(double x, double y) { return std::exp(-(x * x + y * y) * 4); }         // Main.hs:580
enum class FNorm { Nothing, Gaussian, Closure };                                               // Main.hs:581-584 runs on the GPU
using FNormFn = std::function<Vector3D(double, double)>;                                       // Maybe (Double -> Double -> Vector3D)

// `Just fnorm` with an arbitrary closure (the sinc and sine-cosine surfaces, Main.hs:586-605)
inline std::pair<Shape, Light> linearInterpolation(const std::function<double(double, double)> &f, FNorm fnorm,
                                                   std::pair<double, double> p1, std::pair<double, double> p2,
                                                   double step, Vector3D origin, double scale, FNormFn closure);
inline std::pair<Shape, Light> linearInterpolation(const std::function<double(double, double)> &f, FNormFn closure,
                                                   std::pair<double, double> p1, std::pair<double, double> p2,
                                                   double step, Vector3D origin, double scale) {
    return linearInterpolation(f, FNorm::Closure, p1, p2, step, origin, scale, std::move(closure));
}
inline std::pair<Shape, Light> linearInterpolation(const std::function<double(double, double)> &f, FNorm fnorm,
                                                   std::pair<double, double> p1, std::pair<double, double> p2,
                                                   double step, Vector3D origin, double scale, FNormFn closure) {
    const double x1 = p1.first, y1 = p1.second, x2 = p2.first, y2 = p2.second;
    const double x = (x2 + x1) / 2, y = (y2 + y1) / 2, w = x2 - x1, h = y2 - y1, scaleInv = 1.0 / scale;
    const Vector3D offset = origin - scalar(scale, {x, 0, y});
    Material mat = white; mat.mat_specularity = 1;
    auto fp = [&](double a, double b) { return offset + scalar(scale, Vector3D{a, f(a, b), b}); };
    auto steps = [&](double a, double b) { std::vector<double> v; for (double t = a; t < b; t = t + step) v.push_back(t); return v; };
    std::vector<Shape> cells;
    for (double gx : steps(x1, x2))
        for (double gy : steps(y1, y2)) {
            const double xp = gx + step, yp = gy + step;
            Shape two = mconcat(std::vector<Shape>{triangle(mat, fp(gx, yp), fp(xp, yp), fp(gx, gy)), triangle(mat, fp(gx, gy), fp(xp, yp), fp(xp, gy))});
            if (fnorm == FNorm::Gaussian) {
                ShapeNode n; n.kind = ShapeNode::GaussNormals; n.kids = {two}; n.a = offset; n.p[0] = scaleInv;
                two = node(n);
            } else if (fnorm == FNorm::Closure) {                  // `Just fnorm` for any fnorm: evaluated on the host
                if (!closure) throw std::invalid_argument("linearInterpolation: FNorm::Closure needs a function");
                ShapeNode n; n.kind = ShapeNode::FnNormals; n.kids = {two}; n.a = offset; n.p[0] = scaleInv; n.fnorm = closure;
                two = node(n);
            }
            cells.push_back(two);
        }
    Shape shape = mconcat(std::vector<Shape>{mconcat(cells), rectangle(aquamarine, origin, {1.5 * scale * w, 0, 0}, {0, 0, -(1.5 * scale * h)})});
    Light light = mconcat(std::vector<Light>{pointLight(0.3, 0.6, origin + Vector3D{scale * w / 4, scale * 10, scale * w / 4}), ambient(0.3)});
    return {shape, light};
}

inline std::pair<Shape, Light> linearInterpolation(const std::function<double(double, double)> &f, FNorm fnorm,
                                                   std::pair<double, double> p1, std::pair<double, double> p2,
                                                   double step, Vector3D origin, double scale) {
    return linearInterpolation(f, fnorm, p1, p2, step, origin, scale, nullptr);
}

// ---- rendering ----------------------------------------------------------------------------------
struct Camera { int width, height; };
inline Camera fixedCamera(int width, int height) { return {width, height}; }                  // Main.hs:506
struct Tracer { int depth; Light lights; Shape world; };
inline Tracer trace(int depth, Light lights, Shape world) { return {depth, std::move(lights), std::move(world)}; }   // Main.hs:541

struct Image { int w = 0, h = 0; std::vector<uint8_t> rgba; std::vector<int32_t> hit_id;
               PixelRGB8 at(int x, int y) const { const uint8_t *p = &rgba[4 * ((size_t)y * w + x)]; return {p[0], p[1], p[2]}; } };

namespace detail {
inline void check(int rc, const char *what) { if (rc < 0) throw std::runtime_error(std::string(what) + ": " + rt_last_error()); }
inline rt_material cmat(const Material &m) {
    rt_material c{};
    c.diffuse[0] = m.mat_diffuse.r; c.diffuse[1] = m.mat_diffuse.g; c.diffuse[2] = m.mat_diffuse.b;
    c.specular[0] = m.mat_specular.r; c.specular[1] = m.mat_specular.g; c.specular[2] = m.mat_specular.b;
    c.specularity = m.mat_specularity; c.reflectivity = m.mat_reflectivity;
    return c;
}
// rt_normal_fn trampoline: `user` is the std::function of a FnNormals node (alive as long as the Shape is)
inline void fnorm_trampoline(int n, const double *xy, double *normals, void *user) {
    const FNormFn &fn = *static_cast<const FNormFn *>(user);
    for (int i = 0; i < n; ++i) {
        const Vector3D v = fn(xy[2 * i], xy[2 * i + 1]);
        normals[3 * i] = v.x; normals[3 * i + 1] = v.y; normals[3 * i + 2] = v.z;
    }
}
inline void emit(rt_scene *s, const Shape &sh) {
    const ShapeNode &n = *sh;
    const double a[3] = {n.a.x, n.a.y, n.a.z}, b[3] = {n.b.x, n.b.y, n.b.z}, c[3] = {n.c.x, n.c.y, n.c.z};
    switch (n.kind) {
    case ShapeNode::Empty: check(rt_begin_mconcat(s), "mempty"); check(rt_end_mconcat(s), "mempty"); break;
    case ShapeNode::Rect: { rt_material m = cmat(n.mats[0]); check(rt_add_rectangle(s, &m, a, b, c), "rectangle"); break; }
    case ShapeNode::Tri: { rt_material m = cmat(n.mats[0]); check(rt_add_triangle(s, &m, a, b, c), "triangle"); break; }
    case ShapeNode::Sph: { rt_material m = cmat(n.mats[0]); check(rt_add_sphere(s, &m, a, n.p[0]), "sphere"); break; }
    case ShapeNode::Cuboid: {
        std::vector<rt_material> ms; for (const Material &m : n.mats) ms.push_back(cmat(m));
        check(rt_add_cuboid(s, ms.data(), (int)ms.size(), a, n.p[0], n.p[1], n.p[2]), "colorcuboid"); break; }
    case ShapeNode::Cyl: { rt_material t = cmat(n.mats[0]), bo = cmat(n.mats[1]), ma = cmat(n.mats[2]);
        check(rt_add_cylinder(s, &t, &bo, &ma, a, n.n, n.p[0], n.p[1]), "cylinder"); break; }
    case ShapeNode::Append: check(rt_begin_mappend(s), "mappend"); emit(s, n.kids[0]); emit(s, n.kids[1]); check(rt_end_mappend(s), "mappend"); break;
    case ShapeNode::Concat: check(rt_begin_mconcat(s), "mconcat"); for (const Shape &k : n.kids) emit(s, k); check(rt_end_mconcat(s), "mconcat"); break;
    case ShapeNode::GaussNormals: {
        const int first = rt_scene_num_leaves(s);
        emit(s, n.kids[0]);
        const double params[4] = {n.p[0], n.a.x, n.a.y, n.a.z};
        check(rt_set_normal_mode(s, first, rt_scene_num_leaves(s) - first, RT_NORMAL_GAUSSIAN, params, 4), "normals"); break; }
    case ShapeNode::FnNormals: {
        const int first = rt_scene_num_leaves(s);
        emit(s, n.kids[0]);
        const double params[4] = {n.p[0], n.a.x, n.a.y, n.a.z};
        check(rt_set_normal_callback(s, first, rt_scene_num_leaves(s) - first, params, fnorm_trampoline,
                                     const_cast<FNormFn *>(&n.fnorm)), "normals"); break; }
    }
}
// the lights; a point light that closed over a Shape other than the traced world gets that Shape as a shadow world (once per Shape)
inline void emit_lights(rt_scene *s, const Tracer &tr) {
    std::vector<std::pair<const ShapeNode *, int>> own;
    auto world_of = [&](const LightSrc &l) {
        if (!l.point || !l.world || l.world == tr.world) return 0;
        for (const auto &kv : own) if (kv.first == l.world.get()) return kv.second;
        const int id = rt_begin_shadow_world(s);
        check(id, "rt_begin_shadow_world");
        emit(s, l.world);
        check(rt_end_shadow_world(s), "rt_end_shadow_world");
        own.emplace_back(l.world.get(), id);
        return id;
    };
    std::vector<int> worlds;
    for (const LightSrc &l : tr.lights) worlds.push_back(world_of(l));
    for (size_t i = 0; i < tr.lights.size(); ++i) {
        const LightSrc &l = tr.lights[i];
        const double p[3] = {l.pos.x, l.pos.y, l.pos.z};
        check(!l.point ? rt_add_ambient(s, l.f) : (worlds[i] ? rt_add_point_light_world(s, worlds[i], l.diffuse, l.specular, p)
                                                             : rt_add_point_light(s, l.diffuse, l.specular, p)), "light");
    }
}
} // namespace detail

// parallelTrace w h camera (trace depth lights world) — Main.hs:623-637: the whole image in one call.
inline Image parallelTrace(int w, int h, Camera camera, const Tracer &tr, bool want_hit_ids = false) {
    if (camera.width != w || camera.height != h) throw std::invalid_argument("parallelTrace: camera and image size differ");
    rt_scene *s = nullptr;
    detail::check(rt_scene_create(&s), "rt_scene_create");
    std::unique_ptr<rt_scene, void (*)(rt_scene *)> guard(s, rt_scene_destroy);
    detail::emit(s, tr.world);
    detail::emit_lights(s, tr);
    Image img; img.w = w; img.h = h; img.rgba.resize((size_t)w * h * 4);
    if (want_hit_ids) img.hit_id.resize((size_t)w * h);
    detail::check(rt_render(s, w, h, tr.depth, 1, 0, 0, h, img.rgba.data(), want_hit_ids ? img.hit_id.data() : nullptr), "rt_render");
    return img;
}

// The same frame on several GPUs of this process at once (rt_render_multi): bands of equal measured cost, gathered on
// devices[0] over xGMI, copied to host memory from there.
inline Image parallelTraceMulti(const std::vector<int> &devices, int w, int h, Camera camera, const Tracer &tr, bool want_hit_ids = false) {
    if (camera.width != w || camera.height != h) throw std::invalid_argument("parallelTrace: camera and image size differ");
    rt_scene *s = nullptr;
    detail::check(rt_scene_create(&s), "rt_scene_create");
    std::unique_ptr<rt_scene, void (*)(rt_scene *)> guard(s, rt_scene_destroy);
    detail::emit(s, tr.world);
    detail::emit_lights(s, tr);
    Image img; img.w = w; img.h = h; img.rgba.resize((size_t)w * h * 4);
    if (want_hit_ids) img.hit_id.resize((size_t)w * h);
    detail::check(rt_render_multi(s, w, h, tr.depth, 1, 0, (int)devices.size(), devices.data(), img.rgba.data(),
                                  want_hit_ids ? img.hit_id.data() : nullptr), "rt_render_multi");
    return img;
}

inline void saveBmpImage(const std::string &path, const Image &img) {                          // Main.hs:621
    detail::check(rt_write_bmp(path.c_str(), img.rgba.data(), img.w, img.h), "saveBmpImage");
}

} // namespace raytrace
