// examples/main.cpp — the reference's `main` (src/Main.hs:569-621) on the MI355X render path:
// the gaussian height field of linearInterpolation, 1024 x 1024, trace depth 4, written to trace.bmp.
//   usage: main [width height [out.bmp [hit_ids.bin]]]
#include <cstdio>
#include <cstdlib>

#include "raytrace.hpp"

using namespace raytrace;

int main(int argc, char **argv) {
    const int w = argc > 2 ? std::atoi(argv[1]) : 1024, h = argc > 2 ? std::atoi(argv[2]) : 1024;   // Main.hs:619-620
    const char *out = argc > 3 ? argv[3] : "trace.bmp";
    try {
        auto [world, lights] = linearInterpolation(gaussian, FNorm::Gaussian, {-1.0, -1.0}, {1.0, 1.0}, 0.2,
                                                   Vector3D{0.01, -7, -25}, 10);                     // Main.hs:578-585
        const Camera camera = fixedCamera(w, h);                                                    // Main.hs:606
        const Image img = parallelTrace(w, h, camera, trace(4, lights, world), argc > 4);           // Main.hs:571
        saveBmpImage(out, img);                                                                     // Main.hs:621
        if (argc > 4) {
            FILE *f = std::fopen(argv[4], "wb");
            if (f) { std::fwrite(img.hit_id.data(), sizeof(int32_t), img.hit_id.size(), f); std::fclose(f); }
        }
        const PixelRGB8 c = img.at(w / 2, h / 2);
        std::printf("wrote %s (%d x %d), centre pixel (%d, %d, %d)\n", out, w, h, c.r, c.g, c.b);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
