// TEST INFRASTRUCTURE.  The two files of the reference that compile without GLM -- bitmap_image.hpp (the BMP writer main.cpp:107,
// 150, 156 uses and the skybox reader of scene.cpp:21-23, 136-139) and config.h (the flag parser, config.h:35-99) -- built from
// the sources WHERE THEY LIE under /root/reference (oracle/Makefile, target ref: -I/root/reference, output only in oracle/_ref/), with
// this driver around them.  The rest of the reference (main.cpp, scene.*, triangles.h, material.h, ray.h) needs GLM, which this image
// lacks, and is not built.  Nothing of the product links or runs this; tests/test_ref_parts.py compares the oracle's and the
// product's BMP writer, the skybox texel order and the front end's flag parser with it.
//
//   ref_parts bmpwrite W H in.rgb out.bmp   in.rgb = W*H*3 bytes (r, g, b), row-major from the top row: bitmap_image image(W, H);
//                                           image.clear(); image.set_pixel(x, y, r, g, b) for every pixel; image.save_image(out)
//   ref_parts bmpread in.bmp out.rgb        bitmap_image(in) as scene.cpp:22 loads the skybox; prints "W H"; out.rgb = get_pixel(x, y)
//                                           .red / .green / .blue for every pixel, row-major from y = 0
//   ref_parts config [flags ...]            Config::get().set_config(argc, argv); prints every field as "name value" lines
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "bitmap_image.hpp"
#include "config.h"

int main(int argc, char **argv) {
    if (argc >= 6 && std::strcmp(argv[1], "bmpwrite") == 0) {
        const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
        std::vector<unsigned char> rgb(static_cast<size_t>(w) * h * 3);
        std::ifstream in(argv[4], std::ios::binary);
        in.read(reinterpret_cast<char *>(rgb.data()), static_cast<std::streamsize>(rgb.size()));
        if (!in) return 2;
        bitmap_image image(w, h);
        image.clear();
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const unsigned char *p = &rgb[(static_cast<size_t>(y) * w + x) * 3];
                image.set_pixel(x, y, p[0], p[1], p[2]);
            }
        image.save_image(argv[5]);
        return 0;
    }
    if (argc >= 4 && std::strcmp(argv[1], "bmpread") == 0) {
        bitmap_image image{std::string(argv[2])};
        if (!image) return 3;
        std::printf("%u %u\n", image.width(), image.height());
        std::vector<unsigned char> rgb(static_cast<size_t>(image.width()) * image.height() * 3);
        for (unsigned y = 0; y < image.height(); ++y)
            for (unsigned x = 0; x < image.width(); ++x) {
                const auto c = image.get_pixel(x, y);
                unsigned char *p = &rgb[(static_cast<size_t>(y) * image.width() + x) * 3];
                p[0] = c.red; p[1] = c.green; p[2] = c.blue;
            }
        std::ofstream out(argv[3], std::ios::binary);
        out.write(reinterpret_cast<const char *>(rgb.data()), static_cast<std::streamsize>(rgb.size()));
        return out ? 0 : 4;
    }
    if (argc >= 2 && std::strcmp(argv[1], "config") == 0) {
        Config &c = Config::get();
        c.set_config(argc - 1, argv + 1);      // (argv[1] = "config" plays the program name)
        std::printf("height %d\nwidth %d\nrays_per_pixel %d\nmax_ray_reflections %d\nmedian %d\ngauss %d\neps %.9g\nerror %.9g\nupdate %d\n"
                    "gamma_correction %.9g\nmodel_path %s\nmodel_name %s\nskybox %s\ntime_limit %d\nseed %u\n",
                    c.height, c.width, c.rays_per_pixel, c.max_ray_reflections, c.median, c.gauss, static_cast<double>(c.eps),
                    static_cast<double>(c.error), c.update, static_cast<double>(c.gamma_correction), c.model_path.c_str(),
                    c.model_name.c_str(), c.skybox.c_str(), c.time_limit, c.getSeed());
        return 0;
    }
    std::fprintf(stderr, "usage: ref_parts bmpwrite W H in.rgb out.bmp | bmpread in.bmp out.rgb | config [flags]\n");
    return 1;
}
