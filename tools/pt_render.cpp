// pt_render -- command-line front end with the reference's flag set (config.h:35-99) on top of libpt_hip.so.
//
// It plays the part of the reference's main() (main.cpp:87-215): parse flags, load the model, run the passes, write
// previews every `-UPDATE` passes, stop at the `-TL` time limit, print the per-pass progress lines, then resolve,
// and write "<date>  <ms>   <n> of <rpp>  max_disp .. min_disp .. aver_disp ...bmp" plus "../result.bmp".
// The passes themselves run on the GPU through the C ABI; nothing here computes radiance.
//
// The accumulators stay on the device for the whole frame (pt_session): they cross PCIe only when a preview or the final
// image needs them.  With -TL the pass slices are kept to about 75 ms (from the measured time per pass, never less than
// one pass), so the time limit is checked at the reference's granularity -- before a pass starts (main.cpp:111-114) --
// whatever -UPDATE is.
//
// Extra flags (not in the reference): -OUT <file> writes only that file instead of the two reference outputs,
// -DEVICE <n> selects the HIP device, -QUIET 1 drops the per-pass lines, -TIMING 1 prints one JSON line with the
// seconds spent in each phase (HIP start-up, load, render, read-back, resolve, BMP write) on stderr.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <string>
#include <vector>

#include "pt_hip.h"

namespace {

struct Options {   // defaults: config.h:16-29
    int height = 512, width = 512, rays_per_pixel = 20, max_ray_reflections = 8, median = 0, gauss = 0;
    float eps = 1e-4f, error = 0.001f;
    int update = 32;
    float gamma_correction = 1 / 2.2f;
    std::string model_path = "../models/", model_name = "Tor.obj", skybox;
    int seed = 42, time_limit = 0;
    std::string out;
    int device = 0, quiet = 0, timing = 0;
};

long long now_ms() {
    using namespace std::chrono;
    return duration_cast<milliseconds>(system_clock::now().time_since_epoch()).count();
}

void parse(int argc, char **argv, Options &o) {   // pairs `flag value` from argv[1] on, unknown flags ignored
    for (int i = 1; i < argc - 1; i += 2) {
        const std::string f = argv[i];
        const char *v = argv[i + 1];
        if (f == "--H") o.height = std::atoi(v);
        if (f == "--W") o.width = std::atoi(v);
        if (f == "-RPP") o.rays_per_pixel = std::atoi(v);
        if (f == "-MRR") o.max_ray_reflections = std::atoi(v);
        if (f == "-EPS") o.eps = static_cast<float>(std::atof(v));
        if (f == "-ERR") o.error = static_cast<float>(std::atof(v));
        if (f == "-MEDIAN") { o.median = std::atoi(v); o.gauss = 0; }
        if (f == "-UPDATE") o.update = std::atoi(v);
        if (f == "-MODEL_PATH") o.model_path = v;
        if (f == "-MODEL_NAME") o.model_name = v;
        if (f == "-GAUSS") { o.gauss = std::atoi(v); o.median = 0; }
        if (f == "-GAMMA") o.gamma_correction = static_cast<float>(std::atof(v));
        if (f == "-SKYBOX") o.skybox = v;
        if (f == "-SEED") o.seed = std::atoi(v);
        if (f == "-TL") o.time_limit = std::atoi(v);
        if (f == "-OUT") o.out = v;
        if (f == "-DEVICE") o.device = std::atoi(v);
        if (f == "-QUIET") o.quiet = std::atoi(v);
        if (f == "-TIMING") o.timing = std::atoi(v);
    }
}

int die(const char *what) {
    std::cerr << what << ": " << pt_last_error() << std::endl;
    return 1;
}

}  // namespace

int main(int argc, char **argv) {
    const long long start_time = now_ms();
    Options o;
    parse(argc, argv, o);
    if (o.width <= 0 || o.height <= 0) {
        std::cerr << "pt_render: --W and --H must be positive" << std::endl;
        return 2;
    }
    const unsigned seed = o.seed < 0 ? static_cast<unsigned>(std::time(nullptr)) : static_cast<unsigned>(o.seed);   // config.h:101-104

    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const clk::time_point t_begin = clk::now();
    if (pt_device_count() < 1) {   // first HIP call: runtime start-up
        std::cerr << "pt_render: no HIP device (the integrator has no CPU fallback)" << std::endl;
        return 1;
    }
    const clk::time_point t_hip = clk::now();
    pt_scene *scene = nullptr;
    if (pt_scene_load_obj(o.model_path.c_str(), o.model_name.c_str(), o.device, &scene) != PT_OK) return die("pt_render");
    if (!o.skybox.empty() && pt_scene_set_skybox_bmp(scene, o.skybox.c_str()) != PT_OK) return die("pt_render");   // scene.cpp:20-22
    const clk::time_point t_load = clk::now();

    const size_t px = static_cast<size_t>(o.width) * o.height;
    // page-locked accumulators: the read-back then runs at PCIe speed without staging copies
    struct Pinned {
        void *p;
        explicit Pinned(size_t bytes) : p(pt_host_alloc(bytes)) {}
        ~Pinned() { pt_host_free(p); }
    } pin_sum(3 * px * sizeof(float)), pin_sum2(3 * px * sizeof(float)), pin_count(px * sizeof(int32_t));
    if (!pin_sum.p || !pin_sum2.p || !pin_count.p) return die("pt_render");
    float *const sum = static_cast<float *>(pin_sum.p), *const sum2 = static_cast<float *>(pin_sum2.p);
    int32_t *const count = static_cast<int32_t *>(pin_count.p);
    std::vector<uint8_t> bgr(3 * px);
    float disp[3] = {0, INFINITY, 0};
    double read_s = 0, preview_s = 0;

    pt_session *session = nullptr;
    if (pt_session_create(scene, o.width, o.height, 0, o.height, &session) != PT_OK) return die("pt_render");
    pt_render_params rp;
    std::memset(&rp, 0, sizeof rp);
    rp.width = o.width; rp.height = o.height; rp.row_begin = 0; rp.row_end = o.height;
    rp.max_ray_reflections = o.max_ray_reflections;
    rp.eps = o.eps; rp.error = o.error; rp.seed = seed;
    auto read_back = [&]() {
        const clk::time_point a = clk::now();
        const int rc = pt_session_read(session, sum, sum2, count);
        read_s += secs(a, clk::now());
        return rc;
    };

    // Pass slices end exactly where the reference writes a preview (after every pass p with p % update == 0,
    // main.cpp:144-158) so that previews happen between GPU calls; with a time limit they are also kept short.
    int rays_count = 0;
    double ms_per_pass = 0;   // measured on the previous slice (0 = not yet known)
    while (rays_count < o.rays_per_pixel) {
        const long long elapsed_ms = now_ms() - start_time;
        if (o.time_limit != 0 && elapsed_ms >= 1000LL * o.time_limit) break;   // main.cpp:111-114
        int slice_end = o.rays_per_pixel;
        if (o.update != 0) {
            const int next_preview = (rays_count % o.update == 0) ? rays_count : (rays_count / o.update + 1) * o.update;
            slice_end = std::min(o.rays_per_pixel, next_preview + 1);
        }
        if (o.time_limit != 0) {
            // the reference would start every pass that begins before the deadline: run as many as are expected to,
            // at most ~75 ms worth, at least one
            int n = 1;
            if (ms_per_pass > 0) {
                const double left_ms = 1000.0 * o.time_limit - static_cast<double>(elapsed_ms);
                n = static_cast<int>(std::min(75.0, left_ms) / ms_per_pass);
                n = std::max(1, n);
            }
            slice_end = std::min(slice_end, rays_count + n);
        }
        rp.pass_begin = rays_count;
        rp.pass_count = slice_end - rays_count;
        pt_render_stats st;
        const clk::time_point a = clk::now();
        if (pt_session_render(session, &rp, o.time_limit != 0 ? &st : nullptr) != PT_OK) return die("pt_render");
        if (o.time_limit != 0) ms_per_pass = 1e3 * secs(a, clk::now()) / rp.pass_count;   // the call waited for the kernel
        for (int p = rays_count; p < slice_end; ++p) {
            if (o.update != 0 && p % o.update == 0) {
                const clk::time_point b = clk::now();
                if (read_back() != PT_OK) return die("pt_render");
                pt_resolve(o.width, o.height, sum, sum2, count, o.gamma_correction, bgr.data(), nullptr);
                if (o.out.empty() && pt_write_bmp("../result.bmp", o.width, o.height, bgr.data()) != PT_OK)
                    std::cerr << pt_last_error() << std::endl;   // the reference's save_image only prints, too
                std::cerr << "Image update" << std::endl;
                preview_s += secs(b, clk::now());
            }
            if (!o.quiet) std::cerr << p + 1 << " rays per pixel were sent" << std::endl;
        }
        rays_count = slice_end;
    }
    if (pt_session_wait(session) != PT_OK) return die("pt_render");   // the last slice (and, in a fresh process, the
    const clk::time_point t_kernels = clk::now();                      // one-time load of the kernels' code object)
    if (read_back() != PT_OK) return die("pt_render");
    const clk::time_point t_render = clk::now();

    if (o.gauss || o.median) {   // main.cpp:187-201: filters act on the tonemapped float image, then set_pixel
        std::vector<float> rgb(3 * px);
        pt_resolve_float(o.width, o.height, sum, sum2, count, o.gamma_correction, rgb.data(), disp);
        if (pt_post_filter_host(o.device, o.width, o.height, rgb.data(), o.gauss, o.median) != PT_OK) return die("pt_render");
        pt_quantize(o.width, o.height, rgb.data(), count, bgr.data());
    } else {
        pt_resolve(o.width, o.height, sum, sum2, count, o.gamma_correction, bgr.data(), disp);
    }
    const clk::time_point t_resolve = clk::now();
    const long long end_time = now_ms();
    const std::time_t t = std::time(nullptr);
    const std::tm *now = std::localtime(&t);
    const std::string name =   // main.cpp:206-213
        std::to_string(now->tm_year + 1900) + '-' + std::to_string(now->tm_mon + 1) + '-' + std::to_string(now->tm_mday) + '-' +
        std::to_string(now->tm_hour) + '-' + std::to_string(now->tm_min) + '-' + std::to_string(now->tm_sec) + "  " +
        std::to_string(end_time - start_time) + "   " + std::to_string(rays_count) + " of " + std::to_string(o.rays_per_pixel) +
        "  max_disp " + std::to_string(disp[0]) + "  min_disp " + std::to_string(disp[1]) + "  aver_disp " + std::to_string(disp[2]);
    int rc = 0;
    if (!o.out.empty()) {
        if (pt_write_bmp(o.out.c_str(), o.width, o.height, bgr.data()) != PT_OK) rc = die("pt_render");
    } else {
        if (pt_write_bmp((name + ".bmp").c_str(), o.width, o.height, bgr.data()) != PT_OK) rc = die("pt_render");
        if (pt_write_bmp("../result.bmp", o.width, o.height, bgr.data()) != PT_OK) rc = die("pt_render");
    }
    std::cout << name << std::endl;
    if (o.timing) {
        const clk::time_point t_end = clk::now();
        std::fprintf(stderr, "{\"hip_startup_s\": %.4f, \"load_s\": %.4f, \"render_s\": %.4f, \"read_back_s\": %.4f, \"previews_s\": %.4f, "
                             "\"resolve_s\": %.4f, \"bmp_write_s\": %.4f, \"total_s\": %.4f}\n",
                     secs(t_begin, t_hip), secs(t_hip, t_load), secs(t_load, t_kernels) - preview_s, secs(t_kernels, t_render), preview_s,
                     secs(t_render, t_resolve), secs(t_resolve, t_end), secs(t_begin, t_end));
    }
    pt_session_destroy(session);
    pt_scene_destroy(scene);
    return rc;
}
