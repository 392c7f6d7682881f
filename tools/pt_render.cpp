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
// Several GPUs: the reference splits the image's rows over its OpenMP threads inside the pass loop (main.cpp:115,132,141);
// `-GPUS N` splits them over the first N devices of the node (or `-DEVICES 0,2,5`) through pt_frame_*: N row bands, every
// pass slice enqueued on all devices before anything waits, the bands' accumulators gathered to the first device by one RCCL
// group of sends / receives, then the unmodified resolve.  The image is bit-identical for any N.  On a box with fewer
// devices `-REHEARSE 1` runs the N-band code path anyway, several bands per device, with device-to-device copies in place of
// the collective -- and says so on stderr; without it such a request is refused.
//
// Extra flags (not in the reference): -OUT <file> writes only that file instead of the two reference outputs,
// -DEVICE <n> selects the HIP device (one GPU), -GPUS / -DEVICES / -REHEARSE as above, -QUIET 1 drops the per-pass lines,
// -TIMING 1 prints one JSON line with the seconds spent in each phase (HIP start-up, load, render, read-back, resolve, BMP
// write) on stderr, -BENCH_STEPS k [-BENCH_WARMUP w] times k whole frames (clear, all passes on all devices, gather, wait)
// after w untimed ones and prints one JSON line on stdout instead of writing an image, -SELFCOLL 1 (test aid, one GPU)
// routes the band through an RCCL send / receive to self, -FASTEXIT 1 leaves with _Exit once the files are written (skips the
// runtime's teardown).
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
    int gpus = 0, rehearse = 0, selfcoll = 0, bench_steps = 0, bench_warmup = 1, fast_exit = 0;
    std::string devices;
    long long t0_ns = 0;   // -T0_NS: CLOCK_REALTIME of the parent just before it started this process (bench.py), for the start-up phase
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
        if (f == "-GPUS") o.gpus = std::atoi(v);
        if (f == "-DEVICES") o.devices = v;
        if (f == "-REHEARSE") o.rehearse = std::atoi(v);
        if (f == "-SELFCOLL") o.selfcoll = std::atoi(v);
        if (f == "-BENCH_STEPS") o.bench_steps = std::atoi(v);
        if (f == "-BENCH_WARMUP") o.bench_warmup = std::atoi(v);
        if (f == "-T0_NS") o.t0_ns = std::atoll(v);
        if (f == "-FASTEXIT") o.fast_exit = std::atoi(v);
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
    if (std::getenv("PT_RENDER_PRINT_CONFIG")) {   // tests/test_ref_parts.py: the parsed Config fields, as the reference's own parser is asked for them
        const unsigned sd = o.seed < 0 ? static_cast<unsigned>(std::time(nullptr)) : static_cast<unsigned>(o.seed);
        std::printf("height %d\nwidth %d\nrays_per_pixel %d\nmax_ray_reflections %d\nmedian %d\ngauss %d\neps %.9g\nerror %.9g\nupdate %d\n"
                    "gamma_correction %.9g\nmodel_path %s\nmodel_name %s\nskybox %s\ntime_limit %d\nseed %u\n",
                    o.height, o.width, o.rays_per_pixel, o.max_ray_reflections, o.median, o.gauss, static_cast<double>(o.eps),
                    static_cast<double>(o.error), o.update, static_cast<double>(o.gamma_correction), o.model_path.c_str(),
                    o.model_name.c_str(), o.skybox.c_str(), o.time_limit, sd);
        return 0;
    }
    if (o.width <= 0 || o.height <= 0) {
        std::cerr << "pt_render: --W and --H must be positive" << std::endl;
        return 2;
    }
    const unsigned seed = o.seed < 0 ? static_cast<unsigned>(std::time(nullptr)) : static_cast<unsigned>(o.seed);   // config.h:101-104

    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const clk::time_point t_begin = clk::now();
    double pre_main_s = 0;   // exec + dynamic linking, when the parent told us when it started us
    if (o.t0_ns > 0) {
        timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        pre_main_s = (static_cast<long long>(ts.tv_sec) * 1000000000LL + ts.tv_nsec - o.t0_ns) * 1e-9;
    }
    // The model is parsed before the first HIP call (host-only scene): nothing below depends on the device yet.
    pt_scene *scene = nullptr;
    if (pt_scene_load_obj(o.model_path.c_str(), o.model_name.c_str(), -1, &scene) != PT_OK) return die("pt_render");
    if (!o.skybox.empty() && pt_scene_set_skybox_bmp(scene, o.skybox.c_str()) != PT_OK) return die("pt_render");   // scene.cpp:20-22
    const clk::time_point t_parse = clk::now();
    const int n_dev = pt_device_count();   // first HIP call: runtime start-up
    if (n_dev < 1) {
        std::cerr << "pt_render: no HIP device (the integrator has no CPU fallback)" << std::endl;
        return 1;
    }
    const clk::time_point t_hip = clk::now();

    // Row bands -> devices (main.cpp:115,132,141: the reference's split of the rows over its threads).
    std::vector<int32_t> devices;
    if (!o.devices.empty()) {
        for (size_t at = 0; at <= o.devices.size();) {
            const size_t comma = std::min(o.devices.find(',', at), o.devices.size());
            devices.push_back(std::atoi(o.devices.substr(at, comma - at).c_str()));
            at = comma + 1;
        }
    } else if (o.gpus > 0) {
        for (int b = 0; b < o.gpus; ++b) devices.push_back(o.rehearse ? b % n_dev : b);
    } else {
        devices.push_back(o.device);
    }
    uint32_t flags = 0;
    if (o.rehearse) flags |= PT_FRAME_REHEARSE;
    if (o.selfcoll) flags |= PT_FRAME_SELF_COLLECTIVE;
    pt_frame *frame = nullptr;
    if (pt_frame_create(scene, devices.data(), static_cast<int32_t>(devices.size()), o.width, o.height, flags, &frame) != PT_OK)
        return die("pt_render");
    int32_t transport = 0;
    pt_frame_info(frame, nullptr, nullptr, nullptr, &transport);
    const char *transport_name = transport == PT_FRAME_TRANSPORT_RCCL ? "rccl" : transport == PT_FRAME_TRANSPORT_DEVICE_COPIES ? "device_copies" : "none";
    if (transport == PT_FRAME_TRANSPORT_DEVICE_COPIES)
        std::cerr << "pt_render: REHEARSAL -- " << devices.size() << " row bands on " << n_dev << " device(s); the gather is device-to-device "
                     "copies, not the RCCL collective" << std::endl;
    const clk::time_point t_load = clk::now();

    pt_render_params rp;
    std::memset(&rp, 0, sizeof rp);
    rp.width = o.width; rp.height = o.height; rp.row_begin = 0; rp.row_end = o.height;
    rp.max_ray_reflections = o.max_ray_reflections;
    rp.eps = o.eps; rp.error = o.error; rp.seed = seed;

    if (o.bench_steps > 0) {
        // k whole frames: zero the accumulators, every pass on every device, the gather, wait for all of it
        auto one_frame = [&]() {
            rp.pass_begin = 0;
            rp.pass_count = o.rays_per_pixel;
            return pt_frame_clear(frame) == PT_OK && pt_frame_render(frame, &rp, nullptr) == PT_OK && pt_frame_gather(frame) == PT_OK;
        };
        for (int i = 0; i < std::max(o.bench_warmup, 0); ++i)
            if (!one_frame()) return die("pt_render");
        if (pt_frame_wait(frame) != PT_OK) return die("pt_render");
        const clk::time_point a = clk::now();
        for (int i = 0; i < o.bench_steps; ++i)
            if (!one_frame()) return die("pt_render");
        if (pt_frame_wait(frame) != PT_OK) return die("pt_render");
        const double dt = secs(a, clk::now());
        const double samples = static_cast<double>(o.width) * o.height * o.rays_per_pixel * o.bench_steps;
        // With more than one band: one more frame, untimed, taken apart -- every band's own kernel time (HIP events on its
        // stream), then, with all kernels done, the gather alone on the host's clock -- so that the first run on several devices
        // says where the time went and not only how long it took.
        std::string diagnosis;
        if (devices.size() > 1) {
            pt_render_stats st;
            rp.pass_begin = 0;
            rp.pass_count = o.rays_per_pixel;
            std::vector<float> band_ms(devices.size(), -1.0f);
            if (pt_frame_clear(frame) != PT_OK || pt_frame_render(frame, &rp, &st) != PT_OK || pt_frame_wait(frame) != PT_OK ||
                pt_frame_band_kernel_ms(frame, band_ms.data()) != PT_OK)
                return die("pt_render");
            const clk::time_point g0 = clk::now();
            if (pt_frame_gather(frame) != PT_OK || pt_frame_wait(frame) != PT_OK) return die("pt_render");
            const double gather_ms = secs(g0, clk::now()) * 1e3;
            char buf[64];
            diagnosis = ", \"band_kernel_ms\": [";
            for (size_t b = 0; b < band_ms.size(); ++b) {
                std::snprintf(buf, sizeof buf, "%s%.3f", b ? ", " : "", static_cast<double>(band_ms[b]));
                diagnosis += buf;
            }
            std::snprintf(buf, sizeof buf, "], \"gather_alone_ms\": %.3f", gather_ms);
            diagnosis += buf;
        }
        std::printf("{\"cxx_frame\": true, \"value\": %.3f, \"unit\": \"Msamples/s\", \"ms_per_step\": %.4f, \"steps\": %d, \"warmup\": %d, "
                    "\"bands\": %zu, \"devices_visible\": %d, \"transport\": \"%s\", \"width\": %d, \"height\": %d, \"spp\": %d, \"mrr\": %d, \"error\": %g%s}\n",
                    samples / dt / 1e6, dt / o.bench_steps * 1e3, o.bench_steps, o.bench_warmup, devices.size(), n_dev, transport_name, o.width, o.height,
                    o.rays_per_pixel, o.max_ray_reflections, static_cast<double>(o.error), diagnosis.c_str());
        pt_frame_destroy(frame);
        pt_scene_destroy(scene);
        return 0;
    }

    const size_t px = static_cast<size_t>(o.width) * o.height;
    // Page-locked accumulators: the read-back then runs at PCIe speed without staging copies.  They are allocated on first
    // use -- normally while the GPUs are busy with the frame (pinning 116 MB takes 20 ms, which the host has nothing else to
    // do with between enqueueing the passes and waiting for them).
    struct Pinned {
        void *p = nullptr;
        ~Pinned() { pt_host_free(p); }
    } pin_sum, pin_sum2, pin_count;
    float *sum = nullptr, *sum2 = nullptr;
    int32_t *count = nullptr;
    std::vector<uint8_t> bgr;
    double alloc_s = 0;
    auto ensure_buffers = [&]() {
        if (sum) return true;
        const clk::time_point a = clk::now();
        pin_sum.p = pt_host_alloc(3 * px * sizeof(float));
        pin_sum2.p = pt_host_alloc(3 * px * sizeof(float));
        pin_count.p = pt_host_alloc(px * sizeof(int32_t));
        if (!pin_sum.p || !pin_sum2.p || !pin_count.p) return false;
        sum = static_cast<float *>(pin_sum.p);
        sum2 = static_cast<float *>(pin_sum2.p);
        count = static_cast<int32_t *>(pin_count.p);
        bgr.resize(3 * px);
        alloc_s += secs(a, clk::now());
        return true;
    };
    float disp[3] = {0, INFINITY, 0};
    double read_s = 0, preview_s = 0;

    auto read_back = [&]() {   // gathers the bands (one collective) if any changed, waits, copies the frame out
        if (!ensure_buffers()) return static_cast<int>(PT_ERR_OUT_OF_MEMORY);
        const clk::time_point a = clk::now();
        const int rc = pt_frame_read(frame, sum, sum2, count);
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
        const clk::time_point a = clk::now();
        if (pt_frame_render(frame, &rp, nullptr) != PT_OK) return die("pt_render");
        if (o.time_limit != 0) {   // wait for the slice (without asking for statistics: the statistics-free kernels are the fast ones)
            if (pt_frame_wait(frame) != PT_OK) return die("pt_render");
            ms_per_pass = 1e3 * secs(a, clk::now()) / rp.pass_count;
        }
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
    const clk::time_point t_enqueued = clk::now();
    if (pt_frame_gather(frame) != PT_OK) return die("pt_render");   // the frame's one collective (nothing to do for one band)
    const double alloc_before = alloc_s;
    if (!ensure_buffers()) return die("pt_render");                 // (while the devices work)
    const double alloc_in_wait = alloc_s - alloc_before;
    if (pt_frame_wait(frame) != PT_OK) return die("pt_render");     // the last slice (and, in a fresh process, the
    const clk::time_point t_kernels = clk::now();                    // one-time load of the kernels' code object)
    if (read_back() != PT_OK) return die("pt_render");
    const clk::time_point t_render = clk::now();

    if (o.gauss || o.median) {   // main.cpp:187-201: filters act on the tonemapped float image, then set_pixel
        std::vector<float> rgb(3 * px);
        pt_resolve_float(o.width, o.height, sum, sum2, count, o.gamma_correction, rgb.data(), disp);
        if (pt_post_filter_host(devices[0], o.width, o.height, rgb.data(), o.gauss, o.median) != PT_OK) return die("pt_render");
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
        double host_s[2] = {0, 0};
        pt_scene_timings(scene, host_s);
        std::fprintf(stderr, "{\"pre_main_s\": %.4f, \"parse_s\": %.4f, \"hip_startup_s\": %.4f, \"frame_setup_s\": %.4f, \"host_alloc_s\": %.4f, "
                             "\"enqueue_s\": %.4f, \"hierarchy_build_s\": %.4f, \"kernels_wait_s\": %.4f, \"read_back_s\": %.4f, \"previews_s\": %.4f, "
                             "\"resolve_s\": %.4f, \"bmp_write_s\": %.4f, \"main_s\": %.4f, \"bands\": %zu, \"transport\": \"%s\"}\n",
                     pre_main_s, secs(t_begin, t_parse), secs(t_parse, t_hip), secs(t_hip, t_load), alloc_in_wait,
                     secs(t_load, t_enqueued) - preview_s, host_s[1], secs(t_enqueued, t_kernels) - alloc_in_wait, secs(t_kernels, t_render), preview_s,
                     secs(t_render, t_resolve), secs(t_resolve, t_end), secs(t_begin, t_end), devices.size(), transport_name);
    }
    if (o.fast_exit) {
        // Every file is written and closed; tearing the HIP runtime down (code objects, device heap, RCCL) is all that a normal
        // exit would add.
        std::cout.flush();
        std::cerr.flush();
        std::_Exit(rc);
    }
    pt_frame_destroy(frame);
    pt_scene_destroy(scene);
    return rc;
}
