// Command line front end with the reference's interface (main.rs:38-198):
//
//     rust-raytracer [-t N | --thread-count N] IN_SCENE_FILE OUT_FILEPATH
//
// reads a scene specification, traces it on the current GPU through the C ABI of librrt_hip.so and writes an 8-bit RGB
// PNG.  As in the reference the output file is created before tracing starts and an existing file is never overwritten;
// a progress bar follows the sample count and the run ends with "Success! Took N seconds".
// -t is accepted and validated like the reference's (a positive integer, default 2) but selects nothing here: the
// samples are traced by the GPU, not by host threads.
// Extensions (not in the reference): --background black|sky (main.rs:259-270 keeps the sky gradient commented out),
// --seed N, --aabb tight|reference|exhaustive, --samples N / --width N / --height N overriding the scene's Logistics,
// --gpus N (the reference's N host threads draining one sample queue, main.rs:136-165, become N GPUs each tracing a share of the
// frame: rrt_render_multi) with --shard samples|tiles, and --stats (the reference's single "Took N seconds" split by phase).
#include <fcntl.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rrt.h"

namespace {

// progress.rs:4-73: "\rProgress: <####      > (NN%)", redrawn when the fraction moved by more than 0.009.
struct Progress {
    uint64_t total, current = 0;
    uint32_t width;
    float last = 0.0f;
    Progress(uint64_t total_work, uint32_t bar) : total(total_work), width(bar) {}
    void update(uint64_t made) {
        current = current + made < total ? current + made : total;
        float pct = (float)((double)current / (double)total);
        if (std::fabs(pct - last) > 0.009f) print(pct);
    }
    void done() {
        current = 0;
        last = 0.0f;
        std::fputc('\n', stdout);
        std::fflush(stdout);
    }
    void print(float pct) {
        last = pct;
        uint32_t filled = (uint32_t)std::lround(pct * (float)width);
        std::string p = "\rProgress: <";
        p.append(filled, '#');
        p.append(width - filled, ' ');
        p += "> (" + std::to_string((uint32_t)std::lround(pct * 100.0f)) + "%)";
        std::fputs(p.c_str(), stdout);
        std::fflush(stdout);
    }
};

[[noreturn]] void die(const std::string& msg) {
    std::fprintf(stderr, "%s\n", msg.c_str());
    std::exit(1);
}

[[noreturn]] void usage(const char* why) {
    std::fprintf(stderr,
                 "error: %s\n\nUSAGE:\n    rust-raytracer [OPTIONS] <IN_SCENE_FILE> <OUT_FILEPATH>\n\nOPTIONS:\n"
                 "    -t, --thread-count <THREAD_COUNT>    Number of threads to use while tracing\n"
                 "        --background <black|sky>         Colour of rays that leave the scene (default black)\n"
                 "        --seed <N>                       Seed of the per-sample random streams (default 0)\n"
                 "        --aabb <tight|reference|exhaustive>  Bounding-box test of the BVH walk (default tight)\n"
                 "        --tree <host|gpu|clean>          Where and how a \"BVH\" aggregate is built: the reference's tree on the host\n"
                 "                                         (default) or on the GPU, or the clean tree with the List aggregate's results\n"
                 "        --samples <N> --width <N> --height <N>   Override the scene's Logistics\n"
                 "        --gpus <N>                       GPUs of this node that share the frame (default 1)\n"
                 "        --shard <samples|tiles>          How they share it: sample ranges (default) or interleaved 32x32 tiles\n"
                 "        --stats                          Print the seconds spent per phase\n\nARGS:\n"
                 "    <IN_SCENE_FILE>    The scene specification to render\n"
                 "    <OUT_FILEPATH>     The relative filepath to write the output image to\n",
                 why);
    std::exit(2);
}

bool parse_u32(const char* s, uint32_t* out) {
    if (!s || !*s) return false;
    char* end = nullptr;
    unsigned long long v = std::strtoull(s, &end, 10);
    if (*end || s[0] == '-' || v > 0xffffffffull) return false;
    *out = (uint32_t)v;
    return true;
}

}  // namespace

int main(int argc, char** argv) {
    const char* thread_count = "2";
    uint32_t seed = 0, bg = RRT_BG_BLACK, aabb = RRT_AABB_TIGHT, o_samples = 0, o_width = 0, o_height = 0;
    int tree = RRT_AGGREGATE_BVH;
    uint32_t gpus = 1, shard = RRT_SHARD_SAMPLES;
    bool print_stats = false;
    std::vector<const char*> positional;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto value = [&](const char* name) -> const char* {
            if (i + 1 >= argc) usage((std::string("The argument '") + name + " <value>' requires a value but none was supplied").c_str());
            return argv[++i];
        };
        if (a == "-t" || a == "--thread-count") thread_count = value("--thread-count");
        else if (a.rfind("--thread-count=", 0) == 0) thread_count = argv[i] + 15;
        else if (a == "--seed") { if (!parse_u32(value("--seed"), &seed)) usage("--seed requires a non-negative integer"); }
        else if (a == "--samples") { if (!parse_u32(value("--samples"), &o_samples)) usage("--samples requires a positive integer"); }
        else if (a == "--width") { if (!parse_u32(value("--width"), &o_width)) usage("--width requires a positive integer"); }
        else if (a == "--height") { if (!parse_u32(value("--height"), &o_height)) usage("--height requires a positive integer"); }
        else if (a == "--background") {
            std::string v = value("--background");
            if (v == "black") bg = RRT_BG_BLACK;
            else if (v == "sky") bg = RRT_BG_SKY;
            else usage("--background takes black or sky");
        } else if (a == "--aabb") {
            std::string v = value("--aabb");
            if (v == "tight") aabb = RRT_AABB_TIGHT;
            else if (v == "reference") aabb = RRT_AABB_REFERENCE;
            else if (v == "exhaustive") aabb = RRT_AABB_EXHAUSTIVE;
            else usage("--aabb takes tight, reference or exhaustive");
        } else if (a == "--tree") {
            std::string v = value("--tree");
            if (v == "host") tree = RRT_AGGREGATE_BVH;
            else if (v == "gpu") tree = RRT_AGGREGATE_BVH_GPU;
            else if (v == "clean") tree = RRT_AGGREGATE_BVH_CLEAN;
            else usage("--tree takes host, gpu or clean");
        } else if (a == "--gpus") {
            if (!parse_u32(value("--gpus"), &gpus) || gpus == 0) usage("--gpus requires a positive integer");
        } else if (a == "--shard") {
            std::string v = value("--shard");
            if (v == "samples") shard = RRT_SHARD_SAMPLES;
            else if (v == "tiles") shard = RRT_SHARD_TILES;
            else usage("--shard takes samples or tiles");
        } else if (a == "--stats") print_stats = true;
        else if (a == "-h" || a == "--help") usage("help requested");
        else if (a.size() > 1 && a[0] == '-') usage(("Found argument '" + a + "' which wasn't expected, or isn't valid in this context").c_str());
        else positional.push_back(argv[i]);
    }
    if (positional.size() < 2) usage("The following required arguments were not provided:\n    <IN_SCENE_FILE>\n    <OUT_FILEPATH>");
    if (positional.size() > 2) usage("Found an argument which wasn't expected");

    const auto program_start = std::chrono::steady_clock::now();  // main.rs:64

    uint32_t num_threads = 0;  // main.rs:68-75
    if (!parse_u32(thread_count, &num_threads)) die("thread-count requires a valid positive integer");
    if (num_threads == 0) die("Need a thread count greater than zero");

    rrt_host_scene* host = nullptr;  // main.rs:77-95
    if (rrt_host_scene_load_as(positional[0], o_width, o_height, o_samples, NAN, tree, &host) != RRT_OK)
        die(std::string("Failed to parse scene spec JSON.: ") + rrt_last_error());
    const rrt_scene_desc* desc = rrt_host_desc(host);
    const uint32_t W = desc->resolution_x, H = desc->resolution_y, spp = desc->samples;

    // main.rs:97-107: the output file is created (and must not exist) before any tracing happens
    int fd = open(positional[1], O_WRONLY | O_CREAT | O_EXCL, 0644);
    if (fd < 0) die(std::string("Failed to create new file: ") + positional[1] + ": " + std::strerror(errno));
    close(fd);
    auto fail = [&](const std::string& msg) {
        unlink(positional[1]);  // leave nothing behind for a run that produced no image
        die(msg);
    };

    // the scene is replicated: one upload per GPU (it is small next to 288 GB of HBM)
    int visible = 0;
    if (rrt_device_count(&visible) != RRT_OK) fail(std::string("Failed to count the GPUs: ") + rrt_last_error());
    if ((int)gpus > visible) fail("--gpus " + std::to_string(gpus) + " but only " + std::to_string(visible) + " GPU(s) are visible");
    std::vector<rrt_scene*> scenes(gpus, nullptr);
    double upload_seconds = 0.0;
    for (uint32_t i = 0; i < gpus; ++i) {
        if (rrt_scene_create_on(desc, (int)i, &scenes[i]) != RRT_OK) fail(std::string("Failed to upload the scene: ") + rrt_last_error());
        double t = 0.0;
        rrt_scene_upload_seconds(scenes[i], &t);
        upload_seconds += t;
    }
    rrt_stats total;
    std::memset(&total, 0, sizeof total);

    std::vector<float> sums((size_t)W * H * 3, 0.0f);
    Progress progress((uint64_t)W * H * spp, 20);  // main.rs:130-134
    // The frame is traced in batches of samples so the progress bar can follow; ACCUMULATE keeps the per-pixel sums
    // bit-identical to a single call.
    const uint32_t batch = spp >= 64 * 16 ? ((spp / 16 + 63) / 64) * 64 : (spp >= 128 ? 64 : (spp ? spp : 1));
    for (uint32_t done = 0; done < spp; done += batch) {
        rrt_render_params p;
        std::memset(&p, 0, sizeof p);
        p.width = W;
        p.height = H;
        p.samples = spp - done < batch ? spp - done : batch;
        p.first_sample = done;
        p.seed = seed;
        p.background = bg;
        p.max_depth = -1;
        p.aabb_mode = aabb;
        p.flags = done ? RRT_RENDER_ACCUMULATE : 0;
        rrt_stats st;
        if (rrt_render_multi(scenes.data(), gpus, shard, &p, sums.data(), &st) != RRT_OK) fail(std::string("Tracing failed: ") + rrt_last_error());
        total.rays += st.rays;
        total.paths += st.paths;
        total.render_seconds += st.render_seconds;
        total.trace_seconds += st.trace_seconds;
        total.accumulate_seconds += st.accumulate_seconds;
        total.reduce_seconds += st.reduce_seconds;
        total.copy_seconds += st.copy_seconds;
        progress.update((uint64_t)W * H * p.samples);
    }
    progress.done();  // main.rs:166-169

    std::vector<uint8_t> rgb((size_t)W * H * 3);  // main.rs:171-186
    const auto resolve_start = std::chrono::steady_clock::now();
    if (rrt_resolve_rgb8(sums.data(), W, H, spp, rgb.data()) != RRT_OK) fail(std::string("Failed to finalize the image: ") + rrt_last_error());
    const auto png_start = std::chrono::steady_clock::now();
    if (rrt_write_png(positional[1], rgb.data(), W, H, 0) != RRT_OK) {  // main.rs:187-197
        std::fprintf(stderr, "Failed to encode the png for output: %s\n", rrt_last_error());
        unlink(positional[1]);
        return 1;
    }
    const auto finished = std::chrono::steady_clock::now();
    std::printf("Success! Took %g seconds\n", std::chrono::duration<double>(finished - program_start).count());
    if (print_stats) {
        double load_seconds = 0.0, build_seconds = 0.0;
        rrt_host_scene_times(host, &load_seconds, &build_seconds);
        std::printf("  load %.6f s | bvh + flatten %.6f s | upload %.6f s (%u GPU%s) | render %.6f s (trace %.6f, accumulate %.6f) | "
                    "reduce %.6f s | frame copies %.6f s | resolve %.6f s | png %.6f s\n",
                    load_seconds, build_seconds, upload_seconds, gpus, gpus == 1 ? "" : "s", total.render_seconds, total.trace_seconds,
                    total.accumulate_seconds, total.reduce_seconds, total.copy_seconds,
                    std::chrono::duration<double>(png_start - resolve_start).count(), std::chrono::duration<double>(finished - png_start).count());
        std::printf("  %llu rays, %llu paths, %.1f Mrays/s over the render phase\n", (unsigned long long)total.rays, (unsigned long long)total.paths,
                    total.render_seconds > 0.0 ? (double)total.rays / total.render_seconds * 1e-6 : 0.0);
    }
    for (rrt_scene* sc : scenes) rrt_scene_destroy(sc);
    rrt_host_scene_destroy(host);
    return 0;
}
