// Host-only robustness driver for the scene loader (JSON reader, .obj reader, scene construction, tree build), built with
// -fsanitize=address,undefined by test_host_scene.py from the library's host sources.  Feeds rrt_host_scene_load a valid scene
// file cut short at every length, with seeded random byte flips, with numbers replaced by hostile ones (huge, negative, NaN
// spellings, long digit strings) and with its .obj file treated the same way; the loader may refuse a file with a message but
// must never read or write out of bounds, overflow an index or recurse without end.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "rrt.h"
#include "../rust_raytracer_amd/csrc/rrt_bvh_build.hpp"

// The device-side tree builder lives in the GPU library; files cannot ask for it, so the loader never gets here.
int rrt_bvh_build_gpu(const float*, uint32_t, bool, std::vector<rrt_node>*, std::vector<uint32_t>*, uint32_t*) { return RRT_ERR_DEVICE; }

static std::string slurp(const std::string& path) {
    std::ifstream in(path, std::ios::binary);
    return std::string((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}
static void spit(const std::string& path, const std::string& text) {
    std::ofstream out(path, std::ios::binary | std::ios::trunc);
    out.write(text.data(), (std::streamsize)text.size());
}

static uint64_t g_state = 0x2545f4914f6cdd1dull;
static uint32_t rnd() {
    g_state ^= g_state << 13;
    g_state ^= g_state >> 7;
    g_state ^= g_state << 17;
    return (uint32_t)(g_state >> 16);
}

struct Tally {
    long ok = 0, refused = 0;
};

static void load(const std::string& scene_path, Tally* t) {
    rrt_host_scene* hs = nullptr;
    const int rc = rrt_host_scene_load(scene_path.c_str(), 0, 0, 0, 0.0f / 0.0f, &hs);
    if (rc == RRT_OK) {
        ++t->ok;
        const rrt_scene_desc* d = rrt_host_desc(hs);
        if (!d || (d->n_slots && !d->slots) || (d->n_nodes && !d->nodes)) {
            std::fprintf(stderr, "loader accepted a file but returned no description\n");
            std::exit(2);
        }
        // every index the description holds must be in range: walk what a consumer would walk
        uint64_t sum = 0;
        for (uint32_t i = 0; i < d->n_slots; ++i) {
            const rrt_slot& s = d->slots[i];
            if (RRT_INFO_MATERIAL(s.info) >= d->n_materials) std::exit(3);
            const uint32_t kind = RRT_INFO_KIND(s.info);
            if (kind == RRT_KIND_SPHERE && s.aux >= d->n_spheres) std::exit(3);
            if (kind == RRT_KIND_MEDIUM && s.aux >= d->n_media) std::exit(3);
            sum += s.prim;
        }
        for (uint32_t i = 0; i < d->n_media; ++i) {
            const rrt_medium& m = d->media[i];
            if (m.boundary_kind == RRT_KIND_MEDIUM && m.boundary_index >= i) std::exit(3);
            if (m.boundary_kind == RRT_KIND_SPHERE && m.boundary_index >= d->n_spheres) std::exit(3);
            if (m.boundary_kind == RRT_KIND_TRIANGLE && m.boundary_index >= d->n_boundary_tris) std::exit(3);
        }
        for (uint32_t i = 0; i < d->n_textures; ++i)
            if (d->textures[i].type == RRT_TEX_IMAGE) sum += d->texels[d->textures[i].u3];
        if (sum == 0x7fffffffffffffffull) std::puts("");
        rrt_host_scene_destroy(hs);
    } else {
        ++t->refused;
        if (hs) {
            std::fprintf(stderr, "a refused load left a handle behind\n");
            std::exit(2);
        }
        if (!rrt_last_error() || !*rrt_last_error()) {
            std::fprintf(stderr, "a refused load left no message\n");
            std::exit(2);
        }
    }
}

// Positions where a JSON or .obj number starts.
static std::vector<size_t> number_starts(const std::string& s) {
    std::vector<size_t> out;
    for (size_t i = 0; i < s.size(); ++i) {
        const bool digit = (s[i] >= '0' && s[i] <= '9') || s[i] == '-';
        const bool fresh = i == 0 || !((s[i - 1] >= '0' && s[i - 1] <= '9') || s[i - 1] == '.' || s[i - 1] == '-' || s[i - 1] == 'e' || s[i - 1] == 'E' ||
                                       (s[i - 1] >= 'a' && s[i - 1] <= 'z') || (s[i - 1] >= 'A' && s[i - 1] <= 'Z') || s[i - 1] == '_');
        if (digit && fresh) out.push_back(i);
    }
    return out;
}
static size_t number_end(const std::string& s, size_t i) {
    while (i < s.size() && ((s[i] >= '0' && s[i] <= '9') || s[i] == '.' || s[i] == '-' || s[i] == '+' || s[i] == 'e' || s[i] == 'E')) ++i;
    return i;
}

static void torture(const std::string& good, const std::string& write_to, const std::string& scene_path, Tally* t) {
    static const char* kHostile[] = {"0", "-1", "1e39", "-1e39", "4294967296", "18446744073709551616", "1e-50", "NaN", "-0", "1e", "99999999999999999999999999",
                                     "2147483648", "0.0000000000000000000000000000000000000000001", "[]", "{}", "\"x\"", "null", "true", ""};
    // 1. cut short: every length for small files, a sample of lengths for large ones
    if (good.size() <= 4000) {
        for (size_t n = 0; n < good.size(); ++n) {
            spit(write_to, good.substr(0, n));
            load(scene_path, t);
        }
    } else {
        for (int k = 0; k < 300; ++k) {
            spit(write_to, good.substr(0, rnd() % good.size()));
            load(scene_path, t);
        }
    }
    // 2. every number replaced by hostile values (a sample of the numbers when there are many)
    const std::vector<size_t> nums = number_starts(good);
    const size_t stride = nums.size() > 120 ? nums.size() / 120 : 1;
    for (size_t k = 0; k < nums.size(); k += stride)
        for (const char* h : kHostile) {
            std::string f = good.substr(0, nums[k]) + h + good.substr(number_end(good, nums[k]));
            spit(write_to, f);
            load(scene_path, t);
        }
    // 3. seeded random damage: bytes overwritten, ranges deleted, ranges duplicated, brackets nested deeply
    for (int k = 0; k < 1500; ++k) {
        std::string f = good;
        const int what = (int)(rnd() % 4);
        if (what == 0) {
            const int flips = 1 + (int)(rnd() % 6);
            for (int j = 0; j < flips; ++j) f[rnd() % f.size()] = (char)rnd();
        } else if (what == 1) {
            const size_t a = rnd() % f.size(), n = rnd() % 40;
            f.erase(a, n);
        } else if (what == 2) {
            const size_t a = rnd() % f.size(), n = rnd() % 200;
            f.insert(a, f.substr(a, n));
        } else {
            const size_t a = rnd() % f.size();
            f.insert(a, std::string(1 + rnd() % 20000, (rnd() & 1) ? '[' : '{'));
        }
        if (f.empty()) f = " ";
        spit(write_to, f);
        load(scene_path, t);
    }
    // 4. nothing but opening brackets: a reader that recurses per level must stop at a depth of its own choosing
    for (const char* unit : {"[", "{\"a\":", "{\"Shapes\":[{\"boundary\":"}) {
        std::string f;
        for (int k = 0; k < 200000; ++k) f += unit;
        spit(write_to, f);
        load(scene_path, t);
    }
    spit(write_to, good);
}

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: fuzz_scene_loader scene.json [file the scene names, e.g. mesh.obj ...]\n");
        return 1;
    }
    const std::string scene_path = argv[1];
    Tally t;
    load(scene_path, &t);
    if (t.ok != 1) {
        std::fprintf(stderr, "the unmodified scene was refused: %s\n", rrt_last_error());
        return 1;
    }
    for (int a = 1; a < argc; ++a) {
        const std::string good = slurp(argv[a]);
        if (good.empty()) {
            std::fprintf(stderr, "cannot read %s\n", argv[a]);
            return 1;
        }
        torture(good, argv[a], scene_path, &t);
        Tally again;
        load(scene_path, &again);
        if (again.ok != 1) {
            std::fprintf(stderr, "the restored scene was refused: %s\n", rrt_last_error());
            return 1;
        }
    }
    std::printf("fuzz ok: %ld accepted, %ld refused\n", t.ok, t.refused);
    return 0;
}
