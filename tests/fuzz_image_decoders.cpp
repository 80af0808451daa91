// Host-only robustness driver for the texture decoders (built with -fsanitize=address,undefined by test_host_scene.py).
// Feeds decode_jpeg / decode_png a valid file cut short at every length, with hostile table selectors and segment lengths,
// and with seeded random byte flips; a decoder may refuse a file with a message but must never read or write out of bounds.
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

namespace rrt_host {
std::string decode_jpeg(const std::vector<uint8_t>& file, uint32_t* w, uint32_t* h, uint32_t* channels, std::vector<uint8_t>* px);
std::string decode_png(const std::vector<uint8_t>& file, uint32_t* w, uint32_t* h, uint32_t* channels, std::vector<uint8_t>* px);
}  // namespace rrt_host

typedef std::string (*Decoder)(const std::vector<uint8_t>&, uint32_t*, uint32_t*, uint32_t*, std::vector<uint8_t>*);

static std::vector<uint8_t> slurp(const char* path) {
    std::ifstream in(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}

static uint64_t g_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() {
    g_state ^= g_state << 13;
    g_state ^= g_state >> 7;
    g_state ^= g_state << 17;
    return (uint32_t)(g_state >> 16);
}

struct Tally {
    long ok = 0, refused = 0;
};

static void feed(Decoder dec, const std::vector<uint8_t>& f, Tally* t) {
    uint32_t w = 0, h = 0, c = 0;
    std::vector<uint8_t> px;
    const std::string e = dec(f, &w, &h, &c, &px);
    if (e.empty()) {
        ++t->ok;
        if (px.size() != (size_t)w * h * c) {
            std::fprintf(stderr, "decoder accepted a file but returned %zu texels for %ux%ux%u\n", px.size(), w, h, c);
            std::exit(2);
        }
    } else {
        ++t->refused;
    }
}

// Offsets of the JPEG marker segments (marker code, offset of the length field) up to the start of scan.
static std::vector<std::pair<int, size_t>> jpeg_segments(const std::vector<uint8_t>& f) {
    std::vector<std::pair<int, size_t>> out;
    size_t i = 2;
    while (i + 4 <= f.size() && f[i] == 0xFF) {
        const int m = f[i + 1];
        const size_t len = ((size_t)f[i + 2] << 8) | f[i + 3];
        out.push_back({m, i + 2});
        if (m == 0xDA) break;
        i += 2 + len;
    }
    return out;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: fuzz_image_decoders file.jpg file.png [file.jpg ...]\n");
        return 1;
    }
    Tally t;
    for (int a = 1; a < argc; ++a) {
        const std::vector<uint8_t> good = slurp(argv[a]);
        if (good.size() < 16) {
            std::fprintf(stderr, "cannot read %s\n", argv[a]);
            return 1;
        }
        const bool jpeg = good[0] == 0xFF;
        const Decoder dec = jpeg ? rrt_host::decode_jpeg : rrt_host::decode_png;
        Tally before = t;
        feed(dec, good, &t);
        if (t.ok != before.ok + 1) {
            std::fprintf(stderr, "%s: the unmodified file was refused\n", argv[a]);
            return 1;
        }
        // 1. cut short: every length through the headers, then a sample of lengths through the entropy-coded data
        const size_t header_end = good.size() < 700 ? good.size() : 700;
        for (size_t n = 0; n < header_end; ++n) feed(dec, std::vector<uint8_t>(good.begin(), good.begin() + n), &t);
        for (int k = 0; k < 60; ++k) feed(dec, std::vector<uint8_t>(good.begin(), good.begin() + header_end + rnd() % (good.size() - header_end + 1)), &t);
        if (jpeg) {
            const auto segs = jpeg_segments(good);
            for (const auto& sg : segs) {
                const size_t at = sg.second;
                // 2. segment lengths: too short for the fields the segment must hold, and running past the end of the file
                for (int len : {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 0xFFFF}) {
                    std::vector<uint8_t> f = good;
                    f[at] = (uint8_t)(len >> 8);
                    f[at + 1] = (uint8_t)len;
                    feed(dec, f, &t);
                    f.resize(at + 2 + (size_t)(len > 2 && len < 64 ? len - 2 : 0));  // the file ends inside the segment
                    feed(dec, f, &t);
                }
                // 3. every byte of the segment's first 24 bytes set to small values, to every value with one nibble zero and to
                //    every value with equal nibbles: table ids, component counts, sampling factors, table selectors
                for (size_t o = 2; o < 26 && at + o < good.size(); ++o)
                    for (int v = 0; v < 256; ++v) {
                        if (!(v < 20 || (v & 15) == 0 || (v >> 4) == (v & 15))) continue;
                        std::vector<uint8_t> f = good;
                        f[at + o] = (uint8_t)v;
                        feed(dec, f, &t);
                    }
            }
        }
        // 4. seeded random flips, 1..8 bytes per file, weighted towards the headers
        for (int k = 0; k < 400; ++k) {
            std::vector<uint8_t> f = good;
            const int flips = 1 + (int)(rnd() % 8);
            for (int j = 0; j < flips; ++j) {
                const size_t span = (rnd() & 1) ? header_end : f.size();
                f[rnd() % span] = (uint8_t)rnd();
            }
            feed(dec, f, &t);
        }
    }
    std::printf("fuzz ok: %ld accepted, %ld refused\n", t.ok, t.refused);
    return 0;
}
