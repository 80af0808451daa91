// Image file decoding for Image textures (resources.rs:16-51 calls image::open).
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "scene_host.hpp"

namespace rrt_host {

std::string decode_jpeg(const std::vector<uint8_t>& file, uint32_t* w, uint32_t* h, uint32_t* channels, std::vector<uint8_t>* px);
std::string decode_png(const std::vector<uint8_t>& file, uint32_t* w, uint32_t* h, uint32_t* channels, std::vector<uint8_t>* px);

static std::string decode_pnm(const std::vector<uint8_t>& f, uint32_t* w, uint32_t* h, uint32_t* channels, std::vector<uint8_t>* px) {
    // binary PGM (P5) / PPM (P6), maxval 255
    size_t i = 2;
    auto next_int = [&](uint32_t* v) -> bool {
        for (;;) {
            while (i < f.size() && (f[i] == ' ' || f[i] == '\n' || f[i] == '\r' || f[i] == '\t')) ++i;
            if (i < f.size() && f[i] == '#') {
                while (i < f.size() && f[i] != '\n') ++i;
                continue;
            }
            break;
        }
        if (i >= f.size() || f[i] < '0' || f[i] > '9') return false;
        uint64_t x = 0;
        while (i < f.size() && f[i] >= '0' && f[i] <= '9') x = x * 10 + (f[i++] - '0');
        *v = (uint32_t)x;
        return true;
    };
    uint32_t maxv;
    if (!next_int(w) || !next_int(h) || !next_int(&maxv) || maxv != 255) return "unsupported PNM header";
    ++i;
    *channels = f[1] == '5' ? 1 : 3;
    size_t bytes = (size_t)*w * *h * *channels;
    if (i + bytes > f.size()) return "truncated PNM data";
    px->assign(f.begin() + i, f.begin() + i + bytes);
    return "";
}

std::string decode_image_file(const std::string& path, uint32_t* w, uint32_t* h, uint32_t* channels, std::vector<uint8_t>* px) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return "No such file or directory (os error 2)";
    std::vector<uint8_t> f((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    if (f.size() >= 3 && f[0] == 0xFF && f[1] == 0xD8) return decode_jpeg(f, w, h, channels, px);
    if (f.size() >= 8 && std::memcmp(f.data(), "\x89PNG\r\n\x1a\n", 8) == 0) return decode_png(f, w, h, channels, px);
    if (f.size() >= 3 && f[0] == 'P' && (f[1] == '5' || f[1] == '6')) return decode_pnm(f, w, h, channels, px);
    return "Unsupported image format";
}

}  // namespace rrt_host
