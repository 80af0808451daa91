// Image file decoding for Image textures (resources.rs:16-51 calls image::open).
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include "../rrt_error.hpp"
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

// PNG, 8-bit RGB, non-interlaced, filter 0 on every row, one IDAT chunk (≙ PNGEncoder::encode(.., ColorType::RGB(8)), main.rs:188).
std::vector<uint8_t> encode_png_rgb8(const uint8_t* rgb, uint32_t w, uint32_t h) {
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    auto be32 = [](std::vector<uint8_t>* v, uint32_t x) {
        v->push_back((uint8_t)(x >> 24));
        v->push_back((uint8_t)(x >> 16));
        v->push_back((uint8_t)(x >> 8));
        v->push_back((uint8_t)x);
    };
    auto chunk = [&](const char* type, const std::vector<uint8_t>& data) {
        be32(&out, (uint32_t)data.size());
        size_t start = out.size();
        out.insert(out.end(), type, type + 4);
        out.insert(out.end(), data.begin(), data.end());
        be32(&out, (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start)));
    };
    std::vector<uint8_t> ihdr;
    be32(&ihdr, w);
    be32(&ihdr, h);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});  // bit depth 8, colour type 2 (RGB), deflate, adaptive filtering, no interlace
    chunk("IHDR", ihdr);
    const size_t row = (size_t)w * 3;
    std::vector<uint8_t> raw((row + 1) * h);
    for (uint32_t y = 0; y < h; ++y) {
        raw[(row + 1) * y] = 0;
        std::memcpy(&raw[(row + 1) * y + 1], rgb + row * y, row);
    }
    uLongf bound = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(bound);
    if (compress2(z.data(), &bound, raw.data(), (uLong)raw.size(), 6) != Z_OK) return {};
    z.resize(bound);
    chunk("IDAT", z);
    chunk("IEND", {});
    return out;
}

}  // namespace rrt_host

extern "C" int rrt_write_png(const char* path, const uint8_t* rgb, uint32_t width, uint32_t height, int create_new) {
    if (!path || !rgb || width == 0 || height == 0) return rrt_set_error(RRT_ERR_INVALID, "rrt_write_png: invalid argument");
    // OpenOptions::new().write(true).create_new(true) (main.rs:98-106): never overwrite an existing file
    int fd = open(path, O_WRONLY | O_CREAT | (create_new ? O_EXCL : O_TRUNC), 0644);
    if (fd < 0) return rrt_set_error(RRT_ERR_IO, std::string("Failed to create new file: ") + path + ": " + std::strerror(errno));
    std::vector<uint8_t> png = rrt_host::encode_png_rgb8(rgb, width, height);
    if (png.empty()) {
        close(fd);
        return rrt_set_error(RRT_ERR_IO, "Failed to encode the png for output: deflate failed");
    }
    size_t off = 0;
    while (off < png.size()) {
        ssize_t n = write(fd, png.data() + off, png.size() - off);
        if (n <= 0) {
            close(fd);
            return rrt_set_error(RRT_ERR_IO, std::string("Failed to encode the png for output: ") + std::strerror(errno));
        }
        off += (size_t)n;
    }
    close(fd);
    return RRT_OK;
}
