// JPEG (8-bit Huffman-coded: baseline and multi-scan sequential, progressive) and PNG (plain and Adam7-interlaced) decoders for
// Image textures (texture.rs:121-141 reads texels through image::open, resources.rs:41).
//
// The JPEG path follows the IJG integer pipeline — the "slow but accurate" 8x8 inverse DCT with 13-bit
// constants and the 16-bit fixed-point YCbCr->RGB tables — so that decoded texels agree with the widely
// deployed libjpeg decoders for non-subsampled and greyscale files.  Chroma-subsampled files are upsampled
// by replication.  The reference's own decoder (jpeg-decoder 0.1.18) is not part of the reference tree and
// may differ by one code value per texel.
#include <zlib.h>

#include <cstring>

#include "scene_host.hpp"

namespace rrt_host {

namespace {

// Largest image either decoder accepts (16384 x 16384): keeps every size computation far inside size_t and the
// allocations bounded for files whose headers lie.
constexpr uint64_t kMaxImagePixels = 1ull << 28;

struct Huff {
    // canonical code tables: for each code length, the smallest code and the index of its first symbol
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    bool present = false;
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int pred = 0;
    int bw = 0, bh = 0;  // blocks per row / column in the padded plane
    std::vector<uint8_t> plane;
};

struct JpegReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t bitbuf = 0;
    int bitcnt = 0;
    bool hit_marker = false;

    int next_byte() {
        if (p >= end) return -1;
        return *p++;
    }
    void fill() {
        while (bitcnt <= 24) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) {
                        p += 2;
                    } else {
                        hit_marker = true;  // leave the marker for the caller; feed zeros
                        b = 0;
                    }
                } else {
                    ++p;
                }
            } else {
                b = 0;
            }
            bitbuf |= (uint32_t)b << (24 - bitcnt);
            bitcnt += 8;
        }
    }
    int get_bits(int n) {
        if (n == 0) return 0;
        if (bitcnt < n) fill();
        int v = (int)(bitbuf >> (32 - n));
        bitbuf <<= n;
        bitcnt -= n;
        return v;
    }
    int decode(const Huff& h) {
        int code = 0;
        for (int len = 1; len <= 16; ++len) {
            code = (code << 1) | get_bits(1);
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.vals[h.valptr[len] + code - h.mincode[len]];
        }
        return -1;
    }
    void reset_bits() {
        bitbuf = 0;
        bitcnt = 0;
        hit_marker = false;
    }
};

int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

const int kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline int descale(long x, int n) { return (int)((x + (1L << (n - 1))) >> n); }
inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// 8x8 inverse DCT, 13-bit fixed point, two passes with 2 extra bits kept between them.
void idct_block(const int* coef, const uint16_t* quant, uint8_t* out, int stride) {
    const int CB = 13, P1 = 2;
    const long F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633, F_1_501 = 12299, F_1_847 = 15137,
               F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
    long ws[64];
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 8; ++i) {
            long in[8];
            for (int k = 0; k < 8; ++k) in[k] = pass == 0 ? (long)coef[k * 8 + i] * quant[k * 8 + i] : ws[i * 8 + k];
            long z2 = in[2], z3 = in[6];
            long z1 = (z2 + z3) * F_0_541;
            long t2 = z1 + z3 * (-F_1_847);
            long t3 = z1 + z2 * F_0_765;
            long t0 = (in[0] + in[4]) * (1L << CB);  // (a shift of a negative value is undefined before C++20)
            long t1 = (in[0] - in[4]) * (1L << CB);
            long t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
            t0 = in[7]; t1 = in[5]; t2 = in[3]; t3 = in[1];
            z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
            long z4 = t1 + t3;
            long z5 = (z3 + z4) * F_1_175;
            t0 *= F_0_298; t1 *= F_2_053; t2 *= F_3_072; t3 *= F_1_501;
            z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
            z3 += z5; z4 += z5;
            t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
            long o[8] = {t10 + t3, t11 + t2, t12 + t1, t13 + t0, t13 - t0, t12 - t1, t11 - t2, t10 - t3};
            if (pass == 0) {
                for (int k = 0; k < 8; ++k) ws[k * 8 + i] = descale(o[k], CB - P1);
            } else {
                for (int k = 0; k < 8; ++k) out[i * stride + k] = clamp8(descale(o[k], CB + P1 + 3) + 128);
            }
        }
    }
}

}  // namespace

// Decodes every scan of the file into per-component coefficient arrays - one scan of all components (baseline files), several
// scans of some components each (multi-scan sequential files), or the spectral-selection / successive-approximation scans of a
// progressive file - and runs the inverse transform once, at the end of the image.
std::string decode_jpeg(const std::vector<uint8_t>& f, uint32_t* w, uint32_t* h, uint32_t* channels, std::vector<uint8_t>* px) {
    uint16_t quant[4][64] = {};
    Huff dc[4], ac[4];
    Component comp[4];
    std::vector<int16_t> coefs[4];  // [block][64], natural (row-major) order within a block
    int ncomp = 0, width = 0, height = 0, restart_interval = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    bool have_frame = false, progressive = false, any_scan = false;
    size_t i = 2;
    auto be16 = [&](size_t at) { return (int)((f[at] << 8) | f[at + 1]); };
    while (i + 4 <= f.size()) {
        if (f[i] != 0xFF) return "JPEG: expected a marker";
        int marker = f[i + 1];
        i += 2;
        if (marker == 0xD8 || (marker >= 0xD0 && marker <= 0xD7) || marker == 0x01 || marker == 0xFF) {
            if (marker == 0xFF) --i;
            continue;
        }
        if (marker == 0xD9) break;
        if (i + 2 > f.size()) return "JPEG: truncated segment";
        int len = be16(i);
        if (len < 2 || i + (size_t)len > f.size()) return "JPEG: bad segment length";
        size_t seg = i + 2, seg_end = i + (size_t)len;
        if (marker == 0xDB) {  // quantisation tables
            while (seg < seg_end) {
                int pq = f[seg] >> 4, tq = f[seg] & 15;
                ++seg;
                if (tq > 3) return "JPEG: bad quantisation table id";
                for (int k = 0; k < 64; ++k) {
                    if (seg + (pq ? 2 : 1) > seg_end) return "JPEG: truncated quantisation table";
                    quant[tq][kZigzag[k]] = (uint16_t)(pq ? be16(seg) : f[seg]);
                    seg += pq ? 2 : 1;
                }
            }
        } else if (marker == 0xC4) {  // Huffman tables
            while (seg < seg_end) {
                int tc = f[seg] >> 4, th = f[seg] & 15;
                ++seg;
                if (th > 3 || tc > 1 || seg + 16 > seg_end) return "JPEG: bad Huffman table";
                Huff& t = tc ? ac[th] : dc[th];
                int counts[17] = {0};
                int total = 0;
                for (int k = 1; k <= 16; ++k) {
                    counts[k] = f[seg + k - 1];
                    total += counts[k];
                }
                seg += 16;
                if (total > 256 || seg + (size_t)total > seg_end) return "JPEG: bad Huffman table";
                std::memcpy(t.vals, &f[seg], (size_t)total);
                seg += (size_t)total;
                int code = 0, k = 0;
                for (int l = 1; l <= 16; ++l) {
                    t.valptr[l] = k;
                    t.mincode[l] = code;
                    code += counts[l];
                    k += counts[l];
                    t.maxcode[l] = counts[l] ? code - 1 : -1;
                    code <<= 1;
                }
                t.present = true;
            }
        } else if (marker == 0xC0 || marker == 0xC1 || marker == 0xC2) {  // baseline / extended sequential / progressive, Huffman
            if (have_frame) return "JPEG: more than one frame";
            if (len < 8) return "JPEG: truncated frame header";
            if (f[seg] != 8) return "JPEG: only 8-bit samples are supported";
            progressive = marker == 0xC2;
            height = be16(seg + 1);
            width = be16(seg + 3);
            ncomp = f[seg + 5];
            if (width <= 0 || height <= 0 || (ncomp != 1 && ncomp != 3)) return "JPEG: unsupported frame";
            if (len < 8 + 3 * ncomp) return "JPEG: truncated frame header";
            if ((uint64_t)width * (uint64_t)height > kMaxImagePixels) return "JPEG: image too large";
            // An 8x8 block costs at least two bits of entropy-coded data in a sequential file (a one-bit DC code and a one-bit
            // end-of-block) and one bit in a progressive one (its DC scan), so a file of n bytes describes at most 256 n (512 n)
            // pixels: a frame header that claims more is lying, and is refused before anything is allocated.
            if ((uint64_t)width * (uint64_t)height > (uint64_t)f.size() * (progressive ? 512u : 256u)) return "JPEG: frame larger than the file's data can describe";
            for (int c = 0; c < ncomp; ++c) {
                comp[c].id = f[seg + 6 + 3 * c];
                comp[c].h = f[seg + 7 + 3 * c] >> 4;
                comp[c].v = f[seg + 7 + 3 * c] & 15;
                comp[c].tq = f[seg + 8 + 3 * c];
                if (comp[c].h < 1 || comp[c].h > 2 || comp[c].v < 1 || comp[c].v > 2 || comp[c].tq > 3) return "JPEG: unsupported sampling";
            }
            if (ncomp == 1) comp[0].h = comp[0].v = 1;
            for (int c = 0; c < ncomp; ++c) {
                hmax = comp[c].h > hmax ? comp[c].h : hmax;
                vmax = comp[c].v > vmax ? comp[c].v : vmax;
            }
            mcux = (width + 8 * hmax - 1) / (8 * hmax);
            mcuy = (height + 8 * vmax - 1) / (8 * vmax);
            for (int c = 0; c < ncomp; ++c) {
                comp[c].bw = mcux * comp[c].h;
                comp[c].bh = mcuy * comp[c].v;
                coefs[c].assign((size_t)comp[c].bw * comp[c].bh * 64, 0);
            }
            have_frame = true;
        } else if (marker == 0xC3 || (marker >= 0xC5 && marker <= 0xCF && marker != 0xC8 && marker != 0xCC)) {
            return "JPEG: lossless, hierarchical and arithmetic-coded files are not supported";
        } else if (marker == 0xDD) {
            if (len < 4) return "JPEG: truncated restart interval";
            restart_interval = be16(seg);
        } else if (marker == 0xDA) {  // start of scan
            if (!have_frame) return "JPEG: scan before frame";
            if (len < 3) return "JPEG: truncated scan header";
            const int ns = f[seg];
            if (ns < 1 || ns > ncomp) return "JPEG: bad number of components in scan";
            if (len < 6 + 2 * ns) return "JPEG: truncated scan header";
            int in_scan[4];
            for (int k = 0; k < ns; ++k) {
                const int cid = f[seg + 1 + 2 * k];
                int found = -1;
                for (int c = 0; c < ncomp; ++c)
                    if (comp[c].id == cid) found = c;
                if (found < 0) return "JPEG: scan names a component the frame does not have";
                for (int q = 0; q < k; ++q)
                    if (in_scan[q] == found) return "JPEG: a component appears twice in a scan";
                in_scan[k] = found;
                comp[found].td = f[seg + 2 + 2 * k] >> 4;
                comp[found].ta = f[seg + 2 + 2 * k] & 15;
                if (comp[found].td > 3 || comp[found].ta > 3) return "JPEG: bad Huffman table id in scan";
            }
            int Ss = f[seg + 1 + 2 * ns], Se = f[seg + 2 + 2 * ns], Ah = f[seg + 3 + 2 * ns] >> 4, Al = f[seg + 3 + 2 * ns] & 15;
            if (!progressive) {
                Ss = 0;
                Se = 63;
                Ah = Al = 0;
            } else {
                if (Ss > Se || Se > 63 || Ah > 13 || Al > 13) return "JPEG: bad spectral selection or successive approximation";
                if (Ss == 0 && Se != 0) return "JPEG: a progressive scan mixes DC and AC coefficients";
                if (Ss != 0 && ns != 1) return "JPEG: a progressive AC scan of several components";
                if (Ah != 0 && Ah != Al + 1) return "JPEG: bad successive approximation";
            }
            const bool needs_dc = Ss == 0 && Ah == 0, needs_ac = Se > 0;
            for (int k = 0; k < ns; ++k) {
                const Component& cc = comp[in_scan[k]];
                if (needs_dc && !dc[cc.td].present) return "JPEG: missing Huffman table";
                if (needs_ac && !ac[cc.ta].present) return "JPEG: missing Huffman table";
                comp[in_scan[k]].pred = 0;
            }
            // the units the scan walks over: MCUs when it holds several components, otherwise the blocks of the one component
            // that cover the image (not the padding that rounds the component up to whole MCUs)
            int ux = mcux, uy = mcuy;
            if (ns == 1) {
                const Component& cc = comp[in_scan[0]];
                ux = ((width * cc.h + hmax - 1) / hmax + 7) / 8;
                uy = ((height * cc.v + vmax - 1) / vmax + 7) / 8;
            }
            JpegReader rd{f.data() + seg_end, f.data() + f.size()};
            int unit_count = 0, eobrun = 0;
            const int p1 = 1 << Al, m1 = -(1 << Al);
            for (int my = 0; my < uy; ++my) {
                for (int mx = 0; mx < ux; ++mx) {
                    if (restart_interval && unit_count && unit_count % restart_interval == 0) {
                        // byte-align and consume the RSTn marker
                        rd.reset_bits();
                        while (rd.p + 1 < rd.end && !(rd.p[0] == 0xFF && rd.p[1] >= 0xD0 && rd.p[1] <= 0xD7)) ++rd.p;
                        if (rd.p + 1 < rd.end) rd.p += 2;
                        for (int k = 0; k < ns; ++k) comp[in_scan[k]].pred = 0;
                        eobrun = 0;
                    }
                    for (int k = 0; k < ns; ++k) {
                        Component& cc = comp[in_scan[k]];
                        const int nbx = ns == 1 ? 1 : cc.h, nby = ns == 1 ? 1 : cc.v;
                        for (int by = 0; by < nby; ++by) {
                            for (int bx = 0; bx < nbx; ++bx) {
                                const int gx = ns == 1 ? mx : mx * cc.h + bx, gy = ns == 1 ? my : my * cc.v + by;
                                int16_t* blk = &coefs[in_scan[k]][((size_t)gy * cc.bw + gx) * 64];
                                if (Ss == 0) {
                                    if (Ah == 0) {  // the DC coefficient (whole, or its high bits)
                                        int sz = rd.decode(dc[cc.td]);
                                        if (sz < 0 || sz > 15) return "JPEG: bad DC code";
                                        int diff = sz ? extend(rd.get_bits(sz), sz) : 0;
                                        cc.pred += diff;
                                        blk[0] = (int16_t)(cc.pred * (1 << Al));
                                    } else if (rd.get_bits(1)) {  // one more bit of it
                                        blk[0] = (int16_t)(blk[0] | p1);
                                    }
                                }
                                if (Se == 0) continue;
                                int kk = Ss == 0 ? 1 : Ss;
                                if (Ah == 0) {  // AC coefficients Ss..Se (whole, or their high bits)
                                    if (eobrun > 0) {
                                        --eobrun;
                                        continue;
                                    }
                                    for (; kk <= Se; ++kk) {
                                        int rs = rd.decode(ac[cc.ta]);
                                        if (rs < 0) return "JPEG: bad AC code";
                                        int r = rs >> 4, sz = rs & 15;
                                        if (sz) {
                                            kk += r;
                                            if (kk > Se) return "JPEG: coefficient index out of range";
                                            blk[kZigzag[kk]] = (int16_t)(extend(rd.get_bits(sz), sz) * (1 << Al));
                                        } else if (r == 15) {
                                            kk += 15;
                                        } else {
                                            if (!progressive && r != 0) return "JPEG: bad AC code";
                                            eobrun = 1 << r;
                                            if (r) eobrun += rd.get_bits(r);
                                            --eobrun;
                                            break;
                                        }
                                    }
                                } else {  // one more bit of the AC coefficients Ss..Se
                                    if (eobrun == 0) {
                                        for (; kk <= Se; ++kk) {
                                            int rs = rd.decode(ac[cc.ta]);
                                            if (rs < 0) return "JPEG: bad AC code";
                                            int r = rs >> 4, sz = rs & 15, value = 0;
                                            if (sz) {
                                                if (sz != 1) return "JPEG: bad AC refinement code";
                                                value = rd.get_bits(1) ? p1 : m1;
                                            } else if (r != 15) {
                                                eobrun = 1 << r;
                                                if (r) eobrun += rd.get_bits(r);
                                                break;  // the rest of the band is handled below
                                            }
                                            // pass the coefficients that are already non-zero (each takes a correction bit) and r that are still zero
                                            for (; kk <= Se; ++kk) {
                                                int16_t& cf = blk[kZigzag[kk]];
                                                if (cf != 0) {
                                                    if (rd.get_bits(1) && (cf & p1) == 0) cf = (int16_t)(cf + (cf >= 0 ? p1 : m1));
                                                } else if (--r < 0) break;
                                            }
                                            if (value) {
                                                if (kk > Se) return "JPEG: coefficient index out of range";
                                                blk[kZigzag[kk]] = (int16_t)value;
                                            }
                                        }
                                    }
                                    if (eobrun > 0) {  // to the end of the band: correction bits for what is already non-zero
                                        for (; kk <= Se; ++kk) {
                                            int16_t& cf = blk[kZigzag[kk]];
                                            if (cf != 0 && rd.get_bits(1) && (cf & p1) == 0) cf = (int16_t)(cf + (cf >= 0 ? p1 : m1));
                                        }
                                        --eobrun;
                                    }
                                }
                            }
                        }
                    }
                    ++unit_count;
                }
            }
            any_scan = true;
            // the entropy-coded data runs up to the next marker that is not a restart marker or a stuffed byte
            size_t at = (size_t)(rd.p - f.data());
            if (rd.hit_marker) at = (size_t)(rd.p - f.data());
            while (at + 1 < f.size() && !(f[at] == 0xFF && f[at + 1] != 0x00 && !(f[at + 1] >= 0xD0 && f[at + 1] <= 0xD7) && f[at + 1] != 0xFF)) ++at;
            i = at;
            continue;
        }
        i = seg_end;
    }
    if (!have_frame || !any_scan) return "JPEG: no image data found";
    // inverse transform and assembly
    for (int c = 0; c < ncomp; ++c) {
        comp[c].plane.assign((size_t)comp[c].bw * 8 * comp[c].bh * 8, 0);
        const int stride = comp[c].bw * 8;
        for (int gy = 0; gy < comp[c].bh; ++gy)
            for (int gx = 0; gx < comp[c].bw; ++gx) {
                int coef[64];
                const int16_t* blk = &coefs[c][((size_t)gy * comp[c].bw + gx) * 64];
                for (int k = 0; k < 64; ++k) coef[k] = blk[k];
                idct_block(coef, quant[comp[c].tq], &comp[c].plane[(size_t)gy * 8 * stride + (size_t)gx * 8], stride);
            }
    }
    *w = (uint32_t)width;
    *h = (uint32_t)height;
    *channels = ncomp == 1 ? 1u : 3u;
    px->resize((size_t)width * height * *channels);
    if (ncomp == 1) {
        int stride = comp[0].bw * 8;
        for (int y = 0; y < height; ++y) std::memcpy(&(*px)[(size_t)y * width], &comp[0].plane[(size_t)y * stride], (size_t)width);
    } else {
        for (int y = 0; y < height; ++y) {
            for (int x = 0; x < width; ++x) {
                int sv[3];
                for (int c = 0; c < 3; ++c) {
                    int sx = x * comp[c].h / hmax, sy = y * comp[c].v / vmax;
                    sv[c] = comp[c].plane[(size_t)sy * comp[c].bw * 8 + sx];
                }
                int Y = sv[0], cb = sv[1] - 128, cr = sv[2] - 128;
                int r = Y + (int)((91881L * cr + 32768) >> 16);
                int g = Y + (int)((-22554L * cb + 32768 + -46802L * cr) >> 16);
                int b = Y + (int)((116130L * cb + 32768) >> 16);
                uint8_t* q = &(*px)[((size_t)y * width + x) * 3];
                q[0] = clamp8(r);
                q[1] = clamp8(g);
                q[2] = clamp8(b);
            }
        }
    }
    return "";
}

std::string decode_png(const std::vector<uint8_t>& f, uint32_t* w, uint32_t* h, uint32_t* channels, std::vector<uint8_t>* px) {
    auto be32 = [&](size_t at) { return ((uint32_t)f[at] << 24) | ((uint32_t)f[at + 1] << 16) | ((uint32_t)f[at + 2] << 8) | (uint32_t)f[at + 3]; };
    size_t i = 8;
    uint32_t width = 0, height = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, palette;
    while (i + 12 <= f.size()) {
        uint32_t len = be32(i);
        if (i + 12 + (size_t)len > f.size()) return "PNG: truncated chunk";
        const uint8_t* tag = &f[i + 4];
        const uint8_t* data = &f[i + 8];
        if (!std::memcmp(tag, "IHDR", 4)) {
            if (len < 13) return "PNG: bad IHDR";
            width = be32(i + 8);
            height = be32(i + 12);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
        } else if (!std::memcmp(tag, "PLTE", 4)) {
            palette.assign(data, data + len);
        } else if (!std::memcmp(tag, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(tag, "IEND", 4)) {
            break;
        }
        i += 12 + (size_t)len;
    }
    if (width == 0 || height == 0) return "PNG: missing IHDR";
    if ((uint64_t)width * (uint64_t)height > kMaxImagePixels) return "PNG: image too large";
    if (interlace > 1) return "PNG: unknown interlace method";
    int samples = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!samples || (depth != 8 && depth != 16 && !((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))))
        return "PNG: unsupported colour type / bit depth";
    const size_t bpp_bits = (size_t)samples * depth;
    const size_t bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
    // The image is one pass over all pixels, or the seven passes of Adam7 (each a smaller image of every xstep-th pixel of every
    // ystep-th row, with filter bytes and a "previous row" of its own).
    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass kAdam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass kWhole[1] = {{0, 0, 1, 1}};
    const Pass* passes = interlace ? kAdam7 : kWhole;
    const int n_passes = interlace ? 7 : 1;
    auto pass_size = [&](const Pass& ps, uint32_t* pw, uint32_t* ph) {
        *pw = width > ps.x0 ? (width - ps.x0 + ps.dx - 1) / ps.dx : 0;
        *ph = height > ps.y0 ? (height - ps.y0 + ps.dy - 1) / ps.dy : 0;
    };
    uint64_t raw_bytes = 0;
    for (int k = 0; k < n_passes; ++k) {
        uint32_t pw, ph;
        pass_size(passes[k], &pw, &ph);
        if (pw && ph) raw_bytes += (uint64_t)(((size_t)pw * bpp_bits + 7) / 8 + 1) * ph;
    }
    // deflate expands by at most a factor of 1032, so the header's size can be checked against the data before allocating
    if (raw_bytes > (uint64_t)idat.size() * 1032u + 1024u) return "PNG: image larger than its data can describe";
    std::vector<uint8_t> raw((size_t)raw_bytes);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) return "PNG: inflate failed";
    *w = width;
    *h = height;
    *channels = (ctype == 0 || ctype == 4) ? 1u : 3u;
    px->assign((size_t)width * height * *channels, 0);
    size_t at = 0;
    for (int k = 0; k < n_passes; ++k) {
        const Pass& ps = passes[k];
        uint32_t pw, ph;
        pass_size(ps, &pw, &ph);
        if (!pw || !ph) continue;
        const size_t row_bytes = ((size_t)pw * bpp_bits + 7) / 8;
        std::vector<uint8_t> prev(row_bytes, 0), cur(row_bytes);
        for (uint32_t py = 0; py < ph; ++py) {
            const uint8_t* line = &raw[at];
            at += row_bytes + 1;
            int filter = line[0];
            for (size_t j = 0; j < row_bytes; ++j) {
                int a = j >= bpp ? cur[j - bpp] : 0, b = prev[j], c = j >= bpp ? prev[j - bpp] : 0;
                int x = line[1 + j];
                switch (filter) {
                    case 0: break;
                    case 1: x += a; break;
                    case 2: x += b; break;
                    case 3: x += (a + b) >> 1; break;
                    case 4: {
                        int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
                        x += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                        break;
                    }
                    default: return "PNG: bad filter";
                }
                cur[j] = (uint8_t)x;
            }
            const uint32_t y = ps.y0 + py * ps.dy;
            for (uint32_t qx = 0; qx < pw; ++qx) {
                const uint32_t x = ps.x0 + qx * ps.dx;
                uint8_t sv[4] = {0, 0, 0, 0};
                for (int c = 0; c < samples; ++c) {
                    if (depth == 8) sv[c] = cur[(size_t)qx * samples + c];
                    else if (depth == 16) sv[c] = cur[((size_t)qx * samples + c) * 2];
                    else {
                        size_t bit = (size_t)qx * depth;
                        int v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
                        sv[c] = ctype == 3 ? (uint8_t)v : (uint8_t)(v * 255 / ((1 << depth) - 1));
                    }
                }
                uint8_t* q = &(*px)[((size_t)y * width + x) * *channels];
                if (ctype == 3) {
                    size_t pi = (size_t)sv[0] * 3;
                    if (pi + 3 > palette.size()) return "PNG: palette index out of range";
                    q[0] = palette[pi]; q[1] = palette[pi + 1]; q[2] = palette[pi + 2];
                } else if (*channels == 1) {
                    q[0] = sv[0];
                } else {
                    q[0] = sv[0]; q[1] = sv[1]; q[2] = sv[2];
                }
            }
            prev.swap(cur);
        }
    }
    return "";
}

}  // namespace rrt_host
