#include "png_min.h"

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put32(std::vector<unsigned char>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
int paeth(int a, int b, int c) { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
void chunk(std::vector<unsigned char>& out, const char* type, const std::vector<unsigned char>& data) {
    put32(out, (uint32_t)data.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start)));
}
} // namespace

bool png_read_planar_f32(const std::string& path, std::vector<float>& img, size_t& w, size_t& h, size_t& c) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<unsigned char> buf;
    unsigned char tmp[65536];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    std::fclose(f);
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (buf.size() < 33 || std::memcmp(buf.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    std::vector<unsigned char> idat;
    int depth = 0, ctype = 0, interlace = 0;
    w = h = 0;
    while (pos + 12 <= buf.size()) {
        const uint32_t len = be32(&buf[pos]);
        const char* type = (const char*)&buf[pos + 4];
        if (pos + 12 + len > buf.size()) return false;
        const unsigned char* d = &buf[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) { w = be32(d); h = be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12]; }
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + len;
    }
    if (!w || !h || depth != 8 || interlace != 0) return false;
    c = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;
    if (!c) return false;
    const size_t stride = w * c;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return false;
    std::vector<unsigned char> px(stride * h);
    for (size_t y = 0; y < h; y++) {
        const unsigned char ft = raw[y * (stride + 1)];
        const unsigned char* in = &raw[y * (stride + 1) + 1];
        unsigned char* out = &px[y * stride];
        const unsigned char* up = y ? &px[(y - 1) * stride] : nullptr;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= c ? out[i - c] : 0, b = up ? up[i] : 0, cc = (up && i >= c) ? up[i - c] : 0;
            int v = in[i];
            switch (ft) { case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) / 2; break; case 4: v += paeth(a, b, cc); break; default: break; }
            out[i] = (unsigned char)v;
        }
    }
    img.resize(w * h * c);
    for (size_t ch = 0; ch < c; ch++)
        for (size_t i = 0; i < w * h; i++) img[ch * w * h + i] = (float)px[i * c + ch];
    return true;
}

bool png_write_planar_f32(const std::string& path, const float* img, size_t w, size_t h, size_t c) {
    if (c != 1 && c != 3) return false;
    const size_t stride = w * c;
    std::vector<unsigned char> raw((stride + 1) * h);
    for (size_t y = 0; y < h; y++) {
        raw[y * (stride + 1)] = 0;
        for (size_t x = 0; x < w; x++)
            for (size_t ch = 0; ch < c; ch++) {
                const float v = img[ch * w * h + y * w + x];
                raw[y * (stride + 1) + 1 + x * c + ch] = (unsigned char)(v + 0.5f); /* io_png.c: round to nearest */
            }
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    z.resize(zlen);
    std::vector<unsigned char> out = {137, 80, 78, 71, 13, 10, 26, 10}, ihdr;
    put32(ihdr, (uint32_t)w); put32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(c == 1 ? 0 : 2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok;
}
