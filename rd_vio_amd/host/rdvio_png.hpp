// rdvio_png.hpp -- minimal PNG reader for the EuRoC harness (libpng / OpenCV are not in this build image; zlib is).
// Supports what EuRoC's cam0/data/*.png and the synthetic mav0 writer produce: non-interlaced, 8 bits per sample, colour
// types 0 (gray), 2 (RGB), 4 (gray + alpha), 6 (RGBA).  Output: rows x cols x channels u8 (channels 1 or 3; alpha dropped,
// RGB order kept).  Stands where the reference calls cv::imread(IMREAD_UNCHANGED) (/root/reference/examples/dataset.hpp:585-592).
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace rdvio_hip {

struct PngImage {
    int width = 0, height = 0, channels = 0;
    std::vector<uint8_t> pixels;
};

inline PngImage read_png(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("file not found: " + path);
    std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (buf.size() < 8 || std::equal(sig, sig + 8, buf.begin()) == false) throw std::runtime_error("not a PNG file: " + path);
    auto be32 = [&](size_t o) { return ((uint32_t)buf[o] << 24) | ((uint32_t)buf[o + 1] << 16) | ((uint32_t)buf[o + 2] << 8) | buf[o + 3]; };
    size_t pos = 8;
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat;
    while (pos + 12 <= buf.size()) {
        const uint32_t len = be32(pos);
        const std::string type(buf.begin() + (long)pos + 4, buf.begin() + (long)pos + 8);
        if (pos + 12 + len > buf.size()) throw std::runtime_error("truncated PNG: " + path);
        const size_t data = pos + 8;
        if (type == "IHDR") {
            if (len != 13) throw std::runtime_error("malformed PNG (IHDR is not 13 bytes): " + path);
            w = (int)be32(data);
            h = (int)be32(data + 4);
            depth = buf[data + 8];
            ctype = buf[data + 9];
            interlace = buf[data + 12];
        } else if (type == "IDAT") {
            idat.insert(idat.end(), buf.begin() + (long)data, buf.begin() + (long)(data + len));
        } else if (type == "IEND") {
            break;
        }
        pos += 12 + len;
    }
    if (w <= 0 || h <= 0 || depth != 8 || interlace != 0 || (ctype != 0 && ctype != 2 && ctype != 4 && ctype != 6))
        throw std::runtime_error("unsupported PNG (need 8-bit, non-interlaced, gray / RGB with optional alpha): " + path);
    // the header sizes every allocation below: refuse what no camera of this pipeline produces before trusting it
    if (w > 16384 || h > 16384) throw std::runtime_error("unreasonable PNG dimensions: " + path);
    const int spp = ctype == 0 ? 1 : (ctype == 2 ? 3 : (ctype == 4 ? 2 : 4));
    const size_t stride = (size_t)w * spp;
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size())
        throw std::runtime_error("PNG inflate failed: " + path);
    // undo the per-row filters (PNG spec 9.2)
    std::vector<uint8_t> img(stride * (size_t)h);
    for (int y = 0; y < h; ++y) {
        const uint8_t *src = &raw[(stride + 1) * (size_t)y];
        uint8_t *dst = &img[stride * (size_t)y];
        const uint8_t *up = y > 0 ? dst - stride : nullptr;
        const int ft = src[0];
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= (size_t)spp ? dst[x - spp] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)spp) ? up[x - spp] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) / 2; break;
                case 4: {
                    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: throw std::runtime_error("bad PNG filter type: " + path);
            }
            dst[x] = (uint8_t)(src[1 + x] + pred);
        }
    }
    PngImage out;
    out.width = w;
    out.height = h;
    out.channels = (ctype == 0 || ctype == 4) ? 1 : 3;
    out.pixels.resize((size_t)w * h * out.channels);
    for (size_t i = 0; i < (size_t)w * h; ++i)
        for (int ch = 0; ch < out.channels; ++ch) out.pixels[i * out.channels + ch] = img[i * spp + ch];
    return out;
}

}  // namespace rdvio_hip
