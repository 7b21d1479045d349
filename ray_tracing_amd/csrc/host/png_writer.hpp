// RGBA8 PNG writer: what `image::save_buffer("screenshot.png", &pixels, w, h, ColorType::Rgba8)` does for
// the reference's image mode (src/cli.rs:31-38). One IDAT chunk, filter type 0 on every row, zlib deflate.
#pragma once
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace rt {

inline void png_chunk(std::vector<uint8_t>& out, const char tag[4], const uint8_t* data, size_t n) {
    auto be32 = [&](uint32_t v) { for (int s = 24; s >= 0; s -= 8) out.push_back((uint8_t)(v >> s)); };
    be32((uint32_t)n);
    const size_t start = out.size();
    out.insert(out.end(), tag, tag + 4);
    if (n) out.insert(out.end(), data, data + n);
    be32((uint32_t)crc32(0L, out.data() + start, (uInt)(n + 4)));
}

inline std::vector<uint8_t> png_encode_rgba8(const uint8_t* rgba, uint32_t width, uint32_t height) {
    if (!rgba || width == 0 || height == 0) throw std::invalid_argument("png: empty image");
    std::vector<uint8_t> raw((size_t)height * ((size_t)width * 4 + 1));
    for (uint32_t y = 0; y < height; ++y) {
        uint8_t* row = raw.data() + (size_t)y * ((size_t)width * 4 + 1);
        row[0] = 0;  // filter: None
        std::copy(rgba + (size_t)y * width * 4, rgba + (size_t)(y + 1) * width * 4, row + 1);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) throw std::runtime_error("png: deflate failed");
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    uint8_t ihdr[13];
    for (int i = 0; i < 4; ++i) { ihdr[i] = (uint8_t)(width >> (24 - 8 * i)); ihdr[4 + i] = (uint8_t)(height >> (24 - 8 * i)); }
    ihdr[8] = 8;   // bit depth
    ihdr[9] = 6;   // colour type: RGBA
    ihdr[10] = ihdr[11] = ihdr[12] = 0;
    png_chunk(out, "IHDR", ihdr, sizeof ihdr);
    png_chunk(out, "IDAT", comp.data(), clen);
    png_chunk(out, "IEND", nullptr, 0);
    return out;
}

inline void png_save_rgba8(const std::string& path, const uint8_t* rgba, uint32_t width, uint32_t height) {
    const std::vector<uint8_t> bytes = png_encode_rgba8(rgba, width, height);
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("png: cannot open " + path);
    const size_t n = std::fwrite(bytes.data(), 1, bytes.size(), f);
    if (std::fclose(f) != 0 || n != bytes.size()) throw std::runtime_error("png: short write to " + path);
}

}  // namespace rt
