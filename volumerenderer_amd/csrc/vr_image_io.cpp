/*
 * vr_image_io.cpp -- writes a finished RGB24 frame to disk: PNG (what the reference driver produces
 * through zigimg, src/main.zig:116,137: "renders/out_iso{d}.png") and binary PPM.  No compression
 * library is needed: the PNG carries its scanlines in stored (uncompressed) deflate blocks, which
 * every PNG reader accepts.  Host code only.
 */
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/vr.h"

namespace {

uint32_t crc_table[256];
bool crc_ready = false;

void crc_init()
{
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? (0xedb88320u ^ (c >> 1)) : (c >> 1);
        crc_table[n] = c;
    }
    crc_ready = true;
}

uint32_t crc_update(uint32_t c, const uint8_t *p, size_t n)
{
    for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xffu] ^ (c >> 8);
    return c;
}

void put_be32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

bool write_chunk(FILE *f, const char type[4], const std::vector<uint8_t> &data)
{
    std::vector<uint8_t> head;
    put_be32(head, (uint32_t)data.size());
    head.insert(head.end(), type, type + 4);
    uint32_t c = crc_update(0xffffffffu, head.data() + 4, 4);
    c = crc_update(c, data.data(), data.size()) ^ 0xffffffffu;
    std::vector<uint8_t> tail;
    put_be32(tail, c);
    return std::fwrite(head.data(), 1, head.size(), f) == head.size() &&
           (data.empty() || std::fwrite(data.data(), 1, data.size(), f) == data.size()) &&
           std::fwrite(tail.data(), 1, 4, f) == 4;
}

} // namespace

extern "C" {

int vr_write_png_rgb24(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb24)
{
    if (!path || !rgb24 || !width || !height) return VR_EARG;
    if (!crc_ready) crc_init();
    /* raw image data: every scanline prefixed by filter type 0 */
    const size_t row = (size_t)width * 3u, raw_len = (row + 1u) * height;
    std::vector<uint8_t> raw(raw_len);
    for (uint32_t y = 0; y < height; y++) {
        raw[(row + 1u) * y] = 0;
        std::memcpy(&raw[(row + 1u) * y + 1u], rgb24 + row * y, row);
    }
    /* zlib stream: header, stored blocks of at most 65535 bytes, Adler-32 */
    std::vector<uint8_t> z;
    z.reserve(raw_len + raw_len / 65535u * 5u + 16u);
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw_len;) {
        const size_t n = raw_len - off < 65535u ? raw_len - off : 65535u;
        z.push_back(off + n == raw_len ? 1 : 0);
        z.push_back((uint8_t)(n & 0xffu)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xffu)); z.push_back((uint8_t)((~n >> 8) & 0xffu));
        z.insert(z.end(), raw.begin() + (long)off, raw.begin() + (long)(off + n));
        for (size_t i = 0; i < n; i++) { a = (a + raw[off + i]) % 65521u; b = (b + a) % 65521u; }
        off += n;
    }
    put_be32(z, (b << 16) | a);
    FILE *f = std::fopen(path, "wb");
    if (!f) return VR_EIO;
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, width); put_be32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0); /* 8-bit RGB */
    bool ok = std::fwrite(sig, 1, 8, f) == 8 && write_chunk(f, "IHDR", ihdr) && write_chunk(f, "IDAT", z) &&
              write_chunk(f, "IEND", std::vector<uint8_t>());
    ok = (std::fclose(f) == 0) && ok;
    return ok ? VR_OK : VR_EIO;
}

int vr_write_ppm_rgb24(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb24)
{
    if (!path || !rgb24 || !width || !height) return VR_EARG;
    FILE *f = std::fopen(path, "wb");
    if (!f) return VR_EIO;
    bool ok = std::fprintf(f, "P6\n%u %u\n255\n", width, height) > 0;
    const size_t n = (size_t)width * height * 3u;
    ok = ok && std::fwrite(rgb24, 1, n, f) == n;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? VR_OK : VR_EIO;
}

} // extern "C"
