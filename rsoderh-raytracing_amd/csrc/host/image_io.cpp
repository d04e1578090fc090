// image_io.cpp — Radiance .hdr reader, PNG / PFM writers, camera state string, CPU display transform.
//
//  rsrt_load_hdr           what `image::load_from_memory(..).into_rgb32f()` gives the reference for its
//                          HDRIs (src/state.rs:119-132; image 0.25 `hdr` decoder, un-vendored): RGBE,
//                          value = mantissa * 2^(e - 136), all zero when e == 0, rows top to bottom,
//                          new-style per-channel RLE and flat scanlines.
//  rsrt_camera_serialize / rsrt_camera_deserialize
//                          Camera::serialize / deserialize (src/camera.rs:30-89): 24 little-endian bytes
//                          [pos.xyz, yaw, pitch, fov_y] as standard base64 (the --state flag, src/cli.rs:39-43).
//  rsrt_write_png / rsrt_write_pfm
//                          image output (the reference only presents to a window).
//  rsrt_display_srgb8_host the display transform of include/rsrt_tonemap.h on the CPU, for hosts that
//                          downloaded the f32 sums.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/rsrt_host.h"
#include "../../../include/rsrt_tonemap.h"

namespace {

void set_err(char *err, size_t n, const std::string &m)
{
    if (err && n) std::snprintf(err, n, "%s", m.c_str());
}

bool read_all(const char *path, std::vector<unsigned char> &out)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    unsigned char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    std::fclose(f);
    return true;
}

// ---- PNG: stored (uncompressed) deflate blocks
uint32_t crc_table[256];
bool crc_ready = false;
void crc_init()
{
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
    crc_ready = true;
}
uint32_t crc32(uint32_t crc, const unsigned char *p, size_t n)
{
    if (!crc_ready) crc_init();
    crc ^= 0xffffffffu;
    for (size_t i = 0; i < n; i++) crc = crc_table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return crc ^ 0xffffffffu;
}
void put32(std::vector<unsigned char> &v, uint32_t x)
{
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16)); v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}
void chunk(std::vector<unsigned char> &png, const char *type, const std::vector<unsigned char> &data)
{
    put32(png, (uint32_t)data.size());
    size_t start = png.size();
    png.insert(png.end(), type, type + 4);
    png.insert(png.end(), data.begin(), data.end());
    put32(png, crc32(0, png.data() + start, png.size() - start));
}

const char B64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";

} // namespace

extern "C" {

int rsrt_load_hdr(const char *path, uint32_t *width, uint32_t *height, float **rgb_out, char *err, size_t err_len)
{
    if (!path || !width || !height || !rgb_out) { set_err(err, err_len, "null argument"); return 1; }
    *rgb_out = nullptr;
    std::vector<unsigned char> d;
    if (!read_all(path, d)) { set_err(err, err_len, std::string("cannot open ") + path); return 2; }
    size_t p = 0;
    auto line = [&](std::string &out) -> bool {
        out.clear();
        while (p < d.size() && d[p] != '\n') out += (char)d[p++];
        if (p >= d.size()) return false;
        p++;
        return true;
    };
    std::string l;
    if (!line(l) || (l.rfind("#?RADIANCE", 0) != 0 && l.rfind("#?RGBE", 0) != 0)) { set_err(err, err_len, "not a Radiance HDR file"); return 3; }
    bool format_ok = false;
    for (;;) {
        if (!line(l)) { set_err(err, err_len, "truncated header"); return 3; }
        if (l.empty()) break;
        if (l.rfind("FORMAT=", 0) == 0) format_ok = (l == "FORMAT=32-bit_rle_rgbe");
    }
    if (!format_ok) { set_err(err, err_len, "unsupported FORMAT (need 32-bit_rle_rgbe)"); return 3; }
    if (!line(l)) { set_err(err, err_len, "missing resolution line"); return 3; }
    int h = 0, w = 0;
    if (std::sscanf(l.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) { set_err(err, err_len, "unsupported orientation '" + l + "' (need -Y h +X w)"); return 3; }
    float *rgb = (float *)std::malloc((size_t)w * h * 3 * sizeof(float));
    if (!rgb) { set_err(err, err_len, "out of memory"); return 4; }
    std::vector<unsigned char> scan((size_t)w * 4);
    for (int y = 0; y < h; y++) {
        bool rle = false;
        if (w >= 8 && w < 32768 && p + 4 <= d.size() && d[p] == 2 && d[p + 1] == 2 && ((d[p + 2] << 8) | d[p + 3]) == w) rle = true;
        if (rle) {
            p += 4;
            for (int c = 0; c < 4; c++) {
                int x = 0;
                while (x < w) {
                    if (p >= d.size()) { std::free(rgb); set_err(err, err_len, "truncated RLE data"); return 3; }
                    int n = d[p++];
                    if (n > 128) { // run
                        n -= 128;
                        if (p >= d.size() || x + n > w) { std::free(rgb); set_err(err, err_len, "bad RLE run"); return 3; }
                        unsigned char v = d[p++];
                        for (int k = 0; k < n; k++) scan[(size_t)(x++) * 4 + c] = v;
                    } else { // literal
                        if (n == 0 || p + n > d.size() || x + n > w) { std::free(rgb); set_err(err, err_len, "bad RLE literal"); return 3; }
                        for (int k = 0; k < n; k++) scan[(size_t)(x++) * 4 + c] = d[p++];
                    }
                }
            }
        } else {
            if (p + (size_t)w * 4 > d.size()) { std::free(rgb); set_err(err, err_len, "truncated pixel data"); return 3; }
            std::memcpy(scan.data(), d.data() + p, (size_t)w * 4);
            p += (size_t)w * 4;
        }
        for (int x = 0; x < w; x++) {
            const unsigned char *q = &scan[(size_t)x * 4];
            float *o = rgb + ((size_t)y * w + x) * 3;
            if (q[3] == 0) { o[0] = o[1] = o[2] = 0.0f; }
            else {
                const float scale = std::ldexp(1.0f, (int)q[3] - 136); // exp2(e - 128 - 8)
                o[0] = (float)q[0] * scale; o[1] = (float)q[1] * scale; o[2] = (float)q[2] * scale;
            }
        }
    }
    *width = (uint32_t)w;
    *height = (uint32_t)h;
    *rgb_out = rgb;
    return 0;
}

void rsrt_free(void *p) { std::free(p); }

int rsrt_write_png(const char *path, uint32_t width, uint32_t height, const uint8_t *rgba8)
{
    if (!path || !rgba8 || width == 0 || height == 0) return 1;
    std::vector<unsigned char> raw;
    raw.reserve((size_t)height * (1 + (size_t)width * 4));
    for (uint32_t y = 0; y < height; y++) {
        raw.push_back(0); // filter: none
        raw.insert(raw.end(), rgba8 + (size_t)y * width * 4, rgba8 + (size_t)(y + 1) * width * 4);
    }
    std::vector<unsigned char> z = {0x78, 0x01}; // zlib header, no compression
    uint32_t a = 1, b = 0; // adler32
    for (size_t i = 0; i < raw.size(); i++) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
    size_t pos = 0;
    while (pos < raw.size()) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((unsigned char)(n & 0xff)); z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xff)); z.push_back((unsigned char)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
    }
    put32(z, (b << 16) | a);
    std::vector<unsigned char> png = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<unsigned char> ihdr;
    put32(ihdr, width); put32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0); // 8-bit RGBA
    chunk(png, "IHDR", ihdr);
    chunk(png, "IDAT", z);
    chunk(png, "IEND", {});
    FILE *f = std::fopen(path, "wb");
    if (!f) return 2;
    const bool ok = std::fwrite(png.data(), 1, png.size(), f) == png.size();
    std::fclose(f);
    return ok ? 0 : 3;
}

// PFM "PF": rows bottom to top, little-endian (scale -1.0)
int rsrt_write_pfm(const char *path, uint32_t width, uint32_t height, const float *rgb, uint32_t stride_floats)
{
    if (!path || !rgb || width == 0 || height == 0 || stride_floats < 3) return 1;
    FILE *f = std::fopen(path, "wb");
    if (!f) return 2;
    std::fprintf(f, "PF\n%u %u\n-1.0\n", width, height);
    std::vector<float> row((size_t)width * 3);
    for (uint32_t y = height; y-- > 0;) {
        for (uint32_t x = 0; x < width; x++)
            for (int c = 0; c < 3; c++) row[(size_t)x * 3 + c] = rgb[((size_t)y * width + x) * stride_floats + c];
        if (std::fwrite(row.data(), sizeof(float), row.size(), f) != row.size()) { std::fclose(f); return 3; }
    }
    std::fclose(f);
    return 0;
}

void rsrt_display_srgb8_host(const float *sum_rgba, size_t n_pixels, uint32_t sample_total, uint8_t *out_rgba8)
{
    for (size_t i = 0; i < n_pixels; i++) {
        unsigned char rgb[3];
        rsrt_display_pixel(sum_rgba + 4 * i, (float)sample_total, rgb);
        out_rgba8[4 * i] = rgb[0]; out_rgba8[4 * i + 1] = rgb[1]; out_rgba8[4 * i + 2] = rgb[2]; out_rgba8[4 * i + 3] = 255;
    }
}

void rsrt_camera_serialize(const rsrt_camera_desc *cam, char out[33])
{
    unsigned char raw[24];
    std::memcpy(raw, cam, 24); // pos.xyz, yaw, pitch, fov_y — little-endian f32, as bytemuck writes them
    for (int i = 0; i < 8; i++) {
        const uint32_t v = (raw[3 * i] << 16) | (raw[3 * i + 1] << 8) | raw[3 * i + 2];
        out[4 * i] = B64[v >> 18]; out[4 * i + 1] = B64[(v >> 12) & 63]; out[4 * i + 2] = B64[(v >> 6) & 63]; out[4 * i + 3] = B64[v & 63];
    }
    out[32] = 0;
}

int rsrt_camera_deserialize(const char *encoded, rsrt_camera_desc *out, char *err, size_t err_len)
{
    if (!encoded || !out) { set_err(err, err_len, "null argument"); return 1; }
    std::vector<unsigned char> raw;
    uint32_t acc = 0;
    int bits = 0;
    size_t n = std::strlen(encoded), pad = 0;
    if (n % 4 != 0) { set_err(err, err_len, "Invalid padding"); return 2; } // base64 STANDARD engine requires padding
    for (size_t i = 0; i < n; i++) {
        const char c = encoded[i];
        if (c == '=') { pad++; if (i + 2 < n) { set_err(err, err_len, "Invalid byte 61, offset " + std::to_string(i) + "."); return 2; } continue; }
        const char *q = std::strchr(B64, c);
        if (!q || pad) { set_err(err, err_len, "Invalid byte " + std::to_string((int)(unsigned char)c) + ", offset " + std::to_string(i) + "."); return 2; }
        acc = (acc << 6) | (uint32_t)(q - B64);
        bits += 6;
        if (bits >= 8) { bits -= 8; raw.push_back((unsigned char)((acc >> bits) & 0xff)); }
    }
    if (raw.size() != 24) { // src/camera.rs:55-60
        set_err(err, err_len, "Couldn't deserialize camera: binary data (" + std::to_string(raw.size()) + " bytes) not 24 bytes");
        return 3;
    }
    std::memcpy(out, raw.data(), 24);
    return 0;
}

} // extern "C"
