// Small PNG reader/writer on top of zlib: non-interlaced greyscale 8/16 bit and RGB 8 bit, which
// is everything the data path needs (TUM depth PNGs in, rendered scene/normal maps out).
// Functions and pixel conventions follow the reference's src/Utilities/PngUtilities.cpp
// (16-bit samples are big-endian in the file and returned as b1*256+b2, :57-66).
#include "PngUtilities.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <vector>

namespace {

const unsigned char kSig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};

uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

struct Image {
    uint32_t width = 0, height = 0;
    int bit_depth = 0, colour_type = 0;
    std::vector<unsigned char> rows;  // unfiltered scanlines, no filter bytes
    size_t stride = 0;
};

int paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return (pb <= pc) ? b : c;
}

bool read_png(const std::string &file_name, Image &img) {
    FILE *fp = std::fopen(file_name.c_str(), "rb");
    if (!fp) return false;
    std::vector<unsigned char> file;
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), fp)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(fp);
    if (file.size() < 8 || memcmp(file.data(), kSig, 8) != 0) return false;

    std::vector<unsigned char> idat;
    bool have_header = false;
    int interlace = 0;
    size_t pos = 8;
    while (pos + 12 <= file.size()) {
        uint32_t len = be32(&file[pos]);
        const unsigned char *type = &file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) return false;
        const unsigned char *data = &file[pos + 8];
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            img.width = be32(data);
            img.height = be32(data + 4);
            img.bit_depth = data[8];
            img.colour_type = data[9];
            interlace = data[12];
            have_header = true;
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_header || interlace != 0 || img.width == 0 || img.height == 0) return false;
    int channels = (img.colour_type == 0) ? 1 : (img.colour_type == 2) ? 3 : (img.colour_type == 4) ? 2 : (img.colour_type == 6) ? 4 : 0;
    if (channels == 0 || (img.bit_depth != 8 && img.bit_depth != 16)) return false;
    const size_t bpp = (size_t)channels * img.bit_depth / 8;
    img.stride = (size_t)img.width * bpp;
    std::vector<unsigned char> raw((img.stride + 1) * img.height);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) return false;

    img.rows.assign(img.stride * img.height, 0);
    for (uint32_t y = 0; y < img.height; y++) {
        const unsigned char *src = &raw[(img.stride + 1) * y];
        unsigned char *dst = &img.rows[img.stride * y];
        const unsigned char *up = y ? dst - img.stride : nullptr;
        const int filter = src[0];
        for (size_t i = 0; i < img.stride; i++) {
            int a = (i >= bpp) ? dst[i - bpp] : 0;
            int b = up ? up[i] : 0;
            int c = (up && i >= bpp) ? up[i - bpp] : 0;
            int x = src[i + 1];
            switch (filter) {
                case 0: break;
                case 1: x += a; break;
                case 2: x += b; break;
                case 3: x += (a + b) / 2; break;
                case 4: x += paeth(a, b, c); break;
                default: return false;
            }
            dst[i] = (unsigned char)x;
        }
    }
    return true;
}

void put_chunk(FILE *fp, const char *type, const unsigned char *data, uint32_t len) {
    unsigned char hdr[8] = {(unsigned char)(len >> 24), (unsigned char)(len >> 16), (unsigned char)(len >> 8), (unsigned char)len,
                            (unsigned char)type[0], (unsigned char)type[1], (unsigned char)type[2], (unsigned char)type[3]};
    fwrite(hdr, 1, 8, fp);
    if (len) fwrite(data, 1, len, fp);
    uLong crc = crc32(0L, Z_NULL, 0);
    crc = crc32(crc, hdr + 4, 4);
    if (len) crc = crc32(crc, data, len);
    unsigned char c[4] = {(unsigned char)(crc >> 24), (unsigned char)(crc >> 16), (unsigned char)(crc >> 8), (unsigned char)crc};
    fwrite(c, 1, 4, fp);
}

// rows: height scanlines of `stride` bytes already in PNG byte order
bool write_png(const std::string &file_name, uint32_t width, uint32_t height, int bit_depth, int colour_type,
               const unsigned char *rows, size_t stride) {
    std::vector<unsigned char> raw((stride + 1) * height);
    for (uint32_t y = 0; y < height; y++) {
        raw[(stride + 1) * y] = 0;  // filter: none
        memcpy(&raw[(stride + 1) * y + 1], rows + stride * y, stride);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<unsigned char> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    FILE *fp = std::fopen(file_name.c_str(), "wb");
    if (!fp) return false;
    fwrite(kSig, 1, 8, fp);
    unsigned char ihdr[13] = {(unsigned char)(width >> 24), (unsigned char)(width >> 16), (unsigned char)(width >> 8), (unsigned char)width,
                              (unsigned char)(height >> 24), (unsigned char)(height >> 16), (unsigned char)(height >> 8), (unsigned char)height,
                              (unsigned char)bit_depth, (unsigned char)colour_type, 0, 0, 0};
    put_chunk(fp, "IHDR", ihdr, 13);
    put_chunk(fp, "IDAT", comp.data(), (uint32_t)clen);
    put_chunk(fp, "IEND", nullptr, 0);
    bool ok = !ferror(fp);
    fclose(fp);
    return ok;
}

}  // namespace

uint16_t *load_png_from_file(const std::string file_name, uint32_t &width, uint32_t &height) {
    width = 0;
    height = 0;
    Image img;
    if (!read_png(file_name, img)) {
        std::cerr << "Problem reading file " << file_name << std::endl;
        return nullptr;
    }
    width = img.width;
    height = img.height;
    if (img.stride != 2 * (size_t)img.width) {  // reference: rowbytes must be 2*width (:54)
        std::cerr << "Expected 16bpp greyscale file" << std::endl;
        return nullptr;
    }
    uint16_t *pixels = new uint16_t[(size_t)width * height];
    for (size_t i = 0; i < (size_t)width * height; i++) pixels[i] = (uint16_t)(img.rows[2 * i] * 256 + img.rows[2 * i + 1]);
    return pixels;
}

uint8_t *load_colour_png_from_file(const std::string file_name, uint32_t &width, uint32_t &height) {
    width = 0;
    height = 0;
    Image img;
    if (!read_png(file_name, img)) {
        std::cerr << "Problem reading file " << file_name << std::endl;
        return nullptr;
    }
    width = img.width;
    height = img.height;
    if (img.stride != 3 * (size_t)img.width) {
        std::cerr << "Expected 24bpp colour file" << std::endl;
        return nullptr;
    }
    uint8_t *pixels = new uint8_t[img.rows.size()];
    memcpy(pixels, img.rows.data(), img.rows.size());
    return pixels;
}

bool save_png_to_file(const std::string file_name, uint32_t width, uint32_t height, const uint16_t *pixel_data) {
    std::vector<unsigned char> rows((size_t)width * height * 2);
    for (size_t i = 0; i < (size_t)width * height; i++) {
        rows[2 * i] = (unsigned char)(pixel_data[i] >> 8);
        rows[2 * i + 1] = (unsigned char)(pixel_data[i] & 0xFF);
    }
    return write_png(file_name, width, height, 16, 0, rows.data(), (size_t)width * 2);
}

bool save_png_to_file(const std::string file_name, uint32_t width, uint32_t height, const uint8_t *pixel_data) {
    return write_png(file_name, width, height, 8, 0, pixel_data, width);
}

bool save_colour_png_to_file(const std::string file_name, uint32_t width, uint32_t height, const uint8_t *pixel_data) {
    return write_png(file_name, width, height, 8, 2, pixel_data, (size_t)width * 3);
}
