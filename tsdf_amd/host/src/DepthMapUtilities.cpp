// reference: src/Utilities/DepthMapUtilities.cpp (+ the .dmap layout of src/Tests/TestTSDF/TestHelpers.cpp:216-225)
#include "DepthMapUtilities.hpp"

#include <cstdio>
#include <fstream>
#include <vector>

#include "PngUtilities.hpp"

uint16_t *read_tum_depth_map(const std::string &file_name, uint32_t &width, uint32_t &height) {
    uint16_t *range_map = load_png_from_file(file_name, width, height);
    if (!range_map) return nullptr;
    const size_t n = (size_t)width * height;
    for (size_t i = 0; i < n; i++) range_map[i] = range_map[i] / 5;
    return range_map;
}

// 16-bit binary PGM ("P5 w h maxval" then big-endian samples); the reference then swaps the two bytes of
// every sample (:29-31), which is reproduced here on top of a host-endian read.
uint16_t *read_nyu_depth_map(const std::string &file_name, uint32_t &width, uint32_t &height) {
    width = height = 0;
    FILE *fp = std::fopen(file_name.c_str(), "rb");
    if (!fp) return nullptr;
    char magic[3] = {0, 0, 0};
    unsigned w = 0, h = 0, maxval = 0;
    if (fscanf(fp, "%2s %u %u %u", magic, &w, &h, &maxval) != 4 || magic[0] != 'P' || magic[1] != '5') {
        fclose(fp);
        return nullptr;
    }
    fgetc(fp);  // the single whitespace after maxval
    const size_t n = (size_t)w * h;
    const size_t bytes = maxval < 256 ? 1 : 2;   // (one byte a sample below 256: src/Utilities/PgmUtilities.cpp:70-75)
    std::vector<unsigned char> raw(n * bytes);
    if (fread(raw.data(), 1, raw.size(), fp) != raw.size()) {
        fclose(fp);
        return nullptr;
    }
    fclose(fp);
    uint16_t *range_map = new uint16_t[n];
    for (size_t i = 0; i < n; i++) {
        uint16_t v = bytes == 1 ? (uint16_t)raw[i] : (uint16_t)(raw[2 * i] * 256 + raw[2 * i + 1]);
        range_map[i] = (uint16_t)((v >> 8) + ((v & 0xFF) * 256));
    }
    width = w;
    height = h;
    return range_map;
}

uint16_t *load_depth_map(std::string file_name, uint16_t &width, uint16_t &height) {
    width = height = 0;
    std::ifstream f{file_name, std::ios::in | std::ios::binary};
    if (!f.good()) return nullptr;
    f.read(reinterpret_cast<char *>(&width), sizeof(width));
    f.read(reinterpret_cast<char *>(&height), sizeof(height));
    if (!f.good() || width == 0 || height == 0) return nullptr;
    uint16_t *pixels = new uint16_t[(size_t)width * height];
    f.read(reinterpret_cast<char *>(pixels), (std::streamsize)width * height * sizeof(uint16_t));
    if (!f.good()) {
        delete[] pixels;
        return nullptr;
    }
    return pixels;
}
