// A 16-bit depth frame in host memory, the class surface src/Tools/kinfu.cpp and the loaders are written against
// (behaviour follows the reference's src/DataLoader/DepthImage.cpp: what is accepted, what is thrown, the arithmetic per pixel).
#include "DepthImage.hpp"

#include <algorithm>
#include <stdexcept>
#include <utility>

#include "FileUtilities.hpp"
#include "PngUtilities.hpp"

namespace {
inline size_t pixel_count(uint16_t w, uint16_t h) { return (size_t)w * h; }
}  // namespace

DepthImage::DepthImage(std::string file_name) : m_width{0}, m_height{0}, m_data{nullptr} {
    bool is_directory = false;
    if (!file_exists(file_name, is_directory) || is_directory) throw std::invalid_argument("File not found or is directory " + file_name);
    uint32_t w = 0, h = 0;
    uint16_t *pixels = load_png_from_file(file_name, w, h);
    if (!pixels) throw std::invalid_argument("Problem reading depth image " + file_name);
    m_data = pixels;
    m_width = (uint16_t)w;
    m_height = (uint16_t)h;
}

DepthImage::DepthImage(const uint16_t width, const uint16_t height, const uint16_t *const data) : m_width{0}, m_height{0}, m_data{nullptr} {
    if (width == 0 || height == 0 || data == nullptr) throw std::invalid_argument("width and height must be non-zero and data must not be null");
    m_data = new uint16_t[pixel_count(width, height)];
    std::copy(data, data + pixel_count(width, height), m_data);
    m_width = width;
    m_height = height;
}

DepthImage::~DepthImage() { delete[] m_data; }

// every pixel <- (uint16)(pixel * factor) in float (reference :61-69); TUM PNGs hold 5000 units per metre: 0.2 gives millimetres
void DepthImage::scale_depth(const float factor) {
    if (!m_data) return;
    std::transform(m_data, m_data + pixel_count(m_width, m_height), m_data, [factor](uint16_t v) { return (uint16_t)((float)v * factor); });
}

// pixels beyond `mm` become "no measurement" (reference :75-84)
void DepthImage::truncate_depth_to(const int mm) {
    if (!m_data) return;
    std::replace_if(m_data, m_data + pixel_count(m_width, m_height), [mm](uint16_t v) { return v > mm; }, (uint16_t)0);
}

// over all pixels, zeros included; (0xFFFF, 0) for an image without data (reference :89-100)
void DepthImage::min_max(uint16_t &min, uint16_t &max) {
    min = 0xFFFF;
    max = 0;
    if (!m_data || pixel_count(m_width, m_height) == 0) return;
    const std::pair<uint16_t *, uint16_t *> mm = std::minmax_element(m_data, m_data + pixel_count(m_width, m_height));
    min = *mm.first;
    max = *mm.second;
}

uint16_t DepthImage::width() const { return m_width; }
uint16_t DepthImage::height() const { return m_height; }
const uint16_t *DepthImage::data() const { return m_data; }
