// DepthImage (reference: src/DataLoader/DepthImage.cpp).
#include "DepthImage.hpp"

#include <sys/stat.h>

#include <cstring>
#include <new>
#include <stdexcept>

#include "PngUtilities.hpp"

DepthImage::DepthImage(std::string file_name) : m_width{0}, m_height{0}, m_data{nullptr} {
    struct stat st;
    if (stat(file_name.c_str(), &st) != 0 || S_ISDIR(st.st_mode))
        throw std::invalid_argument("File not found or is directory " + file_name);
    uint32_t w = 0, h = 0;
    m_data = load_png_from_file(file_name, w, h);
    if (m_data == nullptr) throw std::invalid_argument("Problem reading depth image " + file_name);
    m_width = (uint16_t)w;
    m_height = (uint16_t)h;
}

DepthImage::DepthImage(const uint16_t width, const uint16_t height, const uint16_t *const data)
    : m_width{0}, m_height{0}, m_data{nullptr} {
    if (width > 0 && height > 0 && data != nullptr) {
        m_data = new uint16_t[(size_t)width * height];
        m_width = width;
        m_height = height;
        memcpy(m_data, data, (size_t)width * height * sizeof(uint16_t));
    } else {
        throw std::invalid_argument("width and height must be non-zero and data must not be null");
    }
}

DepthImage::~DepthImage() {
    delete[] m_data;
    m_data = nullptr;
}

// reference: :61-69 -- TUM PNGs hold depth*5000/m, so factor 0.2 gives millimetres
void DepthImage::scale_depth(const float factor) {
    const size_t n = (size_t)m_width * m_height;
    if (m_data)
        for (size_t i = 0; i < n; i++) m_data[i] = (uint16_t)((float)m_data[i] * factor);
}

void DepthImage::truncate_depth_to(const int mm) {
    const size_t n = (size_t)m_width * m_height;
    if (m_data)
        for (size_t i = 0; i < n; i++)
            if (m_data[i] > mm) m_data[i] = 0;
}

void DepthImage::min_max(uint16_t &min, uint16_t &max) {
    min = 0xFFFF;
    max = 0;
    const size_t n = (size_t)m_width * m_height;
    if (m_data)
        for (size_t i = 0; i < n; i++) {
            uint16_t v = m_data[i];
            if (v > max) max = v;
            if (v < min) min = v;
        }
}

uint16_t DepthImage::width() const { return m_width; }
uint16_t DepthImage::height() const { return m_height; }
const uint16_t *DepthImage::data() const { return m_data; }
