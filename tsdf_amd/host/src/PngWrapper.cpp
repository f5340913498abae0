// PngWrapper (reference: src/Utilities/PngWrapper.cpp).
#include "PngWrapper.hpp"

#include <cstring>
#include <stdexcept>

#include "PngUtilities.hpp"

PngWrapper::PngWrapper(const std::string &file_name, PNG_TYPE type) : m_width{0}, m_height{0}, m_data{nullptr}, m_type{type} {
    switch (type) {
        case COLOUR:
            m_data = load_colour_png_from_file(file_name, m_width, m_height);
            break;
        case GREYSCALE_16:
            m_data = reinterpret_cast<const uint8_t *>(load_png_from_file(file_name, m_width, m_height));
            break;
        case GREYSCALE_8:
            break;  // not loadable in the reference either
    }
    if (!m_data) throw std::invalid_argument("Failed to create PNGWrapper");
}

PngWrapper::PngWrapper(const uint16_t width, const uint16_t height, const uint8_t *data, PNG_TYPE type)
    : m_width{width}, m_height{height}, m_data{nullptr}, m_type{type} {
    uint64_t sz = (uint64_t)width * height;
    if (type == GREYSCALE_16) sz *= 2;
    if (type == COLOUR) sz *= 3;
    uint8_t *copy = new uint8_t[sz];
    memcpy(copy, data, sz);
    m_data = copy;
}

PngWrapper::~PngWrapper() {
    delete[] m_data;
    m_data = nullptr;
    m_width = 0;
    m_height = 0;
}

bool PngWrapper::save_to(const std::string &file_name) const {
    switch (m_type) {
        case COLOUR:
            return save_colour_png_to_file(file_name, m_width, m_height, m_data);
        case GREYSCALE_8:
            return save_png_to_file(file_name, m_width, m_height, m_data);
        case GREYSCALE_16:
            return save_png_to_file(file_name, m_width, m_height, reinterpret_cast<const uint16_t *>(m_data));
    }
    return false;
}
