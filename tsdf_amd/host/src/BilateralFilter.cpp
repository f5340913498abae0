// BilateralFilter class surface over the C ABI (reference: src/BilateralFilter.cpp).
#include "BilateralFilter.hpp"

#include "host_common.hpp"

using tsdf_host::check;

BilateralFilter::BilateralFilter(float sigma_colour, float sigma_space)
    : m_sigma_colour{sigma_colour}, m_sigma_space{sigma_space}, m_handle{nullptr} {
    check(tsdf_bilateral_create(sigma_colour, sigma_space, &m_handle), "Couldn't create bilateral filter");
}

BilateralFilter::~BilateralFilter() {
    if (m_handle) tsdf_bilateral_destroy(m_handle);
}

// In place, like the reference (it memcpy's its result over the const input, :116).
void BilateralFilter::filter(const uint8_t *const image, int width, int height) const {
    check(tsdf_bilateral_filter_u8(m_handle, const_cast<uint8_t *>(image), width, height), "Bilateral filter failed");
}

void BilateralFilter::filter(const uint16_t *const image, int width, int height) const {
    check(tsdf_bilateral_filter_u16(m_handle, const_cast<uint16_t *>(image), width, height), "Bilateral filter failed");
}
