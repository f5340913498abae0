// C entry point around the reference's own BilateralFilter class, so tests can call the
// real reference code (compiled from /root/reference/src/BilateralFilter.cpp where it lies;
// see oracle/Makefile target "ref").  Test infrastructure only.
//
// Only the 8-bit overload is exposed: the reference's 16-bit path reads its 256-entry
// similarity table out of bounds and stores one byte per pixel (src/BilateralFilter.cpp:59,99,109),
// i.e. it is undefined behaviour and cannot serve as an oracle.
#include <cstdint>
#include <cstddef>
#include "include/BilateralFilter.hpp"

extern "C" void ref_bilateral_u8(uint8_t *image, int width, int height, float sigma_colour, float sigma_space) {
    BilateralFilter f(sigma_colour, sigma_space);
    f.filter(image, width, height);
}
