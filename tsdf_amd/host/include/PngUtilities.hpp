// PNG file helpers (16-bit greyscale depth images, 8-bit greyscale, 8-bit RGB).
// Same functions as the reference's src/include/PngUtilities.hpp; implemented on zlib alone
// (libpng is not a build dependency here).
#ifndef TSDF_AMD_HOST_PNG_UTILITIES_INCLUDED
#define TSDF_AMD_HOST_PNG_UTILITIES_INCLUDED

#include <cstdint>
#include <iostream>
#include <string>

#include <Eigen/Dense>   // as the reference's header does (its tools rely on what this brings in)

// 16-bit greyscale -> width*height host-endian values (new[]-allocated, caller frees); nullptr on failure
uint16_t *load_png_from_file(const std::string file_name, uint32_t &width, uint32_t &height);
// 8-bit RGB -> width*height*3 bytes (new[]-allocated); nullptr on failure
uint8_t *load_colour_png_from_file(const std::string file_name, uint32_t &width, uint32_t &height);

bool save_png_to_file(const std::string file_name, uint32_t width, uint32_t height, const uint16_t *pixel_data);
bool save_png_to_file(const std::string file_name, uint32_t width, uint32_t height, const uint8_t *pixel_data);
bool save_colour_png_to_file(const std::string file_name, uint32_t width, uint32_t height, const uint8_t *pixel_data);

#endif /* TSDF_AMD_HOST_PNG_UTILITIES_INCLUDED */
