// Depth map file readers.  Same functions as the reference's src/include/DepthMapUtilities.hpp.
#ifndef TSDF_AMD_HOST_DEPTH_MAP_UTILITIES_INCLUDED
#define TSDF_AMD_HOST_DEPTH_MAP_UTILITIES_INCLUDED

#include <cstdint>
#include <string>

// ".dmap": uint16 width, uint16 height, then width*height uint16 values
uint16_t *load_depth_map(std::string file_name, uint16_t &width, uint16_t &height);
// TUM: 16-bit PNG, 5000 units per metre -> mm.  NYU: 16-bit binary PGM (P5), byte-swapped.  new[]-allocated.
uint16_t *read_nyu_depth_map(const std::string &file_name, uint32_t &width, uint32_t &height);
uint16_t *read_tum_depth_map(const std::string &file_name, uint32_t &width, uint32_t &height);

#endif
