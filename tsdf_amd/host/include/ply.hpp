// ASCII PLY mesh writer.  Same function as the reference's src/include/ply.hpp.
#ifndef TSDF_AMD_HOST_PLY_INCLUDED
#define TSDF_AMD_HOST_PLY_INCLUDED
#include <string>
#include <vector>
#include "vector_types.h"

// one "x y z" line per vertex, one "3 a b c" line per triangle
void write_to_ply(const std::string &file_name,
                  const std::vector<float3> &vertices,
                  const std::vector<int3> &triangles);
#endif
