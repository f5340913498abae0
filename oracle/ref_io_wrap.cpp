// C entry points around two more of the reference's files that compile from where they lie: the mesh writer
// (src/Utilities/ply.cpp:6-30 -- needs only vector_types.h, which the CUDA toolkit headers of the image's triton package carry) and
// the PGM reader behind the NYU depth maps (src/Utilities/PgmUtilities.cpp:49-85, plain C++).  SURVEY 8 f4 (loaders / PLY glue).
// oracle/Makefile target "ref" compiles the reference's files; this file only wraps what they define.  Test infrastructure only:
// the host library's write_to_ply / read_nyu_depth_map are checked against these, byte for byte.
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>
#include "include/ply.hpp"
#include "include/PgmUtilities.hpp"

extern "C" {

void ref_write_to_ply(const char *file_name, const float *vertices, size_t n_vertices, const int *triangles, size_t n_triangles) {
    std::vector<float3> v(n_vertices);
    std::vector<int3> t(n_triangles);
    for (size_t i = 0; i < n_vertices; i++) { v[i].x = vertices[3 * i]; v[i].y = vertices[3 * i + 1]; v[i].z = vertices[3 * i + 2]; }
    for (size_t i = 0; i < n_triangles; i++) { t[i].x = triangles[3 * i]; t[i].y = triangles[3 * i + 1]; t[i].z = triangles[3 * i + 2]; }
    write_to_ply(file_name, v, t);
}

// returns the number of samples (width * height); copies them when `out` holds at least that many
size_t ref_read_pgm(const char *file_name, uint32_t *width, uint32_t *height, uint16_t *out, size_t capacity) {
    uint32_t w = 0, h = 0;
    uint16_t *data = read_pgm(file_name, w, h);
    const size_t n = (size_t)w * h;
    *width = w;
    *height = h;
    if (data && out && capacity >= n)
        for (size_t i = 0; i < n; i++) out[i] = data[i];
    delete[] data;
    return n;
}

}  // extern "C"
