// C entry points around more of the reference's files that compile from where they lie: the file helpers of the loaders
// (src/Utilities/FileUtilities.cpp, plain C++), the mesh writer
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
#include "include/FileUtilities.hpp"
#include <cstring>

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

// ---- FileUtilities on C strings; lists come back as one buffer, names / lines separated by '\n' ('\x1f' inside a line stays) ----
static size_t ref_copy_out(const std::string &s, char *out, size_t capacity) {
    if (out && capacity > s.size()) { memcpy(out, s.data(), s.size()); out[s.size()] = 0; }
    return s.size();
}
int ref_match_file_name(const char *prefix, int num_digits, const char *suffix, const char *extension, const char *test_string) {
    return match_file_name(prefix, num_digits, suffix, extension, test_string) ? 1 : 0;
}
// returns the call's own result in *ok and the length of the joined lines; every line is followed by '\x1e'
size_t ref_process_file_by_lines(const char *file_name, int *ok, char *out, size_t capacity) {
    std::string joined;
    *ok = process_file_by_lines(file_name, [&joined](const std::string &line) { joined += line; joined += '\x1e'; }) ? 1 : 0;
    return ref_copy_out(joined, out, capacity);
}
// *ok: the call's result; the text (preset to `preset`, which a call may leave untouched) in out
size_t ref_read_last_line(const char *file_name, const char *preset, int *ok, char *out, size_t capacity) {
    std::string text = preset;
    *ok = read_last_line(file_name, text) ? 1 : 0;
    return ref_copy_out(text, out, capacity);
}
// the names match_file_name(prefix, num_digits, suffix, extension, .) accepts, in the order the call returned them, each followed by '\x1e'
size_t ref_files_in_directory(const char *directory, const char *prefix, int num_digits, const char *suffix, const char *extension, char *out,
                              size_t capacity) {
    std::vector<std::string> files;
    const std::string p = prefix, s = suffix, e = extension;
    files_in_directory(directory, files, [&](const char *name) { return match_file_name(p, num_digits, s, e, name); });
    std::string joined;
    for (const std::string &f : files) { joined += f; joined += '\x1e'; }
    return ref_copy_out(joined, out, capacity);
}
// 0: no such file; 1: exists; *is_directory is preset by the caller (the call writes it for plain files and directories only)
int ref_file_exists(const char *file_name, int *is_directory) {
    bool d = *is_directory != 0;
    const bool e = file_exists(file_name, d);
    *is_directory = d ? 1 : 0;
    return e ? 1 : 0;
}

}  // extern "C"
