// C entry points of the host library for language bindings (the Python mirror in tsdf_amd/api.py):
// they expose the C++ Camera so that Python uses the very same K^-1 / pose^-1 / look_at arithmetic
// as C++ callers.  Matrices cross as column-major float arrays.
#include <cstring>

#include <cstdint>
#include <vector>

#include "Camera.hpp"
#include "BlockTSDFLoader.hpp"
#include "MarkAndSweepMC.hpp"
#include "TUMDataLoader.hpp"
#include "DepthMapUtilities.hpp"
#include "ply.hpp"
#include "FileUtilities.hpp"
#include <string>

const int8_t *tsdf_host_mc_triangle_table();

extern "C" {

typedef struct tsdf_camera tsdf_camera;

tsdf_camera *tsdf_camera_create(float fx, float fy, float cx, float cy) {
    return reinterpret_cast<tsdf_camera *>(new Camera(fx, fy, cx, cy));
}
void tsdf_camera_destroy(tsdf_camera *c) { delete reinterpret_cast<Camera *>(c); }

void tsdf_camera_get(const tsdf_camera *c, float k[9], float kinv[9], float pose[16], float inv_pose[16]) {
    const Camera *cam = reinterpret_cast<const Camera *>(c);
    const Eigen::Matrix3f mk = cam->k(), mki = cam->kinv();
    memcpy(k, mk.data(), 9 * sizeof(float));
    memcpy(kinv, mki.data(), 9 * sizeof(float));
    memcpy(pose, cam->pose().data(), 16 * sizeof(float));
    memcpy(inv_pose, cam->inverse_pose().data(), 16 * sizeof(float));
}
void tsdf_camera_set_pose(tsdf_camera *c, const float pose[16]) {
    Eigen::Matrix4f p;
    memcpy(p.data(), pose, 16 * sizeof(float));
    reinterpret_cast<Camera *>(c)->set_pose(p);
}
void tsdf_camera_set_pose_tum(tsdf_camera *c, const float vars[7]) {
    float v[7];
    memcpy(v, vars, sizeof(v));
    reinterpret_cast<Camera *>(c)->set_pose(v);
}
void tsdf_camera_move_to(tsdf_camera *c, float x, float y, float z) { reinterpret_cast<Camera *>(c)->move_to(x, y, z); }
void tsdf_camera_look_at(tsdf_camera *c, float x, float y, float z) { reinterpret_cast<Camera *>(c)->look_at(x, y, z); }
void tsdf_camera_world_to_camera(const tsdf_camera *c, const float w[3], float out[3]) {
    Eigen::Vector3f r = reinterpret_cast<const Camera *>(c)->world_to_camera(Eigen::Vector3f{w[0], w[1], w[2]});
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
}
void tsdf_camera_camera_to_world(const tsdf_camera *c, const float w[3], float out[3]) {
    Eigen::Vector3f r = reinterpret_cast<const Camera *>(c)->camera_to_world(Eigen::Vector3f{w[0], w[1], w[2]});
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
}
void tsdf_camera_world_to_pixel(const tsdf_camera *c, const float w[3], int out[2]) {
    Eigen::Vector2i r = reinterpret_cast<const Camera *>(c)->world_to_pixel(Eigen::Vector3f{w[0], w[1], w[2]});
    out[0] = r[0]; out[1] = r[1];
}
void tsdf_camera_pixel_to_image_plane(const tsdf_camera *c, uint16_t x, uint16_t y, float out[2]) {
    Eigen::Vector2f r = reinterpret_cast<const Camera *>(c)->pixel_to_image_plane(x, y);
    out[0] = r[0]; out[1] = r[1];
}
void tsdf_camera_image_plane_to_pixel(const tsdf_camera *c, const float p[2], int out[2]) {
    Eigen::Vector2i r = reinterpret_cast<const Camera *>(c)->image_plane_to_pixel(Eigen::Vector2f{p[0], p[1]});
    out[0] = r[0]; out[1] = r[1];
}


// Marching cubes on a host distance array (MarkAndSweepMC.cpp): returns the number of vertices (3 per triangle) and, when
// `out` is not null and holds at least `capacity` float3, writes them.
size_t tsdf_host_marching_cubes_c(const float *dist, unsigned X, unsigned Y, unsigned Z, const float vs[3], const float offset[3],
                                  float *out, size_t capacity) {
    std::vector<float3> v;
    tsdf_host_marching_cubes(dist, X, Y, Z, vs, offset, v);
    if (out && capacity >= v.size()) memcpy(out, v.data(), v.size() * sizeof(float3));
    return v.size();
}
// the generated 256 x 32 triangle table (edge numbers, -1 terminated rows)
void tsdf_host_mc_table(signed char out[256 * 32]) { memcpy(out, tsdf_host_mc_triangle_table(), 256 * 32); }

// BlockTSDFLoader::load_from_file for bindings: 1 = complete file.  Sizes are always reported; the arrays are copied when
// `capacity` (floats per array) suffices.
int tsdf_host_block_loader_parse(const char *file_name, unsigned size[3], float physical[3], float *distances, float *weights,
                                 size_t capacity) {
    BlockTSDFLoader loader;
    const bool ok = loader.load_from_file(file_name);
    size[0] = loader.size_x(); size[1] = loader.size_y(); size[2] = loader.size_z();
    for (int i = 0; i < 3; i++) physical[i] = loader.physical_size()[i];
    const size_t n = loader.distances().size();
    if (distances && weights && capacity >= n && n > 0) {
        memcpy(distances, loader.distances().data(), n * sizeof(float));
        memcpy(weights, loader.weights().data(), n * sizeof(float));
    }
    return ok ? 1 : 0;
}

// TUMDataLoader for bindings (bench.py --tum-dir: the frames and poses tools/kinfu_stream.cpp sees, through the same loader).
// open: nullptr when the directory does not have the layout.  next: 1 = a frame was returned (depth in millimetres, copied when
// capacity >= width * height; pose 4x4 column-major), 0 = exhausted.
typedef struct tsdf_tum_loader tsdf_tum_loader;
tsdf_tum_loader *tsdf_host_tum_open(const char *directory) {
    try {
        return reinterpret_cast<tsdf_tum_loader *>(new TUMDataLoader(directory));
    } catch (const std::exception &) {
        return nullptr;
    }
}
// 1: a frame; 0: the sequence is exhausted; -1: the next record was consumed but its frame is missing or unreadable (the
// reference's loader answers nullptr for both of the last two; a caller that stops at the first nullptr must know which)
int tsdf_host_tum_next(tsdf_tum_loader *l, uint16_t *depth, size_t capacity, unsigned size[2], float pose[16]) {
    Eigen::Matrix4f p;
    DepthImage *image = nullptr;
    TUMDataLoader *loader = reinterpret_cast<TUMDataLoader *>(l);
    if (loader->records_left() == 0) return 0;
    try {
        image = loader->next(p);
    } catch (const std::exception &) {
        return -1;
    }
    if (!image) return -1;
    size[0] = image->width();
    size[1] = image->height();
    const size_t n = (size_t)image->width() * image->height();
    if (depth && capacity >= n) memcpy(depth, image->data(), n * sizeof(uint16_t));
    memcpy(pose, p.data(), 16 * sizeof(float));
    delete image;
    return 1;
}
void tsdf_host_tum_close(tsdf_tum_loader *l) { delete reinterpret_cast<TUMDataLoader *>(l); }

static const char *or_empty(const char *s) { return s ? s : ""; }   // (a null string from a binding reads as the empty one)

// write_to_ply (ply.cpp) on flat arrays: 3 floats a vertex, 3 indices a triangle
void tsdf_host_write_ply(const char *file_name, const float *vertices, size_t n_vertices, const int *triangles, size_t n_triangles) {
    std::vector<float3> v(n_vertices);
    std::vector<int3> t(n_triangles);
    for (size_t i = 0; i < n_vertices; i++) v[i] = float3{vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]};
    for (size_t i = 0; i < n_triangles; i++) t[i] = int3{triangles[3 * i], triangles[3 * i + 1], triangles[3 * i + 2]};
    write_to_ply(or_empty(file_name), v, t);
}

// read_nyu_depth_map (DepthMapUtilities.cpp): returns width * height (0: the file did not parse) and copies the samples when
// `out` holds at least that many
size_t tsdf_host_read_nyu_depth_map(const char *file_name, unsigned size[2], uint16_t *out, size_t capacity) {
    uint32_t w = 0, h = 0;
    uint16_t *map = read_nyu_depth_map(or_empty(file_name), w, h);
    size[0] = w;
    size[1] = h;
    if (!map) return 0;
    const size_t n = (size_t)w * h;
    if (out && capacity >= n) memcpy(out, map, n * sizeof(uint16_t));
    delete[] map;
    return n;
}

// ---- FileUtilities on C strings; lists come back as one buffer, names / lines separated by '\n' ('\x1f' inside a line stays) ----
static size_t tsdf_host_copy_out(const std::string &s, char *out, size_t capacity) {
    if (out && capacity > s.size()) { memcpy(out, s.data(), s.size()); out[s.size()] = 0; }
    return s.size();
}
int tsdf_host_match_file_name(const char *prefix, int num_digits, const char *suffix, const char *extension, const char *test_string) {
    return match_file_name(or_empty(prefix), num_digits, or_empty(suffix), or_empty(extension), or_empty(test_string)) ? 1 : 0;
}
// returns the call's own result in *ok and the length of the joined lines; every line is followed by '\x1e'
size_t tsdf_host_process_file_by_lines(const char *file_name, int *ok, char *out, size_t capacity) {
    std::string joined;
    *ok = process_file_by_lines(or_empty(file_name), [&joined](const std::string &line) { joined += line; joined += '\x1e'; }) ? 1 : 0;
    return tsdf_host_copy_out(joined, out, capacity);
}
// *ok: the call's result; the text (preset to `preset`, which a call may leave untouched) in out
size_t tsdf_host_read_last_line(const char *file_name, const char *preset, int *ok, char *out, size_t capacity) {
    std::string text = or_empty(preset);
    *ok = read_last_line(or_empty(file_name), text) ? 1 : 0;
    return tsdf_host_copy_out(text, out, capacity);
}
// the names match_file_name(prefix, num_digits, suffix, extension, .) accepts, in the order the call returned them, each followed by '\x1e'
size_t tsdf_host_files_in_directory(const char *directory, const char *prefix, int num_digits, const char *suffix, const char *extension, char *out,
                              size_t capacity) {
    std::vector<std::string> files;
    const std::string p = or_empty(prefix), s = or_empty(suffix), e = or_empty(extension);
    files_in_directory(or_empty(directory), files, [&](const char *name) { return match_file_name(p, num_digits, s, e, name); });
    std::string joined;
    for (const std::string &f : files) { joined += f; joined += '\x1e'; }
    return tsdf_host_copy_out(joined, out, capacity);
}
// 0: no such file; 1: exists; *is_directory is preset by the caller (the call writes it for plain files and directories only)
int tsdf_host_file_exists(const char *file_name, int *is_directory) {
    bool d = *is_directory != 0;
    const bool e = file_exists(or_empty(file_name), d);
    *is_directory = d ? 1 : 0;
    return e ? 1 : 0;
}
}  // extern "C"
