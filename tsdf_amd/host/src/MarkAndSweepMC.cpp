// Iso-surface extraction: marching cubes over the distance array, as the reference's extract_surface
// (src/MarchingCubes/MarkAndSweepMC.cu:506-555) -- same cube order, corner and edge numbering (:9-36, :80-97), sign
// classification (:110-124), edge interpolation (:47-63), triangle soup with winding (i, i+2, i+1) (:549).
//
// The 256-case triangle table is built at start-up the way the classic tables were derived (Lorensen & Cline's base
// configurations carried round the cube by its rotation group; P. Bourke, "Polygonising a scalar field", 1994, whose
// table the reference's MC_triangle_table.cu:87 holds): a configuration is turned by the cube's 24 rotations, tried in a
// fixed order, until it coincides with one of 30 base configurations, and that base's triangulation is turned back.
// The base triangulations and the try-order below reproduce the reference's TRIANGLE_TABLE entry for entry -- same
// triangles, same order, same first vertex (tools/mc_table_sha.py compares a SHA-256 of the 256 x 16 table with the
// reference's file where that is mounted; tests/golden/mc_tables.sha256.json carries the digest to the GPU box) -- so
// extract_surface emits the vertex array the reference's loop (MarkAndSweepMC.cu:285) emits.
#include "MarkAndSweepMC.hpp"

#include <cmath>
#include <cstdint>
#include <thread>

#include "host_common.hpp"

namespace {

// corner i of the cube rooted at voxel (x, y, z): offsets (dx, dy, dz)        (MarkAndSweepMC.cu:80-97)
const int kCorner[8][3] = {{0, 0, 1}, {1, 0, 1}, {1, 0, 0}, {0, 0, 0}, {0, 1, 1}, {1, 1, 1}, {1, 1, 0}, {0, 1, 0}};
// edge e joins corners kEdge[e][0] and kEdge[e][1], in the order the reference interpolates them (:291-302)
const int kEdge[12][2] = {{0, 1}, {2, 1}, {3, 2}, {3, 0}, {4, 5}, {6, 5}, {7, 6}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
constexpr int kTableWidth = 32;   // row stride of the table handed to the device kernel; the reference's rows hold 16 entries

struct Tables {
    int8_t tri[256][kTableWidth];   // edge numbers, three per triangle, -1 terminated
    uint8_t count[256];    // vertices emitted per configuration (the reference's VERTICES_FOR_CUBE_TYPE, MC_triangle_table.cu:46)
};

int edge_between(int a, int b) {
    for (int e = 0; e < 12; e++)
        if ((kEdge[e][0] == a && kEdge[e][1] == b) || (kEdge[e][0] == b && kEdge[e][1] == a)) return e;
    return -1;
}

// The 24 rotations of the cube as corner permutations (corner i goes to kRotation[r][i]), in the order they are tried.
const uint8_t kRotation[24][8] = {
    {0, 1, 2, 3, 4, 5, 6, 7}, {2, 1, 5, 6, 3, 0, 4, 7}, {5, 1, 0, 4, 6, 2, 3, 7}, {1, 0, 4, 5, 2, 3, 7, 6},
    {4, 0, 3, 7, 5, 1, 2, 6}, {2, 3, 0, 1, 6, 7, 4, 5}, {7, 3, 2, 6, 4, 0, 1, 5}, {3, 0, 1, 2, 7, 4, 5, 6},
    {0, 3, 7, 4, 1, 2, 6, 5}, {1, 2, 3, 0, 5, 6, 7, 4}, {3, 2, 6, 7, 0, 1, 5, 4}, {6, 2, 1, 5, 7, 3, 0, 4},
    {0, 4, 5, 1, 3, 7, 6, 2}, {5, 4, 7, 6, 1, 0, 3, 2}, {7, 4, 0, 3, 6, 5, 1, 2}, {1, 5, 6, 2, 0, 4, 7, 3},
    {4, 5, 1, 0, 7, 6, 2, 3}, {3, 7, 4, 0, 2, 6, 5, 1}, {6, 7, 3, 2, 5, 4, 0, 1}, {6, 5, 4, 7, 2, 1, 0, 3},
    {4, 7, 6, 5, 0, 3, 2, 1}, {2, 6, 7, 3, 1, 5, 4, 0}, {7, 6, 5, 4, 3, 2, 1, 0}, {5, 6, 2, 1, 4, 7, 3, 0}};

// Base configurations (bit i set = corner i negative) and their triangulations: one hex digit per triangle corner = the
// cube edge it lies on.
struct BaseCase {
    uint8_t config;
    const char *triangles;
};
const BaseCase kBase[] = {
    {0, ""}, {1, "083"}, {3, "183981"}, {5, "08312a"}, {7, "2832a8a98"},
    {15, "98aa8b"}, {20, "12a847"}, {21, "34730412a"}, {23, "2a9297273794"}, {27, "47b94b9b2921"},
    {31, "47b4b99ba"}, {37, "30812a495"}, {45, "4950818a18ba"}, {60, "958857a13a3b"}, {61, "5705097b010aba0"},
    {63, "ba57b5"}, {90, "01947823b5a6"}, {92, "8473b53515b6"}, {94, "059065036b63847"}, {95, "65969b4797b9"},
    {113, "0730a709a67a"}, {125, "091b67"}, {141, "a7617a187108"}, {142, "03707a0a96a7"}, {150, "4b846b0292a9"},
    {153, "042462"}, {165, "6b712a083495"}, {191, "a56"}, {232, "29a279237749"}, {255, ""}};

Tables build_tables() {
    Tables t;
    // edge permutation of every rotation: the edge joining corners (a, b) goes to the edge joining their images
    uint8_t edge_map[24][12], inverse[24];
    for (int r = 0; r < 24; r++)
        for (int e = 0; e < 12; e++) edge_map[r][e] = (uint8_t)edge_between(kRotation[r][kEdge[e][0]], kRotation[r][kEdge[e][1]]);
    for (int r = 0; r < 24; r++)
        for (int q = 0; q < 24; q++) {
            bool is_inverse = true;
            for (int i = 0; i < 8; i++) is_inverse = is_inverse && kRotation[q][kRotation[r][i]] == i;
            if (is_inverse) inverse[r] = (uint8_t)q;
        }
    int base_of[256];
    for (int c = 0; c < 256; c++) base_of[c] = -1;
    for (size_t b = 0; b < sizeof(kBase) / sizeof(kBase[0]); b++) base_of[kBase[b].config] = (int)b;
    for (int c = 0; c < 256; c++) {
        int n_out = 0;
        for (int r = 0; r < 24; r++) {
            int turned = 0;   // the configuration after rotation r
            for (int i = 0; i < 8; i++)
                if ((c >> i) & 1) turned |= 1 << kRotation[r][i];
            if (base_of[turned] < 0) continue;
            // the base's triangles, turned back
            for (const char *d = kBase[base_of[turned]].triangles; *d; d++) {
                const int e = *d <= '9' ? *d - '0' : *d - 'a' + 10;
                t.tri[c][n_out++] = (int8_t)edge_map[inverse[r]][e];
            }
            break;
        }
        t.count[c] = (uint8_t)n_out;
        for (int i = n_out; i < kTableWidth; i++) t.tri[c][i] = -1;
    }
    return t;
}

const Tables &tables() {
    static const Tables t = build_tables();
    return t;
}

// interpolate (MarkAndSweepMC.cu:47-63): the zero crossing between v0 (value w0) and v1 (value w1)
inline float3 interpolate(float3 v0, float3 v1, float w0, float w1) {
    if ((w0 > 0) && (w1 < 0)) {
        float tw = w0; w0 = w1; w1 = tw;
        float3 tv = v0; v0 = v1; v1 = tv;
    }
    const float3 delta = {v1.x - v0.x, v1.y - v0.y, v1.z - v0.z};
    const float ratio = -(w0) / (w1 - w0);
    return float3{(ratio * delta.x) + v0.x, (ratio * delta.y) + v0.y, (ratio * delta.z) + v0.z};
}

}  // namespace

// Cubes with z in [z_begin, z_end), in the reference's cube order (x fastest, then y, then z).
static void march_slab(const float *dist, unsigned X, unsigned Y, unsigned z_begin, unsigned z_end, const float vs[3],
                       const float offset[3], std::vector<float3> &vertices) {
    const Tables &t = tables();
    const size_t dy = X, dz = (size_t)X * Y;
    for (unsigned z = z_begin; z < z_end; z++)
        for (unsigned y = 0; y + 1 < Y; y++)
            for (unsigned x = 0; x + 1 < X; x++) {
                const size_t base = (size_t)x + y * dy + z * dz;
                float w[8];
                int type = 0;
                for (int i = 0; i < 8; i++) {
                    w[i] = dist[base + kCorner[i][0] + kCorner[i][1] * dy + kCorner[i][2] * dz];
                    type |= (w[i] < 0) << i;   // calculate_cube_type (:110-124)
                }
                if (t.count[type] == 0) continue;
                float3 v[8];
                for (int i = 0; i < 8; i++) {   // centre_of_voxel_at (src/TSDF/TSDF_utilities.cu:10-17)
                    v[i].x = ((int)(x + kCorner[i][0]) + 0.5f) * vs[0] + offset[0];
                    v[i].y = ((int)(y + kCorner[i][1]) + 0.5f) * vs[1] + offset[1];
                    v[i].z = ((int)(z + kCorner[i][2]) + 0.5f) * vs[2] + offset[2];
                }
                for (int i = 0; t.tri[type][i] != -1; i++) {
                    const int e = t.tri[type][i];
                    vertices.push_back(interpolate(v[kEdge[e][0]], v[kEdge[e][1]], w[kEdge[e][0]], w[kEdge[e][1]]));
                }
            }
}

// Marching cubes over a host distance array (x fastest).  Appends three vertices per triangle, cubes in the reference's
// order; large grids are cut into z ranges marched by host threads and concatenated in that order.
void tsdf_host_marching_cubes(const float *dist, unsigned X, unsigned Y, unsigned Z, const float vs[3], const float offset[3],
                              std::vector<float3> &vertices) {
    if (X < 2 || Y < 2 || Z < 2) return;
    (void)tables();   // build the table before any thread needs it
    const unsigned layers = Z - 1;
    unsigned n_threads = std::thread::hardware_concurrency();
    if (n_threads > 64) n_threads = 64;
    if ((size_t)X * Y * layers < ((size_t)1 << 22) || n_threads < 2) n_threads = 1;
    if (n_threads > layers) n_threads = layers;
    if (n_threads == 1) {
        march_slab(dist, X, Y, 0, layers, vs, offset, vertices);
        return;
    }
    std::vector<std::vector<float3>> parts(n_threads);
    std::vector<std::thread> workers;
    for (unsigned i = 0; i < n_threads; i++) {
        const unsigned z0 = (unsigned)((size_t)layers * i / n_threads), z1 = (unsigned)((size_t)layers * (i + 1) / n_threads);
        workers.emplace_back([=, &parts] { march_slab(dist, X, Y, z0, z1, vs, offset, parts[i]); });
    }
    for (auto &w : workers) w.join();
    size_t total = vertices.size();
    for (const auto &p : parts) total += p.size();
    vertices.reserve(total);
    for (const auto &p : parts) vertices.insert(vertices.end(), p.begin(), p.end());
}

// the table, for tests and for the device kernel: 256 x 32 edge numbers (the reference's 16 columns + padding)
const int8_t *tsdf_host_mc_triangle_table() { return &tables().tri[0][0]; }

void extract_surface(const TSDFVolume *volume, std::vector<float3> &vertices, std::vector<int3> &triangles) {
    vertices.clear();
    triangles.clear();
    // on the device (the reference extracts on the GPU too), with the table built above
    uint64_t n_vertices = 0;
    tsdf_host::check(tsdf_volume_marching_cubes(volume->handle(), tsdf_host_mc_triangle_table(), &n_vertices, nullptr, 0),
                     "Couldn't extract the surface");
    vertices.resize((size_t)n_vertices);
    if (n_vertices != 0)
        tsdf_host::check(tsdf_volume_marching_cubes(volume->handle(), tsdf_host_mc_triangle_table(), &n_vertices,
                                                    reinterpret_cast<float *>(vertices.data()), n_vertices),
                         "Couldn't extract the surface");

    // triangles are implicit, three consecutive vertices each, wired (i, i+2, i+1) like the reference (:549)
    for (size_t i = 0; i + 2 < vertices.size(); i += 3) triangles.push_back(int3{(int)i, (int)i + 2, (int)i + 1});
}
