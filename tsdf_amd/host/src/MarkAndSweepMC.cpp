// Iso-surface extraction: marching cubes over the distance array, as the reference's extract_surface
// (src/MarchingCubes/MarkAndSweepMC.cu:506-555) -- same cube order, corner and edge numbering (:9-36, :80-97), sign
// classification (:110-124), edge interpolation (:47-63), triangle soup with winding (i, i+2, i+1) (:549).
//
// The 256-case table is not taken from the reference's MC_triangle_table.cu: it is GENERATED here, once, from the cube's
// geometry.  For a sign configuration, on every face the points where the surface crosses the face's edges are joined
// pairwise (a face whose corners alternate in sign is resolved by cutting off its negative corners -- the rule depends on
// the face alone, so two cubes sharing a face agree and the mesh has no cracks), each segment directed so that the
// negative side lies to its left seen from outside (the orientation of the classic table: configuration 1, only corner
// 0 negative, comes out as edges 0, 8, 3); the segments chain into closed loops, and every loop is
// triangulated as a fan from its lowest-numbered edge.  For the unambiguous configurations this is the same surface patch
// as the classic table's, possibly fanned from another corner; the set of mesh vertices (one per sign-changing cube edge
// per triangle corner using it) lies on the same edges at the same interpolated positions.
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
// the six faces, corners in cyclic order
const int kFace[6][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 1, 5, 4}, {3, 2, 6, 7}, {0, 3, 7, 4}, {1, 2, 6, 5}};

constexpr int kTableWidth = 32;   // 12 crossing edges bound a configuration to 10 triangles; the generated table needs 5

struct Tables {
    int8_t tri[256][kTableWidth];   // edge numbers, three per triangle, -1 terminated
    uint8_t count[256];    // vertices emitted per configuration
};

int edge_between(int a, int b) {
    for (int e = 0; e < 12; e++)
        if ((kEdge[e][0] == a && kEdge[e][1] == b) || (kEdge[e][0] == b && kEdge[e][1] == a)) return e;
    return -1;
}

void edge_midpoint(int e, double m[3]) {
    for (int k = 0; k < 3; k++) m[k] = 0.5 * (kCorner[kEdge[e][0]][k] + kCorner[kEdge[e][1]][k]);
}

// Is corner q on the LEFT of the directed segment a -> b, seen from outside the face with outward normal n?
bool on_left(const double a[3], const double b[3], const double n[3], int q) {
    const double d[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    const double left[3] = {n[1] * d[2] - n[2] * d[1], n[2] * d[0] - n[0] * d[2], n[0] * d[1] - n[1] * d[0]};  // n x d
    const double r[3] = {kCorner[q][0] - a[0], kCorner[q][1] - a[1], kCorner[q][2] - a[2]};
    return left[0] * r[0] + left[1] * r[1] + left[2] * r[2] > 0.0;
}

Tables build_tables() {
    Tables t;
    for (int c = 0; c < 256; c++) {
        int next[12];
        for (int e = 0; e < 12; e++) next[e] = -1;
        auto negative = [c](int corner) { return ((c >> corner) & 1) != 0; };
        auto join = [&](int e_from, int e_to, const double n[3], int negative_corner) {
            double a[3], b[3];
            edge_midpoint(e_from, a);
            edge_midpoint(e_to, b);
            if (on_left(a, b, n, negative_corner)) next[e_from] = e_to; else next[e_to] = e_from;
        };
        for (int f = 0; f < 6; f++) {
            const int *q = kFace[f];
            // outward normal: from the cube centre to the face centre
            double n[3] = {0, 0, 0};
            for (int i = 0; i < 4; i++)
                for (int k = 0; k < 3; k++) n[k] += 0.25 * kCorner[q[i]][k];
            for (int k = 0; k < 3; k++) n[k] -= 0.5;
            int crossing[4], n_cross = 0;   // index i: the face edge q[i] - q[i+1] changes sign
            for (int i = 0; i < 4; i++)
                if (negative(q[i]) != negative(q[(i + 1) & 3])) crossing[n_cross++] = i;
            if (n_cross == 2) {
                // one segment; any negative corner of the face tells the side
                int neg = -1;
                for (int i = 0; i < 4; i++)
                    if (negative(q[i])) neg = q[i];
                join(edge_between(q[crossing[0]], q[(crossing[0] + 1) & 3]), edge_between(q[crossing[1]], q[(crossing[1] + 1) & 3]), n, neg);
            } else if (n_cross == 4) {
                // alternating corners: cut off each negative corner with its own segment
                for (int i = 0; i < 4; i++)
                    if (negative(q[i])) join(edge_between(q[(i + 3) & 3], q[i]), edge_between(q[i], q[(i + 1) & 3]), n, q[i]);
            }
        }
        // closed loops -> fans
        int n_out = 0;
        bool used[12] = {false};
        for (int e0 = 0; e0 < 12; e0++) {
            if (next[e0] < 0 || used[e0]) continue;
            int loop[12], len = 0;
            for (int e = e0; !used[e]; e = next[e]) {
                used[e] = true;
                loop[len++] = e;
            }
            for (int i = 1; i + 1 < len; i++) {
                t.tri[c][n_out++] = (int8_t)loop[0];
                t.tri[c][n_out++] = (int8_t)loop[i];
                t.tri[c][n_out++] = (int8_t)loop[i + 1];
            }
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

// the generated table, for tests: 256 x 32 edge numbers
const int8_t *tsdf_host_mc_triangle_table() { return &tables().tri[0][0]; }

void extract_surface(const TSDFVolume *volume, std::vector<float3> &vertices, std::vector<int3> &triangles) {
    vertices.clear();
    triangles.clear();
    // on the device (the reference extracts on the GPU too); the table is the one generated above
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
