// Host-side iso-surface extraction.  The reference triangulates each cube with the classic 256-case marching
// cubes tables (src/MarchingCubes/MC_triangle_table.cu); this implementation splits each cube into six
// tetrahedra around its main diagonal and triangulates those (16 cases, derived below, no tables), which
// yields the same zero level set of the same trilinear samples with a different -- denser -- triangulation.
// Like the reference (src/MarchingCubes/MarkAndSweepMC.cu:506-555) it emits a triangle soup: three fresh
// vertices per triangle, vertex i of triangle t at index 3t+i, and classifies corners by the sign of the
// distance alone (:110-124), unobserved voxels (distance = +truncation) included.
#include "MarkAndSweepMC.hpp"

#include <cmath>

#include "host_common.hpp"

namespace {

struct P {
    float x, y, z, d;
};

inline float3 lerp(const P &a, const P &b) {
    // zero crossing on the edge a-b (a.d and b.d have opposite signs)
    float t = a.d / (a.d - b.d);
    return float3{a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z)};
}

inline void emit(std::vector<float3> &V, std::vector<int3> &T, const float3 &a, const float3 &b, const float3 &c) {
    int base = (int)V.size();
    V.push_back(a);
    V.push_back(b);
    V.push_back(c);
    T.push_back(int3{base, base + 2, base + 1});  // winding as the reference's (i, i+2, i+1), :549
}

// one tetrahedron: separate the vertices with d < 0 from those with d >= 0
void tetra(const P &p0, const P &p1, const P &p2, const P &p3, std::vector<float3> &V, std::vector<int3> &T) {
    const P *p[4] = {&p0, &p1, &p2, &p3};
    int in[4], out[4], ni = 0, no = 0;
    for (int i = 0; i < 4; i++) {
        if (p[i]->d < 0) in[ni++] = i; else out[no++] = i;
    }
    if (ni == 0 || ni == 4) return;
    if (ni == 1 || ni == 3) {
        const int apex = (ni == 1) ? in[0] : out[0];
        const int *others = (ni == 1) ? out : in;
        emit(V, T, lerp(*p[apex], *p[others[0]]), lerp(*p[apex], *p[others[1]]), lerp(*p[apex], *p[others[2]]));
    } else {  // 2 + 2: a quad, split into two triangles
        float3 a = lerp(*p[in[0]], *p[out[0]]), b = lerp(*p[in[0]], *p[out[1]]);
        float3 c = lerp(*p[in[1]], *p[out[1]]), d = lerp(*p[in[1]], *p[out[0]]);
        emit(V, T, a, b, c);
        emit(V, T, a, c, d);
    }
}

}  // namespace

void extract_surface(const TSDFVolume *volume, std::vector<float3> &vertices, std::vector<int3> &triangles) {
    vertices.clear();
    triangles.clear();
    const TSDFVolume::UInt3 size = volume->size();
    const TSDFVolume::Float3 vs = volume->voxel_size();
    const TSDFVolume::Float3 off = volume->offset();
    const size_t n = (size_t)size.x * size.y * size.z;
    std::vector<float> dist(n);
    tsdf_host::check(tsdf_volume_get_distance_data(volume->handle(), dist.data()), "Couldn't read distance data");

    // cube corner offsets and the six tetrahedra sharing the diagonal corner 0 - corner 6
    static const int corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
    static const int tets[6][4] = {{0, 5, 1, 6}, {0, 1, 2, 6}, {0, 2, 3, 6}, {0, 3, 7, 6}, {0, 7, 4, 6}, {0, 4, 5, 6}};

    for (unsigned z = 0; z + 1 < size.z; z++)
        for (unsigned y = 0; y + 1 < size.y; y++)
            for (unsigned x = 0; x + 1 < size.x; x++) {
                P c[8];
                bool any_neg = false, any_pos = false;
                for (int i = 0; i < 8; i++) {
                    unsigned cx = x + corner[i][0], cy = y + corner[i][1], cz = z + corner[i][2];
                    size_t idx = (size_t)cx + (size_t)cy * size.x + (size_t)cz * size.x * size.y;
                    c[i].x = (cx + 0.5f) * vs.x + off.x;  // voxel centres carry the samples
                    c[i].y = (cy + 0.5f) * vs.y + off.y;
                    c[i].z = (cz + 0.5f) * vs.z + off.z;
                    c[i].d = dist[idx];
                    if (c[i].d < 0) any_neg = true; else any_pos = true;
                }
                if (!(any_neg && any_pos)) continue;
                for (int t = 0; t < 6; t++) tetra(c[tets[t][0]], c[tets[t][1]], c[tets[t][2]], c[tets[t][3]], vertices, triangles);
            }
}
