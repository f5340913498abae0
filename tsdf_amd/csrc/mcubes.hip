// Marching cubes over the volume's distance array on the device, for extract_surface
// (src/MarchingCubes/MarkAndSweepMC.cu:506-555 of the reference, which also runs it on the GPU, slab by slab).
//
// Same output as the host implementation (tsdf_amd/host/src/MarkAndSweepMC.cpp), vertex for vertex and bit for bit:
// cubes in the reference's order (x fastest, then y, then z), corner and edge numbering of MarkAndSweepMC.cu:9-36 /
// :80-97, sign classification (:110-124), interpolate() arithmetic (:47-63), three consecutive vertices per triangle.
// The 256-case table comes from the caller (the host library generates it, see there): one source for both.
//
// Two passes over the distances, both with one wave per row of cubes (fixed y, z; lanes along x, coalesced):
//   count: vertices per row -> exclusive scan over the rows -> emit: every lane writes its cube's vertices at the row's
//   offset + the prefix of the cubes before it in the row.
#include <cstring>
#include <vector>

#include "common.hpp"

namespace tsdf {

struct McTable {
    int8_t tri[256][32];   // edge numbers, three per triangle, -1 terminated
    uint8_t count[256];    // vertices per configuration
};

// corner i of the cube rooted at voxel (x, y, z): offsets (dx, dy, dz)        (MarkAndSweepMC.cu:80-97)
__constant__ int kMcCorner[8][3] = {{0, 0, 1}, {1, 0, 1}, {1, 0, 0}, {0, 0, 0}, {0, 1, 1}, {1, 1, 1}, {1, 1, 0}, {0, 1, 0}};
// edge e joins corners kMcEdge[e][0] and kMcEdge[e][1], in the order the reference interpolates them (:291-302)
__constant__ int kMcEdge[12][2] = {{0, 1}, {2, 1}, {3, 2}, {3, 0}, {4, 5}, {6, 5}, {7, 6}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};

// The 8 corner values of the cube at (x, y, z) and its configuration (bit i: corner i negative, :110-124).
__device__ inline int cube_type(const float *__restrict__ dist, size_t base, size_t dy, size_t dz, float w[8]) {
    w[0] = dist[base + dz];          w[1] = dist[base + 1 + dz];
    w[2] = dist[base + 1];           w[3] = dist[base];
    w[4] = dist[base + dy + dz];     w[5] = dist[base + 1 + dy + dz];
    w[6] = dist[base + 1 + dy];      w[7] = dist[base + dy];
    int type = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) type |= (w[i] < 0) << i;
    return type;
}

// EMIT = false: row_count[row] = vertices of the row.  EMIT = true: writes them at row_offset[row].
template <bool EMIT>
__global__ __launch_bounds__(256) void mc_rows_kernel(const float *__restrict__ dist, uint32_t X, uint32_t Y, uint32_t z_first, uint32_t n_layers,
                                                      uint32_t z_stored, F3 vs, F3 offset,
                                                      const McTable *__restrict__ table, uint32_t *__restrict__ row_count,
                                                      const uint64_t *__restrict__ row_offset, float *__restrict__ out) {
    __shared__ uint8_t count[256];
    count[threadIdx.x] = table->count[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, n_rows = (Y - 1) * n_layers;
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const uint32_t y = row % (Y - 1), z = z_first + row / (Y - 1);   // (z: the grid's plane; z - z_stored: where it is stored)
    const size_t dy = X, dz = (size_t)X * Y;
    uint64_t running = EMIT ? row_offset[row] : 0;   // vertices of the row so far (EMIT: absolute position)
    for (uint32_t x0 = 0; x0 + 1 < X; x0 += 64) {
        const uint32_t x = x0 + lane;
        const bool valid = x + 1 < X;
        float w[8];
        int type = 0;
        if (valid) type = cube_type(dist, (size_t)x + y * dy + (z - z_stored) * dz, dy, dz, w);
        const uint32_t n = valid ? count[type] : 0u;
        uint32_t incl = n;   // inclusive prefix over the wave
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if ((int)lane >= o) incl += up;
        }
        if (EMIT && n != 0) {
            float3 v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {   // centre_of_voxel_at (src/TSDF/TSDF_utilities.cu:10-17)
                v[i].x = ((int)(x + kMcCorner[i][0]) + 0.5f) * vs.x + offset.x;
                v[i].y = ((int)(y + kMcCorner[i][1]) + 0.5f) * vs.y + offset.y;
                v[i].z = ((int)(z + kMcCorner[i][2]) + 0.5f) * vs.z + offset.z;
            }
            float *dst = out + (running + (incl - n)) * 3;
            for (uint32_t i = 0; i < n; i++) {
                const int e = table->tri[type][i];
                const int a = kMcEdge[e][0], b = kMcEdge[e][1];
                // interpolate (MarkAndSweepMC.cu:47-63): the zero crossing between v0 (value w0) and v1 (value w1)
                float3 v0 = v[a], v1 = v[b];
                float w0 = w[a], w1 = w[b];
                if ((w0 > 0) && (w1 < 0)) {
                    const float tw = w0; w0 = w1; w1 = tw;
                    const float3 tv = v0; v0 = v1; v1 = tv;
                }
                const float ratio = -(w0) / (w1 - w0);
                dst[i * 3 + 0] = (ratio * (v1.x - v0.x)) + v0.x;
                dst[i * 3 + 1] = (ratio * (v1.y - v0.y)) + v0.y;
                dst[i * 3 + 2] = (ratio * (v1.z - v0.z)) + v0.z;
            }
        }
        running += __shfl(incl, 63);
    }
    if (!EMIT && lane == 0) row_count[row] = (uint32_t)running;
}

// Exclusive scan of the row counts (one workgroup; a few hundred thousand rows): row_offset[i] = sum of count[0..i),
// row_offset[n] = total.
__global__ __launch_bounds__(1024) void mc_scan_kernel(const uint32_t *__restrict__ count, uint32_t n, uint64_t *__restrict__ offset) {
    __shared__ uint64_t wave_sum[16];
    __shared__ uint64_t carry;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < n ? count[i] : 0;
        uint64_t incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t up = __shfl_up(incl, o);
            if ((int)lane >= o) incl += up;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        uint64_t before = carry;
        for (uint32_t w = 0; w < wave; w++) before += wave_sum[w];
        if (i < n) offset[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) offset[n] = carry;
}

}  // namespace tsdf

using namespace tsdf;

extern "C" int tsdf_volume_marching_cubes(const tsdf_volume *cv, const int8_t *table, uint64_t *n_vertices, float *host_vertices,
                                          uint64_t capacity) {
    TSDF_REQUIRE(cv && table && n_vertices, "tsdf_volume_marching_cubes: null argument");
    tsdf_volume *v = const_cast<tsdf_volume *>(cv);
    const Geom &g = v->g;
    *n_vertices = 0;
    if (g.X < 2 || g.Y < 2 || g.Z < 2) return TSDF_OK;
    // a Z-slab marches the cube layers rooted in the planes it owns (the layer above its last plane reads the halo plane
    // it stores): concatenated in slab order the slabs' vertices are the whole volume's
    const uint32_t z_first = v->z_begin, z_last = v->z_end < g.Z - 1 ? v->z_end : g.Z - 1;   // cube layers [z_first, z_last)
    if (z_last <= z_first) return TSDF_OK;
    const uint32_t n_layers = z_last - z_first;
    McTable t;
    memset(&t, 0, sizeof(t));
    for (int c = 0; c < 256; c++) {
        int n = 0;
        while (n < 32 && table[c * 32 + n] >= 0) {
            TSDF_REQUIRE(table[c * 32 + n] < 12, "tsdf_volume_marching_cubes: bad edge number in the table");
            t.tri[c][n] = table[c * 32 + n];
            n++;
        }
        TSDF_REQUIRE(n % 3 == 0, "tsdf_volume_marching_cubes: a configuration's vertices are not whole triangles");
        for (int i = n; i < 32; i++) t.tri[c][i] = -1;
        t.count[c] = (uint8_t)n;
    }
    const uint32_t n_rows = (g.Y - 1) * n_layers;
    McTable *d_table = nullptr;
    uint32_t *d_count = nullptr;
    uint64_t *d_offset = nullptr;
    float *d_out = nullptr;
    int rc = TSDF_OK;
    hipError_t e = hipMalloc((void **)&d_table, sizeof(McTable));
    if (e == hipSuccess) e = hipMalloc((void **)&d_count, (size_t)n_rows * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&d_offset, ((size_t)n_rows + 1) * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemcpyAsync(d_table, &t, sizeof(t), hipMemcpyHostToDevice, v->stream);
    uint64_t total = 0;
    if (e == hipSuccess) {
        const dim3 grid((n_rows + 3) / 4);
        hipLaunchKernelGGL((mc_rows_kernel<false>), grid, dim3(256), 0, v->stream, v->dist, g.X, g.Y, z_first, n_layers, g.z_store_begin, g.vs, g.offset, d_table, d_count,
                           (const uint64_t *)nullptr, (float *)nullptr);
        hipLaunchKernelGGL(mc_scan_kernel, dim3(1), dim3(1024), 0, v->stream, d_count, n_rows, d_offset);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&total, d_offset + n_rows, sizeof(total), hipMemcpyDeviceToHost, v->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(v->stream);
        *n_vertices = total;
        if (e == hipSuccess && host_vertices && total != 0) {
            if (capacity < total) {
                set_error("tsdf_volume_marching_cubes: %llu vertices do not fit the buffer of %llu", (unsigned long long)total,
                          (unsigned long long)capacity);
                rc = TSDF_ERR_INVALID;
            } else {
                e = hipMalloc((void **)&d_out, total * 3 * sizeof(float));
                if (e == hipSuccess) {
                    hipLaunchKernelGGL((mc_rows_kernel<true>), grid, dim3(256), 0, v->stream, v->dist, g.X, g.Y, z_first, n_layers, g.z_store_begin, g.vs, g.offset, d_table,
                                       (uint32_t *)nullptr, d_offset, d_out);
                    e = hipGetLastError();
                }
                if (e == hipSuccess) e = hipMemcpyAsync(host_vertices, d_out, total * 3 * sizeof(float), hipMemcpyDeviceToHost, v->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(v->stream);
            }
        }
    }
    if (d_out) (void)hipFree(d_out);
    if (d_offset) (void)hipFree(d_offset);
    if (d_count) (void)hipFree(d_count);
    if (d_table) (void)hipFree(d_table);
    if (e != hipSuccess) return hip_fail(e, "marching cubes failed");
    return rc;
}
