// Volume lifecycle and data access behind the C ABI (include/tsdf_amd.h).
// Replaces the host side of src/TSDF/TSDFVolume.cu (ctors, set_size, clear, accessors) of the
// reference.  HBM layout: two dense fp32 arrays (distance, weight), x fastest, one contiguous
// range of planes per object; the 24-byte/voxel deformation grid stays implicit until a caller
// asks for it and the colour array, which no kernel of the path touches, is
// not allocated at all.
#include <cmath>
#include <cstring>
#include <new>

#include "common.hpp"

namespace tsdf {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? TSDF_ERR_NOMEM : TSDF_ERR_DEVICE;
}

// ---- fill kernels ------------------------------------------------------------------------
// clear() is a pure streaming write: 16 B per lane per store, grid-stride, 2048 blocks.
__global__ __launch_bounds__(256) void fill2_kernel(float *__restrict__ dist, float *__restrict__ weight,
                                                    size_t n, float dval, float wval) {
    size_t n4 = n >> 2;
    float4 d4 = make_float4(dval, dval, dval, dval);
    float4 w4 = make_float4(wval, wval, wval, wval);
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        reinterpret_cast<float4 *>(dist)[i] = d4;
        reinterpret_cast<float4 *>(weight)[i] = w4;
    }
    // tail (n not a multiple of 4)
    size_t t = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        dist[t] = dval;
        weight[t] = wval;
    }
}

// initialise_deformation (src/TSDF/TSDFVolume.cu:768-794) for the materialised node array.
__global__ __launch_bounds__(256) void init_nodes_kernel(tsdf_deformation_node *nodes, Geom g) {
    uint32_t vx = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t vy = blockIdx.y;
    uint32_t vz = g.z_store_begin + blockIdx.z;
    if (vx >= g.X) return;
    size_t idx = (size_t)g.X * g.Y * (vz - g.z_store_begin) + (size_t)g.X * vy + vx;
    tsdf_deformation_node nd;
    nd.translation[0] = (((int)vx + 0.5f) * g.vs.x) + g.offset.x;
    nd.translation[1] = (((int)vy + 0.5f) * g.vs.y) + g.offset.y;
    nd.translation[2] = (((int)vz + 0.5f) * g.vs.z) + g.offset.z;
    nd.rotation[0] = 0.0f;
    nd.rotation[1] = 0.0f;
    nd.rotation[2] = 0.0f;
    nodes[idx] = nd;
}

static int init_nodes(tsdf_volume *v) {
    dim3 block(256, 1, 1);
    dim3 grid((v->g.X + 255) / 256, v->g.Y, v->g.z_store_end - v->g.z_store_begin);
    hipLaunchKernelGGL(init_nodes_kernel, grid, block, 0, v->stream, v->nodes, v->g);
    TSDF_HIP(hipGetLastError(), "initialise deformation nodes");
    return TSDF_OK;
}

}  // namespace tsdf

using namespace tsdf;

extern "C" {

const char *tsdf_last_error(void) { return g_err; }

const char *tsdf_build_arch(void) { return "gfx950"; }

int tsdf_device_count(int *count) {
    TSDF_REQUIRE(count, "tsdf_device_count: null argument");
    TSDF_HIP(hipGetDeviceCount(count), "hipGetDeviceCount");
    return TSDF_OK;
}

int tsdf_set_device(int device) {
    TSDF_HIP(hipSetDevice(device), "hipSetDevice");
    return TSDF_OK;
}

int tsdf_get_device(int *device) {
    TSDF_REQUIRE(device, "tsdf_get_device: null argument");
    TSDF_HIP(hipGetDevice(device), "hipGetDevice");
    return TSDF_OK;
}

int tsdf_volume_create_slab(uint32_t sx, uint32_t sy, uint32_t sz, float px, float py, float pz,
                            uint32_t z_begin, uint32_t z_end, tsdf_volume **out) {
    TSDF_REQUIRE(out, "tsdf_volume_create: null out pointer");
    *out = nullptr;
    // src/TSDF/TSDFVolume.cu:431-436: all sizes and physical sizes must be > 0
    TSDF_REQUIRE(sx > 0 && sy > 0 && sz > 0 && px > 0 && py > 0 && pz > 0,
                 "Attempt to construct TSDFVolume with zero or negative size");
    // the reference's set_size takes uint16_t dimensions (src/include/TSDFVolume.hpp:112)
    TSDF_REQUIRE(sx <= 65535 && sy <= 65535 && sz <= 65535, "TSDFVolume dimensions must fit in 16 bits");
    TSDF_REQUIRE(z_begin < z_end && z_end <= sz, "invalid slab [%u,%u) of %u planes", z_begin, z_end, sz);

    tsdf_volume *v = new (std::nothrow) tsdf_volume();
    if (!v) {
        set_error("out of host memory");
        return TSDF_ERR_NOMEM;
    }
    std::memset(v, 0, sizeof(*v));
    Geom &g = v->g;
    g.X = sx; g.Y = sy; g.Z = sz;
    g.phys = {px, py, pz};
    // src/TSDF/TSDFVolume.cu:690 f3_div_elem(float3, dim3); :693 trunc = 1.1f * f3_norm(voxel_size)
    g.vs.x = px / (float)sx;
    g.vs.y = py / (float)sy;
    g.vs.z = pz / (float)sz;
    g.trunc = 1.1f * sqrtf(g.vs.x * g.vs.x + g.vs.y * g.vs.y + g.vs.z * g.vs.z);
    g.offset = {0.0f, 0.0f, 0.0f};
    g.offset_clear = {0.0f, 0.0f, 0.0f};
    v->z_begin = z_begin;
    v->z_end = z_end;
    g.z_store_begin = z_begin;
    g.z_store_end = (z_end < sz) ? z_end + 1 : sz;  // one halo plane above (trilinear reads lower.z+1)
    v->max_weight = 15.0f;                           // src/TSDF/TSDFVolume.cu:717
    v->stream = nullptr;

    hipError_t e = hipGetDevice(&v->device);
    size_t bytes = v->resident_voxels() * sizeof(float);
    if (e == hipSuccess) e = hipMalloc((void **)&v->dist, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&v->weight, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&v->counter_dev, 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(v->counter_dev, 0, 4 * sizeof(unsigned long long));
    if (e != hipSuccess) {
        int rc = hip_fail(e, "Couldn't allocate space for TSDF data");
        tsdf_volume_destroy(v);
        return rc;
    }
    int rc = tsdf_volume_clear(v);
    if (rc == TSDF_OK) rc = tsdf_volume_synchronize(v);
    if (rc != TSDF_OK) {
        tsdf_volume_destroy(v);
        return rc;
    }
    *out = v;
    return TSDF_OK;
}

int tsdf_volume_create(uint32_t sx, uint32_t sy, uint32_t sz, float px, float py, float pz, tsdf_volume **out) {
    TSDF_REQUIRE(sz > 0, "Attempt to construct TSDFVolume with zero or negative size");
    return tsdf_volume_create_slab(sx, sy, sz, px, py, pz, 0, sz, out);
}

int tsdf_volume_destroy(tsdf_volume *v) {
    if (!v) return TSDF_OK;
    if (v->dist) (void)hipFree(v->dist);
    if (v->weight) (void)hipFree(v->weight);
    if (v->nodes) (void)hipFree(v->nodes);
    if (v->depth_buf) (void)hipFree(v->depth_buf);
    if (v->vert_buf) (void)hipFree(v->vert_buf);
    if (v->norm_buf) (void)hipFree(v->norm_buf);
    if (v->counter_dev) (void)hipFree(v->counter_dev);
    delete v;
    return TSDF_OK;
}

int tsdf_volume_set_stream(tsdf_volume *v, void *hip_stream) {
    TSDF_REQUIRE(v, "null volume");
    v->stream = (hipStream_t)hip_stream;
    return TSDF_OK;
}

int tsdf_volume_synchronize(const tsdf_volume *v) {
    TSDF_REQUIRE(v, "null volume");
    TSDF_HIP(hipStreamSynchronize(v->stream), "stream synchronize");
    return TSDF_OK;
}

int tsdf_volume_clear(tsdf_volume *v) {
    TSDF_REQUIRE(v, "null volume");
    size_t n = v->resident_voxels();
    hipLaunchKernelGGL(fill2_kernel, dim3(2048), dim3(256), 0, v->stream, v->dist, v->weight, n, v->g.trunc, 0.0f);
    TSDF_HIP(hipGetLastError(), "Couldn't clear TSDF data");
    // initialise_deformation bakes the CURRENT offset into the node translations (Q1)
    v->g.offset_clear = v->g.offset;
    if (v->nodes) return init_nodes(v);
    return TSDF_OK;
}

int tsdf_volume_get_info(const tsdf_volume *v, tsdf_volume_info *info) {
    TSDF_REQUIRE(v && info, "null argument");
    const Geom &g = v->g;
    info->size[0] = g.X; info->size[1] = g.Y; info->size[2] = g.Z;
    info->z_begin = v->z_begin; info->z_end = v->z_end;
    info->z_store_begin = g.z_store_begin; info->z_store_end = g.z_store_end;
    const F3 *src[5] = {&g.phys, &g.vs, &g.offset, &g.offset_clear, nullptr};
    float *dst[4] = {info->physical_size, info->voxel_size, info->offset, info->offset_at_clear};
    for (int i = 0; i < 4; i++) {
        dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z;
    }
    info->truncation_distance = g.trunc;
    info->max_weight = v->max_weight;
    for (int i = 0; i < 3; i++) {
        info->global_translation[i] = v->global_translation[i];
        info->global_rotation[i] = v->global_rotation[i];
    }
    info->deformation_materialised = v->nodes ? 1 : 0;
    return TSDF_OK;
}

int tsdf_volume_set_offset(tsdf_volume *v, float ox, float oy, float oz) {
    TSDF_REQUIRE(v, "null volume");
    v->g.offset = {ox, oy, oz};
    return TSDF_OK;
}

int tsdf_volume_set_header(tsdf_volume *v, const float offset[3], float trunc, float max_weight,
                           const float gt[3], const float gr[3]) {
    TSDF_REQUIRE(v && offset && gt && gr, "null argument");
    v->g.offset = {offset[0], offset[1], offset[2]};
    v->g.trunc = trunc;
    v->max_weight = max_weight;
    for (int i = 0; i < 3; i++) {
        v->global_translation[i] = gt[i];
        v->global_rotation[i] = gr[i];
    }
    return TSDF_OK;
}

int tsdf_volume_distances(const tsdf_volume *v, float **p) {
    TSDF_REQUIRE(v && p, "null argument");
    *p = v->dist;
    return TSDF_OK;
}

int tsdf_volume_weights(const tsdf_volume *v, float **p) {
    TSDF_REQUIRE(v && p, "null argument");
    *p = v->weight;
    return TSDF_OK;
}

int tsdf_volume_deformation(tsdf_volume *v, tsdf_deformation_node **p) {
    TSDF_REQUIRE(v && p, "null argument");
    if (!v->nodes) {
        TSDF_HIP(hipMalloc((void **)&v->nodes, v->resident_voxels() * sizeof(tsdf_deformation_node)),
                 "Couldn't allocate space for deformation nodes for TSDF");
        // materialise what clear() would have written: centres + the offset at the time of clear()
        Geom saved = v->g;
        v->g.offset = v->g.offset_clear;
        int rc = init_nodes(v);
        v->g = saved;
        if (rc != TSDF_OK) return rc;
        TSDF_HIP(hipStreamSynchronize(v->stream), "initialise deformation nodes");
    }
    *p = v->nodes;
    return TSDF_OK;
}

static int copy_in(tsdf_volume *v, void *dst, const void *src, size_t bytes, const char *what) {
    TSDF_REQUIRE(v && src, "null argument");
    TSDF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, v->stream), what);
    TSDF_HIP(hipStreamSynchronize(v->stream), what);
    return TSDF_OK;
}

int tsdf_volume_set_distance_data(tsdf_volume *v, const float *host) {
    TSDF_REQUIRE(v, "null volume");
    return copy_in(v, v->dist, host, v->resident_voxels() * sizeof(float), "Couldn't set distance data");
}

int tsdf_volume_set_weight_data(tsdf_volume *v, const float *host) {
    TSDF_REQUIRE(v, "null volume");
    return copy_in(v, v->weight, host, v->resident_voxels() * sizeof(float), "Couldn't set weight data");
}

int tsdf_volume_set_deformation(tsdf_volume *v, const tsdf_deformation_node *host) {
    TSDF_REQUIRE(v && host, "null argument");
    tsdf_deformation_node *p;
    int rc = tsdf_volume_deformation(v, &p);
    if (rc != TSDF_OK) return rc;
    return copy_in(v, p, host, v->resident_voxels() * sizeof(tsdf_deformation_node), "Couldn't set deformation");
}

static int copy_out(const tsdf_volume *v, void *dst, const void *src, size_t bytes, const char *what) {
    TSDF_REQUIRE(v && dst, "null argument");
    TSDF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, v->stream), what);
    TSDF_HIP(hipStreamSynchronize(v->stream), what);
    return TSDF_OK;
}

int tsdf_volume_get_distance_data(const tsdf_volume *v, float *host) {
    TSDF_REQUIRE(v, "null volume");
    return copy_out(v, host, v->dist, v->resident_voxels() * sizeof(float), "Couldn't read distance data");
}

int tsdf_volume_get_weight_data(const tsdf_volume *v, float *host) {
    TSDF_REQUIRE(v, "null volume");
    return copy_out(v, host, v->weight, v->resident_voxels() * sizeof(float), "Couldn't read weight data");
}

int tsdf_volume_set_counting(tsdf_volume *v, int enabled) {
    TSDF_REQUIRE(v, "null volume");
    v->counting = enabled ? 1 : 0;
    return TSDF_OK;
}

int tsdf_volume_last_updated_voxels(const tsdf_volume *v, uint64_t *count) {
    TSDF_REQUIRE(v && count, "null argument");
    unsigned long long c = 0;
    TSDF_HIP(hipMemcpyAsync(&c, v->counter_dev, sizeof(c), hipMemcpyDeviceToHost, v->stream), "read counter");
    TSDF_HIP(hipStreamSynchronize(v->stream), "read counter");
    *count = c;
    return TSDF_OK;
}

}  // extern "C"
