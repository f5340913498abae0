// TSDF ray casting + normals for gfx950 (wave64).  Replaces GPURaycaster::raycast, get_vertices /
// process_ray and compute_normals of the reference (src/RayCaster/GPURaycaster.cu:14-547).
//
// One lane per pixel; a wave is an 8x8 pixel tile (coherent rays share cache lines of the
// distance array), a 256-thread workgroup is a 16x16 tile, 1200 workgroups at 640x480.  The
// march is the reference's, step for step, including its quirks (direction not normalised: Q6,
// previous_tsdf == trunc: Q7, 4402-sample cap: Q8, t accumulated by repeated float adds: Q9,
// unclamped point in the interpolation weights: Q10); fp contraction is off so every sample is
// bit-identical.  Lanes leave the loop individually; the wave iterates while any lane is
// still marching (ballot), so a tile costs as much as its longest ray.
#include <cmath>

#include "common.hpp"

namespace tsdf {

struct RayParams {
    F3 origin;
    Mat33 rot;
    Mat33 kinv;
    F3 space_min;
    F3 space_max;
    uint32_t width, height;
    uint32_t own_lo, own_hi;  // slab ownership (planes of the lower trilinear tap)
};

// tsdf_value_at (src/TSDF/TSDF_utilities.cu:29-37): uint16_t coordinates clamped to the grid.
template <bool STATS>
__device__ inline float tsdf_value_at(int xi, int yi, int zi, const float *__restrict__ dist, const Geom &g,
                                      unsigned int *__restrict__ touched) {
    uint32_t x = (uint16_t)xi, y = (uint16_t)yi, z = (uint16_t)zi;
    x = min(x, g.X - 1);
    y = min(y, g.Y - 1);
    z = min(z, g.Z - 1);
    if (STATS) {
        size_t gi = (size_t)g.X * g.Y * z + (size_t)g.X * y + x;
        atomicOr(&touched[gi >> 5], 1u << (gi & 31));
    }
    size_t idx = (size_t)g.X * g.Y * (z - g.z_store_begin) + (size_t)g.X * y + x;
    return dist[idx];
}

// trilinearly_interpolate (src/RayCaster/GPURaycaster.cu:53-124).  For SLAB, samples whose lower
// tap plane is not owned are not evaluated (owned=false, result NaN).
template <bool SLAB, bool STATS>
__device__ inline float trilinear(float px, float py, float pz, const float *__restrict__ dist, const Geom &g,
                                  const RayParams &rp, bool &owned, unsigned int *__restrict__ touched) {
    float max_x = g.X * g.vs.x, max_y = g.Y * g.vs.y, max_z = g.Z * g.vs.z;
    float ax = px, ay = py, az = pz;
    if (px >= max_x) ax = max_x - (g.vs.x / 10.0f);
    if (py >= max_y) ay = max_y - (g.vs.y / 10.0f);
    if (pz >= max_z) az = max_z - (g.vs.z / 10.0f);
    if (px < 0.0f) ax = 0.0f;
    if (py < 0.0f) ay = 0.0f;
    if (pz < 0.0f) az = 0.0f;

    // voxel_for_point (src/TSDF/TSDF_utilities.cu:45-53)
    int vx = f2i_sat(floorf(ax / g.vs.x));
    int vy = f2i_sat(floorf(ay / g.vs.y));
    int vz = f2i_sat(floorf(az / g.vs.z));

    owned = true;
    if (vx < 0 || vy < 0 || vz < 0 || (uint32_t)vx >= g.X || (uint32_t)vy >= g.Y || (uint32_t)vz >= g.Z) {
        return NAN;  // the reference also printf's here (:78)
    }

    // centre_of_voxel_at with its default zero offset (src/TSDF/TSDF_utilities.cu:10-17)
    float ccx = (vx + 0.5f) * g.vs.x + 0.0f;
    float ccy = (vy + 0.5f) * g.vs.y + 0.0f;
    float ccz = (vz + 0.5f) * g.vs.z + 0.0f;

    int lx = (px < ccx) ? vx - 1 : vx;
    int ly = (py < ccy) ? vy - 1 : vy;
    int lz = (pz < ccz) ? vz - 1 : vz;
    lx = max(lx, 0);
    ly = max(ly, 0);
    lz = max(lz, 0);

    if (SLAB) {
        if (!((uint32_t)lz >= rp.own_lo && (uint32_t)lz < rp.own_hi)) {
            owned = false;
            return NAN;
        }
    }

    float lcx = (lx + 0.5f) * g.vs.x + 0.0f;
    float lcy = (ly + 0.5f) * g.vs.y + 0.0f;
    float lcz = (lz + 0.5f) * g.vs.z + 0.0f;
    float u = (px - lcx) / g.vs.x;
    float v = (py - lcy) / g.vs.y;
    float w = (pz - lcz) / g.vs.z;

    float c000 = tsdf_value_at<STATS>(lx + 0, ly + 0, lz + 0, dist, g, touched);
    float c001 = tsdf_value_at<STATS>(lx + 0, ly + 0, lz + 1, dist, g, touched);
    float c010 = tsdf_value_at<STATS>(lx + 0, ly + 1, lz + 0, dist, g, touched);
    float c011 = tsdf_value_at<STATS>(lx + 0, ly + 1, lz + 1, dist, g, touched);
    float c100 = tsdf_value_at<STATS>(lx + 1, ly + 0, lz + 0, dist, g, touched);
    float c101 = tsdf_value_at<STATS>(lx + 1, ly + 0, lz + 1, dist, g, touched);
    float c110 = tsdf_value_at<STATS>(lx + 1, ly + 1, lz + 0, dist, g, touched);
    float c111 = tsdf_value_at<STATS>(lx + 1, ly + 1, lz + 1, dist, g, touched);

    float interpolated = c000 * (1 - u) * (1 - v) * (1 - w) +
                         c001 * (1 - u) * (1 - v) * w +
                         c010 * (1 - u) * v * (1 - w) +
                         c011 * (1 - u) * v * w +
                         c100 * u * (1 - v) * (1 - w) +
                         c101 * u * (1 - v) * w +
                         c110 * u * v * (1 - w) +
                         c111 * u * v * w;
    return interpolated;
}

// can_intersect_in_dimension (src/RayCaster/GPURaycaster.cu:138-181)
__device__ inline bool can_intersect_in_dimension(float space_min, float space_max, float origin, float direction,
                                                  float &near_t, float &far_t) {
    bool can_intersect = true;
    if (direction == 0) {
        if (origin < space_min || origin > space_max) can_intersect = false;
    } else {
        float dmin = (space_min - origin) / direction;
        float dmax = (space_max - origin) / direction;
        if (dmin > dmax) {
            float t = dmin;
            dmin = dmax;
            dmax = t;
        }
        if (dmin > near_t) near_t = dmin;
        if (dmax < far_t) far_t = dmax;
        if (near_t > far_t) can_intersect = false;
        else if (far_t < 0) can_intersect = false;
    }
    return can_intersect;
}

// compute_near_and_far_t (src/RayCaster/GPURaycaster.cu:197-251)
__device__ inline bool compute_near_and_far_t(const F3 &o, const F3 &d, const F3 &smin, const F3 &smax,
                                              float &near_t, float &far_t) {
    bool intersects = false;
    if (o.x >= smin.x && o.x <= smax.x && o.y >= smin.y && o.y <= smax.y && o.z >= smin.z && o.z <= smax.z) {
        near_t = 0;
        float x_t = NAN, y_t = NAN, z_t = NAN;
        if (d.x > 0) x_t = (smax.x - o.x) / d.x; else if (d.x < 0) x_t = (smin.x - o.x) / d.x;
        if (d.y > 0) y_t = (smax.y - o.y) / d.y; else if (d.y < 0) y_t = (smin.y - o.y) / d.y;
        if (d.z > 0) z_t = (smax.z - o.z) / d.z; else if (d.z < 0) z_t = (smin.z - o.z) / d.z;
        if (x_t < y_t) {
            if (x_t < z_t) far_t = x_t; else far_t = z_t;
        } else {
            if (y_t < z_t) far_t = y_t; else far_t = z_t;
        }
        intersects = true;
    } else {
        near_t = -INFINITY;
        far_t = INFINITY;
        if (can_intersect_in_dimension(smin.x, smax.x, o.x, d.x, near_t, far_t) &&
            can_intersect_in_dimension(smin.y, smax.y, o.y, d.y, near_t, far_t) &&
            can_intersect_in_dimension(smin.z, smax.z, o.z, d.z, near_t, far_t)) {
            intersects = true;
        }
    }
    return intersects;
}

// process_ray (src/RayCaster/GPURaycaster.cu:265-377).
//   SLAB=false: out = packed float3 vertices.  SLAB=true: out = float4 records {k, x, y, z}.
//   STATS: counters[1] += samples, counters[2] += hits, touched bitmap marked per tap.
template <bool SLAB, bool STATS>
__global__ __launch_bounds__(256) void process_ray_kernel(const float *__restrict__ dist, const Geom g,
                                                          const RayParams rp, float *__restrict__ out,
                                                          unsigned long long *__restrict__ counters,
                                                          unsigned int *__restrict__ touched) {
    // 16x16 pixel tile per workgroup, 8x8 per wave
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const int imx = blockIdx.x * 16 + (wave & 1u) * 8 + (lane & 7u);
    const int imy = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool in_image = imx < (int)rp.width && imy < (int)rp.height;

    float ix = NAN, iy = NAN, iz = NAN;
    float hit_k = INFINITY;
    uint32_t samples = 0;

    // compute_ray_direction_at_pixel (:24-44); f3_normalise is a no-op (by-value argument): Q6
    uint16_t pix_x = (uint16_t)imx, pix_y = (uint16_t)imy;
    float rcx = pix_x * rp.kinv.m11 + pix_y * rp.kinv.m12 + rp.kinv.m13;
    float rcy = pix_x * rp.kinv.m21 + pix_y * rp.kinv.m22 + rp.kinv.m23;
    float rcz = pix_x * rp.kinv.m31 + pix_y * rp.kinv.m32 + rp.kinv.m33;
    F3 dir;
    dir.x = rp.rot.m11 * rcx + rp.rot.m12 * rcy + rp.rot.m13 * rcz;
    dir.y = rp.rot.m21 * rcx + rp.rot.m22 * rcy + rp.rot.m23 * rcz;
    dir.z = rp.rot.m31 * rcx + rp.rot.m32 * rcy + rp.rot.m33 * rcz;

    float near_t = 0.f, far_t = 0.f;
    bool intersects = in_image && compute_near_and_far_t(rp.origin, dir, rp.space_min, rp.space_max, near_t, far_t);

    // start point in grid coordinates (:306)
    const float sx = ((near_t * dir.x) + rp.origin.x) - rp.space_min.x;
    const float sy = ((near_t * dir.y) + rp.origin.y) - rp.space_min.y;
    const float sz = ((near_t * dir.z) + rp.origin.z) - rp.space_min.z;

    const float previous_tsdf = g.trunc;                      // Q7
    const float step_size = (float)((double)g.trunc * 0.05);  // :324 (double literal)
    const float max_t = far_t - near_t;
    float t = 0;
    int count = 0;
    bool done = !intersects;

    while (__ballot(!done) != 0ull) {
        if (!done) {
            float px = (t * dir.x) + sx;
            float py = (t * dir.y) + sy;
            float pz = (t * dir.z) + sz;
            bool owned;
            float tsdf = trilinear<SLAB, STATS>(px, py, pz, dist, g, rp, owned, touched);
            if (STATS && owned) samples++;
            if (tsdf <= 0) {
                if (tsdf < 0) {
                    t = t - step_size;
                    t = t + (previous_tsdf / (previous_tsdf - tsdf)) * step_size;
                }
                px = (t * dir.x) + sx;
                py = (t * dir.y) + sy;
                pz = (t * dir.z) + sz;
                ix = px + rp.space_min.x;
                iy = py + rp.space_min.y;
                iz = pz + rp.space_min.z;
                hit_k = (float)count;
                done = true;
            } else if (previous_tsdf < 0) {
                done = true;
            } else {
                t = t + step_size;
                if (t >= max_t) done = true;
            }
            if (count++ > 4400) done = true;  // :369
        }
    }

    if (in_image) {
        size_t idx = (size_t)imy * rp.width + imx;
        if (SLAB) {
            reinterpret_cast<float4 *>(out)[idx] = make_float4(hit_k, ix, iy, iz);
        } else {
            out[idx * 3 + 0] = ix;
            out[idx * 3 + 1] = iy;
            out[idx * 3 + 2] = iz;
        }
    }
    if (STATS) {
        uint32_t h = (in_image && ix == ix) ? 1u : 0u;
        for (int o = 32; o > 0; o >>= 1) {
            samples += __shfl_down(samples, o);
            h += __shfl_down(h, o);
        }
        if (lane == 0) {
            atomicAdd(&counters[1], (unsigned long long)samples);
            atomicAdd(&counters[2], (unsigned long long)h);
        }
    }
}

// compute_normals (src/RayCaster/GPURaycaster.cu:393-427): Q11
__global__ __launch_bounds__(256) void normals_kernel(uint32_t width, uint32_t height,
                                                      const float *__restrict__ V, float *__restrict__ N) {
    const uint32_t imx = blockIdx.x * 64 + (threadIdx.x & 63u);
    const uint32_t imy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (imx >= width || imy >= height) return;
    size_t idx = (size_t)imy * width + imx;
    float nx = 0, ny = 0, nz = 0;
    if (imy != height - 1 && imx != width - 1) {
        const float *a = V + idx * 3, *r = V + (idx + 1) * 3, *b = V + (idx + width) * 3;
        float v2x = r[0] - a[0], v2y = r[1] - a[1], v2z = r[2] - a[2];
        float v1x = b[0] - a[0], v1y = b[1] - a[1], v1z = b[2] - a[2];
        float cx = v1y * v2z - v1z * v2y;
        float cy = v1z * v2x - v1x * v2z;
        float cz = v1x * v2y - v1y * v2x;
        float l = sqrtf(cx * cx + cy * cy + cz * cz);
        nx = cx / l;
        ny = cy / l;
        nz = cz / l;
    }
    N[idx * 3 + 0] = nx;
    N[idx * 3 + 1] = ny;
    N[idx * 3 + 2] = nz;
}

// Per pixel, keep the record with the smallest k among n_slabs gathered buffers
// (layout [slab][pixel][4]).  Ties cannot occur: a sample has exactly one owner.
__global__ __launch_bounds__(256) void merge_hits_kernel(const float4 *__restrict__ hits, uint32_t n_slabs,
                                                         uint32_t n_pixels, float *__restrict__ V) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    float4 best = hits[i];
    for (uint32_t s = 1; s < n_slabs; s++) {
        float4 h = hits[(size_t)s * n_pixels + i];
        if (h.x < best.x) best = h;
    }
    V[(size_t)i * 3 + 0] = best.y;
    V[(size_t)i * 3 + 1] = best.z;
    V[(size_t)i * 3 + 2] = best.w;
}

__global__ __launch_bounds__(256) void popcount_kernel(const unsigned int *__restrict__ words, size_t n,
                                                       unsigned long long *__restrict__ counter) {
    unsigned long long c = 0;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) c += __popc(words[i]);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63u) == 0 && c) atomicAdd(counter, c);
}

static RayParams make_params(const tsdf_volume *v, uint32_t width, uint32_t height, const float pose[16],
                             const float kinv[9]) {
    RayParams rp;
    // get_vertices (src/RayCaster/GPURaycaster.cu:432-464): origin = camera.position(), rot = pose 3x3
    rp.origin = {pose[12], pose[13], pose[14]};
    rp.rot = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
    memcpy(&rp.kinv, kinv, sizeof(Mat33));
    const Geom &g = v->g;
    rp.space_min = g.offset;
    rp.space_max = {g.offset.x + g.phys.x, g.offset.y + g.phys.y, g.offset.z + g.phys.z};
    rp.width = width;
    rp.height = height;
    rp.own_lo = v->z_begin;
    rp.own_hi = v->z_end;
    return rp;
}

static int launch_normals(uint32_t width, uint32_t height, const float *V, float *N, hipStream_t s) {
    dim3 grid((width + 63) / 64, (height + 3) / 4);
    hipLaunchKernelGGL(normals_kernel, grid, dim3(256), 0, s, width, height, V, N);
    TSDF_HIP(hipGetLastError(), "compute_normals failed");
    return TSDF_OK;
}

static int check_ray_args(const tsdf_volume *v, uint32_t width, uint32_t height, const float *pose, const float *kinv) {
    TSDF_REQUIRE(v && pose && kinv, "tsdf_raycast: null argument");
    // the reference's raycaster stores width/height as uint16_t (src/include/Raycaster.hpp:35-36)
    TSDF_REQUIRE(width > 0 && height > 0 && width <= 65535 && height <= 65535, "tsdf_raycast: bad image size");
    return TSDF_OK;
}

}  // namespace tsdf

using namespace tsdf;

extern "C" {

int tsdf_raycast_device(const tsdf_volume *v, uint32_t width, uint32_t height, const float pose[16],
                        const float kinv[9], float *device_vertices, float *device_normals) {
    int rc = check_ray_args(v, width, height, pose, kinv);
    if (rc != TSDF_OK) return rc;
    TSDF_REQUIRE(device_vertices, "tsdf_raycast: null vertex buffer");
    TSDF_REQUIRE(v->z_begin == 0 && v->z_end == v->g.Z, "tsdf_raycast on a slab: use tsdf_raycast_slab_device");
    RayParams rp = make_params(v, width, height, pose, kinv);
    dim3 grid((width + 15) / 16, (height + 15) / 16);
    hipLaunchKernelGGL((process_ray_kernel<false, false>), grid, dim3(256), 0, v->stream, v->dist, v->g, rp,
                       device_vertices, (unsigned long long *)nullptr, (unsigned int *)nullptr);
    TSDF_HIP(hipGetLastError(), "process_ray failed");
    if (device_normals) return launch_normals(width, height, device_vertices, device_normals, v->stream);
    return TSDF_OK;
}

int tsdf_raycast(const tsdf_volume *cv, uint32_t width, uint32_t height, const float pose[16], const float kinv[9],
                 float *host_vertices, float *host_normals) {
    int rc = check_ray_args(cv, width, height, pose, kinv);
    if (rc != TSDF_OK) return rc;
    TSDF_REQUIRE(host_vertices, "tsdf_raycast: null vertex buffer");
    tsdf_volume *v = const_cast<tsdf_volume *>(cv);  // per-call temporaries are cached in the handle
    size_t bytes = (size_t)width * height * 3 * sizeof(float);
    if (v->ray_cap < bytes) {
        if (v->vert_buf) (void)hipFree(v->vert_buf);
        if (v->norm_buf) (void)hipFree(v->norm_buf);
        v->vert_buf = v->norm_buf = nullptr;
        v->ray_cap = 0;
        TSDF_HIP(hipMalloc((void **)&v->vert_buf, bytes), "Vertices alloc failed");
        TSDF_HIP(hipMalloc((void **)&v->norm_buf, bytes), "Normals alloc failed");
        v->ray_cap = bytes;
    }
    rc = tsdf_raycast_device(v, width, height, pose, kinv, v->vert_buf, host_normals ? v->norm_buf : nullptr);
    if (rc != TSDF_OK) return rc;
    TSDF_HIP(hipMemcpyAsync(host_vertices, v->vert_buf, bytes, hipMemcpyDeviceToHost, v->stream), "Vertices Memcpy failed");
    if (host_normals)
        TSDF_HIP(hipMemcpyAsync(host_normals, v->norm_buf, bytes, hipMemcpyDeviceToHost, v->stream), "Normals Memcpy failed");
    TSDF_HIP(hipStreamSynchronize(v->stream), "process_ray failed");
    return TSDF_OK;
}

int tsdf_normals_device(uint32_t width, uint32_t height, const float *device_vertices, float *device_normals,
                        void *hip_stream) {
    TSDF_REQUIRE(device_vertices && device_normals && width > 0 && height > 0, "tsdf_normals: bad argument");
    return launch_normals(width, height, device_vertices, device_normals, (hipStream_t)hip_stream);
}

int tsdf_raycast_stats(const tsdf_volume *v, uint32_t width, uint32_t height, const float pose[16],
                       const float kinv[9], uint64_t *samples, uint64_t *touched_voxels, uint64_t *hits) {
    int rc = check_ray_args(v, width, height, pose, kinv);
    if (rc != TSDF_OK) return rc;
    TSDF_REQUIRE(v->z_begin == 0 && v->z_end == v->g.Z, "tsdf_raycast_stats needs a whole volume");
    RayParams rp = make_params(v, width, height, pose, kinv);
    size_t n = (size_t)v->g.X * v->g.Y * v->g.Z;
    size_t words = (n + 31) / 32;
    unsigned int *bitmap = nullptr;
    float *verts = nullptr;
    TSDF_HIP(hipMalloc((void **)&bitmap, words * sizeof(unsigned int)), "stats bitmap alloc");
    hipError_t e = hipMalloc((void **)&verts, (size_t)width * height * 3 * sizeof(float));
    if (e != hipSuccess) {
        (void)hipFree(bitmap);
        return hip_fail(e, "stats vertex alloc");
    }
    (void)hipMemsetAsync(bitmap, 0, words * sizeof(unsigned int), v->stream);
    (void)hipMemsetAsync(v->counter_dev, 0, 4 * sizeof(unsigned long long), v->stream);
    dim3 grid((width + 15) / 16, (height + 15) / 16);
    hipLaunchKernelGGL((process_ray_kernel<false, true>), grid, dim3(256), 0, v->stream, v->dist, v->g, rp, verts,
                       v->counter_dev, bitmap);
    hipLaunchKernelGGL(popcount_kernel, dim3(1024), dim3(256), 0, v->stream, bitmap, words, v->counter_dev + 3);
    unsigned long long c[4] = {0, 0, 0, 0};
    e = hipMemcpyAsync(c, v->counter_dev, sizeof(c), hipMemcpyDeviceToHost, v->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(v->stream);
    (void)hipFree(bitmap);
    (void)hipFree(verts);
    if (e != hipSuccess) return hip_fail(e, "raycast stats");
    if (samples) *samples = c[1];
    if (hits) *hits = c[2];
    if (touched_voxels) *touched_voxels = c[3];
    return TSDF_OK;
}

int tsdf_raycast_slab_device(const tsdf_volume *v, uint32_t width, uint32_t height, const float pose[16],
                             const float kinv[9], float *device_hits) {
    int rc = check_ray_args(v, width, height, pose, kinv);
    if (rc != TSDF_OK) return rc;
    TSDF_REQUIRE(device_hits, "tsdf_raycast_slab: null hit buffer");
    RayParams rp = make_params(v, width, height, pose, kinv);
    dim3 grid((width + 15) / 16, (height + 15) / 16);
    hipLaunchKernelGGL((process_ray_kernel<true, false>), grid, dim3(256), 0, v->stream, v->dist, v->g, rp,
                       device_hits, (unsigned long long *)nullptr, (unsigned int *)nullptr);
    TSDF_HIP(hipGetLastError(), "process_ray (slab) failed");
    return TSDF_OK;
}

int tsdf_merge_hits_device(const float *device_hits_all, uint32_t n_slabs, uint32_t width, uint32_t height,
                           float *device_vertices, void *hip_stream) {
    TSDF_REQUIRE(device_hits_all && device_vertices && n_slabs > 0 && width > 0 && height > 0,
                 "tsdf_merge_hits: bad argument");
    uint32_t n = width * height;
    hipLaunchKernelGGL(merge_hits_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream,
                       reinterpret_cast<const float4 *>(device_hits_all), n_slabs, n, device_vertices);
    TSDF_HIP(hipGetLastError(), "merge hits failed");
    return TSDF_OK;
}

}  // extern "C"
