// Shared host/device definitions for the gfx950 TSDF library (internal; the public surface
// is include/tsdf_amd.h).  Compile every translation unit with -ffp-contract=off: the parity
// contract is that each fp32 operation is rounded on its own, in the order the reference
// writes it.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "tsdf_amd.h"

namespace tsdf {

// ---- error plumbing ----------------------------------------------------------------------
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);

#define TSDF_HIP(call, what)                                  \
    do {                                                      \
        hipError_t e__ = (call);                              \
        if (e__ != hipSuccess) return tsdf::hip_fail(e__, what); \
    } while (0)

#define TSDF_REQUIRE(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            tsdf::set_error(__VA_ARGS__);    \
            return TSDF_ERR_INVALID;         \
        }                                    \
    } while (0)

// ---- PODs passed by value as kernel arguments (land in SGPRs through the kernarg segment) ----
// Same member order as the reference's Mat44 / Mat33 (src/include/cuda_utilities.hpp:12-23)
// so a column-major float[16] / float[9] can be memcpy'd in.
struct Mat44 {
    float m11, m21, m31, m41;
    float m12, m22, m32, m42;
    float m13, m23, m33, m43;
    float m14, m24, m34, m44;
};
struct Mat33 {
    float m11, m21, m31;
    float m12, m22, m32;
    float m13, m23, m33;
};
struct F3 {
    float x, y, z;
};

// Geometry of the (global) grid plus the resident plane range of this object.
struct Geom {
    uint32_t X, Y, Z;          // global grid
    uint32_t z_store_begin;    // first resident plane
    uint32_t z_store_end;      // one past the last resident plane
    F3 vs;                     // voxel size
    F3 offset;                 // m_offset now
    F3 offset_clear;           // m_offset when clear() last ran (Q1)
    F3 phys;                   // physical size
    float trunc;
};

// float -> int with the reference target's semantics (CUDA cvt.rzi: saturating, NaN -> 0);
// written out so the result does not depend on what an out-of-range fptosi lowers to.
__host__ __device__ inline int f2i_sat(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

// ---- object state ------------------------------------------------------------------------
}  // namespace tsdf

struct tsdf_volume {
    tsdf::Geom g;
    uint32_t z_begin, z_end;  // owned planes
    float max_weight;
    float global_translation[3];
    float global_rotation[3];
    int device;
    hipStream_t stream;
    float *dist;
    float *weight;
    tsdf_deformation_node *nodes;  // nullptr while implicit
    // cached per-call temporaries (the reference mallocs/frees these every call)
    uint16_t *depth_buf;
    size_t depth_cap;
    float *vert_buf;
    float *norm_buf;
    size_t ray_cap;
    // diagnostics
    int counting;
    unsigned long long *counter_dev;  // [0] = updated voxels, [1] = samples, [2] = hits
    uint64_t last_updated;
    size_t resident_voxels() const { return (size_t)g.X * g.Y * (g.z_store_end - g.z_store_begin); }
};

struct tsdf_bilateral {
    float sigma_colour, sigma_space;
    int radius;
    int device;
    float *kernel_dev;       // (2r+1)^2
    float *similarity_dev;   // 65536 entries (first 256 = the reference's table)
    void *img_in;            // cached device images for the host variants
    void *img_out;
    size_t img_cap;
};
