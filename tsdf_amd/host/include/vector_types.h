// Plain host definitions of the short-vector type names the reference's class surface exposes
// (float3, int3, uint3, uchar3, dim3): its headers include CUDA's "vector_types.h"
// (src/include/TSDFVolume.hpp:15, ply.hpp) and callers such as src/Tools/kinfu.cpp:206-207 use
// float3 / int3 directly.  Here the host side is ordinary C++ over a C ABI, so these are PODs.
#ifndef TSDF_AMD_VECTOR_TYPES_H
#define TSDF_AMD_VECTOR_TYPES_H

#if defined(__HIPCC__) || defined(HIP_INCLUDE_HIP_HIP_RUNTIME_H)
#include <hip/hip_vector_types.h>
#else
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int3 { int x, y, z; };
struct uint3 { unsigned int x, y, z; };
struct uchar3 { unsigned char x, y, z; };
struct dim3 {
    unsigned int x, y, z;
    dim3(unsigned int vx = 1, unsigned int vy = 1, unsigned int vz = 1) : x(vx), y(vy), z(vz) {}
};
static inline float3 make_float3(float x, float y, float z) { float3 f = {x, y, z}; return f; }
static inline int3 make_int3(int x, int y, int z) { int3 i = {x, y, z}; return i; }
#endif

#endif
