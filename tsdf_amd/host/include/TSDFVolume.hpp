// Dense TSDF voxel grid resident in GPU memory (MI355X): same public surface as the reference's
// TSDFVolume (src/include/TSDFVolume.hpp:21-304) so callers such as src/Tools/kinfu.cpp compile
// unchanged.  All device work goes through the C ABI of include/tsdf_amd.h.
#ifndef TSDF_AMD_HOST_TSDF_VOLUME_INCLUDED
#define TSDF_AMD_HOST_TSDF_VOLUME_INCLUDED

#include "Camera.hpp"

#include <Eigen/Core>
#include "vector_types.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

struct tsdf_volume;  // C-ABI handle (include/tsdf_amd.h)

class TSDFVolume {
public:
    // 24-byte deformation node: where the voxel centre sits, and a rotation (unused by the path)
    struct DeformationNode {
        float3 translation;
        float3 rotation;
    };

    // float triple, implicitly convertible from/to float3
    struct Float3 {
        float x, y, z;
        inline Float3(const float3 &v) : x(v.x), y(v.y), z(v.z) {}
        inline Float3(float fx = 0.0f, float fy = 0.0f, float fz = 0.0f) : x(fx), y(fy), z(fz) {}
        inline operator float3() const { return float3{x, y, z}; }
        inline Float3 operator-(const Float3 &o) const { return float3{x - o.x, y - o.y, z - o.z}; }
        inline Float3 operator+(const Float3 &o) const { return float3{x + o.x, y + o.y, z + o.z}; }
        inline Float3 operator/(const float s) const { return float3{x / s, y / s, z / s}; }
        inline Float3 operator*(const Float3 &o) const { return float3{x * o.x, y * o.y, z * o.z}; }
        inline float norm() const { return std::sqrt(x * x + y * y + z * z); }
    };

    struct Int3 {
        int16_t x, y, z;
    };

    // unsigned triple, implicitly convertible from/to dim3
    struct UInt3 {
        unsigned int x, y, z;
        inline UInt3(const dim3 &d) : x{d.x}, y{d.y}, z{d.z} {}
        inline UInt3(uint32_t vx, uint32_t vy, uint32_t vz) : x{vx}, y{vy}, z{vz} {}
        inline operator dim3() const { return dim3{x, y, z}; }
    };

    ~TSDFVolume();

    // size in voxels, physical size in mm; throws std::invalid_argument on zero / negative sizes
    TSDFVolume(const UInt3 &size = UInt3{64, 64, 64}, const Float3 &physical_size = Float3{3000.0f, 3000.0f, 3000.0f});
    TSDFVolume(uint16_t volume_x, uint16_t volume_y, uint16_t volume_z, float psize_x, float psize_y, float psize_z);
    // load a volume written by save_to_file; throws std::invalid_argument on failure
    TSDFVolume(const std::string &file_name);

    // drop the contents, re-allocate and clear; the offset is kept
    void set_size(uint16_t volume_x, uint16_t volume_y, uint16_t volume_z, float psize_x, float psize_y, float psize_z);

    inline UInt3 size() const { return (UInt3)m_size; }
    inline Float3 voxel_size() const { return (Float3)m_voxel_size; }
    inline Float3 physical_size() const { return (Float3)m_physical_size; }
    inline float truncation_distance() const { return m_truncation_distance; }

    // world position of the corner of voxel (0,0,0); setting it does not move the deformation grid
    void offset(float ox, float oy, float oz);
    inline Float3 offset() const { return (Float3)m_offset; }

    // weights <- 0, distances <- truncation distance, deformation grid <- regular voxel centres
    void clear();

    inline size_t index(int x, int y, int z) const { return x + (y * m_size.x) + (z * m_size.x * m_size.y); }

    // Per-voxel arrays: DEVICE pointers (x fastest), blocking whole-array uploads from host memory
    DeformationNode *deformation() const;
    void set_deformation(DeformationNode *deformation);
    const float *distance_data() const;
    void set_distance_data(const float *distance_data);
    const float *weight_data() const;
    void set_weight_data(const float *weight_data);

    inline float3 global_rotation() const { return m_global_rotation; }
    inline float3 global_translation() const { return m_global_translation; }

    // apply the deformation field to mesh points in place (host memory)
    void deform_mesh(const int num_points, float3 *points) const;

    // fuse one depth frame (uint16 mm, 0 = invalid) seen from `camera`
    void integrate(const uint16_t *depth_map, uint32_t width, uint32_t height, const Camera &camera);

    bool save_to_file(const std::string &file_name) const;
    bool load_from_file(const std::string &file_name);

    // ray cast the zero crossing from `camera`: 3 x (width*height) vertices and normals
    void raycast(uint16_t width, uint16_t height, const Camera &camera,
                 Eigen::Matrix<float, 3, Eigen::Dynamic> &vertices,
                 Eigen::Matrix<float, 3, Eigen::Dynamic> &normals) const;

    // C-ABI handle, for the other classes of this library
    inline tsdf_volume *handle() const { return m_handle; }

private:
    void deallocate();
    void refresh_from_handle();

    tsdf_volume *m_handle;

    // host mirror of the handle's geometry (kept so the accessors above stay inline)
    dim3 m_size;
    float3 m_physical_size;
    float3 m_offset;
    float3 m_voxel_size;
    float m_truncation_distance;
    float m_max_weight;
    float3 m_global_translation;
    float3 m_global_rotation;
};
#endif /* TSDF_AMD_HOST_TSDF_VOLUME_INCLUDED */
