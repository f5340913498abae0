// GPURaycaster class surface over the C ABI (reference host glue: src/RayCaster/GPURaycaster.cu:432-606).
#include "GPURaycaster.hpp"

#include <cmath>
#include <vector>

#include "host_common.hpp"

using tsdf_host::check;

// reference: src/RayCaster/GPURaycaster.cu:519-547
void GPURaycaster::raycast(const TSDFVolume &volume, const Camera &camera,
                           Eigen::Matrix<float, 3, Eigen::Dynamic> &vertices,
                           Eigen::Matrix<float, 3, Eigen::Dynamic> &normals) const {
    vertices.resize(3, m_width * m_height);
    normals.resize(3, m_width * m_height);
    const Eigen::Matrix3f kinv = camera.kinv();
    // 3 x N column-major == packed float3[N], the layout the C ABI writes
    check(tsdf_raycast(volume.handle(), m_width, m_height, camera.pose().data(), kinv.data(), vertices.data(),
                       normals.data()),
          "process_ray failed ");
}

// reference: src/RayCaster/GPURaycaster.cu:555-606 (without its debug PNG dump to a hard-coded path)
DepthImage *GPURaycaster::render_to_depth_image(const TSDFVolume &volume, const Camera &camera) const {
    if (tsdf_host::verbose()) std::cout << "Rendering depth map" << std::endl;
    const size_t n = (size_t)m_width * m_height;
    std::vector<float> verts(n * 3);
    const Eigen::Matrix3f kinv = camera.kinv();
    check(tsdf_raycast(volume.handle(), m_width, m_height, camera.pose().data(), kinv.data(), verts.data(), nullptr),
          "process_ray failed ");
    std::vector<uint16_t> depth(n);
    for (size_t i = 0; i < n; i++) {
        Eigen::Vector3f cam = camera.world_to_camera(Eigen::Vector3f{verts[i * 3], verts[i * 3 + 1], verts[i * 3 + 2]});
        float z = roundf(cam.z());
        // misses are NaN; the reference's (uint16_t)roundf(NaN) is undefined -> store 0 (= invalid depth)
        depth[i] = (z == z && z > 0.0f && z < 65536.0f) ? (uint16_t)z : 0;
    }
    return new DepthImage(m_width, m_height, depth.data());
}
