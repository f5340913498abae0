// Host-side pinhole camera (input provider of the hot path: K, K^-1, pose, pose^-1).
// Behaviour follows the reference's src/Camera.cpp (cited per function); written against the
// Eigen API subset of eigen_compat/ or a real Eigen.
#include "Camera.hpp"

#include <cmath>

#include "Definitions.hpp"

namespace {
const float kEps = 1e-6f;
}

// reference: src/Camera.cpp:20-24
void Camera::init() {
    m_k_inverse = m_k.inverse();
    set_pose(Eigen::Matrix4f::Identity());
}

// reference: src/Camera.cpp:33-42
Camera::Camera(const float focal_x, const float focal_y, const float centre_x, const float centre_y) {
    m_k = Eigen::Matrix3f::Zero();
    m_k(0, 0) = focal_x;
    m_k(1, 1) = focal_y;
    m_k(0, 2) = centre_x;
    m_k(1, 2) = centre_y;
    m_k(2, 2) = 1.0f;
    init();
}

// reference: src/Camera.cpp:48-51
Camera::Camera(const Eigen::Matrix3f &k) {
    m_k = k;
    init();
}

// reference: src/Camera.cpp:60-66 (focal lengths from the fields of view)
Camera::Camera(const int image_width, const int image_height, const float fov_x, const float fov_y) {
    float focal_x = -image_width / (2 * std::tan(fov_x / 2.0f));
    float focal_y = -image_height / (2 * std::tan(fov_y / 2.0f));
    m_k << -focal_x, 0.0f, (image_width / 2.0f), 0.0f, -focal_y, (image_height / 2.0f), 0.0f, 0.0f, 1.0f;
    init();
}

const Eigen::Matrix3f Camera::k() const { return m_k; }
const Eigen::Matrix3f Camera::kinv() const { return m_k_inverse; }
const Eigen::Matrix4f &Camera::pose() const { return m_pose; }
const Eigen::Matrix4f &Camera::inverse_pose() const { return m_pose_inverse; }

// reference: src/Camera.cpp:108-111
void Camera::set_pose(const Eigen::Matrix4f &pose) {
    m_pose = pose;
    m_pose_inverse = m_pose.inverse();
}

// Declared by the reference (src/include/Camera.hpp:104) but never defined there; the obvious
// meaning is the TUM record conversion of src/DataLoader/TUMDataLoader.cpp:47-76
// (tx ty tz in metres -> mm, quaternion qx qy qz qw -> rotation).
void Camera::set_pose(float vars[7]) {
    float w = vars[6], x = vars[3], y = vars[4], z = vars[5];
    Eigen::Matrix4f pose = Eigen::Matrix4f::Zero();
    pose(0, 0) = 1 - 2 * (y * y + z * z);
    pose(0, 1) = 2 * (x * y - w * z);
    pose(0, 2) = 2 * (x * z + w * y);
    pose(1, 0) = 2 * (x * y + w * z);
    pose(1, 1) = 1 - 2 * (x * x + z * z);
    pose(1, 2) = 2 * (y * z - w * x);
    pose(2, 0) = 2 * (x * z - w * y);
    pose(2, 1) = 2 * (y * z + w * x);
    pose(2, 2) = 1 - 2 * (x * x + y * y);
    pose(0, 3) = vars[0] * 1000.0f;
    pose(1, 3) = vars[1] * 1000.0f;
    pose(2, 3) = vars[2] * 1000.0f;
    pose(3, 3) = 1.0f;
    set_pose(pose);
}

void Camera::move_to(const Eigen::Vector3f &p) { move_to(p.x(), p.y(), p.z()); }

// reference: src/Camera.cpp:117-123
void Camera::move_to(float wx, float wy, float wz) {
    m_pose(0, 3) = wx;
    m_pose(1, 3) = wy;
    m_pose(2, 3) = wz;
    m_pose_inverse = m_pose.inverse();
}

// reference: src/Camera.cpp:125-180.  Columns of the rotation become left / up / forward.
void Camera::look_at(const Eigen::Vector3f &target) {
    using Eigen::Vector3f;
    Vector3f here{m_pose(0, 3), m_pose(1, 3), m_pose(2, 3)};
    Vector3f forward = target - here;
    forward.normalize();

    Vector3f up{0.0f, 0.0f, 0.0f};
    const bool vertical = (std::fabs(forward.x()) < kEps) && (std::fabs(forward.z()) < kEps);
    if (vertical) {
        if (forward.y() < 0) up = Vector3f{0.0f, 0.0f, 1.0f};        // looking straight down
        else if (forward.y() > 0) up = Vector3f{0.0f, 0.0f, -1.0f};  // straight up
    } else {
        up = Vector3f{0.0f, 1.0f, 0.0f};
    }
    Vector3f left = up.cross(forward);
    left.normalize();
    up = forward.cross(left);
    up.normalize();

    for (int r = 0; r < 3; r++) {
        m_pose(r, 0) = left[r];
        m_pose(r, 1) = up[r];
        m_pose(r, 2) = forward[r];
    }
    m_pose(3, 0) = 0.0f;
    m_pose(3, 1) = 0.0f;
    m_pose(3, 2) = 0.0f;
    m_pose(3, 3) = 1.0f;
    m_pose_inverse = m_pose.inverse();
}

void Camera::look_at(float wx, float wy, float wz) { look_at(Eigen::Vector3f{wx, wy, wz}); }

// reference: src/Camera.cpp:211-213
Eigen::Vector3f Camera::position() const { return Eigen::Vector3f{m_pose(0, 3), m_pose(1, 3), m_pose(2, 3)}; }

Eigen::Vector2f Camera::pixel_to_image_plane(const Eigen::Vector2i &p) const {
    return pixel_to_image_plane((uint16_t)p.x(), (uint16_t)p.y());
}

// reference: src/Camera.cpp:228-238
Eigen::Vector2f Camera::pixel_to_image_plane(const uint16_t x, const uint16_t y) const {
    Eigen::Vector3f h{static_cast<float>(x), static_cast<float>(y), 1.0f};
    Eigen::Vector3f c = m_k_inverse * h;
    return Eigen::Vector2f{c[0] / c[2], c[1] / c[2]};
}

// reference: src/Camera.cpp:245-255
Eigen::Vector2i Camera::image_plane_to_pixel(const Eigen::Vector2f &cam) const {
    Eigen::Vector3f h{cam.x(), cam.y(), 1.0f};
    Eigen::Vector3f img = m_k * h;
    Eigen::Vector2i px;
    px.x() = (int)std::round(img.x());
    px.y() = (int)std::round(img.y());
    return px;
}

// reference: src/Camera.cpp:264-271
Eigen::Vector3f Camera::camera_to_world(const Eigen::Vector3f &c) const {
    Eigen::Vector4f h{c.x(), c.y(), c.z(), 1.0f};
    Eigen::Vector4f w = m_pose * h;
    return Eigen::Vector3f{w[0] / w[3], w[1] / w[3], w[2] / w[3]};
}

// reference: src/Camera.cpp:278-280
Eigen::Vector3f Camera::world_to_camera_normal(const Eigen::Vector3f &n) const {
    Eigen::Matrix3f r = m_pose_inverse.block(0, 0, 3, 3);
    return r * n;
}

// reference: src/Camera.cpp:287-294
Eigen::Vector3f Camera::world_to_camera(const Eigen::Vector3f &w) const {
    Eigen::Vector4f h{w.x(), w.y(), w.z(), 1.0f};
    Eigen::Vector4f c = m_pose_inverse * h;
    return Eigen::Vector3f{c[0] / c[3], c[1] / c[3], c[2] / c[3]};
}

// reference: src/Camera.cpp:303-322
Eigen::Vector2i Camera::world_to_pixel(const Eigen::Vector3f &w) const {
    Eigen::Vector3f cam = world_to_camera(w);
    Eigen::Vector3f img = m_k * cam;
    img = img / img[2];
    Eigen::Vector2i px;
    px.x() = (int)std::round(img[0]);
    px.y() = (int)std::round(img[1]);
    return px;
}

// reference: src/Camera.cpp:335-391.  Walks from the bottom-right pixel backwards so the right
// and lower neighbours needed for the normal already exist.
void Camera::depth_image_to_vertices_and_normals(const uint16_t *depth_image, const uint32_t width,
                                                 const uint32_t height,
                                                 Eigen::Matrix<float, 3, Eigen::Dynamic> &vertices,
                                                 Eigen::Matrix<float, 3, Eigen::Dynamic> &normals) const {
    using Eigen::Vector3f;
    vertices.resize(3, width * height);
    normals.resize(3, width * height);
    int32_t idx = (int32_t)(width * height) - 1;
    for (int32_t y = (int32_t)height - 1; y >= 0; y--) {
        for (int32_t x = (int32_t)width - 1; x >= 0; x--, idx--) {
            Vector3f vertex = BAD_VERTEX;
            Vector3f normal{0.0f, 0.0f, 0.0f};
            const uint16_t depth = depth_image[idx];
            if (depth != 0) {
                Eigen::Vector2f ip = pixel_to_image_plane((uint16_t)x, (uint16_t)y);
                vertex = Vector3f{ip.x(), ip.y(), 1.0f} * (float)depth;
                if (y < (int32_t)height - 1 && x < (int32_t)width - 1) {
                    Vector3f right{vertices(0, idx + 1), vertices(1, idx + 1), vertices(2, idx + 1)};
                    Vector3f below{vertices(0, idx + width), vertices(1, idx + width), vertices(2, idx + width)};
                    if (right != BAD_VERTEX && below != BAD_VERTEX) {
                        right -= vertex;
                        below -= vertex;
                        normal = right.cross(below).normalized();
                    }
                }
            }
            for (int i = 0; i < 3; i++) {
                vertices(i, idx) = vertex[i];
                normals(i, idx) = normal[i];
            }
        }
    }
}
