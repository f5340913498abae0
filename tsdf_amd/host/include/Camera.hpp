// Pinhole camera: intrinsics, pose and the coordinate transforms the hot path consumes.
// Same public surface as the reference's Camera (src/include/Camera.hpp:17-215); host-only.
#ifndef TSDF_AMD_HOST_CAMERA_INCLUDED
#define TSDF_AMD_HOST_CAMERA_INCLUDED

#include <Eigen/Dense>
#include <cstdint>
#include <deque>

class Camera {
public:
    // Kinect depth camera, Freiburg-1 IR calibration (reference: Camera.hpp:41-44)
    static Camera *default_depth_camera() { return new Camera{591.1f, 590.1f, 331.0f, 234.6f}; }

    explicit Camera(const float focal_x, const float focal_y, const float centre_x, const float centre_y);
    Camera(const Eigen::Matrix3f &k);
    explicit Camera(const int image_width, const int image_height, const float fov_x, const float fov_y);

    // intrinsics
    const Eigen::Matrix3f k() const;
    const Eigen::Matrix3f kinv() const;

    // pose (camera -> world) and its inverse
    const Eigen::Matrix4f &pose() const;
    const Eigen::Matrix4f &inverse_pose() const;
    void set_pose(const Eigen::Matrix4f &pose);
    void set_pose(float vars[7]);  // tx ty tz qx qy qz qw (TUM order)
    void move_to(const Eigen::Vector3f &world_coordinate);
    void move_to(float wx, float wy, float wz);
    void look_at(const Eigen::Vector3f &world_coordinate);
    void look_at(float wx, float wy, float wz);
    Eigen::Vector3f position() const;

    // coordinate transforms
    Eigen::Vector2f pixel_to_image_plane(const Eigen::Vector2i &image_coordinate) const;
    Eigen::Vector2f pixel_to_image_plane(const uint16_t x, const uint16_t y) const;
    Eigen::Vector2i image_plane_to_pixel(const Eigen::Vector2f &camera_coordinate) const;
    Eigen::Vector3f camera_to_world(const Eigen::Vector3f &camera_coordinate) const;
    Eigen::Vector3f world_to_camera_normal(const Eigen::Vector3f &world_normal) const;
    Eigen::Vector3f world_to_camera(const Eigen::Vector3f &world_coordinate) const;
    Eigen::Vector2i world_to_pixel(const Eigen::Vector3f &world_coordinate) const;

    // depth map -> camera-space vertices and normals
    void depth_image_to_vertices_and_normals(const uint16_t *depth_image, const uint32_t width, const uint32_t height,
                                             Eigen::Matrix<float, 3, Eigen::Dynamic> &vertices,
                                             Eigen::Matrix<float, 3, Eigen::Dynamic> &normals) const;

private:
    void init();
    Eigen::Matrix3f m_k;
    Eigen::Matrix3f m_k_inverse;
    Eigen::Matrix4f m_pose;
    Eigen::Matrix4f m_pose_inverse;
};

#endif /* TSDF_AMD_HOST_CAMERA_INCLUDED */
