// Rendering of ray-cast vertex / normal maps to images.  Same functions as the reference's
// src/include/RenderUtilities.hpp.
#ifndef TSDF_AMD_HOST_RENDER_UTILITIES_INCLUDED
#define TSDF_AMD_HOST_RENDER_UTILITIES_INCLUDED

#include <Eigen/Dense>
#include <cstdint>
#include <string>

#include "PngWrapper.hpp"

class Camera;

// normals as RGB: (n/2 + 0.5) * 255 per channel, |nz|
PngWrapper *normals_as_png(uint16_t width, uint16_t height, const Eigen::Matrix<float, 3, Eigen::Dynamic> &normals);

// Lambertian shading from a point light: 0.2 ambient + 0.8 * max(0, n . l), 8-bit grey
PngWrapper *scene_as_png(uint16_t width, uint16_t height, const Eigen::Matrix<float, 3, Eigen::Dynamic> &vertices,
                         const Eigen::Matrix<float, 3, Eigen::Dynamic> &normals, const Camera &camera,
                         const Eigen::Vector3f &light_source);

void save_normals_as_colour_png(std::string filename, uint16_t width, uint16_t height,
                                const Eigen::Matrix<float, 3, Eigen::Dynamic> &normals);
void save_rendered_scene_as_png(std::string filename, uint16_t width, uint16_t height,
                                const Eigen::Matrix<float, 3, Eigen::Dynamic> &vertices,
                                const Eigen::Matrix<float, 3, Eigen::Dynamic> &normals, const Camera &camera,
                                const Eigen::Vector3f &light_source);

#endif  // TSDF_AMD_HOST_RENDER_UTILITIES_INCLUDED
