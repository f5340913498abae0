// Vertex / normal map rendering (reference: src/Utilities/RenderUtilities.cpp:39-112).
#include "RenderUtilities.hpp"

#include <cmath>
#include <vector>

#include "Camera.hpp"

void save_normals_as_colour_png(std::string filename, uint16_t width, uint16_t height,
                                const Eigen::Matrix<float, 3, Eigen::Dynamic> &normals) {
    PngWrapper *p = normals_as_png(width, height, normals);
    p->save_to(filename);
    delete p;
}

void save_rendered_scene_as_png(std::string filename, uint16_t width, uint16_t height,
                                const Eigen::Matrix<float, 3, Eigen::Dynamic> &vertices,
                                const Eigen::Matrix<float, 3, Eigen::Dynamic> &normals, const Camera &camera,
                                const Eigen::Vector3f &light_source) {
    PngWrapper *p = scene_as_png(width, height, vertices, normals, camera, light_source);
    p->save_to(filename);
    delete p;
}

// reference: :39-78.  NaN vertices / normals (ray misses) shade to the ambient level: fmax(0, NaN) = 0.
PngWrapper *scene_as_png(uint16_t width, uint16_t height, const Eigen::Matrix<float, 3, Eigen::Dynamic> &vertices,
                         const Eigen::Matrix<float, 3, Eigen::Dynamic> &normals, const Camera &camera,
                         const Eigen::Vector3f &light_source) {
    (void)camera;
    const uint64_t num_pixels = (uint64_t)width * height;
    std::vector<uint8_t> image(num_pixels);
    const float ambient = 0.2f;
    const float diffuse = 1.0f - ambient;
    for (uint64_t idx = 0; idx < num_pixels; idx++) {
        Eigen::Vector3f vertex{vertices(0, idx), vertices(1, idx), vertices(2, idx)};
        Eigen::Vector3f to_light = (light_source - vertex).normalized();
        Eigen::Vector3f n{normals(0, idx), normals(1, idx), normals(2, idx)};
        float shade = (float)std::fmax(0.0, (double)n.dot(to_light));
        shade = ambient + diffuse * shade;
        image[idx] = (uint8_t)std::floor(shade * 255);
    }
    return new PngWrapper(width, height, image.data(), PngWrapper::GREYSCALE_8);
}

// reference: :80-112
PngWrapper *normals_as_png(uint16_t width, uint16_t height, const Eigen::Matrix<float, 3, Eigen::Dynamic> &normals) {
    const uint64_t num_pixels = (uint64_t)width * height;
    std::vector<uint8_t> image(num_pixels * 3);
    uint64_t w = 0;
    for (uint64_t idx = 0; idx < num_pixels; idx++) {
        float n[3] = {normals(0, idx), normals(1, idx), normals(2, idx)};
        if (n[2] < 0) n[2] = -n[2];
        for (int c = 0; c < 3; c++) {
            float v = ((n[c] / 2.0f) + 0.5f) * 255;
            image[w++] = (v == v) ? (uint8_t)std::floor(v) : 0;  // NaN (ray miss) -> black
        }
    }
    return new PngWrapper(width, height, image.data(), PngWrapper::COLOUR);
}
