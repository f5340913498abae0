// Abstract ray caster: renders vertex and normal maps of a TSDFVolume from a camera.
// Same surface as the reference's Raycaster (src/include/Raycaster.hpp:17-39).
#ifndef TSDF_AMD_HOST_RAYCASTER_INCLUDED
#define TSDF_AMD_HOST_RAYCASTER_INCLUDED

#include <Eigen/Core>

#include "Camera.hpp"
#include "TSDFVolume.hpp"

class Raycaster {
public:
    Raycaster(int width = 640, int height = 480) {
        m_width = width;
        m_height = height;
    }
    virtual ~Raycaster() {}

    virtual void raycast(const TSDFVolume &volume, const Camera &camera,
                         Eigen::Matrix<float, 3, Eigen::Dynamic> &vertices,
                         Eigen::Matrix<float, 3, Eigen::Dynamic> &normals) const = 0;

protected:
    uint16_t m_width;
    uint16_t m_height;
};
#endif /* TSDF_AMD_HOST_RAYCASTER_INCLUDED */
