// GPU ray caster (HIP, MI355X).  Same surface as the reference's GPURaycaster
// (src/include/GPURaycaster.hpp:19-41).
#ifndef TSDF_AMD_HOST_GPU_RAYCASTER_INCLUDED
#define TSDF_AMD_HOST_GPU_RAYCASTER_INCLUDED

#include <Eigen/Core>

#include "DepthImage.hpp"
#include "Raycaster.hpp"
#include "TSDFVolume.hpp"

class GPURaycaster : public Raycaster {
public:
    GPURaycaster(int width = 640, int height = 480) : Raycaster{width, height} {}

    virtual void raycast(const TSDFVolume &volume, const Camera &camera,
                         Eigen::Matrix<float, 3, Eigen::Dynamic> &vertices,
                         Eigen::Matrix<float, 3, Eigen::Dynamic> &normals) const;

    // ray cast, then camera-space z of every vertex rounded to uint16 mm; caller deletes the image
    DepthImage *render_to_depth_image(const TSDFVolume &volume, const Camera &camera) const;
};
#endif /* TSDF_AMD_HOST_GPU_RAYCASTER_INCLUDED */
