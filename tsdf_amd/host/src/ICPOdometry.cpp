// ICPOdometry of the reference (third_party/ICP_CUDA/ICPOdometry.cpp) over the C ABI.
#include "ICP_CUDA/ICPOdometry.h"

#include "host_common.hpp"
#include "tsdf_amd.h"

ICPOdometry::ICPOdometry(int width, int height, float cx, float cy, float fx, float fy, float distThresh, float angleThresh)
    : lastError(0), lastInliers(width * height), m_hip(nullptr), width(width), height(height) {
    // (the reference's constructor cannot fail short of a CUDA allocation error, which exits)
    tsdf_host::check(tsdf_icp_create(width, height, cx, cy, fx, fy, distThresh, angleThresh, &m_hip), "ICPOdometry");
}

ICPOdometry::~ICPOdometry() { tsdf_icp_destroy(m_hip); }

void ICPOdometry::initICP(unsigned short *depth, const float depthCutoff) {
    tsdf_host::check(tsdf_icp_init(m_hip, 0, depth, depthCutoff), "initICP");
}

void ICPOdometry::initICPModel(unsigned short *depth, const float depthCutoff) {
    tsdf_host::check(tsdf_icp_init(m_hip, 1, depth, depthCutoff), "initICPModel");
}

void ICPOdometry::getIncrementalTransformation(Sophus::SE3d &T_prev_curr, int /*threads*/, int /*blocks*/) {
    Eigen::Matrix<double, 4, 4> m = T_prev_curr.matrix();  // column-major, as the C ABI expects
    tsdf_host::check(tsdf_icp_get_incremental_transformation(m_hip, m.data(), &lastError, &lastInliers),
                     "getIncrementalTransformation");
    T_prev_curr = Sophus::SE3d(m);
}
