// The one shared constant of the class surface (the reference declares it in src/include/Definitions.hpp and defines
// it in src/Utilities/Definitions.cpp): the vertex value that stands for "nothing here".
#ifndef TSDF_AMD_HOST_DEFINITIONS_INCLUDED
#define TSDF_AMD_HOST_DEFINITIONS_INCLUDED
#include <Eigen/Core>
extern const Eigen::Matrix<float, 3, 1> BAD_VERTEX;   // (= Eigen::Vector3f)
#endif
