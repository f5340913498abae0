#ifndef Definitions_hpp
#define Definitions_hpp
#include <Eigen/Core>
// Marker for "no vertex here" (reference: src/Utilities/Definitions.cpp)
extern const Eigen::Vector3f BAD_VERTEX;
#endif
