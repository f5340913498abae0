#include "Definitions.hpp"

#include <limits>

// reference: src/Utilities/Definitions.cpp
const Eigen::Vector3f BAD_VERTEX{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(),
                                 std::numeric_limits<float>::max()};
