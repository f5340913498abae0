// make_float3 & co. live in vector_types.h here; kept so `#include "vector_functions.h"` resolves.
#ifndef TSDF_AMD_VECTOR_FUNCTIONS_H
#define TSDF_AMD_VECTOR_FUNCTIONS_H
#include "vector_types.h"
#endif
