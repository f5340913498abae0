// Internal helpers of the host library: C-ABI status -> the reference's error behaviour.
#pragma once
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>

#include "tsdf_amd.h"

namespace tsdf_host {

// TSDF_VERBOSE=1 re-enables the reference's stdout chatter (quiet by default).
inline bool verbose() {
    static const bool v = [] {
        const char *e = std::getenv("TSDF_VERBOSE");
        return e && *e && *e != '0';
    }();
    return v;
}

// Invalid arguments throw std::invalid_argument (as the reference's constructors do,
// src/TSDF/TSDFVolume.cu:435,455,720); any device failure prints and exits(-1) like
// check_cuda_error (src/Utilities/cuda_utilities.cu:5-11).
inline void check(int rc, const char *message) {
    if (rc == TSDF_OK) return;
    if (rc == TSDF_ERR_INVALID) throw std::invalid_argument(tsdf_last_error());
    std::cout << message << std::endl;
    std::cout << tsdf_last_error() << std::endl;
    std::exit(-1);
}

}  // namespace tsdf_host
