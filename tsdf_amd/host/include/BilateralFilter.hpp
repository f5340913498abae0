// Bilateral filter for 8/16-bit single-channel images, run on the GPU.  Same surface as the
// reference's BilateralFilter (src/include/BilateralFilter.hpp:12-35): filter() works IN PLACE on
// the caller's host buffer despite the const pointer.
#ifndef TSDF_AMD_HOST_BILATERAL_FILTER_INCLUDED
#define TSDF_AMD_HOST_BILATERAL_FILTER_INCLUDED

#include <cstdint>

struct tsdf_bilateral;  // C-ABI handle (include/tsdf_amd.h)

class BilateralFilter {
public:
    BilateralFilter(float sigma_colour, float sigma_space);
    ~BilateralFilter();

    void filter(const uint8_t *depth_image, int width, int height) const;
    void filter(const uint16_t *depth_image, int width, int height) const;

private:
    BilateralFilter(const BilateralFilter &);
    BilateralFilter &operator=(const BilateralFilter &);
    float m_sigma_colour;
    float m_sigma_space;
    tsdf_bilateral *m_handle;
};

#endif /* TSDF_AMD_HOST_BILATERAL_FILTER_INCLUDED */
