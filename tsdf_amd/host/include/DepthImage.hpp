// 16-bit depth image held in host memory.  Same surface as the reference's DepthImage
// (src/include/DepthImage.hpp).
#ifndef TSDF_AMD_HOST_DEPTH_IMAGE_INCLUDED
#define TSDF_AMD_HOST_DEPTH_IMAGE_INCLUDED

#include <cstdint>
#include <string>

class DepthImage {
public:
    // load a 16-bit greyscale PNG; throws std::invalid_argument if it cannot be read
    DepthImage(std::string file_name);
    // copy width*height values
    DepthImage(const uint16_t width, const uint16_t height, const uint16_t *const data);
    ~DepthImage();

    void scale_depth(const float factor);       // every pixel <- (uint16)(pixel * factor)
    void truncate_depth_to(const int mm);       // pixels beyond mm <- 0
    void min_max(uint16_t &min, uint16_t &max); // over all pixels

    uint16_t width() const;
    uint16_t height() const;
    const uint16_t *data() const;

private:
    DepthImage(const DepthImage &);
    DepthImage &operator=(const DepthImage &);
    uint16_t m_width;
    uint16_t m_height;
    uint16_t *m_data;
};
#endif
