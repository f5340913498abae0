// An image held in host memory that can be written as (or read from) a PNG file.
// Same surface as the reference's PngWrapper (src/include/PngWrapper.hpp).
#ifndef TSDF_AMD_HOST_PNG_WRAPPER_INCLUDED
#define TSDF_AMD_HOST_PNG_WRAPPER_INCLUDED

#include <cstdint>
#include <string>

class PngWrapper {
public:
    enum PNG_TYPE { GREYSCALE_8, GREYSCALE_16, COLOUR };

    // load from file; throws std::invalid_argument on failure
    PngWrapper(const std::string &file_name, PNG_TYPE type = GREYSCALE_16);
    // copy width*height pixels of the given type (1, 2 or 3 bytes per pixel)
    PngWrapper(const uint16_t width, const uint16_t height, const uint8_t *data, PNG_TYPE);
    virtual ~PngWrapper();

    inline uint32_t width() const { return m_width; };
    inline uint32_t height() const { return m_height; };
    bool save_to(const std::string &file_name) const;

private:
    PngWrapper(const PngWrapper &);
    PngWrapper &operator=(const PngWrapper &);
    uint32_t m_width;
    uint32_t m_height;
    const uint8_t *m_data;
    PNG_TYPE m_type;
};

#endif  // TSDF_AMD_HOST_PNG_WRAPPER_INCLUDED
