// Importer of the text ("block") TSDF format.  Same class as the reference's src/include/BlockTSDFLoader.hpp:
//
//   # comment lines and empty lines are skipped
//   <anything>=<size x> <size y> <size z>            voxel grid size
//   <anything>=<physical x> <physical y> <physical z>
//   then, for y = 0..size_y-1, for x = 0..size_x-1 (x fastest), two lines each:
//     size_z distances of the column (x, y, z = 0..size_z-1)
//     size_z weights of the same column
//
// (reference: src/TSDF/BlockTSDFLoader.cpp:24-100); anything after the last column is ignored.
#ifndef TSDF_AMD_HOST_BLOCK_TSDF_LOADER_INCLUDED
#define TSDF_AMD_HOST_BLOCK_TSDF_LOADER_INCLUDED

#include <cstdint>
#include <string>
#include <vector>

#include "TSDFVolume.hpp"

class BlockTSDFLoader {
public:
    BlockTSDFLoader();
    ~BlockTSDFLoader();

    // true when the file held the two header lines and every column
    bool load_from_file(const std::string &file_name);
    void process_line(const std::string &line);

    // a new volume of the file's size holding its distances and weights (the caller owns it)
    TSDFVolume *to_tsdf() const;

    // what was read (tests, tools)
    uint16_t size_x() const { return m_size[0]; }
    uint16_t size_y() const { return m_size[1]; }
    uint16_t size_z() const { return m_size[2]; }
    const float *physical_size() const { return m_physical; }
    const std::vector<float> &distances() const { return m_distances; }
    const std::vector<float> &weights() const { return m_weights; }

private:
    enum class Expect { GridSize, PhysicalSize, Distances, Weights, Nothing, Ignoring };
    size_t column_index(uint16_t z) const;
    std::vector<float> m_distances, m_weights;
    uint16_t m_size[3];
    float m_physical[3];
    uint16_t m_x, m_y;
    Expect m_expect;
};

#endif
