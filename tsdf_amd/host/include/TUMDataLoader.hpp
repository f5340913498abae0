// Replays a TUM RGB-D style directory: <dir>/ground_truth.txt with lines "<depth file stem> tx ty tz qx qy qz qw"
// and <dir>/depth/<stem>.png (16-bit, 5000 units per metre).  Same surface as the reference's TUMDataLoader.
#ifndef TSDF_AMD_HOST_TUM_DATA_LOADER_INCLUDED
#define TSDF_AMD_HOST_TUM_DATA_LOADER_INCLUDED

#include <Eigen/Dense>
#include <string>
#include <vector>

#include "DepthImage.hpp"

class TUMDataLoader {
public:
    TUMDataLoader(const std::string &directory);  // throws std::invalid_argument if the layout is missing
    ~TUMDataLoader();

    // next depth frame in millimetres and its camera pose (translation in mm); nullptr when exhausted.
    // The caller deletes the image.
    DepthImage *next(Eigen::Matrix4f &pose);

private:
    struct DATA_RECORD {
        std::string file_name;
        float data[7];
    };
    Eigen::Matrix4f to_pose(float vars[7]) const;
    void process_line(const std::string &line);
    void load_data_from(const std::string &gt_file_name);

    size_t m_current_idx;
    std::vector<struct DATA_RECORD> m_data_records;
    std::string m_directory_name;
};
#endif
