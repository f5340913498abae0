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

    // records of ground_truth.txt not consumed yet (not in the reference's class: lets a caller tell "exhausted" from "this
    // record's PNG is missing", which next() answers with nullptr alike, as the reference does)
    size_t records_left() const { return m_frames.size() - m_next; }

private:
    struct Frame {
        std::string png;   // <dir>/depth/<stem>.png
        float tq[7];       // tx ty tz (metres), qx qy qz qw
    };
    static Eigen::Matrix4f pose_of(const Frame &f);

    size_t m_next;
    std::vector<Frame> m_frames;
    std::string m_root;
};
#endif
