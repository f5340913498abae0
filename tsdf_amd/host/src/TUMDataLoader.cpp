// Replays a TUM RGB-D directory for the kinfu loop: "<stem> tx ty tz qx qy qz qw" per line of <dir>/ground_truth.txt, the frame
// in <dir>/depth/<stem>.png.  Behaviour (what is accepted, what is thrown, the poses bit for bit) follows the reference's
// src/DataLoader/TUMDataLoader.cpp; the class surface is the one src/Tools/kinfu.cpp is written against.
#include "TUMDataLoader.hpp"

#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include "FileUtilities.hpp"

namespace {
bool is_plain_file(const std::string &path) {
    bool dir = false;
    return file_exists(path, dir) && !dir;
}
}  // namespace

TUMDataLoader::TUMDataLoader(const std::string &directory) : m_next{0}, m_root{directory} {
    bool dir = false;
    if (!file_exists(directory, dir) || !dir) throw std::invalid_argument("Directory not found " + directory);
    const std::string index = m_root + "/ground_truth.txt";
    // (the reference asks with the flag the directory test left set, and file_exists writes it for plain files and directories only:
    // anything else of that name -- a FIFO, a device -- is "not found" here, and a frame of that kind in next() is opened, as there)
    if (!file_exists(index, dir) || dir) throw std::invalid_argument("Ground truth file not found " + index);
    std::ifstream in(index);
    if (!in.is_open()) throw std::runtime_error("Failed to parse the ground truth file");
    // one record per line that is neither empty nor a '#' comment (reference :111-128); fields that fail to parse read as the
    // stream leaves them, like there
    for (std::string line; std::getline(in, line);) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream fields(line);
        std::string stem;
        Frame f = Frame();
        fields >> stem;
        for (float &value : f.tq) fields >> value;
        f.png = m_root + "/depth/" + stem + ".png";
        m_frames.push_back(f);
    }
}

TUMDataLoader::~TUMDataLoader() {}

// Unit quaternion (qx, qy, qz, qw) + translation in metres -> camera pose in millimetres (reference :47-76).  Every entry is the
// reference's float expression -- products first, their sum or difference, the doubling, then 1 - (...) on the diagonal -- so the
// poses kinfu integrates with are the same bits.
Eigen::Matrix4f TUMDataLoader::pose_of(const Frame &f) {
    const float qx = f.tq[3], qy = f.tq[4], qz = f.tq[5], qw = f.tq[6];
    const float xx = qx * qx, yy = qy * qy, zz = qz * qz;
    const float xy = qx * qy, xz = qx * qz, yz = qy * qz;
    const float wx = qw * qx, wy = qw * qy, wz = qw * qz;
    const float rot[3][3] = {{1 - 2 * (yy + zz), 2 * (xy - wz), 2 * (xz + wy)},
                             {2 * (xy + wz), 1 - 2 * (xx + zz), 2 * (yz - wx)},
                             {2 * (xz - wy), 2 * (yz + wx), 1 - 2 * (xx + yy)}};
    Eigen::Matrix4f pose = Eigen::Matrix4f::Zero();
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) pose(r, c) = rot[r][c];
        pose(r, 3) = f.tq[r] * 1000.0f;
    }
    pose(3, 3) = 1.0f;
    return pose;
}

// The next frame in millimetres (TUM stores 5000 units per metre: x 0.2) and its pose; a record whose PNG is missing is reported,
// consumed and answered with nullptr, as the reference does (:84-108).
DepthImage *TUMDataLoader::next(Eigen::Matrix4f &pose) {
    if (m_next >= m_frames.size()) return nullptr;
    const Frame &f = m_frames[m_next++];
    if (!is_plain_file(f.png)) {
        std::cerr << "Couldn't find file " << f.png << std::endl;
        return nullptr;
    }
    DepthImage *image = new DepthImage(f.png);
    image->scale_depth(0.2f);
    pose = pose_of(f);
    return image;
}
