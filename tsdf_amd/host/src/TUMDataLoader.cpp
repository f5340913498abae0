// reference: src/DataLoader/TUMDataLoader.cpp
#include "TUMDataLoader.hpp"

#include <functional>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include "FileUtilities.hpp"

TUMDataLoader::TUMDataLoader(const std::string &directory) : m_current_idx{0} {
    bool is_directory = false;
    if (!(file_exists(directory, is_directory) && is_directory)) throw std::invalid_argument("Directory not found " + directory);
    m_directory_name = directory;
    const std::string gt = directory + "/ground_truth.txt";
    if (!(file_exists(gt, is_directory) && !is_directory)) throw std::invalid_argument("Ground truth file not found " + gt);
    load_data_from(gt);
}

TUMDataLoader::~TUMDataLoader() {}

// reference: :47-76 -- quaternion (qx qy qz qw) to rotation, translation metres -> millimetres
Eigen::Matrix4f TUMDataLoader::to_pose(float vars[7]) const {
    const float w = vars[6], x = vars[3], y = vars[4], z = vars[5];
    Eigen::Matrix4f pose = Eigen::Matrix4f::Zero();
    pose(0, 0) = 1 - 2 * (y * y + z * z);
    pose(0, 1) = 2 * (x * y - w * z);
    pose(0, 2) = 2 * (x * z + w * y);
    pose(1, 0) = 2 * (x * y + w * z);
    pose(1, 1) = 1 - 2 * (x * x + z * z);
    pose(1, 2) = 2 * (y * z - w * x);
    pose(2, 0) = 2 * (x * z - w * y);
    pose(2, 1) = 2 * (y * z + w * x);
    pose(2, 2) = 1 - 2 * (x * x + y * y);
    pose(0, 3) = vars[0] * 1000.0f;
    pose(1, 3) = vars[1] * 1000.0f;
    pose(2, 3) = vars[2] * 1000.0f;
    pose(3, 3) = 1.0f;
    return pose;
}

// reference: :84-108
DepthImage *TUMDataLoader::next(Eigen::Matrix4f &pose) {
    DepthImage *image = nullptr;
    if (m_current_idx < m_data_records.size()) {
        struct DATA_RECORD dr = m_data_records[m_current_idx];
        bool is_directory;
        if (file_exists(dr.file_name, is_directory) && !is_directory) {
            image = new DepthImage(dr.file_name);
            image->scale_depth(0.2f);  // TUM: 5000 units per metre -> mm
            pose = to_pose(dr.data);
        } else {
            std::cerr << "Couldn't find file " << dr.file_name << std::endl;
        }
        m_current_idx++;
    }
    return image;
}

// reference: :111-128
void TUMDataLoader::process_line(const std::string &line) {
    if (line.size() > 0 && line[0] != '#') {
        std::stringstream iss(line);
        struct DATA_RECORD dr;
        std::string stem;
        iss >> stem;
        dr.file_name = m_directory_name + "/depth/" + stem + ".png";
        for (int i = 0; i < 7; i++) iss >> dr.data[i];
        m_data_records.push_back(dr);
    }
}

void TUMDataLoader::load_data_from(const std::string &gt_file_name) {
    std::function<void(const std::string &)> f = std::bind(&TUMDataLoader::process_line, this, std::placeholders::_1);
    if (!process_file_by_lines(gt_file_name, f)) throw std::runtime_error("Failed to parse the ground truth file");
    m_current_idx = 0;
}
