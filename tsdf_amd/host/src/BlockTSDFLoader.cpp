// Text-format TSDF importer (reference: src/TSDF/BlockTSDFLoader.cpp).  A small line-driven state machine.
#include "BlockTSDFLoader.hpp"

#include <sstream>

#include "FileUtilities.hpp"

BlockTSDFLoader::BlockTSDFLoader() : m_size{0, 0, 0}, m_physical{0, 0, 0}, m_x(0), m_y(0), m_expect(Expect::GridSize) {}

BlockTSDFLoader::~BlockTSDFLoader() {}

size_t BlockTSDFLoader::column_index(uint16_t z) const {
    return ((size_t)m_size[0] * m_size[1]) * z + (size_t)m_size[0] * m_y + m_x;
}

// "<label>=<a> <b> <c>": three numbers after the first '='
template <typename T>
static void three_after_equals(const std::string &line, T out[3]) {
    std::stringstream in(line);
    std::string label;
    std::getline(in, label, '=');
    in >> out[0] >> out[1] >> out[2];
}

void BlockTSDFLoader::process_line(const std::string &line) {
    if (line.empty() || line[0] == '#' || m_expect == Expect::Ignoring) return;
    switch (m_expect) {
    case Expect::GridSize:
        three_after_equals(line, m_size);
        m_distances.assign((size_t)m_size[0] * m_size[1] * m_size[2], 0.0f);
        m_weights.assign(m_distances.size(), 0.0f);
        m_expect = Expect::PhysicalSize;
        break;
    case Expect::PhysicalSize:
        three_after_equals(line, m_physical);
        m_x = m_y = 0;
        m_expect = Expect::Distances;
        break;
    case Expect::Distances:
    case Expect::Weights: {
        std::vector<float> &target = (m_expect == Expect::Distances) ? m_distances : m_weights;
        std::stringstream in(line);
        for (uint16_t z = 0; z < m_size[2]; z++) {
            float value = 0.0f;
            in >> value;
            target[column_index(z)] = value;
        }
        if (m_expect == Expect::Distances) {
            m_expect = Expect::Weights;
        } else {
            m_expect = Expect::Distances;
            if (++m_x == m_size[0]) {   // columns run along x first, then y
                m_x = 0;
                if (++m_y == m_size[1]) m_expect = Expect::Nothing;
            }
        }
        break;
    }
    case Expect::Nothing:   // data after the last column: from here on everything is ignored
        m_expect = Expect::Ignoring;
        break;
    case Expect::Ignoring:
        break;
    }
}

bool BlockTSDFLoader::load_from_file(const std::string &file_name) {
    process_file_by_lines(file_name, [this](const std::string &line) { process_line(line); });
    // the reference demands the state right after the last column; trailing lines flip it to "ignored" there too
    return m_expect == Expect::Nothing;
}

TSDFVolume *BlockTSDFLoader::to_tsdf() const {
    TSDFVolume *volume = new TSDFVolume(m_size[0], m_size[1], m_size[2], m_physical[0], m_physical[1], m_physical[2]);
    volume->set_distance_data(m_distances.data());
    volume->set_weight_data(m_weights.data());
    return volume;
}
