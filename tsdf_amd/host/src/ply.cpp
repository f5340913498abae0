// reference: src/Utilities/ply.cpp:6-30 (same header lines and record layout)
#include "ply.hpp"

#include <fstream>
#include <iostream>

void write_to_ply(const std::string &file_name, const std::vector<float3> &vertices, const std::vector<int3> &triangles) {
    std::ofstream f{file_name};
    if (!f.is_open()) {
        std::cout << "Problem opening file for write " << file_name << std::endl;
        return;
    }
    f << "ply\nformat ascii 1.0\n";
    f << "element vertex " << vertices.size() << "\n";
    f << "property float x\nproperty float y\nproperty float z\n";
    f << "element face " << triangles.size() << "\n";
    f << "property list uchar int vertex_indices\nend_header\n";
    for (size_t v = 0; v < vertices.size(); v++) f << vertices[v].x << " " << vertices[v].y << " " << vertices[v].z << "\n";
    for (size_t t = 0; t < triangles.size(); t++) f << "3 " << triangles[t].x << " " << triangles[t].y << " " << triangles[t].z << "\n";
}
