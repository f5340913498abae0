// Exercises the C++ class surface (the drop-in boundary) exactly the way src/Tools/kinfu.cpp of the reference
// does -- TSDFVolume(UInt3, Float3), Camera::default_depth_camera, set_pose, integrate, raycast, BilateralFilter,
// save_to_file / file constructor, extract_surface -- and dumps raw results for the Python test to compare with
// what the Python mirror (same C ABI) and the CPU oracle produce.
//
//   test_surface <depth.u16> <pose.f32 (16, column-major)> <out_dir> [grid]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <vector>

#include "BilateralFilter.hpp"
#include "BlockTSDFLoader.hpp"
#include "ICP_CUDA/ICPOdometry.h"
#include "GPURaycaster.hpp"
#include "MarkAndSweepMC.hpp"
#include "TSDFVolume.hpp"
#include "tsdf_amd.h"

static void dump(const std::string &path, const void *p, size_t bytes) {
    std::ofstream f(path, std::ios::binary);
    f.write((const char *)p, (std::streamsize)bytes);
}

int main(int argc, char **argv) {
    if (argc < 4) {
        std::cerr << "usage: test_surface depth.u16 pose.f32 out_dir [grid]" << std::endl;
        return 2;
    }
    const int W = 640, H = 480;
    const unsigned n = argc > 4 ? (unsigned)atoi(argv[4]) : 64;
    std::vector<uint16_t> depth(W * H);
    {
        std::ifstream f(argv[1], std::ios::binary);
        f.read((char *)depth.data(), depth.size() * 2);
        if (!f) return 3;
    }
    Eigen::Matrix4f pose;
    {
        std::ifstream f(argv[2], std::ios::binary);
        f.read((char *)pose.data(), 16 * sizeof(float));
        if (!f) return 3;
    }
    const std::string out = argv[3];

    // invalid sizes throw std::invalid_argument, like the reference
    bool threw = false;
    try {
        TSDFVolume bad(TSDFVolume::UInt3{0, 8, 8}, TSDFVolume::Float3{1.0f, 1.0f, 1.0f});
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    if (!threw) return 4;

    // 8-bit-exact bilateral filter, in place on the caller's buffer
    std::vector<uint16_t> filtered = depth;
    BilateralFilter filter(30.0f, 4.5f);
    filter.filter(filtered.data(), W, H);
    dump(out + "/filtered.u16", filtered.data(), filtered.size() * 2);

    TSDFVolume *volume = new TSDFVolume(TSDFVolume::UInt3{n, n, n}, TSDFVolume::Float3{3000.0f, 3000.0f, 3000.0f});
    Camera *camera = Camera::default_depth_camera();
    camera->set_pose(pose);
    volume->integrate(filtered.data(), W, H, *camera);

    Eigen::Matrix<float, 3, Eigen::Dynamic> vertices, normals;
    volume->raycast(W, H, *camera, vertices, normals);
    if (vertices.cols() != W * H || normals.cols() != W * H) return 5;
    dump(out + "/vertices.f32", vertices.data(), (size_t)W * H * 3 * sizeof(float));
    dump(out + "/normals.f32", normals.data(), (size_t)W * H * 3 * sizeof(float));

    // checkpoint round trip through the reference's file format
    const std::string file = out + "/volume.tsdf";
    if (!volume->save_to_file(file)) return 6;
    TSDFVolume *loaded = new TSDFVolume(file);
    if (loaded->size().x != n || loaded->truncation_distance() != volume->truncation_distance()) return 7;
    Eigen::Matrix<float, 3, Eigen::Dynamic> v2, n2;
    GPURaycaster(W, H).raycast(*loaded, *camera, v2, n2);
    dump(out + "/vertices_loaded.f32", v2.data(), (size_t)W * H * 3 * sizeof(float));

    DepthImage *rendered = GPURaycaster(W, H).render_to_depth_image(*volume, *camera);
    dump(out + "/rendered_depth.u16", rendered->data(), (size_t)W * H * 2);

    // ICP between the input depth (model) and the depth rendered from the volume, the way src/Tools/tsdf_icp.cpp does
    {
        ICPOdometry icp(W, H, 331.0f, 234.6f, 591.1f, 590.1f);
        icp.initICPModel(filtered.data());
        icp.initICP((unsigned short *)rendered->data());
        Sophus::SE3d mesh_to_depth_transform;
        icp.getIncrementalTransformation(mesh_to_depth_transform, 224, 96);
        Eigen::Matrix<double, 4, 4> m = mesh_to_depth_transform.matrix();
        dump(out + "/icp_transform.f64", m.data(), 16 * sizeof(double));
        float stats[2] = {icp.lastError, icp.lastInliers};
        dump(out + "/icp_stats.f32", stats, sizeof(stats));
        Eigen::Matrix4f pose_f = mesh_to_depth_transform.cast<float>().matrix();
        Eigen::Vector3f trans = pose_f.topRightCorner(3, 1);
        std::cout << "icp trans : " << trans[0] << " " << trans[1] << " " << trans[2] << " inliers " << icp.lastInliers << std::endl;
    }
    delete rendered;

    std::vector<float3> mesh_vertices;
    std::vector<int3> mesh_triangles;
    extract_surface(volume, mesh_vertices, mesh_triangles);
    std::cout << "mesh " << mesh_vertices.size() << " vertices " << mesh_triangles.size() << " triangles" << std::endl;
    dump(out + "/mesh_vertices.f32", mesh_vertices.data(), mesh_vertices.size() * sizeof(float3));

    // text-format import (BlockTSDFLoader), when the driver left a file: load -> to_tsdf -> distances back
    {
        std::ifstream probe(out + "/block.txt");
        if (probe.good()) {
            BlockTSDFLoader loader;
            if (!loader.load_from_file(out + "/block.txt")) return 8;
            TSDFVolume *imported = loader.to_tsdf();
            std::vector<float> d((size_t)imported->size().x * imported->size().y * imported->size().z);
            if (tsdf_volume_get_distance_data(imported->handle(), d.data()) != 0) return 9;
            dump(out + "/block_distances.f32", d.data(), d.size() * sizeof(float));
            float dims[4] = {(float)imported->size().x, (float)imported->size().y, (float)imported->size().z, imported->physical_size().x};
            dump(out + "/block_dims.f32", dims, sizeof(dims));
            delete imported;
        }
    }

    delete loaded;
    delete camera;
    delete volume;
    std::cout << "test_surface ok" << std::endl;
    return 0;
}
