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

    // Checkpoints of volumes that are not in the state the constructor leaves them in:
    // (a) offset() then clear(): the node grid carries that offset (Q1, src/TSDF/TSDFVolume.cu:783-785 + :343); after
    //     save -> load an integrate must see the same voxel centres as the original volume does;
    // (b) edited deformation nodes survive save -> load verbatim (the reference writes and reads m_deformation_nodes as is).
    {
        const unsigned m = 32;
        TSDFVolume a(TSDFVolume::UInt3{m, m, m}, TSDFVolume::Float3{3000.0f, 3000.0f, 3000.0f});
        a.offset(37.5f, -20.25f, 11.0f);
        a.clear();
        a.offset(-3.0f, 4.5f, 0.75f);              // changed again after the clear: both offsets are live now
        if (!a.save_to_file(out + "/offset.tsdf")) return 10;
        TSDFVolume b(out + "/offset.tsdf");
        tsdf_volume_info ia, ib;
        if (tsdf_volume_get_info(a.handle(), &ia) != 0 || tsdf_volume_get_info(b.handle(), &ib) != 0) return 11;
        if (ib.deformation_materialised != 0) return 12;          // a regular grid stays implicit ...
        for (int k = 0; k < 3; k++)
            if (ib.offset_at_clear[k] != ia.offset_at_clear[k] || ib.offset[k] != ia.offset[k]) return 13;   // ... with both offsets
        a.integrate(filtered.data(), W, H, *camera);
        b.integrate(filtered.data(), W, H, *camera);
        std::vector<float> da((size_t)m * m * m), db(da.size());
        if (tsdf_volume_get_distance_data(a.handle(), da.data()) != 0 || tsdf_volume_get_distance_data(b.handle(), db.data()) != 0) return 14;
        dump(out + "/offset_dist_original.f32", da.data(), da.size() * sizeof(float));
        dump(out + "/offset_dist_loaded.f32", db.data(), db.size() * sizeof(float));

        const unsigned q = 12;
        TSDFVolume c(TSDFVolume::UInt3{q, q, q}, TSDFVolume::Float3{1200.0f, 1200.0f, 1200.0f});
        std::vector<TSDFVolume::DeformationNode> nodes((size_t)q * q * q);
        if (tsdf_volume_get_deformation_planes(c.handle(), 0, q, reinterpret_cast<tsdf_deformation_node *>(nodes.data())) != 0) return 15;
        for (size_t i = 0; i < nodes.size(); i++) {          // a warp: shift and a rotation entry
            nodes[i].translation.x += 0.125f * (float)(i % 7);
            nodes[i].translation.z -= 0.5f * (float)(i % 3);
            nodes[i].rotation.y = 0.01f * (float)(i % 5);
        }
        c.set_deformation(nodes.data());
        if (!c.save_to_file(out + "/warped.tsdf")) return 16;
        TSDFVolume d(out + "/warped.tsdf");
        std::vector<TSDFVolume::DeformationNode> back(nodes.size());
        if (tsdf_volume_get_deformation_planes(d.handle(), 0, q, reinterpret_cast<tsdf_deformation_node *>(back.data())) != 0) return 17;
        dump(out + "/warp_nodes_set.f32", nodes.data(), nodes.size() * sizeof(nodes[0]));
        dump(out + "/warp_nodes_loaded.f32", back.data(), back.size() * sizeof(back[0]));
        // and the file itself carries them (last block of the file)
        std::ifstream f(out + "/warped.tsdf", std::ios::binary);
        f.seekg((std::streamoff)(68 + (size_t)q * q * q * (4 + 4 + 3)));
        std::vector<TSDFVolume::DeformationNode> in_file(nodes.size());
        f.read((char *)in_file.data(), (std::streamsize)(in_file.size() * sizeof(in_file[0])));
        if (!f) return 18;
        dump(out + "/warp_nodes_in_file.f32", in_file.data(), in_file.size() * sizeof(in_file[0]));
    }

    delete loaded;
    delete camera;
    delete volume;
    std::cout << "test_surface ok" << std::endl;
    return 0;
}
