// TSDFVolume class surface over the C ABI.  Behaviour follows the host side of the reference's
// src/TSDF/TSDFVolume.cu (cited per method); the device work itself is in tsdf_amd/csrc.
#include "TSDFVolume.hpp"

#include <cassert>
#include <cmath>
#include <fstream>
#include <stdexcept>

#include "GPURaycaster.hpp"
#include "host_common.hpp"

using tsdf_host::check;

namespace {
const char *kBadSize = "Attempt to construct TSDFVolume with zero or negative size";
}

// reference: src/TSDF/TSDFVolume.cu:396-423
TSDFVolume::~TSDFVolume() {
    if (tsdf_host::verbose()) std::cout << "Destroying TSDFVolume" << std::endl;
    deallocate();
}

void TSDFVolume::deallocate() {
    if (m_handle) {
        tsdf_volume_destroy(m_handle);
        m_handle = nullptr;
    }
}

// reference: src/TSDF/TSDFVolume.cu:430-437.  set_size takes uint16_t, so 32-bit sizes narrow.
TSDFVolume::TSDFVolume(const UInt3 &size, const Float3 &physical_size) : m_handle{nullptr}, m_offset{0.0f, 0.0f, 0.0f} {
    if (size.x > 0 && size.y > 0 && size.z > 0 && physical_size.x > 0 && physical_size.y > 0 && physical_size.z > 0) {
        set_size(size.x, size.y, size.z, physical_size.x, physical_size.y, physical_size.z);
    } else {
        throw std::invalid_argument(kBadSize);
    }
}

// reference: src/TSDF/TSDFVolume.cu:449-457
TSDFVolume::TSDFVolume(uint16_t volume_x, uint16_t volume_y, uint16_t volume_z, float psize_x, float psize_y,
                       float psize_z)
    : m_handle{nullptr}, m_offset{0.0f, 0.0f, 0.0f} {
    if (volume_x > 0 && volume_y > 0 && volume_z > 0 && psize_x > 0 && psize_y > 0 && psize_z > 0) {
        set_size(volume_x, volume_y, volume_z, psize_x, psize_y, psize_z);
    } else {
        throw std::invalid_argument("Attempt to construct CPUTSDFVolume with zero or negative size");
    }
}

void TSDFVolume::refresh_from_handle() {
    tsdf_volume_info i;
    check(tsdf_volume_get_info(m_handle, &i), "Couldn't query TSDF");
    m_size = dim3{i.size[0], i.size[1], i.size[2]};
    m_physical_size = float3{i.physical_size[0], i.physical_size[1], i.physical_size[2]};
    m_voxel_size = float3{i.voxel_size[0], i.voxel_size[1], i.voxel_size[2]};
    m_offset = float3{i.offset[0], i.offset[1], i.offset[2]};
    m_truncation_distance = i.truncation_distance;
    m_max_weight = i.max_weight;
    m_global_translation = float3{i.global_translation[0], i.global_translation[1], i.global_translation[2]};
    m_global_rotation = float3{i.global_rotation[0], i.global_rotation[1], i.global_rotation[2]};
}

// reference: src/TSDF/TSDFVolume.cu:679-722 (offset is maintained across a resize)
void TSDFVolume::set_size(uint16_t volume_x, uint16_t volume_y, uint16_t volume_z, float psize_x, float psize_y,
                          float psize_z) {
    if ((volume_x != 0 && volume_y != 0 && volume_z != 0) && (psize_x != 0 && psize_y != 0 && psize_z != 0)) {
        float3 keep = m_offset;
        deallocate();
        check(tsdf_volume_create(volume_x, volume_y, volume_z, psize_x, psize_y, psize_z, &m_handle),
              "Couldn't allocate space for TSDF");
        if (keep.x != 0.0f || keep.y != 0.0f || keep.z != 0.0f) {
            // the reference clears with m_offset already in place, so the grid carries it
            check(tsdf_volume_set_offset(m_handle, keep.x, keep.y, keep.z), "Couldn't set offset");
            check(tsdf_volume_clear(m_handle), "Couldn't clear TSDF");
            check(tsdf_volume_synchronize(m_handle), "Couldn't clear TSDF");
        }
        refresh_from_handle();
    } else {
        throw std::invalid_argument("Attempt to set TSDF size or physical size to zero");
    }
}

// reference: src/include/TSDFVolume.hpp:139-143
void TSDFVolume::offset(float ox, float oy, float oz) {
    m_offset = float3{ox, oy, oz};
    check(tsdf_volume_set_offset(m_handle, ox, oy, oz), "Couldn't set offset");
}

// reference: src/TSDF/TSDFVolume.cu:812-845
void TSDFVolume::clear() {
    check(tsdf_volume_clear(m_handle), "Couldn't clear TSDF");
    check(tsdf_volume_synchronize(m_handle), "Couldn't clear TSDF");
}

// ---- data access (reference: src/include/TSDFVolume.hpp:175-203, src/TSDF/TSDFVolume.cu:731-757)
TSDFVolume::DeformationNode *TSDFVolume::deformation() const {
    tsdf_deformation_node *p = nullptr;
    check(tsdf_volume_deformation(m_handle, &p), "Couldn't allocate space for deformation nodes for TSDF");
    return reinterpret_cast<DeformationNode *>(p);
}

void TSDFVolume::set_deformation(DeformationNode *deformation) {
    static_assert(sizeof(DeformationNode) == sizeof(tsdf_deformation_node), "DeformationNode must be 24 bytes");
    check(tsdf_volume_set_deformation(m_handle, reinterpret_cast<const tsdf_deformation_node *>(deformation)),
          "Couldn't set deformation");
}

const float *TSDFVolume::distance_data() const {
    float *p = nullptr;
    check(tsdf_volume_distances(m_handle, &p), "Couldn't get distance data");
    return p;
}

void TSDFVolume::set_distance_data(const float *distance_data) {
    check(tsdf_volume_set_distance_data(m_handle, distance_data), "Couldn't set distance data");
}

const float *TSDFVolume::weight_data() const {
    float *p = nullptr;
    check(tsdf_volume_weights(m_handle, &p), "Couldn't get weight data");
    return p;
}

void TSDFVolume::set_weight_data(const float *weight_data) {
    check(tsdf_volume_set_weight_data(m_handle, weight_data), "Couldn't set weight data");
}

// reference: src/TSDF/TSDFVolume.cu:861-902
void TSDFVolume::integrate(const uint16_t *depth_map, uint32_t width, uint32_t height, const Camera &camera) {
    assert(depth_map);
    if (tsdf_host::verbose()) std::cout << "Integrating depth map size " << width << "x" << height << std::endl;
    // Eigen is column-major: .data() is already the Mat44 / Mat33 image the C ABI expects (:867-877)
    const Eigen::Matrix3f k = camera.k(), kinv = camera.kinv();
    check(tsdf_integrate(m_handle, depth_map, width, height, camera.pose().data(), camera.inverse_pose().data(),
                         k.data(), kinv.data()),
          "Integrate kernel failed");
    if (tsdf_host::verbose()) std::cout << "Integration finished" << std::endl;
}

// reference: src/TSDF/TSDFVolume.cu:1054-1058
void TSDFVolume::raycast(uint16_t width, uint16_t height, const Camera &camera,
                         Eigen::Matrix<float, 3, Eigen::Dynamic> &vertices,
                         Eigen::Matrix<float, 3, Eigen::Dynamic> &normals) const {
    GPURaycaster raycaster(width, height);
    raycaster.raycast(*this, camera, vertices, normals);
}

// reference: src/TSDF/TSDFVolume.cu:265-291 (upload, deformation_kernel, download), in place like there
void TSDFVolume::deform_mesh(const int num_points, float3 *points) const {
    check(tsdf_volume_deform_points(m_handle, num_points, reinterpret_cast<float *>(points)), "Deformation kernel failed");
}

// ---- file format (reference: src/TSDF/TSDFVolume.cu:911-1027 writer, :463-664 reader):
// 68-byte header {dim3 size, float3 physical, float3 offset, float trunc, float max_weight,
// float3 global_translation, float3 global_rotation} then float dist[N], float weight[N],
// uchar3 colour[N], DeformationNode[N]; little-endian, no padding.
bool TSDFVolume::save_to_file(const std::string &file_name) const {
    const size_t n = (size_t)m_size.x * m_size.y * m_size.z;
    std::vector<float> dist(n), weight(n);
    if (tsdf_volume_get_distance_data(m_handle, dist.data()) != TSDF_OK) {
        std::cout << "Failed to copy voxel data from device memory [" << tsdf_last_error() << "] " << std::endl;
        return false;
    }
    if (tsdf_volume_get_weight_data(m_handle, weight.data()) != TSDF_OK) {
        std::cout << "Failed to copy weight data from device memory [" << tsdf_last_error() << "] " << std::endl;
        return false;
    }
    std::ofstream ofs{file_name, std::ios::out | std::ios::binary};
    if (!ofs.good()) return false;
    ofs.write((const char *)&m_size, sizeof(m_size));
    ofs.write((const char *)&m_physical_size, sizeof(m_physical_size));
    ofs.write((const char *)&m_offset, sizeof(m_offset));
    ofs.write((const char *)&m_truncation_distance, sizeof(m_truncation_distance));
    ofs.write((const char *)&m_max_weight, sizeof(m_max_weight));
    ofs.write((const char *)&m_global_translation, sizeof(m_global_translation));
    ofs.write((const char *)&m_global_rotation, sizeof(m_global_rotation));
    ofs.write((const char *)dist.data(), n * sizeof(float));
    ofs.write((const char *)weight.data(), n * sizeof(float));
    // colours: never written by any kernel of the path -> all zero, streamed plane by plane
    {
        std::vector<unsigned char> zeros((size_t)m_size.x * m_size.y * 3, 0);
        for (unsigned z = 0; z < m_size.z; z++) ofs.write((const char *)zeros.data(), zeros.size());
    }
    // deformation nodes, as the reference writes m_deformation_nodes (src/TSDF/TSDFVolume.cu:1003-1018): the device array when
    // a caller has touched it (deformation() / set_deformation()), else the regular grid clear() would have written --
    // tsdf_volume_get_deformation_planes hands out either, plane by plane
    {
        std::vector<DeformationNode> plane((size_t)m_size.x * m_size.y);
        for (unsigned z = 0; z < m_size.z; z++) {
            if (tsdf_volume_get_deformation_planes(m_handle, z, 1, reinterpret_cast<tsdf_deformation_node *>(plane.data())) != TSDF_OK) {
                std::cout << "Failed to copy deformation data from device memory [" << tsdf_last_error() << "] " << std::endl;
                return false;
            }
            ofs.write((const char *)plane.data(), plane.size() * sizeof(DeformationNode));
        }
    }
    ofs.close();
    return ofs.good();
}

TSDFVolume::TSDFVolume(const std::string &file_name) : m_handle{nullptr}, m_offset{0.0f, 0.0f, 0.0f} {
    std::ifstream ifs{file_name, std::ios::in | std::ios::binary};
    std::string why;
    dim3 size;
    float3 phys, offset, gt, gr;
    float trunc = 0, max_weight = 0;
    if (!ifs.read((char *)&size, sizeof(size))) {
        why = "Couldn't load file data";
    } else if (!ifs.read((char *)&phys, sizeof(phys))) {
        why = "Couldn't load physical size";
    } else if (!(ifs.read((char *)&offset, sizeof(offset)) && ifs.read((char *)&trunc, sizeof(trunc)) &&
                 ifs.read((char *)&max_weight, sizeof(max_weight)) && ifs.read((char *)&gt, sizeof(gt)) &&
                 ifs.read((char *)&gr, sizeof(gr)))) {
        why = "Couldn't load header data";
    }
    if (why.empty()) {
        if (tsdf_host::verbose())
            std::cout << "Loading TSDF with size " << size.x << "x" << size.y << "x" << size.z << std::endl;
        if (tsdf_volume_create(size.x, size.y, size.z, phys.x, phys.y, phys.z, &m_handle) != TSDF_OK)
            why = std::string("Failed to allocate device memory for distance data: ") + tsdf_last_error();
    }
    if (why.empty()) {
        const float o[3] = {offset.x, offset.y, offset.z}, t[3] = {gt.x, gt.y, gt.z}, r[3] = {gr.x, gr.y, gr.z};
        tsdf_volume_set_header(m_handle, o, trunc, max_weight, t, r);
        const size_t n = (size_t)size.x * size.y * size.z;
        std::vector<float> buf(n);
        if (!ifs.read((char *)buf.data(), n * sizeof(float))) why = "Failed to read distance data";
        else if (tsdf_volume_set_distance_data(m_handle, buf.data()) != TSDF_OK) why = "Failed to copy distance data to device";
        if (why.empty()) {
            if (!ifs.read((char *)buf.data(), n * sizeof(float))) why = "Failed to read weight data";
            else if (tsdf_volume_set_weight_data(m_handle, buf.data()) != TSDF_OK) why = "Failed to copy weight data to device";
        }
        if (why.empty()) {
            ifs.seekg((std::streamoff)(n * 3), std::ios::cur);  // colours: unused by the path
            if (!ifs.good()) why = "Failed to read colour data";
        }
        if (why.empty()) {
            // deformation nodes: keep the grid implicit when the file holds the regular grid
            std::vector<DeformationNode> nodes(n);
            if (!ifs.read((char *)nodes.data(), n * sizeof(DeformationNode))) {
                why = "Failed to read deformation data";
            } else {
                // The reference keeps the block verbatim.  When it is the regular grid clear() wrote for some constant offset oc
                // (node = ((v + 0.5) * vs) + oc, src/TSDF/TSDFVolume.cu:783-785; oc = the offset current at that clear(), Q1)
                // the nodes stay implicit and the volume is told oc.  oc is read off node 0 and its fp32 neighbours are tried
                // too: 0.5*vs + oc was rounded, so node0 - 0.5*vs need not give oc back, while any oc that reproduces every node
                // bit for bit is as good as the original (the kernels only ever form these sums).
                const float3 vs = float3{phys.x / (float)size.x, phys.y / (float)size.y, phys.z / (float)size.z};
                const size_t stride[3] = {1, (size_t)size.x, (size_t)size.x * size.y};
                const unsigned count[3] = {size.x, size.y, size.z};
                const float vsa[3] = {vs.x, vs.y, vs.z};
                float oc[3] = {0.0f, 0.0f, 0.0f};
                bool regular = true;
                auto component = [](const DeformationNode &nd, int axis) {
                    return axis == 0 ? nd.translation.x : axis == 1 ? nd.translation.y : nd.translation.z;
                };
                for (int axis = 0; axis < 3 && regular; axis++) {
                    const float first = component(nodes[0], axis);
                    const float centre = first - (0.5f * vsa[axis]);
                    bool found = false;
                    for (int k = 0; k < 9 && !found; k++) {          // 0, +1, -1, +2, -2, ... ulps: the nearest fit wins
                        float guess = centre;
                        for (int step = 0; step < (k + 1) / 2; step++) guess = std::nextafter(guess, (k & 1) ? INFINITY : -INFINITY);
                        bool ok = true;
                        for (unsigned i = 0; i < count[axis] && ok; i++)
                            ok = component(nodes[i * stride[axis]], axis) == (((int)i + 0.5f) * vsa[axis]) + guess;
                        if (ok) {
                            oc[axis] = guess;
                            found = true;
                        }
                    }
                    regular = found;
                }
                // the three axis sequences fit: now every node (translations that do not vary across the grid, rotations zero)
                size_t i = 0;
                for (unsigned z = 0; z < size.z && regular; z++)
                    for (unsigned y = 0; y < size.y && regular; y++)
                        for (unsigned x = 0; x < size.x; x++, i++) {
                            const DeformationNode &nd = nodes[i];
                            if (nd.translation.x != (((int)x + 0.5f) * vs.x) + oc[0] ||
                                nd.translation.y != (((int)y + 0.5f) * vs.y) + oc[1] ||
                                nd.translation.z != (((int)z + 0.5f) * vs.z) + oc[2] ||
                                nd.rotation.x != 0.0f || nd.rotation.y != 0.0f || nd.rotation.z != 0.0f) {
                                regular = false;
                                break;
                            }
                        }
                if (regular && tsdf_volume_set_offset_at_clear(m_handle, oc) != TSDF_OK) why = "Failed to restore the node offset";
                if (!regular &&
                    tsdf_volume_set_deformation(m_handle, reinterpret_cast<const tsdf_deformation_node *>(nodes.data())) != TSDF_OK)
                    why = "Failed to copy deformation data to device";
            }
        }
    }
    if (!why.empty()) {
        deallocate();
        throw std::invalid_argument("Failed to load TSDF " + file_name + " " + why);
    }
    refresh_from_handle();
}

// reference: src/TSDF/TSDFVolume.cu:1035-1047 -- a stub there too
bool TSDFVolume::load_from_file(const std::string &file_name) {
    (void)file_name;
    std::cout << "Not yet implemented: load_from_file" << std::endl;
    return false;
}
