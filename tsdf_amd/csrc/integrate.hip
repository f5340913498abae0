// Depth-map integration for gfx950 (wave64).  Replaces TSDFVolume::integrate and
// integrate_kernel of the reference (src/TSDF/TSDFVolume.cu:861-902, 308-392).
//
// Mapping (the reference runs one thread per (y,z) with a serial x loop, so neighbouring
// lanes are 4*X bytes apart): here lane <-> x, so one wave touches 64 consecutive voxels
// = 256 contiguous bytes of the distance array and of the weight array; a 256-thread
// workgroup owns a brick of 64(x) x 4(y) x ZC(z) voxels and walks it plane by plane.
//
//   1. Brick culling (exact): lanes 0..7 project the 8 corner voxel centres of the brick.
//      A projective map sends the convex brick into the convex hull of the projected corners
//      as long as the homogeneous divisor keeps one sign over the brick, so if all corners
//      fall off the same side of the depth image (with a margin covering rounding) no voxel
//      of the brick can pass the reference's frustum test and the workgroup exits without
//      touching memory.  Bricks that straddle the camera plane are never culled (the
//      reference projects voxels behind the camera too: Q2).
//   2. Per voxel, the reference's arithmetic in its operation order (fp contraction is off):
//      world_to_pixel -> depth gather -> pixel_to_camera.z -> world_to_camera.z -> sdf ->
//      running weighted mean.  Distance and weight are loaded only under the update
//      predicate and stored with the same mask, so the algorithmic traffic is
//      16 B per updated voxel + the depth pixels gathered (L2 resident: 614 KB).
//   3. Camera matrices, intrinsics and grid geometry are kernel arguments: they are
//      wave-uniform and live in SGPRs.
#include "common.hpp"

namespace tsdf {

constexpr int kTileX = 64;  // one wave along x
constexpr int kTileY = 4;   // waves per workgroup
constexpr int kChunkZ = 16; // planes walked by one workgroup

struct Projected {
    float ix, iy, iz;  // K * cam
    float cam_z;       // row 3 of inv_pose applied to the point
    float ex, ey, ez;  // absolute error bounds of ix, iy, iz
};

// Corner projection with running error bounds; used only for the culling decision.
__device__ inline Projected project_with_bounds(float px, float py, float pz, const Mat44 &ip, const Mat33 &k) {
    const float u = 6.0e-7f;  // > 8 roundings * 2^-24
    float cx = ip.m11 * px + ip.m12 * py + ip.m13 * pz + ip.m14;
    float cy = ip.m21 * px + ip.m22 * py + ip.m23 * pz + ip.m24;
    float cz = ip.m31 * px + ip.m32 * py + ip.m33 * pz + ip.m34;
    float ecx = u * (fabsf(ip.m11 * px) + fabsf(ip.m12 * py) + fabsf(ip.m13 * pz) + fabsf(ip.m14));
    float ecy = u * (fabsf(ip.m21 * px) + fabsf(ip.m22 * py) + fabsf(ip.m23 * pz) + fabsf(ip.m24));
    float ecz = u * (fabsf(ip.m31 * px) + fabsf(ip.m32 * py) + fabsf(ip.m33 * pz) + fabsf(ip.m34));
    Projected r;
    r.cam_z = cz;
    r.ix = k.m11 * cx + k.m12 * cy + k.m13 * cz;
    r.iy = k.m21 * cx + k.m22 * cy + k.m23 * cz;
    r.iz = k.m31 * cx + k.m32 * cy + k.m33 * cz;
    r.ex = u * (fabsf(k.m11 * cx) + fabsf(k.m12 * cy) + fabsf(k.m13 * cz)) + fabsf(k.m11) * ecx + fabsf(k.m12) * ecy + fabsf(k.m13) * ecz;
    r.ey = u * (fabsf(k.m21 * cx) + fabsf(k.m22 * cy) + fabsf(k.m23 * cz)) + fabsf(k.m21) * ecx + fabsf(k.m22) * ecy + fabsf(k.m23) * ecz;
    r.ez = u * (fabsf(k.m31 * cx) + fabsf(k.m32 * cy) + fabsf(k.m33 * cz)) + fabsf(k.m31) * ecx + fabsf(k.m32) * ecy + fabsf(k.m33) * ecz;
    return r;
}

// Returns true when no voxel centre of the brick [x0,x1]x[y0,y1]x[z0,z1] (inclusive voxel
// indices) can project inside the image.  Wave-uniform result.
__device__ inline bool brick_outside_image(uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1, uint32_t z0,
                                           uint32_t z1, const Geom &g, const Mat44 &ip, const Mat33 &k,
                                           uint32_t width, uint32_t height) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t c = lane & 7u;
    uint32_t vx = (c & 1u) ? x1 : x0;
    uint32_t vy = (c & 2u) ? y1 : y0;
    uint32_t vz = (c & 4u) ? z1 : z0;
    float px = ((((int)vx + 0.5f) * g.vs.x) + g.offset_clear.x) + g.offset.x;
    float py = ((((int)vy + 0.5f) * g.vs.y) + g.offset_clear.y) + g.offset.y;
    float pz = ((((int)vz + 0.5f) * g.vs.z) + g.offset_clear.z) + g.offset.z;
    // voxel centres inside the brick deviate from the exact lattice spanned by the corners by
    // a few ulps of the coordinate; folded into the error bounds below.
    Projected p = project_with_bounds(px, py, pz, ip, k);
    float aiz = fabsf(p.iz);
    bool sign_ok = aiz > 8.0f * p.ez + 1.0e-3f;  // divisor reliably away from zero
    float qx = p.ix / p.iz;
    float qy = p.iy / p.iz;
    // error of the quotient (first order, doubled)
    float mqx = 2.0f * (p.ex + fabsf(qx) * p.ez) / aiz + 1.0e-3f;
    float mqy = 2.0f * (p.ey + fabsf(qy) * p.ez) / aiz + 1.0e-3f;
    // a voxel passes the frustum test iff round(q) in [0, W-1]  <=>  q in [-0.5, W-0.5)
    bool left = qx + mqx < -1.0f;
    bool right = qx - mqx > (float)width;
    bool top = qy + mqy < -1.0f;
    bool bottom = qy - mqy > (float)height;
    const unsigned long long m8 = 0xFFull;
    unsigned long long pos = __ballot(sign_ok && p.iz > 0.0f) & m8;
    unsigned long long neg = __ballot(sign_ok && p.iz < 0.0f) & m8;
    if (pos != m8 && neg != m8) return false;
    if ((__ballot(left) & m8) == m8) return true;
    if ((__ballot(right) & m8) == m8) return true;
    if ((__ballot(top) & m8) == m8) return true;
    if ((__ballot(bottom) & m8) == m8) return true;
    return false;
}

// A voxel whose new distance is not safely positive flags every brick whose grown region
// (brick +- kBrickGrow voxels) contains it: the bricks holding voxel v-2 .. v+2 on each axis.
__device__ inline void mark_occupied(const OccGrid &occ, uint32_t vx, uint32_t vy, uint32_t vz) {
    const uint32_t bx0 = (max(vx, (uint32_t)kBrickGrow) - kBrickGrow) >> kBrickShift;
    const uint32_t by0 = (max(vy, (uint32_t)kBrickGrow) - kBrickGrow) >> kBrickShift;
    const uint32_t bz0 = (max(vz, (uint32_t)kBrickGrow) - kBrickGrow) >> kBrickShift;
    const uint32_t bx1 = min((vx + kBrickGrow) >> kBrickShift, occ.nbx - 1);
    const uint32_t by1 = min((vy + kBrickGrow) >> kBrickShift, occ.nby - 1);
    const uint32_t bz1 = min((vz + kBrickGrow) >> kBrickShift, occ.nbz - 1);
    for (uint32_t bz = bz0; bz <= bz1; bz++)
        for (uint32_t by = by0; by <= by1; by++)
            for (uint32_t bx = bx0; bx <= bx1; bx++) {
                occ.fine[((size_t)bz * occ.nby + by) * occ.nbx + bx] = 1;
                const uint32_t s = kCoarseShift - kBrickShift;
                occ.coarse[((size_t)(bz >> s) * occ.ncy + (by >> s)) * occ.ncx + (bx >> s)] = 1;
            }
}

template <bool DEFORM, bool COUNT>
__global__ __launch_bounds__(256) void integrate_kernel(float *__restrict__ dist, float *__restrict__ weight,
                                                        const tsdf_deformation_node *__restrict__ nodes,
                                                        const Geom g, const Mat44 ip, const Mat33 k,
                                                        const Mat33 kinv, const uint32_t width,
                                                        const uint32_t height,
                                                        const uint16_t *__restrict__ depth,
                                                        unsigned long long *__restrict__ counter,
                                                        const OccGrid occ) {
    const uint32_t x0 = blockIdx.x * kTileX;
    const uint32_t y0 = blockIdx.y * kTileY;
    const uint32_t z0 = g.z_store_begin + blockIdx.z * kChunkZ;
    const uint32_t z1 = min(z0 + kChunkZ, g.z_store_end);  // exclusive
    const uint32_t vx = x0 + threadIdx.x;
    const uint32_t vy = y0 + threadIdx.y;

    if (!DEFORM) {
        uint32_t bx1 = min(x0 + kTileX, g.X) - 1, by1 = min(y0 + kTileY, g.Y) - 1;
        if (brick_outside_image(x0, bx1, y0, by1, z0, z1 - 1, g, ip, k, width, height)) return;
    }
    if (vx >= g.X || vy >= g.Y) return;

    const size_t plane = (size_t)g.X * g.Y;
    size_t idx = plane * (z0 - g.z_store_begin) + (size_t)g.X * vy + vx;

    // voxel centre, x and y parts: initialise_deformation (src/TSDF/TSDFVolume.cu:783-784) then
    // integrate_kernel's offset + translation (:343)
    float cx = 0.f, cy = 0.f;
    // partial row sums of inv_pose * (c,1): the reference evaluates ((m_i1*x + m_i2*y) + m_i3*z) + m_i4
    float r1 = 0.f, r2 = 0.f, r3 = 0.f, r4 = 0.f;
    if (!DEFORM) {
        cx = ((((int)vx + 0.5f) * g.vs.x) + g.offset_clear.x) + g.offset.x;
        cy = ((((int)vy + 0.5f) * g.vs.y) + g.offset_clear.y) + g.offset.y;
        r1 = ip.m11 * cx + ip.m12 * cy;
        r2 = ip.m21 * cx + ip.m22 * cy;
        r3 = ip.m31 * cx + ip.m32 * cy;
        r4 = ip.m41 * cx + ip.m42 * cy;
    }
    const float neg_trunc = -g.trunc;
    uint32_t updated = 0;

    for (uint32_t vz = z0; vz < z1; ++vz, idx += plane) {
        float cz;
        if (DEFORM) {
            const tsdf_deformation_node &nd = nodes[idx];
            cx = nd.translation[0] + g.offset.x;
            cy = nd.translation[1] + g.offset.y;
            cz = nd.translation[2] + g.offset.z;
            r1 = ip.m11 * cx + ip.m12 * cy;
            r2 = ip.m21 * cx + ip.m22 * cy;
            r3 = ip.m31 * cx + ip.m32 * cy;
            r4 = ip.m41 * cx + ip.m42 * cy;
        } else {
            cz = ((((int)vz + 0.5f) * g.vs.z) + g.offset_clear.z) + g.offset.z;
        }
        // world_to_pixel (src/Utilities/cuda_coordinate_transforms.cu:10-30)
        float camx = (r1 + ip.m13 * cz) + ip.m14;
        float camy = (r2 + ip.m23 * cz) + ip.m24;
        float camz = (r3 + ip.m33 * cz) + ip.m34;
        float imx = k.m11 * camx + k.m12 * camy + k.m13 * camz;
        float imy = k.m21 * camx + k.m22 * camy + k.m23 * camz;
        float imz = k.m31 * camx + k.m32 * camy + k.m33 * camz;
        int px = f2i_sat(roundf(imx / imz));
        int py = f2i_sat(roundf(imy / imz));
        bool did = false;
        // frustum test (src/TSDF/TSDFVolume.cu:349)
        if (px >= 0 && (uint32_t)px < width && py >= 0 && (uint32_t)py < height) {
            uint16_t d = depth[(uint32_t)py * width + (uint32_t)px];
            if (d > 0) {
                // pixel_to_camera(...).z (cuda_coordinate_transforms.cu:132-146)
                float ipz = kinv.m31 * px + kinv.m32 * py + kinv.m33;
                float scale = (float)d / ipz;
                float surf_z = ipz * scale;
                // world_to_camera(...).z (cuda_coordinate_transforms.cu:108-121): same numerator as camz
                float w = (r4 + ip.m43 * cz) + ip.m44;
                float voxel_cam_z = camz / w;
                float sdf = surf_z - voxel_cam_z;
                if (sdf >= neg_trunc) {
                    float tsdf = (sdf > 0) ? fminf(sdf, g.trunc) : sdf;
                    float prior_weight = weight[idx];
                    float prior_distance = dist[idx];
                    float new_weight = prior_weight + 1.0f;
                    float new_distance = ((prior_distance * prior_weight) + (tsdf * 1.0f)) / new_weight;
                    weight[idx] = new_weight;
                    dist[idx] = new_distance;
                    if (!(new_distance > occ.tau)) mark_occupied(occ, vx, vy, vz);
                    did = true;
                }
            }
        }
        if (COUNT) updated += did ? 1u : 0u;
    }
    if (COUNT) {
        // wave reduction then one atomic per wave
        for (int o = 32; o > 0; o >>= 1) updated += __shfl_down(updated, o);
        if ((threadIdx.x & 63u) == 0 && updated) atomicAdd(counter, (unsigned long long)updated);
    }
}

static int launch_integrate(tsdf_volume *v, const uint16_t *d_depth, uint32_t width, uint32_t height,
                            const float inv_pose[16], const float k[9], const float kinv[9]) {
    Mat44 ip;
    Mat33 mk, mkinv;
    memcpy(&ip, inv_pose, sizeof(ip));
    memcpy(&mk, k, sizeof(mk));
    memcpy(&mkinv, kinv, sizeof(mkinv));
    const Geom &g = v->g;
    dim3 block(kTileX, kTileY, 1);
    dim3 grid((g.X + kTileX - 1) / kTileX, (g.Y + kTileY - 1) / kTileY,
              (g.z_store_end - g.z_store_begin + kChunkZ - 1) / kChunkZ);
    if (v->counting) TSDF_HIP(hipMemsetAsync(v->counter_dev, 0, sizeof(unsigned long long), v->stream), "reset counter");
#define LAUNCH(DEF, CNT)                                                                                     \
    hipLaunchKernelGGL((integrate_kernel<DEF, CNT>), grid, block, 0, v->stream, v->dist, v->weight, v->nodes, \
                       g, ip, mk, mkinv, width, height, d_depth, v->counter_dev, v->occ)
    if (v->nodes) {
        if (v->counting) LAUNCH(true, true); else LAUNCH(true, false);
    } else {
        if (v->counting) LAUNCH(false, true); else LAUNCH(false, false);
    }
#undef LAUNCH
    TSDF_HIP(hipGetLastError(), "Integrate kernel failed");
    return TSDF_OK;
}

}  // namespace tsdf

using namespace tsdf;

extern "C" {

int tsdf_integrate_device(tsdf_volume *v, const uint16_t *device_depth, uint32_t width, uint32_t height,
                          const float pose[16], const float inv_pose[16], const float k[9], const float kinv[9]) {
    TSDF_REQUIRE(v && device_depth && inv_pose && k && kinv, "tsdf_integrate: null argument");
    TSDF_REQUIRE(width > 0 && height > 0, "tsdf_integrate: empty depth map");
    (void)pose;  // the reference passes pose to its kernel but never reads it (src/TSDF/TSDFVolume.cu:316)
    return launch_integrate(v, device_depth, width, height, inv_pose, k, kinv);
}

int tsdf_integrate(tsdf_volume *v, const uint16_t *host_depth, uint32_t width, uint32_t height,
                   const float pose[16], const float inv_pose[16], const float k[9], const float kinv[9]) {
    TSDF_REQUIRE(v && host_depth && inv_pose && k && kinv, "tsdf_integrate: null argument");
    TSDF_REQUIRE(width > 0 && height > 0, "tsdf_integrate: empty depth map");
    size_t bytes = (size_t)width * height * sizeof(uint16_t);
    if (v->depth_cap < bytes) {
        if (v->depth_buf) (void)hipFree(v->depth_buf);
        v->depth_buf = nullptr;
        v->depth_cap = 0;
        TSDF_HIP(hipMalloc((void **)&v->depth_buf, bytes), "Couldn't allocate storage for depth map");
        v->depth_cap = bytes;
    }
    TSDF_HIP(hipMemcpyAsync(v->depth_buf, host_depth, bytes, hipMemcpyHostToDevice, v->stream),
             "Failed to copy depth map to GPU");
    int rc = tsdf_integrate_device(v, v->depth_buf, width, height, pose, inv_pose, k, kinv);
    if (rc != TSDF_OK) return rc;
    TSDF_HIP(hipStreamSynchronize(v->stream), "Integrate kernel failed");
    return TSDF_OK;
}

}  // extern "C"
