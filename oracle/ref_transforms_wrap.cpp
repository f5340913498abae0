// C entry points around the pieces of the reference's hot path that compile from their own sources with g++ and the CUDA toolkit
// headers this image ships (the NVIDIA backend of the installed triton package carries cuda_runtime.h, vector_types.h,
// vector_functions.h): src/Utilities/cuda_coordinate_transforms.cu -- world_to_pixel, pixel_to_camera, world_to_camera
// (cuda_coordinate_transforms.cu:10-30, 108-146: SURVEY 8a rows 8-10) -- and the inline helpers of src/include/cuda_utilities.hpp
// (m3_f3_mul, f3_normalise, f3_add ...: row 20).  oracle/Makefile target "ref" compiles the reference's files where they lie
// (-x c++ -include cuda_runtime.h, as nvcc's own prelude; -include math.h for the global round / sqrt / floor nvcc declares); this file
// only wraps what they define.  Test infrastructure only: the oracle's restatement is checked against these, bit for bit.
//
// What is NOT here: integrate_kernel (src/TSDF/TSDFVolume.cu) and the ray caster (src/RayCaster/GPURaycaster.cu) include Eigen, which
// the image lacks -- unbuildable, no stand-ins; src/TSDF/TSDF_utilities.cu needs the global min / max overloads of CUDA's
// crt/math_functions.hpp, which only nvcc's front end digests.  ref_integrate_composed below is therefore NOT the reference's kernel:
// it is the loop of integrate_kernel restated (TSDFVolume.cu:337-390, the dozen lines between the calls) around the reference's OWN
// compiled world_to_pixel / pixel_to_camera / world_to_camera -- it pins every projection, rounding and gating decision of the
// oracle's integrate on reference code and leaves the blend lines (:363-381) to the reading.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include "include/cuda_coordinate_transforms.hpp"

static Mat44 m44(const float m[16]) { Mat44 r; std::memcpy(&r, m, sizeof(r)); return r; }   // column-major float[16], as TSDFVolume.cu:867-877 copies Eigen's data()
static Mat33 m33(const float m[9]) { Mat33 r; std::memcpy(&r, m, sizeof(r)); return r; }

extern "C" {

void ref_world_to_pixel(size_t n, const float *points, const float inv_pose[16], const float k[9], int32_t *pixels /* 2 per point */) {
    const Mat44 ip = m44(inv_pose);
    const Mat33 kk = m33(k);
    for (size_t i = 0; i < n; i++) {
        const int3 p = world_to_pixel(make_float3(points[3 * i], points[3 * i + 1], points[3 * i + 2]), ip, kk);
        pixels[2 * i] = p.x;
        pixels[2 * i + 1] = p.y;
    }
}

void ref_world_to_camera(size_t n, const float *points, const float inv_pose[16], float *out) {
    const Mat44 ip = m44(inv_pose);
    for (size_t i = 0; i < n; i++) {
        const float3 c = world_to_camera(make_float3(points[3 * i], points[3 * i + 1], points[3 * i + 2]), ip);
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}

void ref_pixel_to_camera(size_t n, const int32_t *pixels /* 2 per point */, const float *depth, const float kinv[9], float *out) {
    const Mat33 ki = m33(kinv);
    for (size_t i = 0; i < n; i++) {
        int3 p; p.x = pixels[2 * i]; p.y = pixels[2 * i + 1]; p.z = 1;
        const float3 c = pixel_to_camera(p, ki, depth[i]);
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}

// compute_ray_direction_at_pixel (GPURaycaster.cu:24-44) is not compilable (its file includes Eigen); its body is three lines around the
// reference's m3_f3_mul and f3_normalise (cuda_utilities.hpp:84-116), called here exactly as there: the by-value f3_normalise changes nothing (Q6)
void ref_ray_direction(size_t n, const uint16_t *pixels /* 2 per ray */, const float rot[9], const float kinv[9], float *out) {
    const Mat33 r = m33(rot), ki = m33(kinv);
    for (size_t i = 0; i < n; i++) {
        const uint16_t pix_x = pixels[2 * i], pix_y = pixels[2 * i + 1];
        float3 ray_in_cam_space{pix_x * ki.m11 + pix_y * ki.m12 + ki.m13, pix_x * ki.m21 + pix_y * ki.m22 + ki.m23, pix_x * ki.m31 + pix_y * ki.m32 + ki.m33};
        float3 ray_in_world_space = m3_f3_mul(r, ray_in_cam_space);
        f3_normalise(ray_in_world_space);
        out[3 * i] = ray_in_world_space.x; out[3 * i + 1] = ray_in_world_space.y; out[3 * i + 2] = ray_in_world_space.z;
    }
}

// The loop of integrate_kernel (TSDFVolume.cu:337-390) around the reference's compiled transforms; voxel centres as initialise_deformation
// leaves them (:783-785) for a volume whose offset was `offset_at_clear` when it was cleared, plus the offset now (:343, Q1).
// Returns the number of voxels updated.
int64_t ref_integrate_composed(float *dist, float *weight, uint32_t X, uint32_t Y, uint32_t Z, const float voxel_size[3], const float offset_at_clear[3],
                               const float offset_now[3], float trunc, const float inv_pose[16], const float k[9], const float kinv[9],
                               const uint16_t *depth, uint32_t width, uint32_t height) {
    const Mat44 ip = m44(inv_pose);
    const Mat33 kk = m33(k), ki = m33(kinv);
    const float3 offset = make_float3(offset_now[0], offset_now[1], offset_now[2]);
    int64_t updated = 0;
    for (uint32_t vz = 0; vz < Z; vz++)
        for (uint32_t vy = 0; vy < Y; vy++)
            for (uint32_t vx = 0; vx < X; vx++) {
                const size_t voxel_index = ((size_t)vz * Y + vy) * X + vx;
                float3 translation;   // :783-785
                translation.x = (((vx + 0.5f) * voxel_size[0]) + offset_at_clear[0]);
                translation.y = (((vy + 0.5f) * voxel_size[1]) + offset_at_clear[1]);
                translation.z = (((vz + 0.5f) * voxel_size[2]) + offset_at_clear[2]);
                const float3 centre_of_voxel = f3_add(offset, translation);                 // :343
                const int3 centre_of_voxel_in_pix = world_to_pixel(centre_of_voxel, ip, kk);   // :346
                if (centre_of_voxel_in_pix.x >= 0 && centre_of_voxel_in_pix.x < (int)width && centre_of_voxel_in_pix.y >= 0 &&
                    centre_of_voxel_in_pix.y < (int)height) {                                  // :349
                    const uint32_t voxel_pixel_index = centre_of_voxel_in_pix.y * width + centre_of_voxel_in_pix.x;
                    const uint16_t surface_depth = depth[voxel_pixel_index];
                    if (surface_depth > 0) {                                                   // :357
                        const float3 surface_vertex = pixel_to_camera(centre_of_voxel_in_pix, ki, surface_depth);   // :359
                        const float3 voxel_cam = world_to_camera(centre_of_voxel, ip);                               // :362
                        const float sdf = surface_vertex.z - voxel_cam.z;                                             // :363
                        if (sdf >= -trunc) {                                                   // :365
                            float tsdf;
                            if (sdf > 0) tsdf = sdf < trunc ? sdf : trunc;                     // :368-372  min(sdf, trunc)
                            else tsdf = sdf;
                            const float prior_weight = weight[voxel_index], current_weight = 1.0f;
                            const float new_weight = prior_weight + current_weight;            // :376 (the cap of :378 is commented out)
                            const float prior_distance = dist[voxel_index];
                            const float new_distance = ((prior_distance * prior_weight) + (tsdf * current_weight)) / new_weight;   // :381
                            weight[voxel_index] = new_weight;
                            dist[voxel_index] = new_distance;
                            updated++;
                        }
                    }
                }
            }
    return updated;
}

}  // extern "C"
