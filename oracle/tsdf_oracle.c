/*
 * tsdf_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see tsdf_oracle.h).
 *
 * Plain-C restatement of the reference hot path.  Build with
 *     gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp
 * so every fp32 operation is a separately rounded IEEE operation in the order the
 * reference writes it.  "ref:" comments cite /root/reference paths.
 */
#include "tsdf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* column-major accessors: Mat44 = {m11,m21,m31,m41, m12,...}  ref: src/include/cuda_utilities.hpp:12-23 */
#define M4(m, r, c) ((m)[((c)-1) * 4 + ((r)-1)])
#define M3(m, r, c) ((m)[((c)-1) * 3 + ((r)-1)])

/*
 * float -> int conversion with the semantics of the reference's target (CUDA cvt.rzi.s32.f32):
 * saturating, NaN -> 0.  ref: src/Utilities/cuda_coordinate_transforms.cu:25-26 assigns
 * round(float) to an int.  (The C cast would be undefined outside the int range.)
 */
static inline int f2i_sat(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

/* ------------------------------------------------------------------------------------ */
/* volume geometry                                                                        */
/* ------------------------------------------------------------------------------------ */

/* ref: src/TSDF/TSDFVolume.cu:686-693 ; f3_div_elem(float3,dim3) src/include/cuda_utilities.hpp:76-79 ;
 * f3_norm src/include/cuda_utilities.hpp:99-102 */
void orc_geom_init(orc_geom *g, uint32_t X, uint32_t Y, uint32_t Z, float px, float py, float pz) {
    g->dims[0] = X; g->dims[1] = Y; g->dims[2] = Z;
    g->phys[0] = px; g->phys[1] = py; g->phys[2] = pz;
    g->vs[0] = px / (float)X;
    g->vs[1] = py / (float)Y;
    g->vs[2] = pz / (float)Z;
    float n = sqrtf(g->vs[0] * g->vs[0] + g->vs[1] * g->vs[1] + g->vs[2] * g->vs[2]);
    g->trunc = 1.1f * n;
    for (int i = 0; i < 3; i++) { g->offset[i] = 0.0f; g->offset_at_clear[i] = 0.0f; }
}

/* ref: src/TSDF/TSDFVolume.cu:812-830 (set_memory_to_value x2) */
void orc_clear(float *dist, float *weight, size_t n, float trunc) {
    for (size_t i = 0; i < n; i++) {
        weight[i] = 0.0f;
        dist[i] = trunc;
    }
}

/* ref: src/TSDF/TSDFVolume.cu:783-785 then :343 (f3_add(offset, translation) = translation + offset) */
void orc_voxel_centre(const orc_geom *g, int vx, int vy, int vz, float out[3]) {
    const float *vs = g->vs, *oc = g->offset_at_clear, *on = g->offset;
    float tx = ((vx + 0.5f) * vs[0]) + oc[0];
    float ty = ((vy + 0.5f) * vs[1]) + oc[1];
    float tz = ((vz + 0.5f) * vs[2]) + oc[2];
    out[0] = tx + on[0];
    out[1] = ty + on[1];
    out[2] = tz + on[2];
}

/* ------------------------------------------------------------------------------------ */
/* integrate                                                                              */
/* ------------------------------------------------------------------------------------ */

/* ref: src/Utilities/cuda_coordinate_transforms.cu:10-30 (no z>0 test: Q2; roundf: Q3) */
void orc_world_to_pixel(const float p[3], const float ip[16], const float k[9], int pix[2]) {
    float cx = M4(ip, 1, 1) * p[0] + M4(ip, 1, 2) * p[1] + M4(ip, 1, 3) * p[2] + M4(ip, 1, 4);
    float cy = M4(ip, 2, 1) * p[0] + M4(ip, 2, 2) * p[1] + M4(ip, 2, 3) * p[2] + M4(ip, 2, 4);
    float cz = M4(ip, 3, 1) * p[0] + M4(ip, 3, 2) * p[1] + M4(ip, 3, 3) * p[2] + M4(ip, 3, 4);
    float ix = M3(k, 1, 1) * cx + M3(k, 1, 2) * cy + M3(k, 1, 3) * cz;
    float iy = M3(k, 2, 1) * cx + M3(k, 2, 2) * cy + M3(k, 2, 3) * cz;
    float iz = M3(k, 3, 1) * cx + M3(k, 3, 2) * cy + M3(k, 3, 3) * cz;
    pix[0] = f2i_sat(roundf(ix / iz));
    pix[1] = f2i_sat(roundf(iy / iz));
}

/* ref: src/Utilities/cuda_coordinate_transforms.cu:108-121 (the full 4 x 4 product, then the divide by w) */
void orc_world_to_camera(const float p[3], const float ip[16], float out[3]) {
    float cx = (M4(ip, 1, 1) * p[0]) + (M4(ip, 1, 2) * p[1]) + (M4(ip, 1, 3) * p[2]) + M4(ip, 1, 4);
    float cy = (M4(ip, 2, 1) * p[0]) + (M4(ip, 2, 2) * p[1]) + (M4(ip, 2, 3) * p[2]) + M4(ip, 2, 4);
    float cz = (M4(ip, 3, 1) * p[0]) + (M4(ip, 3, 2) * p[1]) + (M4(ip, 3, 3) * p[2]) + M4(ip, 3, 4);
    float w = (M4(ip, 4, 1) * p[0]) + (M4(ip, 4, 2) * p[1]) + (M4(ip, 4, 3) * p[2]) + M4(ip, 4, 4);
    out[0] = cx / w;
    out[1] = cy / w;
    out[2] = cz / w;
}

/* ref: src/Utilities/cuda_coordinate_transforms.cu:132-146 (f3_mul_scalar: vec * scalar, cuda_utilities.hpp:56-58) */
void orc_pixel_to_camera(const int pix[2], float depth, const float kinv[9], float out[3]) {
    float ipx = M3(kinv, 1, 1) * pix[0] + M3(kinv, 1, 2) * pix[1] + M3(kinv, 1, 3);
    float ipy = M3(kinv, 2, 1) * pix[0] + M3(kinv, 2, 2) * pix[1] + M3(kinv, 2, 3);
    float ipz = M3(kinv, 3, 1) * pix[0] + M3(kinv, 3, 2) * pix[1] + M3(kinv, 3, 3);
    float scale = depth / ipz;
    out[0] = ipx * scale;
    out[1] = ipy * scale;
    out[2] = ipz * scale;
}

/* the same, many at a time (tests against oracle/_ref/libref_transforms.so and tests/golden/ref_transforms.npz) */
void orc_world_to_pixel_n(size_t n, const float *points, const float ip[16], const float k[9], int *pixels) {
    for (size_t i = 0; i < n; i++) orc_world_to_pixel(points + 3 * i, ip, k, pixels + 2 * i);
}
void orc_world_to_camera_n(size_t n, const float *points, const float ip[16], float *out) {
    for (size_t i = 0; i < n; i++) orc_world_to_camera(points + 3 * i, ip, out + 3 * i);
}
void orc_pixel_to_camera_n(size_t n, const int *pixels, const float *depth, const float kinv[9], float *out) {
    for (size_t i = 0; i < n; i++) orc_pixel_to_camera(pixels + 2 * i, depth[i], kinv, out + 3 * i);
}
void orc_ray_direction_n(size_t n, const uint16_t *pixels, const float rot[9], const float kinv[9], float *out);

/* ref: src/TSDF/TSDFVolume.cu:308-392 */
int64_t orc_integrate(float *dist, float *weight, const orc_geom *g, const float ip[16],
                      const float k[9], const float kinv[9], const uint16_t *depth,
                      uint32_t width, uint32_t height, const float *translation,
                      uint32_t z_store_begin, uint32_t z_begin, uint32_t z_end, int nthreads) {
    const uint32_t X = g->dims[0], Y = g->dims[1];
    const float *on = g->offset;
    const float trunc = g->trunc;
    int64_t updated = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : updated)
    for (int64_t vz = (int64_t)z_begin; vz < (int64_t)z_end; vz++) {
        for (uint32_t vy = 0; vy < Y; vy++) {
            size_t voxel_index = (size_t)X * Y * (size_t)(vz - z_store_begin) + (size_t)X * vy;
            for (uint32_t vx = 0; vx < X; vx++, voxel_index++) {
                float c[3];
                if (translation) {
                    /* :343  centre = offset + node.translation */
                    c[0] = translation[voxel_index * 3 + 0] + on[0];
                    c[1] = translation[voxel_index * 3 + 1] + on[1];
                    c[2] = translation[voxel_index * 3 + 2] + on[2];
                } else {
                    orc_voxel_centre(g, (int)vx, (int)vy, (int)vz, c);
                }
                /* :346 world_to_pixel */
                int pix[2];
                orc_world_to_pixel(c, ip, k, pix);
                /* :349 frustum test */
                if (pix[0] >= 0 && (uint32_t)pix[0] < width && pix[1] >= 0 && (uint32_t)pix[1] < height) {
                    uint32_t pidx = (uint32_t)pix[1] * width + (uint32_t)pix[0];
                    uint16_t d = depth[pidx];
                    if (d > 0) {
                        /* :359 pixel_to_camera (cuda_coordinate_transforms.cu:132-146); only .z is used */
                        float ipz = M3(kinv, 3, 1) * pix[0] + M3(kinv, 3, 2) * pix[1] + M3(kinv, 3, 3);
                        float scale = (float)d / ipz;
                        float surf_z = ipz * scale; /* f3_mul_scalar: vec.z * scalar */
                        /* :362 world_to_camera (cuda_coordinate_transforms.cu:108-121); only .z is used */
                        float cz = (M4(ip, 3, 1) * c[0]) + (M4(ip, 3, 2) * c[1]) + (M4(ip, 3, 3) * c[2]) + M4(ip, 3, 4);
                        float w = (M4(ip, 4, 1) * c[0]) + (M4(ip, 4, 2) * c[1]) + (M4(ip, 4, 3) * c[2]) + M4(ip, 4, 4);
                        cz /= w;
                        float sdf = surf_z - cz; /* :363 */
                        if (sdf >= -trunc) {     /* :365  (Q5) */
                            float tsdf;
                            if (sdf > 0) tsdf = fminf(sdf, trunc); else tsdf = sdf;
                            float prior_weight = weight[voxel_index];
                            float current_weight = 1.0f;
                            float new_weight = prior_weight + current_weight; /* cap commented out: Q4 */
                            float prior_distance = dist[voxel_index];
                            float new_distance = ((prior_distance * prior_weight) + (tsdf * current_weight)) / new_weight;
                            weight[voxel_index] = new_weight;
                            dist[voxel_index] = new_distance;
                            updated++;
                        }
                    }
                }
            }
        }
    }
    return updated;
}

/* ------------------------------------------------------------------------------------ */
/* raycast                                                                                */
/* ------------------------------------------------------------------------------------ */

/* ref: src/RayCaster/GPURaycaster.cu:24-44.  f3_normalise takes its argument by value
 * (src/include/cuda_utilities.hpp:87-93) so the direction is NOT normalised (Q6). */
void orc_ray_direction(uint16_t px, uint16_t py, const float rot[9], const float kinv[9], float dir[3]) {
    float rx = px * M3(kinv, 1, 1) + py * M3(kinv, 1, 2) + M3(kinv, 1, 3);
    float ry = px * M3(kinv, 2, 1) + py * M3(kinv, 2, 2) + M3(kinv, 2, 3);
    float rz = px * M3(kinv, 3, 1) + py * M3(kinv, 3, 2) + M3(kinv, 3, 3);
    /* m3_f3_mul  src/include/cuda_utilities.hpp:110-116 */
    dir[0] = M3(rot, 1, 1) * rx + M3(rot, 1, 2) * ry + M3(rot, 1, 3) * rz;
    dir[1] = M3(rot, 2, 1) * rx + M3(rot, 2, 2) * ry + M3(rot, 2, 3) * rz;
    dir[2] = M3(rot, 3, 1) * rx + M3(rot, 3, 2) * ry + M3(rot, 3, 3) * rz;
}

/* ref: src/RayCaster/GPURaycaster.cu:138-181 */
void orc_ray_direction_n(size_t n, const uint16_t *pixels, const float rot[9], const float kinv[9], float *out) {
    for (size_t i = 0; i < n; i++) orc_ray_direction(pixels[2 * i], pixels[2 * i + 1], rot, kinv, out + 3 * i);
}

static int can_intersect_in_dimension(float space_min, float space_max, float origin, float direction,
                                      float *near_t, float *far_t) {
    int can_intersect = 1;
    if (direction == 0) {
        if (origin < space_min || origin > space_max) can_intersect = 0;
    } else {
        float dmin = (space_min - origin) / direction;
        float dmax = (space_max - origin) / direction;
        if (dmin > dmax) { float t = dmin; dmin = dmax; dmax = t; }
        if (dmin > *near_t) *near_t = dmin;
        if (dmax < *far_t) *far_t = dmax;
        if (*near_t > *far_t) can_intersect = 0;
        else if (*far_t < 0) can_intersect = 0;
    }
    return can_intersect;
}

/* ref: src/RayCaster/GPURaycaster.cu:197-251 */
int orc_ray_box(const float o[3], const float d[3], const float smin[3], const float smax[3],
                float *near_t, float *far_t) {
    int intersects = 0;
    if (o[0] >= smin[0] && o[0] <= smax[0] && o[1] >= smin[1] && o[1] <= smax[1] && o[2] >= smin[2] && o[2] <= smax[2]) {
        *near_t = 0;
        float x_t = NAN, y_t = NAN, z_t = NAN;
        if (d[0] > 0) x_t = (smax[0] - o[0]) / d[0]; else if (d[0] < 0) x_t = (smin[0] - o[0]) / d[0];
        if (d[1] > 0) y_t = (smax[1] - o[1]) / d[1]; else if (d[1] < 0) y_t = (smin[1] - o[1]) / d[1];
        if (d[2] > 0) z_t = (smax[2] - o[2]) / d[2]; else if (d[2] < 0) z_t = (smin[2] - o[2]) / d[2];
        if (x_t < y_t) {
            if (x_t < z_t) *far_t = x_t; else *far_t = z_t;
        } else {
            if (y_t < z_t) *far_t = y_t; else *far_t = z_t;
        }
        intersects = 1;
    } else {
        *near_t = -INFINITY;
        *far_t = INFINITY;
        if (can_intersect_in_dimension(smin[0], smax[0], o[0], d[0], near_t, far_t) &&
            can_intersect_in_dimension(smin[1], smax[1], o[1], d[1], near_t, far_t) &&
            can_intersect_in_dimension(smin[2], smax[2], o[2], d[2], near_t, far_t)) {
            intersects = 1;
        }
    }
    return intersects;
}

/* tsdf_value_at: ref src/TSDF/TSDF_utilities.cu:29-37 (uint16_t coords, clamp to [0,size-1]) */
static inline float tsdf_value_at(int xi, int yi, int zi, const float *dist, const uint32_t dims[3],
                                  uint32_t z_store_begin, uint8_t *touched) {
    uint16_t x = (uint16_t)xi, y = (uint16_t)yi, z = (uint16_t)zi;
    uint32_t cx = x, cy = y, cz = z;
    if (cx > dims[0] - 1) cx = dims[0] - 1;
    if (cy > dims[1] - 1) cy = dims[1] - 1;
    if (cz > dims[2] - 1) cz = dims[2] - 1;
    if (touched) touched[(size_t)dims[0] * dims[1] * cz + (size_t)dims[0] * cy + cx] = 1;
    size_t idx = (size_t)dims[0] * dims[1] * (size_t)(cz - z_store_begin) + (size_t)dims[0] * cy + cx;
    return dist[idx];
}

/*
 * ref: src/RayCaster/GPURaycaster.cu:53-124 (+ voxel_for_point TSDF_utilities.cu:45-53,
 * centre_of_voxel_at TSDF_utilities.cu:10-17 with its default zero offset).
 * lower_z_out receives lower.z (for the slab ownership rule); own_lo/own_hi: when
 * own_lo <= own_hi the taps are only fetched if lower.z is in [own_lo, own_hi) and
 * *owned says whether they were.
 */
static inline float trilinear_core(const float p[3], const uint32_t dims[3], const float vs[3],
                                   const float *dist, uint32_t z_store_begin, uint8_t *touched,
                                   int slab, uint32_t own_lo, uint32_t own_hi, int *owned) {
    float max_x = dims[0] * vs[0], max_y = dims[1] * vs[1], max_z = dims[2] * vs[2];
    float ax = p[0], ay = p[1], az = p[2];
    if (p[0] >= max_x) ax = max_x - (vs[0] / 10.0f);
    if (p[1] >= max_y) ay = max_y - (vs[1] / 10.0f);
    if (p[2] >= max_z) az = max_z - (vs[2] / 10.0f);
    if (p[0] < 0.0f) ax = 0.0f;
    if (p[1] < 0.0f) ay = 0.0f;
    if (p[2] < 0.0f) az = 0.0f;

    int vx = f2i_sat(floorf(ax / vs[0]));
    int vy = f2i_sat(floorf(ay / vs[1]));
    int vz = f2i_sat(floorf(az / vs[2]));

    if (owned) *owned = 1;
    if (vx < 0 || vy < 0 || vz < 0 || (uint32_t)vx >= dims[0] || (uint32_t)vy >= dims[1] || (uint32_t)vz >= dims[2]) {
        return NAN; /* :77-80 (the printf is dropped) */
    }

    float ccx = (vx + 0.5f) * vs[0] + 0.0f;
    float ccy = (vy + 0.5f) * vs[1] + 0.0f;
    float ccz = (vz + 0.5f) * vs[2] + 0.0f;

    int lx = (p[0] < ccx) ? vx - 1 : vx; /* unclamped point: Q10 */
    int ly = (p[1] < ccy) ? vy - 1 : vy;
    int lz = (p[2] < ccz) ? vz - 1 : vz;
    if (lx < 0) lx = 0;
    if (ly < 0) ly = 0;
    if (lz < 0) lz = 0;

    if (slab) {
        if (!((uint32_t)lz >= own_lo && (uint32_t)lz < own_hi)) {
            *owned = 0;
            return NAN;
        }
    }

    float lcx = (lx + 0.5f) * vs[0] + 0.0f;
    float lcy = (ly + 0.5f) * vs[1] + 0.0f;
    float lcz = (lz + 0.5f) * vs[2] + 0.0f;
    float u = (p[0] - lcx) / vs[0];
    float v = (p[1] - lcy) / vs[1];
    float w = (p[2] - lcz) / vs[2];

    float c000 = tsdf_value_at(lx + 0, ly + 0, lz + 0, dist, dims, z_store_begin, touched);
    float c001 = tsdf_value_at(lx + 0, ly + 0, lz + 1, dist, dims, z_store_begin, touched);
    float c010 = tsdf_value_at(lx + 0, ly + 1, lz + 0, dist, dims, z_store_begin, touched);
    float c011 = tsdf_value_at(lx + 0, ly + 1, lz + 1, dist, dims, z_store_begin, touched);
    float c100 = tsdf_value_at(lx + 1, ly + 0, lz + 0, dist, dims, z_store_begin, touched);
    float c101 = tsdf_value_at(lx + 1, ly + 0, lz + 1, dist, dims, z_store_begin, touched);
    float c110 = tsdf_value_at(lx + 1, ly + 1, lz + 0, dist, dims, z_store_begin, touched);
    float c111 = tsdf_value_at(lx + 1, ly + 1, lz + 1, dist, dims, z_store_begin, touched);

    float interpolated = c000 * (1 - u) * (1 - v) * (1 - w) +
                         c001 * (1 - u) * (1 - v) * w +
                         c010 * (1 - u) * v * (1 - w) +
                         c011 * (1 - u) * v * w +
                         c100 * u * (1 - v) * (1 - w) +
                         c101 * u * (1 - v) * w +
                         c110 * u * v * (1 - w) +
                         c111 * u * v * w;
    return interpolated;
}

float orc_trilinear(const float point[3], const uint32_t dims[3], const float vs[3], const float *dist) {
    return trilinear_core(point, dims, vs, dist, 0, NULL, 0, 0, 0, NULL);
}

/*
 * One ray.  ref: src/RayCaster/GPURaycaster.cu:265-377.
 * slab == 0: full semantics, writes vertex (NaN triple on miss), returns #samples.
 * slab == 1: evaluates only owned samples; out[0] = k of first owned sample with tsdf<=0
 *            (+inf if none), out[1] = the refined ray parameter t of :338-341 (NaN if none); the
 *            vertex is formed from t by orc_merge_hits (:344-347), as every rank can.
 */
static int march_ray(int imx, int imy, const float *dist, const uint32_t dims[3], const float vs[3],
                     const float space_min[3], const float space_max[3], float trunc,
                     const float origin[3], const float rot[9], const float kinv[9],
                     uint32_t z_store_begin, uint8_t *touched, int slab, uint32_t own_lo,
                     uint32_t own_hi, float out[4]) {
    float dir[3];
    orc_ray_direction((uint16_t)imx, (uint16_t)imy, rot, kinv, dir);
    float near_t, far_t;
    int intersects = orc_ray_box(origin, dir, space_min, space_max, &near_t, &far_t);
    float ix = NAN, iy = NAN, iz = NAN;
    float hit_k = INFINITY, hit_t = NAN;
    int samples = 0;
    if (intersects) {
        /* :306 start = (origin + near*dir) - space_min */
        float sx = ((near_t * dir[0]) + origin[0]) - space_min[0];
        float sy = ((near_t * dir[1]) + origin[1]) - space_min[1];
        float sz = ((near_t * dir[2]) + origin[2]) - space_min[2];
        int done = 0;
        const float previous_tsdf = trunc; /* inner "float tsdf" shadows the outer one: Q7 */
        float t = 0;
        float max_t = far_t - near_t;
        int count = 0;
        float step_size = (float)((double)trunc * 0.05); /* :324 double literal */
        while (!done) {
            float p[3];
            p[0] = (t * dir[0]) + sx;
            p[1] = (t * dir[1]) + sy;
            p[2] = (t * dir[2]) + sz;
            int owned = 1;
            float tsdf = trilinear_core(p, dims, vs, dist, z_store_begin, touched, slab, own_lo, own_hi, &owned);
            if (owned) samples++;
            if (tsdf <= 0) {
                if (tsdf < 0) {
                    t = t - step_size;
                    t = t + (previous_tsdf / (previous_tsdf - tsdf)) * step_size;
                }
                p[0] = (t * dir[0]) + sx;
                p[1] = (t * dir[1]) + sy;
                p[2] = (t * dir[2]) + sz;
                ix = p[0] + space_min[0];
                iy = p[1] + space_min[1];
                iz = p[2] + space_min[2];
                hit_k = (float)count;
                hit_t = t;
                done = 1;
            } else if (previous_tsdf < 0) {
                done = 1; /* never taken (Q7) */
            } else {
                /* also the path of a NaN sample: both comparisons above are false */
                t = t + step_size;
                if (t >= max_t) done = 1;
            }
            if (count++ > 4400) done = 1; /* :369  => at most 4402 samples (Q8) */
        }
    }
    if (slab) {
        out[0] = hit_k; out[1] = hit_t; out[2] = 0; out[3] = 0;
    } else {
        out[0] = ix; out[1] = iy; out[2] = iz;
    }
    return samples;
}

/* ref: src/RayCaster/GPURaycaster.cu:441-464 ; Camera::position src/Camera.cpp:211-213 ;
 * space_max = offset + physical_size (Float3 operator+ src/include/TSDFVolume.hpp:47-49) */
static void ray_setup(const float pose[16], const orc_geom *g, float origin[3], float rot[9],
                      float smin[3], float smax[3]) {
    origin[0] = pose[12]; origin[1] = pose[13]; origin[2] = pose[14];
    rot[0] = pose[0]; rot[1] = pose[1]; rot[2] = pose[2];
    rot[3] = pose[4]; rot[4] = pose[5]; rot[5] = pose[6];
    rot[6] = pose[8]; rot[7] = pose[9]; rot[8] = pose[10];
    for (int i = 0; i < 3; i++) {
        smin[i] = g->offset[i];
        smax[i] = g->offset[i] + g->phys[i];
    }
}

void orc_raycast(const float *dist, const orc_geom *g, const float pose[16], const float kinv[9], uint32_t width,
                 uint32_t height, float *vertices, int32_t *sample_count, uint8_t *touched_map,
                 orc_ray_stats *stats, int nthreads) {
    float origin[3], rot[9], smin[3], smax[3];
    ray_setup(pose, g, origin, rot, smin, smax);
    const uint32_t *dims = g->dims;
    const float *vs = g->vs;
    const float trunc = g->trunc;
    int64_t total_samples = 0, hits = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads) reduction(+ : total_samples, hits)
    for (int64_t imy = 0; imy < (int64_t)height; imy++) {
        for (uint32_t imx = 0; imx < width; imx++) {
            float out[4];
            int s = march_ray((int)imx, (int)imy, dist, dims, vs, smin, smax, trunc, origin, rot, kinv, 0,
                              touched_map, 0, 0, 0, out);
            size_t idx = (size_t)imy * width + imx;
            vertices[idx * 3 + 0] = out[0];
            vertices[idx * 3 + 1] = out[1];
            vertices[idx * 3 + 2] = out[2];
            if (sample_count) sample_count[idx] = s;
            total_samples += s;
            if (out[0] == out[0]) hits++;
        }
    }
    if (stats) {
        stats->samples = total_samples;
        stats->hits = hits;
        int64_t t = 0;
        if (touched_map) {
            size_t n = (size_t)dims[0] * dims[1] * dims[2];
            for (size_t i = 0; i < n; i++) t += touched_map[i];
        }
        stats->touched = t;
    }
}

/* Rows y_begin, y_begin+y_step, ... < y_end only (bounded CPU-baseline sample); other pixels untouched. */
int64_t orc_raycast_rows(const float *dist, const orc_geom *g, const float pose[16], const float kinv[9],
                         uint32_t width, uint32_t height, uint32_t y_begin, uint32_t y_end, uint32_t y_step,
                         float *vertices, int nthreads) {
    float origin[3], rot[9], smin[3], smax[3];
    ray_setup(pose, g, origin, rot, smin, smax);
    const uint32_t *dims = g->dims;
    const float *vs = g->vs;
    const float trunc = g->trunc;
    int64_t total_samples = 0;
    if (nthreads < 1) nthreads = 1;
    if (y_step < 1) y_step = 1;
    if (y_end > height) y_end = height;
    int64_t nrows = (y_end > y_begin) ? ((int64_t)(y_end - y_begin) + y_step - 1) / y_step : 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : total_samples)
    for (int64_t r = 0; r < nrows; r++) {
        uint32_t imy = y_begin + (uint32_t)r * y_step;
        for (uint32_t imx = 0; imx < width; imx++) {
            float out[4];
            total_samples += march_ray((int)imx, (int)imy, dist, dims, vs, smin, smax, trunc, origin, rot, kinv, 0,
                                       NULL, 0, 0, 0, out);
            size_t idx = (size_t)imy * width + imx;
            vertices[idx * 3 + 0] = out[0];
            vertices[idx * 3 + 1] = out[1];
            vertices[idx * 3 + 2] = out[2];
        }
    }
    return total_samples;
}

/* One slab's hit records: 8 bytes per pixel {uint32 k, float t} -- k = index of the first owned sample with tsdf <= 0
 * (0xffffffff: none), t = its refined ray parameter (src/RayCaster/GPURaycaster.cu:338-341).  hits = 2 * width * height words. */
void orc_raycast_slab(const float *dist, const orc_geom *g, const float pose[16], const float kinv[9], uint32_t width,
                      uint32_t height, uint32_t z_store_begin, uint32_t z_own_begin,
                      uint32_t z_own_end, uint32_t *hits, int nthreads) {
    float origin[3], rot[9], smin[3], smax[3];
    ray_setup(pose, g, origin, rot, smin, smax);
    const uint32_t *dims = g->dims;
    const float *vs = g->vs;
    const float trunc = g->trunc;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (int64_t imy = 0; imy < (int64_t)height; imy++) {
        for (uint32_t imx = 0; imx < width; imx++) {
            size_t idx = (size_t)imy * width + imx;
            float out[4];
            march_ray((int)imx, (int)imy, dist, dims, vs, smin, smax, trunc, origin, rot, kinv,
                      z_store_begin, NULL, 1, z_own_begin, z_own_end, out);
            hits[idx * 2 + 0] = isinf(out[0]) ? 0xffffffffu : (uint32_t)out[0];
            memcpy(&hits[idx * 2 + 1], &out[1], sizeof(float));
        }
    }
}

/* The merge of n_slabs gathered record arrays (layout [slab][pixel]{k, t}): per pixel the record with the smallest k, its
 * vertex formed as process_ray does from the refined t: start = (origin + near * dir) - space_min (:306), vertex =
 * space_min + (start + t * dir) (:344-347).  Direction, near and start depend on the pixel and the pose only. */
void orc_merge_hits(const uint32_t *hits_all, uint32_t n_slabs, const orc_geom *g, const float pose[16], const float kinv[9],
                    uint32_t width, uint32_t height, float *vertices) {
    float origin[3], rot[9], smin[3], smax[3];
    ray_setup(pose, g, origin, rot, smin, smax);
    const size_t n_pixels = (size_t)width * height;
    for (uint32_t imy = 0; imy < height; imy++) {
        for (uint32_t imx = 0; imx < width; imx++) {
            const size_t idx = (size_t)imy * width + imx;
            uint32_t best_k = 0xffffffffu;
            float t = NAN;
            for (uint32_t s = 0; s < n_slabs; s++) {
                const uint32_t *h = hits_all + ((size_t)s * n_pixels + idx) * 2;
                if (h[0] < best_k) {
                    best_k = h[0];
                    memcpy(&t, &h[1], sizeof(float));
                }
            }
            float ix = NAN, iy = NAN, iz = NAN;
            if (best_k != 0xffffffffu) {
                float dir[3], near_t, far_t;
                orc_ray_direction((uint16_t)imx, (uint16_t)imy, rot, kinv, dir);
                (void)orc_ray_box(origin, dir, smin, smax, &near_t, &far_t);
                float sx = ((near_t * dir[0]) + origin[0]) - smin[0];
                float sy = ((near_t * dir[1]) + origin[1]) - smin[1];
                float sz = ((near_t * dir[2]) + origin[2]) - smin[2];
                ix = ((t * dir[0]) + sx) + smin[0];
                iy = ((t * dir[1]) + sy) + smin[1];
                iz = ((t * dir[2]) + sz) + smin[2];
            }
            vertices[idx * 3 + 0] = ix;
            vertices[idx * 3 + 1] = iy;
            vertices[idx * 3 + 2] = iz;
        }
    }
}

/* ref: src/RayCaster/GPURaycaster.cu:393-427 (Q11) */
void orc_normals(uint32_t width, uint32_t height, const float *V, float *N) {
    for (uint32_t imy = 0; imy < height; imy++) {
        for (uint32_t imx = 0; imx < width; imx++) {
            size_t idx = (size_t)imy * width + imx;
            if (imy == height - 1 || imx == width - 1) {
                N[idx * 3 + 0] = 0; N[idx * 3 + 1] = 0; N[idx * 3 + 2] = 0;
            } else {
                const float *a = V + idx * 3, *r = V + (idx + 1) * 3, *b = V + (idx + width) * 3;
                float v2x = r[0] - a[0], v2y = r[1] - a[1], v2z = r[2] - a[2];
                float v1x = b[0] - a[0], v1y = b[1] - a[1], v1z = b[2] - a[2];
                float nx = v1y * v2z - v1z * v2y;
                float ny = v1z * v2x - v1x * v2z;
                float nz = v1x * v2y - v1y * v2x;
                float l = sqrtf(nx * nx + ny * ny + nz * nz);
                N[idx * 3 + 0] = nx / l;
                N[idx * 3 + 1] = ny / l;
                N[idx * 3 + 2] = nz / l;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* bilateral filter                                                                       */
/* ------------------------------------------------------------------------------------ */

/* ref: src/BilateralFilter.cpp:15-42.  Returns the kernel radius. */
int orc_bilateral_tables(float sigma_colour, float sigma_space, float *kernel, float *similarity, int n_sim) {
    int kernel_radius = (int)ceilf(sigma_space * 1.5f);
    float inv_sigma_colour_squared = 1.0f / (sigma_colour * sigma_colour);
    float inv_sigma_space_squared = 1.0f / (sigma_space * sigma_space);
    int kernel_size = kernel_radius * 2 + 1;
    int center = (kernel_size - 1) / 2;
    if (kernel) {
        int idx = 0;
        for (int x = -center; x < kernel_size - center; x++) {
            for (int y = -center; y < kernel_size - center; y++) {
                float dist_squared = (float)(x * x + y * y);
                kernel[idx] = expf(-dist_squared * inv_sigma_space_squared);
                idx++;
            }
        }
    }
    if (similarity) {
        for (int i = 0; i < n_sim; i++) similarity[i] = expf(-i * inv_sigma_colour_squared);
    }
    return kernel_radius;
}

/* ref: src/BilateralFilter.cpp:53-121 (conv_x outer, conv_y inner; kernel_idx only advances for
 * in-image taps: Q12; double product/accumulate narrowed to float each tap) */
#define BILATERAL_BODY(PIX_T)                                                                     \
    int ksz = kernel_radius * 2 + 1;                                                              \
    (void)ksz;                                                                                    \
    for (int x = 0; x < width; x++) {                                                             \
        int current = image[(size_t)width * y + x];                                               \
        float total_weight = 0;                                                                   \
        float sum = 0;                                                                            \
        int kernel_idx = 0;                                                                       \
        for (int conv_x = x - kernel_radius; conv_x <= x + kernel_radius; conv_x++) {             \
            for (int conv_y = y - kernel_radius; conv_y <= y + kernel_radius; conv_y++) {         \
                if (conv_x >= 0 && conv_x < width && conv_y >= 0 && conv_y < height) {            \
                    int conv = image[(size_t)width * conv_y + conv_x];                            \
                    int delta = abs(conv - current);                                              \
                    double conv_weight = kernel[kernel_idx] * similarity[delta];                  \
                    sum += (conv_weight * conv);                                                  \
                    total_weight += conv_weight;                                                  \
                    kernel_idx++;                                                                 \
                }                                                                                 \
            }                                                                                     \
        }                                                                                         \
        out[(size_t)width * y + x] = (PIX_T)(int)floorf(sum / total_weight);                      \
    }

void orc_bilateral_u8(uint8_t *image, int width, int height, float sigma_colour, float sigma_space) {
    int kernel_radius = orc_bilateral_tables(sigma_colour, sigma_space, NULL, NULL, 0);
    int ks = 2 * kernel_radius + 1;
    float *kernel = (float *)malloc(sizeof(float) * ks * ks);
    float similarity[256];
    orc_bilateral_tables(sigma_colour, sigma_space, kernel, similarity, 256);
    uint8_t *out = (uint8_t *)malloc((size_t)width * height);
    for (int y = 0; y < height; y++) {
        BILATERAL_BODY(uint8_t)
    }
    memcpy(image, out, (size_t)width * height);
    free(out);
    free(kernel);
}

void orc_bilateral_u16(uint16_t *image, int width, int height, float sigma_colour, float sigma_space,
                       int nthreads) {
    int kernel_radius = orc_bilateral_tables(sigma_colour, sigma_space, NULL, NULL, 0);
    int ks = 2 * kernel_radius + 1;
    float *kernel = (float *)malloc(sizeof(float) * ks * ks);
    float *similarity = (float *)malloc(sizeof(float) * 65536);
    orc_bilateral_tables(sigma_colour, sigma_space, kernel, similarity, 65536);
    uint16_t *out = (uint16_t *)malloc((size_t)width * height * 2);
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (int y = 0; y < height; y++) {
        BILATERAL_BODY(uint16_t)
    }
    memcpy(image, out, (size_t)width * height * 2);
    free(out);
    free(similarity);
    free(kernel);
}

/* ------------------------------------------------------------------------------------ */
/* camera maths (host-side input provider)                                                */
/* ------------------------------------------------------------------------------------ */

/*
 * 3x3 inverse the way Eigen 3.x computes Matrix3f::inverse() (un-vendored dependency of
 * src/Camera.cpp:21; compute_inverse_size3: cofactors times 1/det).
 */
static inline float cof3(const float m[9], int i, int j) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
#define E3(r, c) m[(c) * 3 + (r)]
    return E3(i1, j1) * E3(i2, j2) - E3(i1, j2) * E3(i2, j1);
#undef E3
}
void orc_mat3_inverse(const float m[9], float out[9]) {
    float c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
    float det = (c0 * m[0] + c1 * m[1]) + c2 * m[2];
    float invdet = 1.0f / det;
    /* result(r,c) = cofactor(c,r) * invdet ; column-major store */
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) out[c * 3 + r] = cof3(m, c, r) * invdet;
}

/* general 4x4 inverse: adjugate / determinant in fp32 (Eigen's Matrix4f::inverse() is a
 * cofactor-family method too; agreement is to rounding, pinned at 1e-6 by Test_Camera). */
void orc_mat4_inverse(const float m[16], float out[16]) {
#define A(r, c) m[(c) * 4 + (r)]
    float s0 = A(0, 0) * A(1, 1) - A(1, 0) * A(0, 1);
    float s1 = A(0, 0) * A(1, 2) - A(1, 0) * A(0, 2);
    float s2 = A(0, 0) * A(1, 3) - A(1, 0) * A(0, 3);
    float s3 = A(0, 1) * A(1, 2) - A(1, 1) * A(0, 2);
    float s4 = A(0, 1) * A(1, 3) - A(1, 1) * A(0, 3);
    float s5 = A(0, 2) * A(1, 3) - A(1, 2) * A(0, 3);
    float c5 = A(2, 2) * A(3, 3) - A(3, 2) * A(2, 3);
    float c4 = A(2, 1) * A(3, 3) - A(3, 1) * A(2, 3);
    float c3 = A(2, 1) * A(3, 2) - A(3, 1) * A(2, 2);
    float c2 = A(2, 0) * A(3, 3) - A(3, 0) * A(2, 3);
    float c1 = A(2, 0) * A(3, 2) - A(3, 0) * A(2, 2);
    float c0 = A(2, 0) * A(3, 1) - A(3, 0) * A(2, 1);
    /* summed so that, for a pose with last row (0,0,0,1), det is bit-identical to the numerator of O(3,3)
     * below: the inverse's last row is then exactly (0,0,0,1) */
    float det = A(2, 0) * s3 - A(2, 1) * s1 + A(2, 2) * s0;
    if (!(A(3, 0) == 0.0f && A(3, 1) == 0.0f && A(3, 2) == 0.0f && A(3, 3) == 1.0f))
        det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
#define O(r, c) out[(c) * 4 + (r)]
    O(0, 0) = (A(1, 1) * c5 - A(1, 2) * c4 + A(1, 3) * c3) / det;
    O(0, 1) = (-A(0, 1) * c5 + A(0, 2) * c4 - A(0, 3) * c3) / det;
    O(0, 2) = (A(3, 1) * s5 - A(3, 2) * s4 + A(3, 3) * s3) / det;
    O(0, 3) = (-A(2, 1) * s5 + A(2, 2) * s4 - A(2, 3) * s3) / det;
    O(1, 0) = (-A(1, 0) * c5 + A(1, 2) * c2 - A(1, 3) * c1) / det;
    O(1, 1) = (A(0, 0) * c5 - A(0, 2) * c2 + A(0, 3) * c1) / det;
    O(1, 2) = (-A(3, 0) * s5 + A(3, 2) * s2 - A(3, 3) * s1) / det;
    O(1, 3) = (A(2, 0) * s5 - A(2, 2) * s2 + A(2, 3) * s1) / det;
    O(2, 0) = (A(1, 0) * c4 - A(1, 1) * c2 + A(1, 3) * c0) / det;
    O(2, 1) = (-A(0, 0) * c4 + A(0, 1) * c2 - A(0, 3) * c0) / det;
    O(2, 2) = (A(3, 0) * s4 - A(3, 1) * s2 + A(3, 3) * s0) / det;
    O(2, 3) = (-A(2, 0) * s4 + A(2, 1) * s2 - A(2, 3) * s0) / det;
    O(3, 0) = (-A(1, 0) * c3 + A(1, 1) * c1 - A(1, 2) * c0) / det;
    O(3, 1) = (A(0, 0) * c3 - A(0, 1) * c1 + A(0, 2) * c0) / det;
    O(3, 2) = (-A(3, 0) * s3 + A(3, 1) * s1 - A(3, 2) * s0) / det;
    O(3, 3) = (A(2, 0) * s3 - A(2, 1) * s1 + A(2, 2) * s0) / det;
#undef O
#undef A
}

/* ref: src/Camera.cpp:26-36 + init() :20-24 */
void orc_camera_k(float fx, float fy, float cx, float cy, float k[9], float kinv[9]) {
    memset(k, 0, 9 * sizeof(float));
    k[0] = fx;       /* (0,0) */
    k[6] = cx;       /* (0,2) */
    k[4] = fy;       /* (1,1) */
    k[7] = cy;       /* (1,2) */
    k[8] = 1.0f;     /* (2,2) */
    orc_mat3_inverse(k, kinv);
}

static void normalize3(float v[3]) {
    /* Eigen normalize(): v /= sqrt(squaredNorm) when squaredNorm > 0 */
    float n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (n2 > 0) {
        float n = sqrtf(n2);
        v[0] /= n; v[1] /= n; v[2] /= n;
    }
}
static void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* ref: src/Camera.cpp:125-180 */
void orc_look_at(float pose[16], float wx, float wy, float wz) {
    const float EPS = 1e-6f;
    float forward[3] = {wx - pose[12], wy - pose[13], wz - pose[14]};
    normalize3(forward);
    float up[3] = {0, 0, 0};
    if ((fabsf(forward[0]) < EPS) && fabsf(forward[2]) < EPS) {
        if (forward[1] < 0) { up[0] = 0; up[1] = 0; up[2] = 1; }
        else if (forward[1] > 0) { up[0] = 0; up[1] = 0; up[2] = -1; }
    } else {
        up[0] = 0; up[1] = 1; up[2] = 0;
    }
    float left[3];
    cross3(up, forward, left);
    normalize3(left);
    cross3(forward, left, up);
    normalize3(up);
    pose[0] = left[0]; pose[1] = left[1]; pose[2] = left[2]; pose[3] = 0.0f;
    pose[4] = up[0]; pose[5] = up[1]; pose[6] = up[2]; pose[7] = 0.0f;
    pose[8] = forward[0]; pose[9] = forward[1]; pose[10] = forward[2]; pose[11] = 0.0f;
    pose[15] = 1.0f;
}

/* ref: src/Camera.cpp:249-256 */
void orc_camera_world_to_camera(const float ip[16], const float w[3], float c[3]) {
    float h[4];
    for (int r = 0; r < 4; r++)
        h[r] = ip[0 * 4 + r] * w[0] + ip[1 * 4 + r] * w[1] + ip[2 * 4 + r] * w[2] + ip[3 * 4 + r] * 1.0f;
    c[0] = h[0] / h[3]; c[1] = h[1] / h[3]; c[2] = h[2] / h[3];
}

/* ref: src/Camera.cpp:206-216 */
void orc_pixel_to_image_plane(const float kinv[9], uint16_t x, uint16_t y, float out[2]) {
    float hx = (float)x, hy = (float)y;
    float c0 = kinv[0] * hx + kinv[3] * hy + kinv[6] * 1.0f;
    float c1 = kinv[1] * hx + kinv[4] * hy + kinv[7] * 1.0f;
    float c2 = kinv[2] * hx + kinv[5] * hy + kinv[8] * 1.0f;
    out[0] = c0 / c2;
    out[1] = c1 / c2;
}

/* ref: src/Camera.cpp:218-228 */
void orc_image_plane_to_pixel(const float k[9], const float cam[2], int out[2]) {
    float h0 = k[0] * cam[0] + k[3] * cam[1] + k[6] * 1.0f;
    float h1 = k[1] * cam[0] + k[4] * cam[1] + k[7] * 1.0f;
    out[0] = (int)roundf(h0);
    out[1] = (int)roundf(h1);
}
