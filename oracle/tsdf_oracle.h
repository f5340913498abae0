/*
 * tsdf_oracle.h -- CPU oracle for the TSDF integrate / raycast / bilateral hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (tsdf_amd/) never
 * links, imports or calls anything in oracle/.
 *
 * It is a plain-C restatement of the reference algorithm (Scoobadood/TSDF), function by
 * function, keeping the reference's fp32 operation order and its quirks (Q1..Q12 of
 * SURVEY.md section 8).  Every function cites the reference file:line it follows.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - bilateral (8 bit):  pinned bit-for-bit against the reference's own
 *     src/BilateralFilter.cpp compiled natively into oracle/_ref (tests/test_oracle_pins.py).
 *   - camera / geometry helpers: pinned on the known answers of the reference's
 *     Test_Camera.cpp and Test_TSDFMetrics.cpp, and the ray/box expectations kept
 *     (commented out) in Test_TSDF_RayCast.cpp.
 *   - integrate / raycast: the reference's CUDA sources cannot be built in this image
 *     (nvcc, CUDA headers and Eigen are absent) and its tests hold no numeric
 *     expectations for them.  They are pinned on the figures the survey recorded from
 *     a run of the reference's device source (SURVEY.md 8c / BASELINE.md 2: updated-voxel
 *     counts, sign change at the wall, centre-pixel vertex) -- otherwise PARITY UNPINNED.
 *
 * Matrices are column-major float arrays exactly as the reference memcpy's them out of
 * Eigen (src/TSDF/TSDFVolume.cu:867-877): a 4x4 is m[col*4+row], a 3x3 is m[col*3+row].
 */
#ifndef TSDF_ORACLE_H
#define TSDF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- volume geometry (src/TSDF/TSDFVolume.cu:679-722, src/include/TSDFVolume.hpp:269-303) --- */
typedef struct {
    uint32_t dims[3];          /* m_size (global grid)                                     */
    float phys[3];             /* m_physical_size                                          */
    float vs[3];               /* m_voxel_size = physical / size                           */
    float offset[3];           /* m_offset now                                             */
    float offset_at_clear[3];  /* m_offset when clear() last ran (baked into the nodes: Q1) */
    float trunc;               /* m_truncation_distance = 1.1f * |voxel_size|              */
} orc_geom;

/* fills dims, phys, vs, trunc; offsets <- 0 */
void orc_geom_init(orc_geom *g, uint32_t X, uint32_t Y, uint32_t Z, float px, float py, float pz);

/* clear(): weights <- 0, distances <- trunc  (src/TSDF/TSDFVolume.cu:812-845) */
void orc_clear(float *dist, float *weight, size_t n, float trunc);

/* centre of voxel as initialise_deformation writes it (src/TSDF/TSDFVolume.cu:783-785)
 * followed by integrate_kernel's "offset + translation" (src/TSDF/TSDFVolume.cu:343). */
void orc_voxel_centre(const orc_geom *g, int vx, int vy, int vz, float out[3]);

/* ---- integrate (src/TSDF/TSDFVolume.cu:308-392) -------------------------------------- */
/*
 * dist / weight : arrays covering z planes [z_store_begin, z_store_begin + planes) of the
 *                 global X*Y*Z grid, x fastest (index = x + y*X + (z - z_store_begin)*X*Y).
 * z_begin,z_end : planes to integrate (global indices, z_end exclusive).
 * translation   : optional per-voxel DeformationNode translations (3 floats per voxel, same
 *                 indexing as dist); NULL = the implicit grid initialise_deformation creates.
 * returns U     : number of voxels whose weight changed.
 */
int64_t orc_integrate(float *dist, float *weight, const orc_geom *g,
                      const float inv_pose[16], const float k[9], const float kinv[9],
                      const uint16_t *depth, uint32_t width, uint32_t height,
                      const float *translation, uint32_t z_store_begin, uint32_t z_begin,
                      uint32_t z_end, int nthreads);

/* ---- raycast (src/RayCaster/GPURaycaster.cu:24-377, 432-486) ------------------------- */
typedef struct {
    int64_t samples;          /* S: total trilinear samples evaluated             */
    int64_t touched;          /* T: distinct voxels touched by any trilinear tap   */
    int64_t hits;             /* rays that produced a vertex                      */
} orc_ray_stats;

/*
 * pose   : camera pose (cam->world) 4x4; origin = pose[12..14], rot = upper-left 3x3
 *          (src/RayCaster/GPURaycaster.cu:441-453).
 * vertices : out, 3*W*H floats, pixel order y*W+x, NaN triple on a miss.
 * sample_count : optional out, W*H int32, samples evaluated per ray.
 * touched_map  : optional scratch, X*Y*Z bytes (zeroed by the caller); set to 1 per touched voxel.
 */
void orc_raycast(const float *dist, const orc_geom *g, const float pose[16], const float kinv[9], uint32_t width,
                 uint32_t height, float *vertices, int32_t *sample_count, uint8_t *touched_map,
                 orc_ray_stats *stats, int nthreads);

/* Rows y_begin, y_begin+y_step, ... < y_end only (bounded CPU-baseline sample); returns samples evaluated. */
int64_t orc_raycast_rows(const float *dist, const orc_geom *g, const float pose[16], const float kinv[9],
                         uint32_t width, uint32_t height, uint32_t y_begin, uint32_t y_end, uint32_t y_step,
                         float *vertices, int nthreads);

/*
 * Slab variant used to check the multi-GPU protocol: dist holds planes
 * [z_store_begin, ...) and only samples whose lower tap plane lies in [z_own_begin, z_own_end)
 * are evaluated.  hits: W*H records of 8 bytes {uint32 k, float t}; k = index of the first owned
 * sample with tsdf <= 0 (0xffffffff when none), t = that sample's refined ray parameter
 * (src/RayCaster/GPURaycaster.cu:338-341).
 */
void orc_raycast_slab(const float *dist, const orc_geom *g, const float pose[16], const float kinv[9], uint32_t width,
                      uint32_t height, uint32_t z_store_begin, uint32_t z_own_begin,
                      uint32_t z_own_end, uint32_t *hits, int nthreads);
/* Per pixel the record with the smallest k among n_slabs record arrays ([slab][pixel]{k, t}); its vertex
 * = space_min + (start + t * dir) with the pixel's own start and direction (:306, :344-347). */
void orc_merge_hits(const uint32_t *hits_all, uint32_t n_slabs, const orc_geom *g, const float pose[16], const float kinv[9],
                    uint32_t width, uint32_t height, float *vertices);

/* compute_normals kernel (src/RayCaster/GPURaycaster.cu:393-427) */
void orc_normals(uint32_t width, uint32_t height, const float *vertices, float *normals);

/* unit pieces exported for known-answer tests */
int orc_ray_box(const float origin[3], const float dir[3], const float space_min[3],
                const float space_max[3], float *near_t, float *far_t);
float orc_trilinear(const float point[3], const uint32_t dims[3], const float vs[3], const float *dist);
void orc_ray_direction(uint16_t px, uint16_t py, const float rot[9], const float kinv[9], float dir[3]);
void orc_world_to_pixel(const float p[3], const float inv_pose[16], const float k[9], int pix[2]);
void orc_world_to_camera(const float p[3], const float inv_pose[16], float out[3]);
void orc_pixel_to_camera(const int pix[2], float depth, const float kinv[9], float out[3]);
void orc_world_to_pixel_n(size_t n, const float *points, const float inv_pose[16], const float k[9], int *pixels);
void orc_world_to_camera_n(size_t n, const float *points, const float inv_pose[16], float *out);
void orc_pixel_to_camera_n(size_t n, const int *pixels, const float *depth, const float kinv[9], float *out);
void orc_ray_direction_n(size_t n, const uint16_t *pixels, const float rot[9], const float kinv[9], float *out);

/* ---- bilateral filter (src/BilateralFilter.cpp:15-121) -------------------------------- */
/* In place, like the reference.  The 8-bit path reproduces the reference exactly (incl. the
 * border mis-registration Q12).  The 16-bit path is the DEFINED semantics of DESIGN.md (the
 * reference's is undefined behaviour): similarity(d) = expf(-d/sigma_c^2) for every d in
 * 0..65535 and a 16-bit store per pixel. */
void orc_bilateral_u8(uint8_t *image, int width, int height, float sigma_colour, float sigma_space);
void orc_bilateral_u16(uint16_t *image, int width, int height, float sigma_colour, float sigma_space,
                       int nthreads);
/* kernel radius and LUTs exactly as the constructor builds them (src/BilateralFilter.cpp:15-42) */
int orc_bilateral_tables(float sigma_colour, float sigma_space, float *kernel /*(2r+1)^2*/,
                         float *similarity, int n_similarity);

/* ---- camera maths (src/Camera.cpp) --------------------------------------------------- */
void orc_camera_k(float fx, float fy, float cx, float cy, float k[9], float kinv[9]);
void orc_mat3_inverse(const float m[9], float out[9]);
void orc_mat4_inverse(const float m[16], float out[16]);
/* look_at (src/Camera.cpp:125-180): rewrites the rotation part of pose, keeps position */
void orc_look_at(float pose[16], float wx, float wy, float wz);
void orc_camera_world_to_camera(const float inv_pose[16], const float w[3], float c[3]);
void orc_pixel_to_image_plane(const float kinv[9], uint16_t x, uint16_t y, float out[2]);
void orc_image_plane_to_pixel(const float k[9], const float cam[2], int out[2]);

int orc_max_threads(void);

/* ---- ICP tracking (third_party/ICP_CUDA) -- icp_oracle.c, PARITY UNPINNED (see its header) ------ */
void orc_icp_pyr_down(const uint16_t *src, int src_rows, int src_cols, uint16_t *dst);
void orc_icp_vmap(const uint16_t *depth, int rows, int cols, float fx, float fy, float cx, float cy, float depth_cutoff,
                  float *vmap);
void orc_icp_nmap(const float *vmap, int rows, int cols, float *nmap);
void orc_icp_step(const float *R, const float *t, const float *vmap_curr, const float *nmap_curr, const float *vmap_prev,
                  const float *nmap_prev, int rows, int cols, float fx, float fy, float cx, float cy, float dist_thresh,
                  float angle_thresh, float *A, float *b, float *residual_inliers, double *sums29);
void orc_ldlt_solve6(const float *A, const float *b, double *x);
void orc_se3_exp(const double *a, double *T);
void orc_mat4d_mul(const double *A, const double *B, double *C);
void orc_icp_incremental_transformation(const uint16_t *depth_curr, const uint16_t *depth_model, int width, int height,
                                        float cx, float cy, float fx, float fy, float dist_thresh, float angle_thresh,
                                        float depth_cutoff, double *T, float *last_error, float *last_inliers);

/* ---- mesh deformation (src/TSDF/TSDFVolume.cu:101-263) -- icp_oracle.c, PARITY UNPINNED ---------- */
void orc_deform_points(const uint32_t dims[3], const float vs[3], const float offset[3], const float offset_at_clear[3],
                       const float *nodes, const float global_rotation[3], const float global_translation[3], int num_points,
                       float *points);

/* ---- marching cubes (mc_oracle.c; src/MarchingCubes/MarkAndSweepMC.cu) ------------------------------ */
/* TRIANGLE_TABLE / VERTICES_FOR_CUBE_TYPE (MC_triangle_table.cu:87, :46), rebuilt from base configurations + rotations;
 * pinned by SHA-256 on the reference's file (tests/golden/mc_tables.sha256.json) */
void orc_mc_tables(int8_t triangle_table[256][16], uint8_t vertices_for_cube_type[256]);
/* extract_surface (MarkAndSweepMC.cu:506-555): vertices in emission order, three per triangle; vertices == NULL counts.
 * Returns the number of vertices, -1 if capacity (in vertices) is too small. */
int64_t orc_marching_cubes(const float *dist, uint32_t X, uint32_t Y, uint32_t Z, const float vs[3], const float offset[3],
                           float *vertices, int64_t capacity, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
