/*
 * tsdf_amd.h -- C ABI of the MI355X (gfx950) TSDF hot path: volume lifecycle, depth-map
 * integration, TSDF ray casting + normals, bilateral depth filter.
 *
 * This is the drop-in boundary.  Everything above it (the C++ classes in
 * tsdf_amd/host/include that keep the reference's TSDFVolume / GPURaycaster /
 * BilateralFilter surface, and the Python mirror used by tests and bench.py) reaches the
 * GPU only through these entry points; they take plain pointers and sizes, no C++ or
 * torch types.  Each group names the reference interface it replaces (paths relative to
 * the Scoobadood/TSDF tree).
 *
 * Conventions
 *   - Every function returns TSDF_OK or an error code; tsdf_last_error() gives the text of
 *     the last failure on the calling thread.  No exceptions cross the boundary.  (The C++
 *     surface maps TSDF_ERR_INVALID to std::invalid_argument and TSDF_ERR_DEVICE to the
 *     reference's "print and exit(-1)", src/Utilities/cuda_utilities.cu:5-11.)
 *   - Matrices are column-major float arrays, byte-for-byte what the reference memcpy's
 *     out of Eigen into Mat44/Mat33 (src/TSDF/TSDFVolume.cu:867-877,
 *     src/include/cuda_utilities.hpp:12-23): 4x4 = m[col*4+row], 3x3 = m[col*3+row].
 *   - Units are millimetres; depth is uint16 mm with 0 = invalid; pixel order y*width+x;
 *     voxel order x + y*X + z*X*Y (src/include/TSDFVolume.hpp:165-167).
 *   - "_device" variants take device pointers, enqueue on the volume's stream
 *     (tsdf_volume_set_stream) and return without synchronising.  The others take host
 *     pointers and are blocking, like the reference's methods.
 *   - Not thread-safe per object, like the reference.
 */
#ifndef TSDF_AMD_H
#define TSDF_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSDF_OK 0
#define TSDF_ERR_INVALID 1 /* bad argument                                   */
#define TSDF_ERR_DEVICE 2  /* a HIP call or kernel failed                    */
#define TSDF_ERR_NOMEM 3   /* allocation failed                              */

typedef struct tsdf_volume tsdf_volume;       /* opaque: one TSDFVolume (or one Z-slab of it) */
typedef struct tsdf_bilateral tsdf_bilateral; /* opaque: one BilateralFilter                  */
typedef struct tsdf_icp tsdf_icp;             /* opaque: one ICPOdometry                      */

/* Mirror of the reference's private state, src/include/TSDFVolume.hpp:269-303. */
typedef struct tsdf_volume_info {
    uint32_t size[3];            /* m_size: voxels of the GLOBAL grid                          */
    uint32_t z_begin, z_end;     /* planes this object owns (0..Z for a whole volume)           */
    uint32_t z_store_begin, z_store_end; /* planes resident in HBM (owned + one halo plane)     */
    float physical_size[3];      /* m_physical_size (mm)                                        */
    float voxel_size[3];         /* m_voxel_size = physical / size                              */
    float offset[3];             /* m_offset                                                    */
    float offset_at_clear[3];    /* m_offset when clear() last initialised the deformation grid */
    float truncation_distance;   /* m_truncation_distance = 1.1f * |voxel_size|                 */
    float max_weight;            /* m_max_weight = 15 (unused by integrate, as in the reference)*/
    float global_translation[3]; /* m_global_translation                                       */
    float global_rotation[3];    /* m_global_rotation                                          */
    int32_t deformation_materialised; /* 0 while the 24-B/voxel node array is still implicit    */
    int32_t fast_division_verified;   /* 1 = reciprocal division proven bit-equal to IEEE for this voxel size */
} tsdf_volume_info;

/* Layout of one deformation node, src/include/TSDFVolume.hpp:23-26 (2 x float3 = 24 bytes). */
typedef struct tsdf_deformation_node {
    float translation[3];
    float rotation[3];
} tsdf_deformation_node;

/* ---- errors / device ------------------------------------------------------------------ */
const char *tsdf_last_error(void);
int tsdf_device_count(int *count);
int tsdf_set_device(int device);
int tsdf_get_device(int *device);
/* Name of the GPU architecture the library was compiled for ("gfx950"). */
const char *tsdf_build_arch(void);

/* Device memory for callers that keep frames and maps resident in HBM (the _device entry points, tsdf_pipeline_*) without
 * including the HIP headers: hipMalloc / hipFree / blocking hipMemcpy. */
int tsdf_device_alloc(size_t bytes, void **device_ptr);
int tsdf_device_free(void *device_ptr);
int tsdf_device_upload(void *device_dst, const void *host_src, size_t bytes);
int tsdf_device_download(void *host_dst, const void *device_src, size_t bytes);
/* hipStreamSynchronize for the same callers (e.g. a tsdf_exchange_fn that stages the records through the host). */
int tsdf_stream_synchronize(void *hip_stream);

/* ---- volume lifecycle ------------------------------------------------------------------ */
/* Replaces TSDFVolume::TSDFVolume / set_size (src/TSDF/TSDFVolume.cu:430-457, 679-722):
 * validates sizes (TSDF_ERR_INVALID if any is zero/negative), computes voxel size and
 * truncation distance, allocates distances + weights in HBM and clears them.  Dimensions
 * above 65535 are rejected (the reference narrows them to uint16_t). */
int tsdf_volume_create(uint32_t size_x, uint32_t size_y, uint32_t size_z, float physical_x,
                       float physical_y, float physical_z, tsdf_volume **out);
/* One Z-slab [z_begin, z_end) of a size_x*size_y*size_z grid, for one-process-per-GPU
 * sharding (no reference equivalent).  Stores one halo plane above the slab. */
int tsdf_volume_create_slab(uint32_t size_x, uint32_t size_y, uint32_t size_z, float physical_x,
                            float physical_y, float physical_z, uint32_t z_begin, uint32_t z_end,
                            tsdf_volume **out);
/* Replaces TSDFVolume::~TSDFVolume / deallocate (src/TSDF/TSDFVolume.cu:396-423). */
int tsdf_volume_destroy(tsdf_volume *volume);
/* HIP stream (hipStream_t) used by this volume's kernels and copies; NULL = default stream. */
int tsdf_volume_set_stream(tsdf_volume *volume, void *hip_stream);
int tsdf_volume_stream(const tsdf_volume *volume, void **hip_stream);   /* the stream the volume's kernels are enqueued on now */
int tsdf_volume_synchronize(const tsdf_volume *volume);
/* Replaces TSDFVolume::clear (src/TSDF/TSDFVolume.cu:812-845): weights <- 0,
 * distances <- truncation distance, deformation grid <- voxel centres + current offset. */
int tsdf_volume_clear(tsdf_volume *volume);
/* Replaces size()/voxel_size()/physical_size()/truncation_distance()/offset()/
 * global_rotation()/global_translation() (src/include/TSDFVolume.hpp:120-153, 216-225). */
int tsdf_volume_get_info(const tsdf_volume *volume, tsdf_volume_info *info);
/* Replaces TSDFVolume::offset(ox,oy,oz) (src/include/TSDFVolume.hpp:139-143); like the
 * reference it does not re-initialise the deformation grid. */
int tsdf_volume_set_offset(tsdf_volume *volume, float ox, float oy, float oz);

/* Restores the header fields a saved volume carries (file constructor,
 * src/TSDF/TSDFVolume.cu:463-505): offset, truncation distance, max weight, global
 * translation / rotation.  The voxel size stays physical/size, as the reference recomputes it. */
int tsdf_volume_set_header(tsdf_volume *volume, const float offset[3], float truncation_distance,
                           float max_weight, const float global_translation[3],
                           const float global_rotation[3]);

/* ---- volume data access ---------------------------------------------------------------- */
/* Replace distance_data()/weight_data()/deformation() (src/include/TSDFVolume.hpp:175-203):
 * DEVICE pointers to the resident planes.  tsdf_volume_deformation materialises the node
 * array on first use (until then integrate computes voxel centres analytically with the
 * same float expression as initialise_deformation, src/TSDF/TSDFVolume.cu:783-785). */
int tsdf_volume_distances(const tsdf_volume *volume, float **device_ptr);
/* Call after writing distances through the raw device pointer (the reference only ever reads through it):
 * the ray caster's brick-occupancy summary is rebuilt before the next ray cast.  Writes through the raw pointers must be
 * ORDERED ON THE VOLUME'S STREAM (tsdf_volume_set_stream / tsdf_pipeline_streams), after the tsdf_volume_distances /
 * tsdf_volume_weights / tsdf_volume_mark_dirty call that precedes them: those calls make that stream wait for a tightening of
 * the occupancy flags that tsdf_pipeline_step may have left running on its second stream (it reads the distances). */
int tsdf_volume_mark_dirty(tsdf_volume *volume);
/* weight_data(): a device pointer to one fp32 weight per resident voxel, the reference's layout.  Until this is called a volume may
 * hold its weights more compactly (integrate only ever adds 1 to a weight, src/TSDF/TSDFVolume.cu:375-377: 8- or 16-bit counts,
 * tsdf_volume_weight_storage); the call converts them and the volume keeps fp32 weights from then on, clear() included, because the
 * caller may hold the pointer.  set_weight_data / get_weight_data speak fp32 whatever the storage and do not pin it. */
int tsdf_volume_weights(const tsdf_volume *volume, float **device_ptr);
/* How the weights are stored now: 8 or 16 (bits per voxel, counts) or 32 (fp32); *pinned (may be NULL) = 1 once tsdf_volume_weights
 * has handed the fp32 pointer out.  Integration computes the same bits in every storage ((float)count is exact). */
int tsdf_volume_weight_storage(const tsdf_volume *volume, int *bits_per_weight, int *pinned);
/* Widen the storage now -- 8 -> 16 -> 32 bits per weight, values unchanged -- instead of when a count is about to overflow in the middle
 * of a stream (the conversion allocates and synchronises: a caller that knows a session will revisit one region for hundreds of
 * frames pays it up front).  Narrowing is refused; clear() returns an unpinned volume to the starting mode. */
int tsdf_volume_set_weight_storage(tsdf_volume *volume, int bits_per_weight);
/* Self-test of the kernel arithmetic behind packed weights: integrate divides by (count + 1) with a short exact sequence instead of
 * the division instruction sequence (integrate_packed.hip: div_by_count, with the proof); this compares the two bit for bit for every
 * mantissa of the dividend (both signs, three exponents) and every divisor in [b_begin, b_end), 1 <= b_begin < b_end <= 2^17 + 1, and
 * returns the number of differences (0).  About a second for all 65536 divisors the 16-bit counts can reach. */
int tsdf_selftest_count_division(uint32_t b_begin, uint32_t b_end, unsigned long long *mismatches);
int tsdf_volume_deformation(tsdf_volume *volume, tsdf_deformation_node **device_ptr);
/* Replace set_distance_data/set_weight_data/set_deformation (src/TSDF/TSDFVolume.cu:731-757):
 * blocking H2D of every resident voxel. */
int tsdf_volume_set_distance_data(tsdf_volume *volume, const float *host);
int tsdf_volume_set_weight_data(tsdf_volume *volume, const float *host);
int tsdf_volume_set_deformation(tsdf_volume *volume, const tsdf_deformation_node *host);
/* TSDFVolume::deform_mesh (src/TSDF/TSDFVolume.cu:265-291, kernel :226-263): num_points xyz triples are replaced by the
 * trilinear blend of the surrounding deformation nodes' translations, rotated by global_rotation and shifted by
 * global_translation.  Points outside the volume are left unchanged (the reference reads uninitialised memory there). */
int tsdf_volume_deform_points(const tsdf_volume *volume, int num_points, float *host_points);
int tsdf_volume_deform_points_device(const tsdf_volume *volume, int num_points, float *device_points);
/* Blocking D2H of every resident voxel (what save_to_file does, src/TSDF/TSDFVolume.cu:911-1027). */
int tsdf_volume_get_distance_data(const tsdf_volume *volume, float *host);
int tsdf_volume_get_weight_data(const tsdf_volume *volume, float *host);
/* The deformation nodes of resident planes [plane_begin, plane_begin + plane_count) (plane 0 = the first resident one), X*Y
 * nodes each, as save_to_file writes m_deformation_nodes (src/TSDF/TSDFVolume.cu:1003-1018): the device array when it has been
 * materialised (deformation() / set_deformation()), otherwise the regular grid initialise_deformation would have written
 * (src/TSDF/TSDFVolume.cu:783-785: voxel centre + the offset at the last clear(), rotation 0) -- without materialising it. */
int tsdf_volume_get_deformation_planes(const tsdf_volume *volume, uint32_t plane_begin, uint32_t plane_count,
                                       tsdf_deformation_node *host);
/* File constructor (src/TSDF/TSDFVolume.cu:463-664): a saved volume's node block is loaded verbatim.  When that block is the
 * regular grid of some constant offset (what clear() wrote, Q1) the nodes can stay implicit; this tells the volume which offset
 * they carry.  Refused once the node array is materialised. */
int tsdf_volume_set_offset_at_clear(tsdf_volume *volume, const float offset_at_clear[3]);

/* ---- integrate -------------------------------------------------------------------------- */
/* Replaces TSDFVolume::integrate + integrate_kernel (src/TSDF/TSDFVolume.cu:861-902, 308-392).
 * pose / inv_pose: 4x4, k / kinv: 3x3 (camera.pose(), inverse_pose(), k(), kinv()).
 * Host variant: blocking, depth in host memory.  Device variant: depth in HBM, asynchronous. */
int tsdf_integrate(tsdf_volume *volume, const uint16_t *host_depth, uint32_t width, uint32_t height,
                   const float pose[16], const float inv_pose[16], const float k[9], const float kinv[9]);
int tsdf_integrate_device(tsdf_volume *volume, const uint16_t *device_depth, uint32_t width,
                          uint32_t height, const float pose[16], const float inv_pose[16],
                          const float k[9], const float kinv[9]);
/* tsdf_integrate_device for a depth image whose tile maxima the caller already holds in HBM (written by
 * tsdf_bilateral_filter_u16_device_tiles for this very image, earlier on the same stream): same result, one launch fewer.
 * device_tile_max[ty * ceil(width / TSDF_DEPTH_TILE) + tx] must be >= every pixel of tile (tx, ty) and 0 only if all of
 * them are 0; a wrong array makes the culling drop bricks the frame updates. */
int tsdf_integrate_device_tiles(tsdf_volume *volume, const uint16_t *device_depth, uint32_t width,
                                uint32_t height, const float pose[16], const float inv_pose[16],
                                const float k[9], const float kinv[9], const uint16_t *device_tile_max);
/* The first half of tsdf_integrate_device_tiles ahead of time: the brick culling of a frame (it reads the tile maxima and the
 * pose, not the volume) on `hip_stream` -- e.g. a lower-priority stream, while the previous frame's ray cast runs on the
 * volume's stream.  The next tsdf_integrate_device_tiles with the same image, matrices and tile maxima launches only the
 * integrate kernel; any other integrate call ignores the preparation.  The caller orders the streams: the prepare call after
 * the volume's previous integrate has finished, the integrate call after the prepare call's work (events).  Same result.
 * The preparation is recognised by the ARGUMENTS (pointers, sizes, matrices), not by the buffers' content: between the prepare
 * call and the integrate call the image and its tile maxima must not be rewritten -- a caller that does rewrite them calls
 * tsdf_integrate_discard_prepared first (clear, a new offset or header, set_deformation discard it themselves). */
int tsdf_integrate_prepare_device_tiles(tsdf_volume *volume, const uint16_t *device_depth, uint32_t width,
                                        uint32_t height, const float pose[16], const float inv_pose[16],
                                        const float k[9], const float kinv[9], const uint16_t *device_tile_max,
                                        void *hip_stream);
int tsdf_integrate_discard_prepared(tsdf_volume *volume);
/* Optional kernel timing for roofline reports: when enabled, every launch of integrate_kernel (which = 0) and of
 * process_ray_kernel (which = 1) and process_ray_tail_kernel (which = 2) is bracketed by HIP events on the volume's stream; tsdf_volume_kernel_time
 * synchronises the stream and returns the number of bracketed launches and their average duration since timing was
 * (re-)enabled.  enabled = n > 1 brackets every n-th launch only: an event record costs a few microseconds of stream
 * time, which a sub-millisecond step notices (measured: 9 % with every launch bracketed).  Off by default. */
int tsdf_volume_set_timing(tsdf_volume *volume, int enabled);
int tsdf_volume_kernel_time(tsdf_volume *volume, int which, uint32_t *launches, float *average_ms);

/* Diagnostics: when enabled, integrate also counts the voxels whose weight changed (U in the
 * roofline model).  Costs one atomic per wave; leave off when timing. */
int tsdf_volume_set_counting(tsdf_volume *volume, int enabled);
int tsdf_volume_last_updated_voxels(const tsdf_volume *volume, uint64_t *count);

/* ---- raycast ---------------------------------------------------------------------------- */
/* Replaces GPURaycaster::raycast = get_vertices/process_ray + compute_normals
 * (src/RayCaster/GPURaycaster.cu:519-547, 432-486, 265-377, 393-427, 496-510).
 * pose: camera pose (origin = translation column, rot = upper-left 3x3), kinv: 3x3.
 * vertices / normals: 3*width*height floats each (packed float3, pixel order y*width+x);
 * a miss is (NaN,NaN,NaN).  normals may be NULL (get_vertices only, as render_to_depth_image
 * needs, src/RayCaster/GPURaycaster.cu:555-606). */
int tsdf_raycast(const tsdf_volume *volume, uint32_t width, uint32_t height, const float pose[16],
                 const float kinv[9], float *host_vertices, float *host_normals);
int tsdf_raycast_device(const tsdf_volume *volume, uint32_t width, uint32_t height,
                        const float pose[16], const float kinv[9], float *device_vertices,
                        float *device_normals);
/* compute_normals alone (src/RayCaster/GPURaycaster.cu:393-427) on device buffers. */
int tsdf_normals_device(uint32_t width, uint32_t height, const float *device_vertices,
                        float *device_normals, void *hip_stream);
/* The per-pixel part of GPURaycaster::render_to_depth_image (src/RayCaster/GPURaycaster.cu:575-579) on device
 * buffers: depth = (uint16_t)roundf(Camera::world_to_camera(vertex).z) with the vertex map of a ray cast; pixels the
 * cast missed (NaN) and depths outside 1..65535 give 0 (the reference's conversion of NaN is undefined). */
int tsdf_vertices_to_depth_device(uint32_t width, uint32_t height, const float *device_vertices,
                                  const float inv_pose[16], uint16_t *device_depth, void *hip_stream);
/* GPURaycaster::render_to_depth_image (src/RayCaster/GPURaycaster.cu:554-589) in one call on device buffers: the ray cast with the
 * depth formed from every hit as above, without a vertex map in between (device_vertices may be NULL; when given it is filled
 * as tsdf_raycast_device fills it).  Asynchronous on the volume's stream. */
int tsdf_raycast_depth_device(const tsdf_volume *volume, uint32_t width, uint32_t height, const float pose[16],
                              const float inv_pose[16], const float kinv[9], uint16_t *device_depth, float *device_vertices);
/* Which kernels the volume's last ray cast took: 1 = the cell-parallel cast (one wave per flagged brick, no ray is marched), 0 = the
 * march kernels.  Scheduling only -- both produce the same bits -- reported by bench.py beside the kernels' times. */
int tsdf_volume_last_raycast_kind(const tsdf_volume *volume, int *cell_parallel);
/* Diagnostics: how many tasks (flagged bricks in view, large ones counted by their parts) the volume's last cell-parallel cast listed.
 * Waits for the volume's stream.  0 before the first such cast. */
int tsdf_volume_last_cell_list(const tsdf_volume *volume, uint32_t *listed);
/* Diagnostics for the roofline model: S = trilinear samples evaluated, T = distinct voxels
 * touched by any tap, of one raycast with these arguments (runs an instrumented kernel). */
int tsdf_raycast_stats(const tsdf_volume *volume, uint32_t width, uint32_t height, const float pose[16],
                       const float kinv[9], uint64_t *samples, uint64_t *touched_voxels, uint64_t *hits);

/* Diagnostics: trilinear samples the production kernel actually evaluates (the rest of the reference's S
 * samples are passed by exact empty-space skipping), and the state of the brick occupancy it skips on. */
int tsdf_raycast_evaluated_samples(const tsdf_volume *volume, uint32_t width, uint32_t height,
                                   const float pose[16], const float kinv[9], uint64_t *evaluated,
                                   float *host_per_ray /* optional 3*W*H: samples, loop trips, count */);
int tsdf_volume_occupancy(const tsdf_volume *volume, uint64_t *occupied_bricks, uint64_t *total_bricks);
/* Diagnostics / tests: the per-brick bytes themselves (ceil(X/4) * ceil(Y/4) * ceil(Z/4) each, x fastest; any pointer
 * may be NULL).  force_rebuild != 0 recomputes the flags from the distance array first. */
int tsdf_volume_get_occupancy_data(const tsdf_volume *volume, int force_rebuild, uint8_t *host_fine, uint8_t *host_cell,
                                   uint8_t *host_reach);

/* Replaces the device part of extract_surface (src/MarchingCubes/MarkAndSweepMC.cu:506-555): marching cubes over the
 * volume's distance array on the GPU.  Cubes in the reference's order (x fastest, then y, then z), its corner / edge
 * numbering (:9-36, :80-97), sign rule (:110-124) and interpolate() arithmetic (:47-63); three consecutive vertices per
 * triangle (the caller wires them (i, i+2, i+1), :549).  table = 256 x 32 edge numbers, three per triangle, -1
 * terminated (the host library generates it: tsdf_host_mc_table).  *n_vertices receives the vertex count; when
 * host_vertices is not NULL and capacity (in vertices) suffices, it receives 3 floats per vertex.  A Z-slab marches the cube layers
 * rooted in its own planes: the slabs' arrays concatenated in slab order are the whole volume's. */
int tsdf_volume_marching_cubes(const tsdf_volume *volume, const int8_t *table, uint64_t *n_vertices, float *host_vertices,
                               uint64_t capacity);

/* Multi-GPU raycast: a slab evaluates only the samples whose lower trilinear tap plane it owns and writes
 * one 8-byte record per pixel: k = index of the first owned sample with tsdf <= 0 (TSDF_NO_HIT if none), t = that sample's
 * refined ray parameter (src/RayCaster/GPURaycaster.cu:338-341).  After an all-gather of the records (layout
 * [slab][pixel]), tsdf_merge_hits_device keeps, per pixel, the record with the smallest k and forms the vertex
 * space_min + (start + t * dir) with the reference's expressions (:306, :344-347): start and dir are functions of the pixel
 * and the pose, which every rank computes bit-identically -- so the record needs neither of them.  (Up to round 2 the record
 * was {k, x, y, z}, 16 bytes.)  `volume` supplies the grid's offset and physical size (any slab of the grid, or the whole). */
#define TSDF_NO_HIT 0xffffffffu
typedef struct tsdf_hit_record {
    uint32_t k;
    float t;
} tsdf_hit_record;
int tsdf_raycast_slab_device(const tsdf_volume *volume, uint32_t width, uint32_t height,
                             const float pose[16], const float kinv[9], tsdf_hit_record *device_hits);
int tsdf_merge_hits_device(const tsdf_volume *volume, const tsdf_hit_record *device_hits_all, uint32_t n_slabs,
                           uint32_t width, uint32_t height, const float pose[16], const float kinv[9],
                           float *device_vertices, void *hip_stream);
/* The same select and compute_normals (src/RayCaster/GPURaycaster.cu:393-427) on the merged map in one launch. */
int tsdf_merge_hits_normals_device(const tsdf_volume *volume, const tsdf_hit_record *device_hits_all, uint32_t n_slabs,
                                   uint32_t width, uint32_t height, const float pose[16], const float kinv[9],
                                   float *device_vertices, float *device_normals, void *hip_stream);

/* ---- slab exchange: the one collective of a sharded frame (no reference counterpart) ------------------------- */
/* ncclAllGather of the ranks' hit records by RCCL ON THE CALLER'S HIP STREAM: one more launch between the slab ray cast and the
 * merge kernel, no event, no second stream.  librccl is opened at run time: rccl_library = path of the library to use (a process
 * that already holds one, e.g. torch's, passes that path), NULL = the one already loaded if any, else the ROCm installation's.
 * Rank 0 makes the 128-byte id (tsdf_slab_exchange_unique_id) and hands it to the other ranks by whatever means the caller has
 * (torch.distributed, MPI, a file); then EVERY rank calls tsdf_slab_exchange_create (it is collective: ncclCommInitRank). */
typedef struct tsdf_slab_exchange tsdf_slab_exchange;
#define TSDF_EXCHANGE_ID_BYTES 128
int tsdf_slab_exchange_unique_id(uint8_t id[TSDF_EXCHANGE_ID_BYTES], const char *rccl_library);
int tsdf_slab_exchange_create(int rank, int world, const uint8_t id[TSDF_EXCHANGE_ID_BYTES], const char *rccl_library,
                              tsdf_slab_exchange **out);
/* The same object over the caller's own collective (MPI, torch.distributed, a test double): all_gather must leave rank r's
 * n_pixels records at device_all + r * n_pixels on every rank, enqueued on or ordered behind hip_stream; returns TSDF_OK. */
typedef int (*tsdf_exchange_fn)(void *user, const tsdf_hit_record *device_mine, tsdf_hit_record *device_all, uint32_t n_pixels,
                                void *hip_stream);
int tsdf_slab_exchange_create_callback(int rank, int world, tsdf_exchange_fn all_gather, void *user, tsdf_slab_exchange **out);
/* A stand-in for the collective on a box with one GPU (emulation and tests; no reference counterpart): "rank `rank` of `world`" whose
 * all-gather copies this rank's own records into every rank's place, device to device on the caller's stream -- the launches, the
 * bytes landing in this GPU's memory and the merge over `world` record buffers cost what they cost on a node; the wire does not run. */
int tsdf_slab_exchange_create_loopback(int rank, int world, tsdf_slab_exchange **out);
int tsdf_slab_exchange_world(const tsdf_slab_exchange *exchange, int *rank, int *world);
/* How many ranks the communicator itself reports (ncclCommCount; the world handed in at creation for a caller's own collective):
 * what a driver prints beside its timings, so that nobody takes a run of one rank for a scaling point. */
int tsdf_slab_exchange_ranks_seen(const tsdf_slab_exchange *exchange, int *ranks);
/* SURVEY.md 8e, mode B -- the cross-rank validator of the merge path (no reference counterpart: the reference has one GPU).  Every
 * rank gathers every rank's DISTANCE slab through the exchange, assembles the whole volume, casts it the ordinary single-volume way
 * and counts the words in which that picture differs from the merged one (device_merged_*: what tsdf_pipeline_step or
 * tsdf_merge_hits_normals_device left; normals may be NULL); NaN equals NaN.  0 differing words on every rank = the 8-byte {k, t}
 * records + min-k merge reproduce the whole volume's cast bit for bit.  Collective (every rank calls it, on its slab's stream),
 * blocking, and expensive by design: 4 N bytes per rank through the collective and a whole volume resident on every rank. */
int tsdf_slab_validate_merge(tsdf_volume *slab_volume, tsdf_slab_exchange *exchange, uint32_t width, uint32_t height,
                             const float pose[16], const float kinv[9], const float *device_merged_vertices,
                             const float *device_merged_normals, uint64_t *differing_words);
/* device_all: world x n_pixels records, rank r's at [r * n_pixels, (r + 1) * n_pixels).  Asynchronous on hip_stream. */
int tsdf_slab_exchange_all_gather(tsdf_slab_exchange *exchange, const tsdf_hit_record *device_mine, tsdf_hit_record *device_all,
                                  uint32_t n_pixels, void *hip_stream);
int tsdf_slab_exchange_destroy(tsdf_slab_exchange *exchange);

/* ---- the per-frame loop ------------------------------------------------------------------------------------------------- */
/* Replaces the body of the reference's frame loop (src/Tools/kinfu.cpp:32-56: load, integrate, one blocking call each; BASELINE
 * configs[2] adds the bilateral filter and a ray cast per frame) on frames that live in HBM: filter -> integrate -> ray cast +
 * normals per step on a HIP stream of the pipeline's own, and -- with TSDF_PIPELINE_OVERLAP -- the NEXT frame's filter (and its
 * brick culling, when next_camera is given: ground-truth trajectories; not when the pose comes from tracking against this
 * frame's ray cast) on a second stream of lower priority, released by this frame's integrate and awaited by the next step.
 * Same results as the three calls one after the other (tests/test_pipeline.py, tests/cpp/test_stream.cpp); 0.345 -> 0.325 ms per
 * step at 512^3.  With a slab exchange the volume is one rank's Z-slab: a step ray casts the slab, all-gathers the ranks' records
 * on the step's stream and merges them (vertices + normals on every rank); TSDF_PIPELINE_EXCHANGE_STREAM moves the all-gather
 * and the merge to a third stream so that the next frame's integrate need not wait for the collective (results are then
 * ordered behind tsdf_pipeline_synchronize only).  The pipeline owns its streams, events, the two filtered frames + tile
 * maxima and the record buffers; while it lives the volume's stream is the pipeline's (tsdf_pipeline_streams), restored
 * by tsdf_pipeline_destroy.  A volume takes ONE pipeline or tracker at a time: tsdf_pipeline_create / tsdf_tracker_create on a
 * volume that is still attached return TSDF_ERR_INVALID (destroy the previous one first).  A frame announced as `next` is
 * recognised in the following step BY ITS DEVICE POINTER: every frame in flight needs a buffer of its own -- refilling one staging
 * buffer between the announcement and the step integrates the image that was filtered ahead, not the new contents.  All pointers are device pointers: width * height uint16 depth (it must stay valid until the step
 * after the one it was announced to has run), 3 * width * height floats per map (device_normals may be NULL). */
typedef struct tsdf_pipeline tsdf_pipeline;
typedef struct tsdf_camera_matrices {   /* camera.pose(), inverse_pose(), k(), kinv(): column-major, as everywhere in this header */
    float pose[16], inv_pose[16], k[9], kinv[9];
} tsdf_camera_matrices;
#define TSDF_PIPELINE_OVERLAP 1          /* the next frame's filter / culling on a second, lower-priority stream                */
#define TSDF_PIPELINE_EQUAL_PRIORITY 2   /* diagnostics: both streams at the same priority (always so with a slab exchange)      */
#define TSDF_PIPELINE_EXCHANGE_STREAM 4  /* slab exchange + merge on a third stream                                            */
#define TSDF_PIPELINE_NO_TIGHTEN_AHEAD 8 /* diagnostics: the periodic tightening of the ray caster's flags stays in front of the ray cast */
int tsdf_pipeline_create(tsdf_volume *volume, const tsdf_bilateral *filter, uint32_t width, uint32_t height, int flags,
                         tsdf_slab_exchange *exchange /* NULL: a whole volume */, tsdf_pipeline **out);
int tsdf_pipeline_step(tsdf_pipeline *pipeline, const uint16_t *device_depth, const tsdf_camera_matrices *camera,
                       float *device_vertices, float *device_normals, const uint16_t *next_device_depth /* or NULL */,
                       const tsdf_camera_matrices *next_camera /* or NULL */);
int tsdf_pipeline_synchronize(tsdf_pipeline *pipeline);
/* The pipeline's hipStream_t handles (side_stream is NULL without TSDF_PIPELINE_OVERLAP), e.g. to order the caller's own work. */
int tsdf_pipeline_streams(const tsdf_pipeline *pipeline, void **main_stream, void **side_stream);
/* The record buffers of a sharded pipeline (NULL for a whole volume): this rank's n_pixels records, all ranks' world x n_pixels. */
int tsdf_pipeline_hit_buffers(const tsdf_pipeline *pipeline, tsdf_hit_record **device_mine, tsdf_hit_record **device_all);
int tsdf_pipeline_destroy(tsdf_pipeline *pipeline);

/* ---- the tracked loop (BASELINE configs[4]) ------------------------------------------------------------------------------- */
/* Frame-to-model tracking: what src/Tools/tsdf_icp.cpp:115-198 does for one frame (render the volume to a depth image from a pose,
 * ICPOdometry against the new frame) composed with kinfu.cpp's integrate, frame after frame on device buffers:
 *     tsdf_tracker_filter(depth)                      bilateral filter (+ tile maxima) and, from the second frame on, initICP of
 *                                                     the filtered frame -- on a second, lower-priority stream (TSDF_PIPELINE_OVERLAP)
 *     tsdf_tracker_align(previous camera, T, ...)     ray cast from the previous pose, render_to_depth_image, initICPModel,
 *                                                     getIncrementalTransformation; blocks for T (current camera -> previous
 *                                                     camera, metres, column-major double as tsdf_icp_get_incremental_transformation)
 *     tsdf_tracker_integrate(camera)                  the filtered frame at the pose the caller composed (asynchronous)
 * The first frame is only filtered and integrated.  The tracker owns its streams, events and frame buffers; while it lives the
 * volume's and the ICP object's stream are its own. */
typedef struct tsdf_tracker tsdf_tracker;
int tsdf_tracker_create(tsdf_volume *volume, const tsdf_bilateral *filter, tsdf_icp *icp, uint32_t width, uint32_t height,
                        float depth_cutoff, int flags /* TSDF_PIPELINE_OVERLAP or 0 */, tsdf_tracker **out);
int tsdf_tracker_filter(tsdf_tracker *tracker, const uint16_t *device_depth);
int tsdf_tracker_align(tsdf_tracker *tracker, const tsdf_camera_matrices *previous, double T_prev_curr[16] /* in: start, out */,
                       float *last_error, float *last_inliers);
int tsdf_tracker_integrate(tsdf_tracker *tracker, const tsdf_camera_matrices *camera);
int tsdf_tracker_synchronize(tsdf_tracker *tracker);
int tsdf_tracker_streams(const tsdf_tracker *tracker, void **main_stream, void **side_stream);
/* The ICP inputs of the last aligned frame: the rendered model depth and the filtered frame (width * height uint16, device). */
int tsdf_tracker_buffers(const tsdf_tracker *tracker, const uint16_t **device_model, const uint16_t **device_filtered);
int tsdf_tracker_destroy(tsdf_tracker *tracker);

/* ---- ICP tracking: replaces third_party/ICP_CUDA ------------------------------------------ */
/* ICPOdometry::ICPOdometry (third_party/ICP_CUDA/ICPOdometry.cpp:10-57): three pyramid levels of vertex / normal maps for
 * the model ("prev") and the current frame; angle_thresh is the sine of the angle gate. */
int tsdf_icp_create(int width, int height, float cx, float cy, float fx, float fy, float dist_thresh, float angle_thresh,
                    tsdf_icp **out);
void tsdf_icp_destroy(tsdf_icp *icp);
int tsdf_icp_set_stream(tsdf_icp *icp, void *hip_stream);
int tsdf_icp_stream(const tsdf_icp *icp, void **hip_stream);   /* the stream its kernels are enqueued on now */
/* ICPOdometry::initICP (model = 0, ICPOdometry.cpp:64-78) / initICPModel (model = 1, :80-95): upload the depth (uint16 mm),
 * pyrDown twice, createVMap + createNMap per level (Cuda/pyrdown.cu).  The host variant synchronises like the reference;
 * the _device variant takes a device pointer and does not. */
int tsdf_icp_init(tsdf_icp *icp, int model, const uint16_t *host_depth, float depth_cutoff);
int tsdf_icp_init_device(tsdf_icp *icp, int model, const uint16_t *device_depth, float depth_cutoff);
/* estimateStep (Cuda/estimate.cu:215-281) for one pyramid level: R (3x3 column-major, Eigen's data()) and t map current-frame
 * points into the model frame; returns the 6x6 normal matrix A, b, {sum of squared residuals, inliers}. */
int tsdf_icp_estimate_step(tsdf_icp *icp, int level, const float R[9], const float t[3], float A[36], float b[6],
                           float residual_inliers[2]);
/* ICPOdometry::getIncrementalTransformation (ICPOdometry.cpp:97-136): 4/5/10 iterations from the coarsest level down,
 * T <- exp(A^-1 b) * T each (the solve and the exponential run on the device, the pose never leaves it between
 * iterations).  T_prev_curr: 4x4 column-major double (Sophus::SE3d::matrix().data()), in/out. */
int tsdf_icp_get_incremental_transformation(tsdf_icp *icp, double T_prev_curr[16], float *last_error, float *last_inliers);
/* Tests / diagnostics: which = 0 vmap_prev, 1 nmap_prev, 2 vmap_curr, 3 nmap_curr (3*rows x cols floats of the level);
 * the depth pyramid of the last init call. */
int tsdf_icp_get_map(const tsdf_icp *icp, int which, int level, float *host_map);
int tsdf_icp_get_depth_level(const tsdf_icp *icp, int level, uint16_t *host_depth);

/* ---- bilateral filter -------------------------------------------------------------------- */
/* Replaces BilateralFilter::BilateralFilter / ~BilateralFilter (src/BilateralFilter.cpp:15-51):
 * builds the spatial kernel and the similarity table on the host exactly as the reference
 * does and uploads them. */
int tsdf_bilateral_create(float sigma_colour, float sigma_space, tsdf_bilateral **out);
int tsdf_bilateral_destroy(tsdf_bilateral *filter);
/* Replace BilateralFilter::filter (src/BilateralFilter.cpp:124-130, 53-121): in place on a host
 * image, blocking.  8 bit: bit-identical to the reference.  16 bit: the reference's path is
 * undefined behaviour; the semantics implemented are stated in DESIGN.md. */
int tsdf_bilateral_filter_u8(const tsdf_bilateral *filter, uint8_t *host_image, int width, int height);
int tsdf_bilateral_filter_u16(const tsdf_bilateral *filter, uint16_t *host_image, int width, int height);
/* Device variants: in != out, asynchronous on hip_stream. */
int tsdf_bilateral_filter_u8_device(const tsdf_bilateral *filter, const uint8_t *device_in,
                                    uint8_t *device_out, int width, int height, void *hip_stream);
int tsdf_bilateral_filter_u16_device(const tsdf_bilateral *filter, const uint16_t *device_in,
                                     uint16_t *device_out, int width, int height, void *hip_stream);
/* The same filter; each workgroup also leaves the largest filtered value of its TSDF_DEPTH_TILE x TSDF_DEPTH_TILE pixel
 * tile in device_tile_max[tile_y * ceil(width / TSDF_DEPTH_TILE) + tile_x] (0 = the tile holds no valid depth).  Hand the
 * array to tsdf_integrate_device_tiles with the filtered image: integrate's culling then skips its own pass over the image.
 * (No reference counterpart: the reference filters on the host and integrates every voxel.) */
#define TSDF_DEPTH_TILE 16
int tsdf_bilateral_filter_u16_device_tiles(const tsdf_bilateral *filter, const uint16_t *device_in,
                                           uint16_t *device_out, int width, int height,
                                           uint16_t *device_tile_max, void *hip_stream);

/* ---- measurement aid (no reference counterpart) ------------------------------------------ */
/* Device-to-device copy of `bytes` (a multiple of 16; two internal buffers) with a float4 copy kernel, `reps` times on
 * `hip_stream`, timed with HIP events on that stream: *gb_per_s = read + write bytes per second of the best repetition / 1e9.
 * bench.py reports it beside the nominal HBM peak as the practical ceiling of a streaming kernel on this box. */
int tsdf_measure_copy_bandwidth(size_t bytes, int reps, void *hip_stream, double *gb_per_s);
/* The same for the access shape of integrate: two arrays of a 512^3 float grid updated IN PLACE (read, modify, write back)
 * brick by brick -- workgroups of 4 waves walking 32 planes of a 64 x 4 x 32 voxel brick, a wave per 256-byte row segment, 4
 * planes in flight -- with no projection work at all: *gb_per_s = (read + write bytes of both arrays) per second / 1e9, best of
 * `reps`.  The ceiling an in-place update of every voxel could reach with integrate's memory walk. */
int tsdf_measure_update_bandwidth(int reps, void *hip_stream, double *gb_per_s);

#ifdef __cplusplus
}
#endif
#endif /* TSDF_AMD_H */
