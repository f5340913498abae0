// Shared host/device definitions for the gfx950 TSDF library (internal; the public surface
// is include/tsdf_amd.h).  Compile every translation unit with -ffp-contract=off: the parity
// contract is that each fp32 operation is rounded on its own, in the order the reference
// writes it.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "tsdf_amd.h"

struct tsdf_volume;

namespace tsdf {

// What a prepared brick list (tsdf_integrate_prepare_device_tiles) was built for
struct PreparedCull {
    const uint16_t *depth, *tile_max;
    uint32_t width, height;
    float inv_pose[16], k[9], kinv[9];
};

// integrate's brick: one wave along x, kIntBrickY waves per workgroup, kIntBrickZ planes walked by a workgroup (integrate.hip)
#ifndef TSDF_CHUNK_Z
#define TSDF_CHUNK_Z 32
#endif
constexpr int kIntBrickX = 64, kIntBrickY = 4, kIntBrickZ = TSDF_CHUNK_Z;

// ---- error plumbing ----------------------------------------------------------------------
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);
// Every environment variable the library reads, in one place, read once per process (volume.hip: tuning()).  None of them can change a
// result: they move work between launches, lanes and streams (tests/test_parity_raycast.py::test_schedule_knobs_do_not_change_a_bit),
// or switch diagnostics on.  DESIGN.md 5 has the table.
struct Tuning {
    int ray_segments;        // TSDF_RAY_SEGMENTS       sample ranges a whole-volume ray march is cut into (6)
    int ray_slab_ranges;     // TSDF_RAY_SLAB_RANGES    parts of a ray's stretch through a Z-slab (0: by the slab's share of the grid)
    int ray_trip_budget;     // TSDF_RAY_TRIP_BUDGET    passes before a wave hands its unfinished rays to the tail kernel (22)
    int ray_tail_lanes;      // TSDF_RAY_TAIL_LANES     lanes per queue entry in the tail kernel (4)
    int ray_tail_grid;       // TSDF_RAY_TAIL_GRID      workgroups of the tail kernel (2560)
    int ray_tail_piece;      // TSDF_RAY_TAIL_PIECE     shortest piece an unfinished stretch is cut into (64)
    int ray_range_order;     // TSDF_RAY_RANGE_ORDER    0 near to far, 1 far to near (default), 2 last, first, then far to near
    int ray_tile_map;        // TSDF_RAY_TILE_MAP       which tiles an XCD gets: 0 every eighth, 1 a contiguous eighth, 2 one block per block row (default)
    int ray_learned_order;   // TSDF_RAY_LEARNED_ORDER  0: launch order, one workgroup per (range, tile) (default 1: the order learnt from the previous cast)
    int ray_heavy_passes;    // TSDF_RAY_HEAVY_PASSES   passes from which a wave counts as long for that order (0: three quarters of the budget)
    int ray_cells;           // TSDF_RAY_CELLS          the cell-parallel cast (raycast_cells.hpp): 0 never, 1 (default) unless the previous cast listed more than
                             //                          TSDF_RAY_CELLS_LIMIT flagged bricks, 2 whenever the view has a projection
    int ray_cells_limit;     // TSDF_RAY_CELLS_LIMIT    (131072)
    float ray_cells_footprint;   // TSDF_RAY_CELLS_FOOTPRINT  largest voxel footprint (pixels, at the depth of the volume's centre) the cell-parallel cast is taken for (10)
    int ray_cells_pairs;     // TSDF_RAY_CELLS_PAIRS    estimated (cell, pixel) pairs of a brick above which it is listed in several parts (1024; 0: never)
    int ray_cells_look;      // TSDF_RAY_CELLS_LOOK     1 (default): every flagged brick is projected when the list is built -- unseen ones dropped, large ones listed in parts; 0: never
    int ray_cells_grid;      // TSDF_RAY_CELLS_GRID     workgroups of cast_cells_kernel (2048: two or three bricks a wave at 20 000 listed bricks; see raycast_cells.hpp)
    int ray_cells_sort;      // TSDF_RAY_CELLS_SORT     the cell-parallel cast's list front to back: 0 never, 1 for views from inside the volume, 2 always
    int ray_chooser;         // TSDF_RAY_CHOOSER        0 (default): TSDF_RAY_CELLS=1 goes by the static rules alone; 1: the cast is chosen from measured times (choose_cast); 2: a trial of the other cast every few casts (test aid)
    int ray_entry_bound;     // TSDF_RAY_ENTRY_BOUND    0: no per-tile entry bound (default 1: rays start at the nearest flagged block their 16 x 16 tile can see)
    int icp_persistent;      // TSDF_ICP_PERSISTENT     default 0: one launch per ICP iteration (the chain); 1 / 2: all 19 in one launch with a grid barrier (slower, kept for study)
    int occ_rebuild_period;  // TSDF_OCC_REBUILD_PERIOD integrations between tightenings of the ray caster's flags (16; 0: never)
    int occ_scan_all;        // TSDF_OCC_SCAN_ALL       1: every tightening reads the whole distance array
    int reach_lds;           // TSDF_REACH_LDS          1: the workgroup variant of the reach summary on every grid
    int int_grid_per_cu;     // TSDF_INT_GRID_PER_CU    integrate: n > 0 = a resident grid of n workgroups per CU walking the brick list
    int weight_pack;         // TSDF_WEIGHT_PACK        how a volume's weights are stored to begin with (weights.hip): 8 (default) / 16-bit counts, 0 = the reference's fp32 array
    int pipe_release;        // TSDF_PIPE_RELEASE       when tsdf_pipeline_step lets the next frame's filter + culling start on the side stream: 0 after
                             //                          this frame's integrate (beside the bulk ray kernel), 1 after the bulk ray kernel (beside the tail kernel), 2 the filter already after the previous step (beside integrate); both measured slower
    int pipe_word_release;   // TSDF_PIPE_WORD_RELEASE  0 (default): the pipeline's second stream is released by an event; 1: by a word the cast's first kernel stores (no packet in the step's stream, but the runtime's wait is a spinning kernel: 0.2167 -> 0.2149 ms, kept as the measured alternative)
    int pipe_host_wait;      // TSDF_PIPE_HOST_WAIT     1: tsdf_pipeline_step waits on the HOST for the frame filtered ahead instead of putting a wait packet into the step's stream
    int event_scope;         // TSDF_EVENT_SCOPE        the events that order the library's own streams: 0 HIP's default (a system-scope fence when the event completes: cache
                             //                          write-back + invalidate for the host's and other devices' sake), 1 hipEventReleaseToDevice, 2 (default) hipEventDisableSystemFence
    int timing_bracket;      // TSDF_TIMING_BRACKET     1: tsdf_volume_set_timing brackets launches with hipEventRecord
    int verbose;             // TSDF_VERBOSE            the reference's chatter
    int debug_rays;          // TSDF_DEBUG_RAYS         how many pieces went through the tail queue (synchronises)
};
const Tuning &tuning();
// Flags of an event that only orders streams of THIS device against each other: nothing the host or another device reads hangs on
// it (the kernels either side of it release / acquire at device scope at their own boundaries, as consecutive kernels of one stream
// do), so the system-scope fence HIP attaches to an event by default is dropped: 4-6 us of every 270 us step (profiles/r04zz_event_scope_ab.txt).
// The events of the multi-GPU exchange (pipeline.hip: cast, merged) keep HIP's default.
inline unsigned stream_order_event_flags() {
    const int scope = tuning().event_scope;
    return hipEventDisableTiming | (scope == 1 ? hipEventReleaseToDevice : scope == 2 ? hipEventDisableSystemFence : 0u);
}
int occupancy_rebuild(struct ::tsdf_volume *v);  // volume.hip
int occupancy_join(struct ::tsdf_volume *v);     // volume.hip: the volume's stream waits for a tightening enqueued elsewhere
int occupancy_tighten_on(struct ::tsdf_volume *v, hipStream_t stream);  // volume.hip: the periodic rebuild on another stream
struct EntryParams;
// volume.hip: bring fine + reach up to date; with `entry` (a whole-volume ray cast whose camera allows it) the same launch also leaves
// the per-tile entry bound of that view (EntryParams)
int occupancy_refresh(struct ::tsdf_volume *v, const EntryParams *entry = nullptr);
bool raycast_takes_cells(const struct ::tsdf_volume *v, uint32_t width, uint32_t height, const float pose[16], const float kinv[9]);   // raycast.hip: would tsdf_raycast_device take the cell-parallel cast now?
int occupancy_flags_refresh(struct ::tsdf_volume *v);   // volume.hip: the flags only (fine, cell), not the reach summary: what the cell-parallel cast reads
int build_t_table(struct ::tsdf_volume *v);      // volume.hip
// timing helpers (volume.hip).  When timing is on, a launch of kernel `which` carries a start and a stop event that take the
// dispatch's own begin / end timestamps (hipExtLaunchKernel: what rocprofv3's kernel trace reads), TSDF_LAUNCH_TIMED below.
// (Up to round 2h the launch was bracketed with two hipEventRecord calls: that interval also holds the two barrier packets and
// the dispatch latency, 8-10 us more than the kernel at 0.11 ms.  TSDF_TIMING_BRACKET=1 brings the brackets back.)
bool timing_pair(struct ::tsdf_volume *v, int which, hipEvent_t *start, hipEvent_t *stop);
void timing_begin(struct ::tsdf_volume *v, int which);
void timing_end(struct ::tsdf_volume *v, int which);
#define TSDF_LAUNCH_TIMED_LDS(v, which, kernel, grid, block, lds_bytes, ...)                                            \
    do {                                                                                                                \
        hipEvent_t ts0__ = nullptr, ts1__ = nullptr;                                                                    \
        if (tsdf::timing_pair(v, which, &ts0__, &ts1__))                                                                \
            hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)(lds_bytes), (v)->stream, ts0__, ts1__, 0, __VA_ARGS__); \
        else {                                                                                                          \
            tsdf::timing_begin(v, which);                                                                               \
            hipLaunchKernelGGL(kernel, grid, block, (uint32_t)(lds_bytes), (v)->stream, __VA_ARGS__);                   \
            tsdf::timing_end(v, which);                                                                                 \
        }                                                                                                               \
    } while (0)
#define TSDF_LAUNCH_TIMED(v, which, kernel, grid, block, ...) TSDF_LAUNCH_TIMED_LDS(v, which, kernel, grid, block, 0, __VA_ARGS__)
int verify_fast_division(struct ::tsdf_volume *v);  // volume.hip
// weights.hip: how the weights are stored
int weights_create(struct ::tsdf_volume *v);
void weights_destroy(struct ::tsdf_volume *v);
int weights_clear(struct ::tsdf_volume *v);
int weights_require_f32(struct ::tsdf_volume *v);
int weights_make_room(struct ::tsdf_volume *v);
int weights_upload(struct ::tsdf_volume *v, const float *host);
int weights_download(const struct ::tsdf_volume *v, float *host);

#define TSDF_HIP(call, what)                                  \
    do {                                                      \
        hipError_t e__ = (call);                              \
        if (e__ != hipSuccess) return tsdf::hip_fail(e__, what); \
    } while (0)

#define TSDF_REQUIRE(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            tsdf::set_error(__VA_ARGS__);    \
            return TSDF_ERR_INVALID;         \
        }                                    \
    } while (0)

// ---- PODs passed by value as kernel arguments (land in SGPRs through the kernarg segment) ----
// Same member order as the reference's Mat44 / Mat33 (src/include/cuda_utilities.hpp:12-23)
// so a column-major float[16] / float[9] can be memcpy'd in.
struct Mat44 {
    float m11, m21, m31, m41;
    float m12, m22, m32, m42;
    float m13, m23, m33, m43;
    float m14, m24, m34, m44;
};
struct Mat33 {
    float m11, m21, m31;
    float m12, m22, m32;
    float m13, m23, m33;
};
struct F3 {
    float x, y, z;
};

// Geometry of the (global) grid plus the resident plane range of this object.
struct Geom {
    uint32_t X, Y, Z;          // global grid
    uint32_t z_store_begin;    // first resident plane
    uint32_t z_store_end;      // one past the last resident plane
    F3 vs;                     // voxel size
    F3 offset;                 // m_offset now
    F3 offset_clear;           // m_offset when clear() last ran (Q1)
    F3 phys;                   // physical size
    float trunc;
};

// Brick occupancy used by the ray caster for exact empty-space skipping (raycast.hip), over the GLOBAL grid in
// bricks of kBrick^3 voxels:
//   fine[b] == 0 : every resident voxel within the brick grown by kBrickGrow voxels on every side is > tau,
//                  i.e. no trilinear sample whose taps lie in the brick grown by one voxel can be <= 0.
//                  Bricks touching the grid boundary need more: in the outer half-voxel shell the reference extrapolates (Q10:
//                  weights 1 - u in (1, 1.5], u in [-0.5, 0) per axis), so positive taps alone prove nothing.  There the flag
//                  is clear only when every resident voxel of the grown brick is FLAT: in [flat_lo, flat_hi] = [15/16, 1 + 2^-10]
//                  x trunc -- what clear() leaves and what free space in front of a surface is averaged to.  With taps in
//                  [c_min, c_max] and weights summing to 1 of which the negative ones sum to -N (N <= (2^3 - 1) / 2 = 3.5 at a
//                  corner of the grid), a sample is >= c_min - N (c_max - c_min) >= 0.71 trunc, far above the rounding of the
//                  reference's fp32 sum.  (Up to round 2 boundary bricks were flagged for good: every ray spent ~11 passes
//                  on the entry rim -- 5 samples in the shell, then blocks of 1, 1, 2, 4, 8 bricks -- a quarter of the bulk
//                  kernel's wave time.)  Sticky: integrate only ever sets flags, a rebuild (clear / whole-array upload) resets them.
//   cell[b] == 0 : every resident voxel in [4b, 4b+4] on every axis is > tau, i.e. all 8 taps of every dual cell
//                  (lower corner) in [4b, 4b+4) are safely positive: a sample whose cell is KNOWN to lie in that
//                  "cell brick" cannot be <= 0.  Tighter than `fine` (no slack for approximate location: the ray
//                  caster uses it only for samples at least eps away from the cell faces).  Cell bricks that
//                  contain cells beyond the grid are set for good.  Sticky like `fine`.
//   reach[b]     : size class of the largest clear ALIGNED block of bricks containing b (0 = b is set; l >= 1 = the
//                  aligned block of 2^(l-1) bricks per side is clear, up to kReachLevels).  A ray inside b may jump
//                  to the faces of that block.  Recomputed from `fine` before a ray cast when stale (volume.hip).
// numerators with |a| < kFastDivMin (zero included) or non-finite take the IEEE division in raycast.hip
constexpr float kFastDivMin = 1.0e-30f;
constexpr int kBrick = 4;
constexpr int kBrickShift = 2;
constexpr int kBrickGrow = 2;
constexpr int kReachLevels = 5;     // largest block: 2^(5-1) = 16 bricks = 64 voxels per side
constexpr int kSlabSkip = 32;       // voxels: granularity at which a slab passes regions it cannot own
constexpr int kSlabSkipShift = 5;
struct OccGrid {
    uint8_t *fine;
    uint8_t *cell;
    uint8_t *reach;
    uint32_t nbx, nby, nbz;  // bricks per axis = ceil(size / kBrick)
    float tau;               // "safely positive" threshold (a fraction of the truncation distance)
    float flat_lo, flat_hi;  // bricks touching the grid boundary: the band every voxel in reach must lie in (see above)
    // first / last voxel index per axis whose value a boundary brick's flag depends on: v < rim_lo or v >= rim_hi* (brick 0 grown
    // by kBrickGrow reaches voxel kBrick + kBrickGrow - 1; the last brick starts at kBrick * (nb - 1))
    __host__ __device__ bool in_rim_zone(uint32_t v, uint32_t nb) const { return v < (uint32_t)(kBrick + kBrickGrow) || v + kBrickGrow >= (uint32_t)kBrick * (nb - 1u); }
    __host__ __device__ bool not_flat(float d) const { return !(d >= flat_lo && d <= flat_hi); }   // (true for NaN)
    __host__ __device__ size_t fine_count() const { return (size_t)nbx * nby * nbz; }
};

// Entry bound of a whole-volume ray cast (round 4; tried in round 2 when the grid's rim bricks were flagged for good, which ate
// the gain).  Most of a ray's passes are hops through the free space between the grid's face and the first surface (13.9 of 22.9
// per ray on the bench scene).  For a view with camera depth == ray parameter (K^-1 with last row (0, 0, 1), pose [R t; 0 0 0 1])
// that free space is bounded once per 16 x 16 pixel tile instead: the reach summary's launch projects every aligned unit of 4^3
// bricks (16^3 voxels) that holds a flagged brick -- grown by a voxel, by its 8 corners -- and lowers, with atomicMin, the word of
// every tile its pixel rectangle (grown by 2 pixels) touches to the unit's smallest camera depth minus two voxels.  A sample of a
// ray through pixel (u, v) that lies in such a unit projects to (u, v), inside the unit's rectangle, and is no nearer than the
// unit's nearest corner (depth is linear, the unit convex, all its corners in front of the camera): so a sample whose depth
// near + T[k] is below its tile's word lies in no flagged brick -- which is exactly the condition under which the march itself
// passes it unevaluated (SkipCtx / locate in raycast.hip; off-grid samples within rounding of a face read clamped taps of boundary
// bricks, whose clear flag means flat voxels).  A unit wholly behind the camera holds no sample; one that straddles the camera
// plane switches the bound off for the view (word [tiles]).  Scheduling only: the picture is the same bits with or without it
// (TSDF_RAY_ENTRY_BOUND=0), CPU estimate tools/dbg_reach_estimate.py: 13.9 -> 6.1 hops per ray.
constexpr int kEntryTile = 16;                  // pixels per side of a tile (= a workgroup of process_ray_kernel)
constexpr uint32_t kEntryFar = 0x7f7f7f7fu;     // "no unit": what a byte-wise fill leaves (3.4e38 as a float)
struct EntryParams {
    float r[3][4];        // world -> camera, rows 1-3 (the inverse of the pose the ray directions are formed with)
    float r_scale;        // an upper bound of the 2-norm of r's 3 x 3 block: a world distance d is at most r_scale * d in the camera's frame (1 for a rigid pose)
    float k[2][3];        // pixel = k * camera / camera.z (the inverse of kinv), rows 1-2
    F3 vs, offset;        // voxel size, grid origin (the ray caster's space_min)
    uint32_t width, height, tiles_x, tiles_y;
    float slack_z;        // subtracted from a unit's nearest corner (two voxels, mm)
    float z_clip;         // cell-parallel cast: no sample of the view has a camera depth below this (> 0 when the camera is outside the volume, else 0: choose_cell_cast)
    float z_near;         // ... and the depth in front of which a box is projected corner by corner (max(z_clip, a quarter voxel))
    uint32_t cell_pairs;  // ... 0, or the pairs of a listed brick's task: the list's builder projects the bricks (cell_cast_prepare_kernel)
    uint32_t *ztile;      // tiles_x * tiles_y words + the on/off word, all kEntryFar / 1 when the launch starts
    uint32_t *ztile_next; // the copy the NEXT cast uses: this launch resets it
    uint32_t units_x, units_y, units_z;   // units (whole or partial) per axis
};

// float -> int with the reference target's semantics (CUDA cvt.rzi: saturating, NaN -> 0);
// written out so the result does not depend on what an out-of-range fptosi lowers to.
__host__ __device__ inline int f2i_sat(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

// ---- object state ------------------------------------------------------------------------
}  // namespace tsdf

namespace tsdf {
// which cast a volume's casts take, from measured times (raycast.hip: choose_cast)
struct CastChooser {
    float ms[2];                  // whole cast, smoothed: [0] the march, [1] the cell-parallel cast
    uint32_t seen[2];             // casts measured
    uint64_t measured_at[2];      // ... the last one at this cast
    uint64_t casts, next_trial;   // whole-volume casts so far; when the cast not taken is tried next
    uint32_t gap;                 // casts between trials (64 ... 4096)
    bool pending, pending_trial, blocked;
    int pending_kind, last_kind, trial_kind, trial_left;   // trial_left > 0: casts of the running trial still to come (the last one is measured)
    hipEvent_t ev[2];             // around the sampled cast on the volume's stream
    float trial_origin[3], trial_axis[3];   // the view of the last trial (a trial lost by 3 x waits for another view)
};
}  // namespace tsdf

struct tsdf_volume {
    tsdf::Geom g;
    uint32_t z_begin, z_end;  // owned planes
    float max_weight;
    float global_translation[3];
    float global_rotation[3];
    int device;
    hipStream_t stream;
    // the tsdf_pipeline / tsdf_tracker that has put this volume on its own stream (nullptr: none).  One at a time: a second
    // attachment is refused, and only the owner's destroy restores the stream the volume had before (pipeline.hip)
    const void *attached;
    float *dist;
    // weights.hip: wmode 0 = the reference's fp32 array `weight` (wpacked null); 8 / 16 = counts of that many bits in `wpacked`
    // (weight null), the planes of one integrate batch of a lane in one dword
    float *weight;
    uint32_t *wpacked;
    int wmode;
    uint32_t weight_bound;   // packed modes: no count exceeds this (integrations since the weights were last known + their maximum then)
    int weight_pinned;       // the caller holds the fp32 device pointer (tsdf_volume_weights): the volume keeps the reference's layout
    tsdf_deformation_node *nodes;  // nullptr while implicit
    // cached per-call temporaries (the reference mallocs/frees these every call)
    uint16_t *depth_buf;
    size_t depth_cap;
    float *vert_buf;
    float *norm_buf;
    size_t ray_cap;
    // per pixel: {the smallest sample index found <= 0 so far by the ray march, that sample's value} in one 64-bit word (all ones
    // = none); every kernel of the march lowers it with atomicMin, resolve_hits_kernel turns it into the vertex and resets it (raycast.hip)
    uint32_t *release_word;  // when set: the cell-parallel cast's first kernel stores release_value there as it starts (tsdf_pipeline_step, scheduling)
    uint32_t release_value;
    hipEvent_t after_bulk;   // when set: recorded on the volume's stream right behind the bulk ray kernel's launch (tsdf_pipeline_step, scheduling)
    uint64_t *ray_best;      // two copies of ray_best_cap words, used alternately (ray_best_side)
    size_t ray_best_cap;
    int ray_best_side;
    size_t ray_best_pixels;  // image size of the last march (a different one refills both copies)
    int ray_best_dirty;      // 1 = a march was started whose resolve kernel was not launched: refill before the next one
    // stretches of rays the first ray-cast kernel hands to the tail kernel: uint2 per piece + {appended, taken}
    void *tail_entries;
    uint32_t *tail_count;
    size_t tail_cap;
    // the cell-parallel cast (raycast_cells.hpp): one 32-byte record per pixel, the list of flagged bricks (counter in front), and the
    // list's length of the previous cast in pinned host memory (which kernels the next cast takes)
    void *cell_rays;
    size_t cell_rays_cap;
    uint32_t *cell_bricks;
    uint32_t *cell_count_scratch;   // two words for count_cell_bricks_kernel (raycast.hip)
    tsdf::CastChooser chooser;
    int cell_recount_wait;          // casts that kept the march because the list was over the limit, since the last recount
    size_t cell_bricks_cap;
    uint32_t *cell_cast_host;
    int last_cast_cells;     // 1 = the last ray cast of this volume took the cell-parallel kernels (tsdf_volume_last_raycast_kind)
    // Dispatch order of the first ray-cast kernel, learnt from the previous cast (scheduling only): ray_heavy[range][workgroup] = 1
    // when a wave of that (sample range, tile) used its whole pass budget, ray_order[z][i] = range << 16 | tile slot that the i-th
    // workgroup of the z-th slab of the launch takes -- the heavy pairs of its XCD first (raycast.hip: order_ray_tiles).
    uint8_t *ray_heavy;
    uint32_t *ray_order;
    uint32_t ray_order_tiles, ray_order_ranges;   // what the two arrays were sized (and the order was built) for
    int ray_order_valid;
    // per-tile entry bound of a whole-volume cast (EntryParams): two copies of ztile_words words used alternately -- the launch that
    // fills one resets the other
    uint32_t *ztile;
    size_t ztile_words;
    int ztile_side;
    // T[k]: the ray parameter of sample k, T[0] = 0, T[k+1] = T[k] + step in fp32 (raycast.hip)
    float *t_table;
    // 1 = dividing by each voxel edge via the 3-instruction reciprocal sequence was verified exhaustively
    // against IEEE division for this volume's voxel size (volume.hip: verify_fast_division)
    int fast_div;
    unsigned long long fast_div_mismatches;
    // brick occupancy (see OccGrid)
    tsdf::OccGrid occ;
    int occ_dirty;   // 1 = the flags do not cover the distances (upload, new truncation): rebuild before the next ray cast
    // THE INVARIANT every writer of distances keeps (the incremental rebuild of volume.hip skips bricks whose `fine` flag is clear and
    // trusts their summary bits): either it sets fine[b] for every brick b whose grown box holds a voxel it leaves low (or, in the rim
    // zone, not flat) and marks the integrate bricks it wrote in `touched` -- integrate's mark_low_voxels does -- or it sets occ_scan_all
    // (with occ_dirty) so that the next rebuild reads everything: set_distance_data, mark_dirty, clear, a new truncation distance do.
    // A writer that only set occ_dirty would leave stale zero summary bits behind, and both casts would miss hits
    // (tests/test_occupancy.py::test_incremental_rebuild_equals_a_full_scan pins the two rebuilds against each other).
    // 1 = the periodic tightening is due (integrate.hip).  The flags still cover the distances -- integrate only ever sets them --
    // they are just looser than they could be, so this rebuild may also run BESIDE the ray cast that follows (occupancy_tighten_on:
    // the per-frame pipeline puts it on its second stream; every byte it writes is a true statement about the distances integrate
    // left, and so is the byte it replaces) as long as the next writer of flags or distances waits for it (occupancy_join).
    int occ_tighten_due;
    hipEvent_t occ_tightened;   // recorded behind a tightening enqueued on another stream ...
    int occ_tighten_pending;    // ... that the volume's own stream has not waited for yet
    int reach_dirty; // 1 = `fine` changed since `reach` was computed
    uint16_t *occ_bits;              // 16 summary bits per brick, kept between rebuilds (volume.hip)
    uint8_t *occ_rim_bits;           // 8 more: which 2^3-voxel octants hold a voxel that is not flat (read for boundary bricks only)
    // A rebuild reads only the distances integrate may have written since the previous one: integrate_kernel marks its brick
    // in `touched` (one byte per integrate brick, index order), the scan skips the others and their summary bits stand.
    // occ_scan_all = 1: the next rebuild reads everything (first rebuild, clear, set_distance_data, mark_dirty, new truncation).
    uint8_t *touched;
    uint32_t touched_nx, touched_ny, touched_nz;
    int occ_scan_all;
    // integrate only ever SETS flags (a voxel that stops being low is not noticed), so the flags are refreshed from
    // the distance array after 2, 4, 8, 16 integrations and then every kOccRebuildPeriod (integrate.hip)
    uint32_t integrations_since_rebuild;
    uint32_t integrations_total;
    // integrate scratch: compact list of bricks a frame can touch (+ its counter), depth tile maxima
    uint32_t *brick_list;
    size_t brick_list_cap;
    tsdf::PreparedCull prepared;  // the brick list on the device was built for these arguments ...
    int prepared_valid;           // ... by a prepare call that no integrate has used yet
    uint32_t brick_count_side;    // which of the list's two length words the next integration appends behind (integrate.hip)
    uint32_t *brick_boxes;   // uint4 per active brick: pixel box of the brick's projection
    size_t brick_box_cap;
    uint16_t *tile_max;
    size_t tile_max_cap;
    float *plane_const;      // float4 per resident plane (+ padding): z-only terms of the projection
    // the depth image of the brick list on the device inside a ring of zeros, (w + 2) x (h + 2): brick_cull_kernel's side job when the
    // weights are packed; integrate_packed_kernel's bricks without an LDS tile look their depths up in it
    uint16_t *depth_pad;
    uint32_t depth_pad_w, depth_pad_h;
    // optional HIP-event timing of the two dominant kernels on the volume's stream (tsdf_volume_set_timing)
    int timing;                   // 0 = off, n = every n-th launch of each kernel is bracketed
    uint32_t timing_launches[3];
    std::vector<hipEvent_t> *tev[3];  // [0] integrate_kernel, [1] process_ray_kernel, [2] process_ray_tail_kernel: start/stop pairs
    // diagnostics
    int counting;
    unsigned long long *counter_dev;  // [0] = updated voxels, [1] = samples, [2] = hits
    uint64_t last_updated;
    size_t resident_voxels() const { return (size_t)g.X * g.Y * (g.z_store_end - g.z_store_begin); }
};

struct tsdf_bilateral {
    float sigma_colour, sigma_space;
    int radius;
    int device;
    int scale_exact;         // every nonzero tap weight is >= 2^-120: the staged kernel may carry 4 * sum (bilateral.hip)
    float *kernel_dev;       // (2r+1)^2
    float *similarity_dev;   // 65536 entries (first 256 = the reference's table)
    void *img_in;            // cached device images for the host variants
    void *img_out;
    size_t img_cap;
};
