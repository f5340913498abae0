// Volume lifecycle and data access behind the C ABI (include/tsdf_amd.h).
// Replaces the host side of src/TSDF/TSDFVolume.cu (ctors, set_size, clear, accessors) of the
// reference.  HBM layout: two dense fp32 arrays (distance, weight), x fastest, one contiguous
// range of planes per object; the 24-byte/voxel deformation grid stays implicit until a caller
// asks for it and the colour array, which no kernel of the path touches, is
// not allocated at all.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include <algorithm>

#include "common.hpp"

namespace tsdf {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const Tuning &tuning() {
    static const Tuning t = [] {
        auto num = [](const char *name, int fallback) { const char *e = getenv(name); return e ? atoi(e) : fallback; };
        auto clamp = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
        Tuning u;
        u.ray_segments = clamp(num("TSDF_RAY_SEGMENTS", 6), 1, 64);
        u.ray_slab_ranges = clamp(num("TSDF_RAY_SLAB_RANGES", 0), 0, 64);
        u.ray_cells = clamp(num("TSDF_RAY_CELLS", 1), 0, 2);
        u.ray_cells_limit = std::max(num("TSDF_RAY_CELLS_LIMIT", 131072), 0);
        u.ray_cells_grid = clamp(num("TSDF_RAY_CELLS_GRID", 2048), 1, 65535);
        u.ray_cells_sort = clamp(num("TSDF_RAY_CELLS_SORT", 1), 0, 2);
        u.ray_chooser = clamp(num("TSDF_RAY_CHOOSER", 0), 0, 2);
        u.ray_cells_pairs = clamp(num("TSDF_RAY_CELLS_PAIRS", 1024), 0, 1 << 24);
        u.ray_cells_look = clamp(num("TSDF_RAY_CELLS_LOOK", 1), 0, 1);
        { const char *fp = getenv("TSDF_RAY_CELLS_FOOTPRINT"); u.ray_cells_footprint = fp ? (float)atof(fp) : 10.0f; }
        u.ray_trip_budget = std::max(num("TSDF_RAY_TRIP_BUDGET", 22), 1);
        u.ray_tail_lanes = num("TSDF_RAY_TAIL_LANES", 4);
        if (!(u.ray_tail_lanes >= 1 && u.ray_tail_lanes <= 64 && (u.ray_tail_lanes & (u.ray_tail_lanes - 1)) == 0)) u.ray_tail_lanes = 4;
        u.ray_tail_grid = std::max(num("TSDF_RAY_TAIL_GRID", 256 * 10), 1);
        u.ray_tail_piece = std::max(num("TSDF_RAY_TAIL_PIECE", 64), 1);
        u.ray_range_order = clamp(num("TSDF_RAY_RANGE_ORDER", 1), 0, 2);
        u.ray_tile_map = clamp(num("TSDF_RAY_TILE_MAP", 2), 0, 2);
        u.ray_learned_order = num("TSDF_RAY_LEARNED_ORDER", 1) != 0;
        u.ray_heavy_passes = std::max(num("TSDF_RAY_HEAVY_PASSES", 0), 0);
        u.ray_entry_bound = num("TSDF_RAY_ENTRY_BOUND", 1) != 0;
        u.icp_persistent = num("TSDF_ICP_PERSISTENT", 0);
        u.occ_rebuild_period = std::max(num("TSDF_OCC_REBUILD_PERIOD", 16), 0);
        u.occ_scan_all = num("TSDF_OCC_SCAN_ALL", 0) != 0;
        u.reach_lds = num("TSDF_REACH_LDS", 0) != 0;
        u.int_grid_per_cu = num("TSDF_INT_GRID_PER_CU", 0);
        u.weight_pack = num("TSDF_WEIGHT_PACK", 8);
        if (u.weight_pack != 0 && u.weight_pack != 16) u.weight_pack = 8;
        u.pipe_release = clamp(num("TSDF_PIPE_RELEASE", 0), 0, 2);
        u.pipe_host_wait = num("TSDF_PIPE_HOST_WAIT", 0) != 0;
        u.pipe_word_release = num("TSDF_PIPE_WORD_RELEASE", 0) != 0;
        u.event_scope = clamp(num("TSDF_EVENT_SCOPE", 2), 0, 2);
        u.timing_bracket = num("TSDF_TIMING_BRACKET", 0) != 0;
        u.verbose = getenv("TSDF_VERBOSE") != nullptr;
        u.debug_rays = getenv("TSDF_DEBUG_RAYS") != nullptr;
        return u;
    }();
    return t;
}

int hip_fail(hipError_t e, const char *what) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? TSDF_ERR_NOMEM : TSDF_ERR_DEVICE;
}

// ---- fill kernels ------------------------------------------------------------------------
// clear() is a pure streaming write: 16 B per lane per store, grid-stride, 2048 blocks.
// (the weights are zeroed by weights_clear, weights.hip: a memset of however many bytes they take)
__global__ __launch_bounds__(256) void fill_kernel(float *__restrict__ dist, size_t n, float dval) {
    size_t n4 = n >> 2;
    float4 d4 = make_float4(dval, dval, dval, dval);
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) reinterpret_cast<float4 *>(dist)[i] = d4;
    // tail (n not a multiple of 4)
    size_t t = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dist[t] = dval;
}

// float4 streaming copy, one 16-byte element per thread and four per loop trip (tsdf_measure_copy_bandwidth)
__global__ __launch_bounds__(256) void copy4_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// In-place update of two 512^3 float arrays in integrate_kernel's walk (tsdf_measure_update_bandwidth): one workgroup per
// 64 x 4 x 32 brick, lane <-> x, 4 planes in flight, the running-mean arithmetic but no projection.
__global__ __launch_bounds__(256) void update_walk_kernel(float *__restrict__ d, float *__restrict__ w) {
    const unsigned b = blockIdx.x, bx = b % 8, by = (b / 8) % 128, bz = b / (8 * 128);
    const size_t plane = (size_t)512 * 512;
    const size_t idx = (size_t)bz * 32 * plane + (size_t)(by * 4 + threadIdx.y) * 512 + bx * 64 + threadIdx.x;
#pragma unroll 1
    for (int z = 0; z < 32; z += 4) {
        float pd[4], pw[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { pd[j] = d[idx + (z + j) * plane]; pw[j] = w[idx + (z + j) * plane]; }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float nw = pw[j] + 1.0f;
            d[idx + (z + j) * plane] = (pd[j] * pw[j] + 3.0f) / nw;
            w[idx + (z + j) * plane] = nw;
        }
    }
}

// initialise_deformation (src/TSDF/TSDFVolume.cu:768-794) for the materialised node array.
__global__ __launch_bounds__(256) void init_nodes_kernel(tsdf_deformation_node *nodes, Geom g) {
    uint32_t vx = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t vy = blockIdx.y;
    uint32_t vz = g.z_store_begin + blockIdx.z;
    if (vx >= g.X) return;
    size_t idx = (size_t)g.X * g.Y * (vz - g.z_store_begin) + (size_t)g.X * vy + vx;
    tsdf_deformation_node nd;
    nd.translation[0] = (((int)vx + 0.5f) * g.vs.x) + g.offset.x;
    nd.translation[1] = (((int)vy + 0.5f) * g.vs.y) + g.offset.y;
    nd.translation[2] = (((int)vz + 0.5f) * g.vs.z) + g.offset.z;
    nd.rotation[0] = 0.0f;
    nd.rotation[1] = 0.0f;
    nd.rotation[2] = 0.0f;
    nodes[idx] = nd;
}

// Occupancy of a cleared volume: every distance is +trunc, so nothing is flagged (bricks at the grid boundary included: +trunc is
// flat) but the cell bricks that hold dual cells beyond the grid, which are set for good.
__global__ __launch_bounds__(256) void occupancy_init_kernel(OccGrid occ, uint32_t size_x, uint32_t size_y, uint32_t size_z) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < occ.fine_count()) {
        uint32_t bx = i % occ.nbx, by = (i / occ.nbx) % occ.nby, bz = i / ((size_t)occ.nbx * occ.nby);
        occ.fine[i] = 0;
        // cell brick b holds dual cells 4b .. 4b+3; cells exist for lower corners 0 .. size-2
        bool partial = bx * kBrick + kBrick > size_x - 1 || by * kBrick + kBrick > size_y - 1 || bz * kBrick + kBrick > size_z - 1;
        occ.cell[i] = partial ? 1 : 0;
    }
}

// Rebuild of the occupancy from the distance array, in two steps.
//
// 1. occupancy_scan_kernel streams the resident planes once (float4 per lane, coalesced) and leaves 16 bits per
//    4^3 brick: bits 0-7 = which of the brick's eight 2^3-voxel octants hold a value that is not safely positive
//    (octant index xo + 2 yo + 4 zo); bits 8-15 = the same question for the whole brick (A), its x = 0, y = 0, z = 0
//    voxel layers (Fx, Fy, Fz), the three edges shared by two of those layers (Exy, Exz, Eyz) and the corner voxel (C).
//    A second byte per brick (rim_bits) holds the octants with a voxel that is not FLAT (OccGrid): read for boundary bricks only.
// 2. occupancy_flags_kernel combines the 27 neighbours: `fine[b]` = some octant inside the brick grown by kBrickGrow = 2
//    voxels (= one octant) is marked; `cell[b]` = some voxel in [4b, 4b+4]^3 is marked, i.e. the brick itself plus the
//    x/y/z = 0 layers, edges and corner of the bricks on its + side.  A brick touching the grid boundary is flagged when some
//    octant inside its grown box is not flat (common.hpp); partial cell bricks keep their permanent marks.
// Workgroup of scan: 64 bricks along x (lane) x the 4 voxel rows of one brick row (wave); loops over the brick's 4 planes.
// touched != nullptr (z_store_begin a multiple of 4): only the bricks inside integrate bricks marked there are read -- the
// distances of the others have not been written since their summary bits were formed, and those stand.
constexpr uint32_t kScanRows = 8;
__global__ __launch_bounds__(256) void occupancy_scan_kernel(const float *__restrict__ dist, Geom g, OccGrid occ,
                                                             uint16_t *__restrict__ bits, uint8_t *__restrict__ rim_bits,
                                                             const uint8_t *__restrict__ touched,
                                                             const uint32_t tnx, const uint32_t tny, const uint32_t tnz) {
    __shared__ uint32_t acc[64];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // kScanRows brick rows a workgroup, one after the other: the incremental rebuild's launch is all but empty -- 2 % of the bricks
    // integrate touches hold a flag -- and a workgroup that only finds that out costs the dispatcher what a working one does: 32 768
    // workgroups at 512^3 were 35 us of a launch with a few microseconds of reading in it (round 6: every 16th step of a stream was
    // 1.11 x the median through it)
    for (uint32_t row = 0; row < kScanRows; row++) {
    const uint32_t bx = blockIdx.x * 64 + lane, by = blockIdx.y * kScanRows + row, bz = blockIdx.z;
    if (by >= occ.nby) break;   // (uniform)
    const uint32_t y = by * kBrick + wave;
    bool scan = true;
    if (touched) {
        // the integrate brick (64 x 4 x 32 voxels from plane z_store_begin; a slab's halo planes hang on the last layer) that holds
        // this 4^3 brick
        const uint32_t z0 = bz * kBrick;
        const uint32_t ibx = min(bx * kBrick / kIntBrickX, tnx - 1u), iby = min(by * kBrick / kIntBrickY, tny - 1u);
        const uint32_t ibz = z0 >= g.z_store_begin ? min((z0 - g.z_store_begin) / kIntBrickZ, tnz - 1u) : 0u;
        scan = touched[((size_t)ibz * tny + iby) * tnx + ibx] != 0;
        // ... and of those only the bricks whose own flag is set (round 5).  A rebuild can only CLEAR flags -- the flags in force are the
        // exact ones of the last rebuild plus what integrate has set since, and integrate sets fine[b] for every voxel of b it leaves low
        // (or, in the rim zone, not flat) -- so a brick whose flag is clear holds no such voxel now, held none when its flag was last
        // computed (b lies inside its own grown box), and its summary bits, all zero since then, still say so.  2 % of the bricks
        // integrate touches hold a flag: the scan of a 512^3 stream fell from 40-60 us to a few.
        scan = scan && bx < occ.nbx && occ.fine[((size_t)bz * occ.nby + by) * occ.nbx + bx] != 0;
        if (__syncthreads_or(scan) == 0) continue;   // nothing in reach of this row needs a look
    }
    if (threadIdx.x < 64) acc[threadIdx.x] = 0;
    __syncthreads();
    uint32_t oct = 0, low = 0, rim = 0;
    if (scan && bx < occ.nbx && y < g.Y) {
        const uint32_t x0 = bx * kBrick;
        const bool vec = (g.X & 3u) == 0;  // rows are 16-byte aligned and a brick never straddles the row end
        const uint32_t y0f = wave == 0 ? 1u : 0u, yo = wave >> 1;
        for (uint32_t j = 0; j < (uint32_t)kBrick; j++) {
            const uint32_t z = bz * kBrick + j;
            if (z < g.z_store_begin || z >= g.z_store_end) continue;  // not resident (or beyond the grid)
            const float *row = dist + ((size_t)g.X * g.Y * (z - g.z_store_begin) + (size_t)g.X * y + x0);
            uint32_t m = 0;  // bit i: voxel x0 + i is not safely positive (also true for NaN)
            uint32_t nf = 0; // bit i: voxel x0 + i is not flat
            if (vec) {
                const float4 d = *reinterpret_cast<const float4 *>(row);
                m = (!(d.x > occ.tau) ? 1u : 0u) | (!(d.y > occ.tau) ? 2u : 0u) | (!(d.z > occ.tau) ? 4u : 0u) | (!(d.w > occ.tau) ? 8u : 0u);
                nf = (occ.not_flat(d.x) ? 1u : 0u) | (occ.not_flat(d.y) ? 2u : 0u) | (occ.not_flat(d.z) ? 4u : 0u) | (occ.not_flat(d.w) ? 8u : 0u);
            } else {
                for (uint32_t i = 0; i < (uint32_t)kBrick; i++)
                    if (x0 + i < g.X) {
                        if (!(row[i] > occ.tau)) m |= 1u << i;
                        if (occ.not_flat(row[i])) nf |= 1u << i;
                    }
            }
            if (nf) {
                const uint32_t zo = j >> 1;
                if (nf & 3u) rim |= 1u << (0 + 2 * yo + 4 * zo);
                if (nf & 12u) rim |= 1u << (1 + 2 * yo + 4 * zo);
            }
            if (m) {
                const uint32_t zo = j >> 1, z0f = j == 0 ? 1u : 0u, fx = m & 1u;
                if (m & 3u) oct |= 1u << (0 + 2 * yo + 4 * zo);
                if (m & 12u) oct |= 1u << (1 + 2 * yo + 4 * zo);
                low |= 1u | (fx << 1) | (y0f << 2) | (z0f << 3) | ((fx & y0f) << 4) | ((fx & z0f) << 5) | ((y0f & z0f) << 6) |
                       ((fx & y0f & z0f) << 7);
            }
        }
    }
    const uint32_t both = oct | (low << 8) | (rim << 16);
    if (both) atomicOr(&acc[lane], both);
    __syncthreads();
    if (threadIdx.x < 64 && bx < occ.nbx && scan) {
        bits[((size_t)bz * occ.nby + by) * occ.nbx + bx] = (uint16_t)acc[lane];
        rim_bits[((size_t)bz * occ.nby + by) * occ.nbx + bx] = (uint8_t)(acc[lane] >> 16);
    }
    __syncthreads();   // (acc is zeroed again for the next row)
    }
}

__global__ __launch_bounds__(256) void occupancy_flags_kernel(const uint16_t *__restrict__ bits, const uint8_t *__restrict__ rim_bits, OccGrid occ, uint32_t size_x,
                                                              uint32_t size_y, uint32_t size_z, uint8_t *__restrict__ touched,
                                                              const uint32_t n_touched, const bool incremental) {
    // (several bricks a thread: like the scan, the incremental launch mostly finds clear flags and leaves)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < max(occ.fine_count(), (size_t)n_touched); i += (size_t)gridDim.x * 256) {
    if (touched && i < n_touched) touched[i] = 0;   // (the scan before this launch has read the marks: they start again)
    if (i >= occ.fine_count()) continue;
    // incremental: a clear flag stays clear (a rebuild only clears, see the scan), and with it the cell flag, which is either clear too
    // -- [4b, 4b+4]^3 lies inside the grown box -- or one of occupancy_init_kernel's permanent marks
    if (incremental && occ.fine[i] == 0) continue;
    const int bx = (int)(i % occ.nbx), by = (int)((i / occ.nbx) % occ.nby), bz = (int)(i / ((size_t)occ.nbx * occ.nby));
    // a brick touching the grid boundary asks for flat voxels, not just positive ones (OccGrid)
    const bool boundary = bx == 0 || by == 0 || bz == 0 || bx + 1 == (int)occ.nbx || by + 1 == (int)occ.nby || bz + 1 == (int)occ.nbz;
    bool fine = false;
    // the permanent marks of occupancy_init_kernel
    bool cell = bx * kBrick + kBrick > (int)size_x - 1 || by * kBrick + kBrick > (int)size_y - 1 || bz * kBrick + kBrick > (int)size_z - 1;
    // octants of a neighbour at offset d that lie within 2 voxels of this brick: all (d = 0), the high half (d = -1),
    // the low half (d = +1); per axis, as masks over the octant index xo + 2 yo + 4 zo
    // (No branch around the 27 look-ups -- a neighbour outside the grid is read at a clamped index and masked to 0 -- so that
    // they are all requested before the first is waited for: with `continue` for the outside ones the loads went out one by
    // one, 38 us for 2 M bricks at 512^3.)
    uint32_t fine_acc = 0, cell_acc = 0, rim_acc = 0;
#pragma unroll
    for (int dz = -1; dz <= 1; dz++) {
        const int z = bz + dz;
        const bool zin = z >= 0 && z < (int)occ.nbz;
        const uint32_t mz = dz < 0 ? 0xF0u : dz == 0 ? 0xFFu : 0x0Fu;
#pragma unroll
        for (int dy = -1; dy <= 1; dy++) {
            const int y = by + dy;
            const bool yin = zin && y >= 0 && y < (int)occ.nby;
            const uint32_t my = dy < 0 ? 0xCCu : dy == 0 ? 0xFFu : 0x33u;
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) {
                const int x = bx + dx;
                const bool in = yin && x >= 0 && x < (int)occ.nbx;
                const uint32_t mx = dx < 0 ? 0xAAu : dx == 0 ? 0xFFu : 0x55u;
                const size_t at = in ? ((size_t)z * occ.nby + y) * occ.nbx + x : i;
                const uint32_t w = (uint32_t)bits[at] & (in ? 0xffffu : 0u);
                fine_acc |= w & mx & my & mz;
                if (boundary) rim_acc |= (uint32_t)rim_bits[at] & (in ? 0xffu : 0u) & mx & my & mz;
                if (dx >= 0 && dy >= 0 && dz >= 0) {
                    // which summary of the + side neighbour touches [4b, 4b+4]^3: A, Fx, Fy, Exy, Fz, Exz, Eyz, C
                    constexpr uint32_t sel[8] = {1u << 8, 1u << 9, 1u << 10, 1u << 12, 1u << 11, 1u << 13, 1u << 14, 1u << 15};
                    cell_acc |= w & sel[dx + 2 * dy + 4 * dz];
                }
            }
        }
    }
    fine = fine_acc != 0u || rim_acc != 0u;
    cell = cell || cell_acc != 0u;
    occ.fine[i] = fine ? 1 : 0;
    occ.cell[i] = cell ? 1 : 0;
    }
}

// ---- reach[b]: size class of the largest EMPTY aligned block of bricks that contains brick b:
//   0 = b itself is flagged; l >= 1 = the aligned block of 2^(l-1) bricks per side (4 * 2^(l-1) voxels) is clear,
// up to l = kReachLevels (64 voxels).  One workgroup per 16^3-brick super block: OR-reduce in LDS, then every
// brick looks up its ancestors.  Bricks outside the grid count as flagged, so blocks that stick out are never
// reported empty.  O(1) work per brick, ~2 x 2 MiB of traffic at 512^3.
constexpr int kSuper = 16;  // bricks per side of a super block = 2^(kReachLevels - 1)

// ---- entry bound of a whole-volume ray cast (EntryParams, common.hpp): side job of the reach summary's launch, which holds the
// OR of `fine` over every aligned unit of 4^3 bricks anyway.  One lane per flagged unit.
__device__ inline void entry_reset_next(const EntryParams &ep, uint32_t global_thread, uint32_t n_threads) {
    const uint32_t n = ep.tiles_x * ep.tiles_y;
    for (uint32_t i = global_thread; i <= n; i += n_threads) ep.ztile_next[i] = i < n ? kEntryFar : 1u;
}
__device__ inline void entry_project_unit(const EntryParams &ep, uint32_t ux, uint32_t uy, uint32_t uz) {
    constexpr int kUnit = kBrick * 4;   // voxels per side
    const float lo_x = (float)((int)(ux * kUnit) - 1), hi_x = (float)(ux * kUnit + kUnit + 1);   // voxel coordinates, grown by one
    const float lo_y = (float)((int)(uy * kUnit) - 1), hi_y = (float)(uy * kUnit + kUnit + 1);
    const float lo_z = (float)((int)(uz * kUnit) - 1), hi_z = (float)(uz * kUnit + kUnit + 1);
    float zmin = INFINITY, zmax = -INFINITY, umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY;
    bool nan = false;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const float wx = ((c & 1) ? hi_x : lo_x) * ep.vs.x + ep.offset.x;
        const float wy = ((c & 2) ? hi_y : lo_y) * ep.vs.y + ep.offset.y;
        const float wz = ((c & 4) ? hi_z : lo_z) * ep.vs.z + ep.offset.z;
        const float cx = ep.r[0][0] * wx + ep.r[0][1] * wy + ep.r[0][2] * wz + ep.r[0][3];
        const float cy = ep.r[1][0] * wx + ep.r[1][1] * wy + ep.r[1][2] * wz + ep.r[1][3];
        const float cz = ep.r[2][0] * wx + ep.r[2][1] * wy + ep.r[2][2] * wz + ep.r[2][3];
        nan = nan || !(cz == cz);
        zmin = fminf(zmin, cz);
        zmax = fmaxf(zmax, cz);
        const float rz = 1.0f / cz;
        const float u = (ep.k[0][0] * cx + ep.k[0][1] * cy + ep.k[0][2] * cz) * rz, w = (ep.k[1][0] * cx + ep.k[1][1] * cy + ep.k[1][2] * cz) * rz;
        umin = fminf(umin, u); umax = fmaxf(umax, u);
        vmin = fminf(vmin, w); vmax = fmaxf(vmax, w);
    }
    // wholly behind the camera, with room to spare: a sample's depth is near + t >= 0, none lies in it
    if (!nan && zmax < -4.0f * ep.slack_z) return;
    const float bound = zmin - ep.slack_z;
    // a corner that is not safely in front of the camera (the hull argument needs one sign), or NaN anywhere: no bound for this view
    if (nan || !(bound > 4.0f * ep.slack_z) || !(umin <= umax) || !(vmin <= vmax)) {
        ep.ztile[ep.tiles_x * ep.tiles_y] = 0u;
        return;
    }
    const float u0 = umin - 2.0f, u1 = umax + 2.0f, v0 = vmin - 2.0f, v1 = vmax + 2.0f;
    if (u1 < 0.0f || v1 < 0.0f || u0 > (float)(ep.width - 1) || v0 > (float)(ep.height - 1)) return;   // off the image
    const uint32_t tx0 = (uint32_t)fmaxf(u0, 0.0f) / kEntryTile, tx1 = min((uint32_t)fminf(u1, (float)(ep.width - 1)) / kEntryTile, ep.tiles_x - 1);
    const uint32_t ty0 = (uint32_t)fmaxf(v0, 0.0f) / kEntryTile, ty1 = min((uint32_t)fminf(v1, (float)(ep.height - 1)) / kEntryTile, ep.tiles_y - 1);
    const uint32_t word = __float_as_uint(bound);   // (positive floats order like their bit patterns)
    for (uint32_t ty = ty0; ty <= ty1; ty++)
        for (uint32_t tx = tx0; tx <= tx1; tx++) atomicMin(&ep.ztile[ty * ep.tiles_x + tx], word);
}
__global__ __launch_bounds__(256) void reach_mip_kernel(OccGrid occ, const EntryParams ep) {
    __shared__ __align__(16) unsigned char f[kSuper * kSuper * kSuper];  // 4096
    __shared__ unsigned char l1[8 * 8 * 8], l2[4 * 4 * 4], l3[2 * 2 * 2], l4[1];
    const uint32_t ox = blockIdx.x * kSuper, oy = blockIdx.y * kSuper, oz = blockIdx.z * kSuper;
    // a row of 16 bricks along x is 16 contiguous bytes: one 128-bit access per thread when the grid allows it
    const bool vec = (occ.nbx & 15u) == 0 && oy + kSuper <= occ.nby && oz + kSuper <= occ.nbz;
    if (vec) {
        const uint32_t y = oy + (threadIdx.x & 15), z = oz + (threadIdx.x >> 4);
        const uint4 row = *reinterpret_cast<const uint4 *>(occ.fine + ((size_t)z * occ.nby + y) * occ.nbx + ox);
        *reinterpret_cast<uint4 *>(f + threadIdx.x * 16) = row;
    } else {
        for (uint32_t i = threadIdx.x; i < kSuper * kSuper * kSuper; i += 256) {
            const uint32_t x = ox + (i & 15), y = oy + ((i >> 4) & 15), z = oz + (i >> 8);
            f[i] = (x < occ.nbx && y < occ.nby && z < occ.nbz) ? occ.fine[((size_t)z * occ.nby + y) * occ.nbx + x] : (unsigned char)1;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 512; i += 256) {
        const uint32_t x = (i & 7) * 2, y = ((i >> 3) & 7) * 2, z = (i >> 6) * 2;
        unsigned char o = 0;
        for (int c = 0; c < 8; c++) o |= f[((z + (c >> 2)) << 8) | ((y + ((c >> 1) & 1)) << 4) | (x + (c & 1))];
        l1[i] = o;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const uint32_t i = threadIdx.x, x = (i & 3) * 2, y = ((i >> 2) & 3) * 2, z = (i >> 4) * 2;
        unsigned char o = 0;
        for (int c = 0; c < 8; c++) o |= l1[((z + (c >> 2)) << 6) | ((y + ((c >> 1) & 1)) << 3) | (x + (c & 1))];
        l2[i] = o;
    }
    __syncthreads();
    if (ep.ztile) {   // the entry bound of the cast this summary is made for: one thread per flagged unit of 4^3 bricks
        entry_reset_next(ep, ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 256u + threadIdx.x, gridDim.x * gridDim.y * gridDim.z * 256u);
        if (threadIdx.x < 64) {
            // (units partly beyond the grid count as flagged here -- bricks outside it do, above -- and are projected like any other:
            // conservative; units wholly beyond it are left out)
            const uint32_t i = threadIdx.x, ux = ox / 4 + (i & 3), uy = oy / 4 + ((i >> 2) & 3), uz = oz / 4 + (i >> 4);
            if (l2[i] && ux < ep.units_x && uy < ep.units_y && uz < ep.units_z) entry_project_unit(ep, ux, uy, uz);
        }
    }
    if (threadIdx.x < 8) {
        const uint32_t i = threadIdx.x, x = (i & 1) * 2, y = ((i >> 1) & 1) * 2, z = (i >> 2) * 2;
        unsigned char o = 0;
        for (int c = 0; c < 8; c++) o |= l2[((z + (c >> 2)) << 4) | ((y + ((c >> 1) & 1)) << 2) | (x + (c & 1))];
        l3[i] = o;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned char o = 0;
        for (int c = 0; c < 8; c++) o |= l3[c];
        l4[0] = o;
    }
    __syncthreads();
    // every thread classifies one row of 16 bricks (i = its index in the super block)
    unsigned char levels[16];
    const uint32_t row0 = vec ? threadIdx.x * 16 : 0;
    {
        for (uint32_t e = 0; e < 16; e++) {
            const uint32_t i = vec ? row0 + e : threadIdx.x + e * 256;
            const uint32_t lx = i & 15, ly = (i >> 4) & 15, lz = i >> 8;
            unsigned char level = 0;
            if (!f[i]) {
                level = 1;
                if (!l1[((lz >> 1) << 6) | ((ly >> 1) << 3) | (lx >> 1)]) {
                    level = 2;
                    if (!l2[((lz >> 2) << 4) | ((ly >> 2) << 2) | (lx >> 2)]) {
                        level = 3;
                        if (!l3[((lz >> 3) << 2) | ((ly >> 3) << 1) | (lx >> 3)]) level = l4[0] ? 4 : 5;
                    }
                }
            }
            levels[e] = level;
            if (!vec) {
                const uint32_t x = ox + lx, y = oy + ly, z = oz + lz;
                if (x < occ.nbx && y < occ.nby && z < occ.nbz) occ.reach[((size_t)z * occ.nby + y) * occ.nbx + x] = level;
            }
        }
    }
    if (vec) {
        const uint32_t y = oy + (threadIdx.x & 15), z = oz + (threadIdx.x >> 4);
        uint4 out;
        out.x = levels[0] | (levels[1] << 8) | (levels[2] << 16) | ((uint32_t)levels[3] << 24);
        out.y = levels[4] | (levels[5] << 8) | (levels[6] << 16) | ((uint32_t)levels[7] << 24);
        out.z = levels[8] | (levels[9] << 8) | (levels[10] << 16) | ((uint32_t)levels[11] << 24);
        out.w = levels[12] | (levels[13] << 8) | (levels[14] << 16) | ((uint32_t)levels[15] << 24);
        *reinterpret_cast<uint4 *>(occ.reach + ((size_t)z * occ.nby + y) * occ.nbx + ox) = out;
    }
}

// The same summary for grids whose brick counts are multiples of 16 on every axis (every super block whole), one WAVE per super
// block and no LDS, no barrier: lane = (y, zg) holds the 4 rows z = 4 zg .. 4 zg + 3 of 16 flag bytes each (0 / 1, the only
// values the writers store); the OR over an aligned block is built in the byte domain along x (shifts inside the words), along
// z inside the lane and across lanes 16 / 32 apart, along y across neighbouring lanes.  Flagged-ness grows with the block (a
// flagged brick flags every block that holds it), so the level is 5 minus the number of flagged blocks among {brick, 2-, 4-, 8-,
// 16-block}: a byte-wise sum.  reach_mip_kernel above: 512 workgroups at 512^3 that spend their 9.5 us in five barriers.
__device__ inline uint32_t or_x2(uint32_t w) { const uint32_t a = (w | (w >> 8)) & 0x00ff00ffu; return a | (a << 8); }   // byte pairs (x, x ^ 1)
__device__ inline uint32_t or_x4(uint32_t w) { const uint32_t a = (w | (w >> 16)) & 0x0000ffffu; return a | (a << 16); }   // (of a pair-uniform word)
__device__ inline uint32_t or_lanes(uint32_t v, int mask) { return v | (uint32_t)__shfl_xor((int)v, mask); }
__global__ __launch_bounds__(64) void reach_mip_wave_kernel(OccGrid occ, const EntryParams ep) {
    // blockIdx.x counts the super blocks along Z, blockIdx.z those along X: workgroups go to the 8 XCDs round robin in launch order
    // (x fastest), and the super blocks that are neighbours along X read the same 128-byte lines -- with X as the fastest launch
    // index every line was fetched by up to 8 XCDs (16.8 MB fetched for 2 MiB of flags at 512^3)
    const uint32_t lane = threadIdx.x, y = blockIdx.y * kSuper + (lane & 15u), z0 = blockIdx.x * kSuper + (lane >> 4) * 4u;
    const size_t row0 = ((size_t)z0 * occ.nby + y) * occ.nbx + (size_t)blockIdx.z * kSuper, zstride = (size_t)occ.nby * occ.nbx;
    uint32_t f[4][4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint4 r = *reinterpret_cast<const uint4 *>(occ.fine + row0 + j * zstride);
        f[j][0] = r.x; f[j][1] = r.y; f[j][2] = r.z; f[j][3] = r.w;
    }
    uint32_t l1[2][4], l2[4], l3[2], l4;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // 2-blocks: x pair, z pair (rows 2 jp, 2 jp + 1), y pair (lane ^ 1)
#pragma unroll
        for (int jp = 0; jp < 2; jp++) l1[jp][k] = or_lanes(or_x2(f[2 * jp][k]) | or_x2(f[2 * jp + 1][k]), 1);
        // 4-blocks: x quad, the lane's 4 rows, y quad (the pair's neighbour pair: lane ^ 2)
        l2[k] = or_lanes(or_x4(l1[0][k]) | or_x4(l1[1][k]), 2);
    }
    // 8-blocks: two words along x, lanes y ^ 4, the neighbouring z group (lane ^ 16)
#pragma unroll
    for (int h = 0; h < 2; h++) l3[h] = or_lanes(or_lanes(l2[2 * h] | l2[2 * h + 1], 4), 16);
    l4 = or_lanes(or_lanes(l3[0] | l3[1], 8), 32);
    if (ep.ztile) {
        // The entry bound of the cast this summary is made for.  l2[k] of lane (y, zg) is the OR over the unit of 4^3 bricks (k, y / 4, zg)
        // of this super block: the lane with y % 4 == k projects it -- 64 units, one per lane.
        entry_reset_next(ep, ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64u + lane, gridDim.x * gridDim.y * gridDim.z * 64u);
        const uint32_t kx = lane & 3u;
        const uint32_t w = kx == 0u ? l2[0] : kx == 1u ? l2[1] : kx == 2u ? l2[2] : l2[3];
        if (w != 0u) entry_project_unit(ep, blockIdx.z * 4u + kx, blockIdx.y * 4u + ((lane & 15u) >> 2), blockIdx.x * 4u + (lane >> 4));
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = 0x05050505u - (f[j][k] + l1[j >> 1][k] + l2[k] + l3[k >> 1] + l4);
        *reinterpret_cast<uint4 *>(occ.reach + row0 + j * zstride) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

static void set_occupancy_thresholds(tsdf_volume *v) {
    v->occ.tau = 0.01f * v->g.trunc;
    v->occ.flat_lo = 0.9375f * v->g.trunc;
    v->occ.flat_hi = (1.0f + 1.0f / 1024.0f) * v->g.trunc;
}

static int occupancy_reset(tsdf_volume *v) {
    size_t n = v->occ.fine_count();
    hipLaunchKernelGGL(occupancy_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, v->stream, v->occ, v->g.X, v->g.Y, v->g.Z);
    TSDF_HIP(hipGetLastError(), "occupancy reset");
    v->reach_dirty = 1;
    return TSDF_OK;
}

// The volume's stream waits for a tightening that was enqueued on another stream (before anything writes flags or distances,
// or frees them).
int occupancy_join(tsdf_volume *v) {
    if (v->occ_tighten_pending) {
        TSDF_HIP(hipStreamWaitEvent(v->stream, v->occ_tightened, 0), "occupancy: wait for the rebuild on the other stream");
        v->occ_tighten_pending = 0;
        v->reach_dirty = 1;   // `fine` has changed under the last summary
    }
    return TSDF_OK;
}

static int occupancy_rebuild_on(tsdf_volume *v, hipStream_t stream) {
    const size_t n = v->occ.fine_count();
    // (zeroed where they are allocated: an incremental rebuild reads the summary bits of bricks it does not scan, and "all zero" is what a
    // brick that was never scanned since clear() must say -- not left to the first rebuild being a full one)
    if (!v->occ_bits) {
        TSDF_HIP(hipMalloc((void **)&v->occ_bits, n * sizeof(uint16_t)), "occupancy scratch alloc");
        TSDF_HIP(hipMemsetAsync(v->occ_bits, 0, n * sizeof(uint16_t), stream), "occupancy scratch reset");
    }
    if (!v->occ_rim_bits) {
        TSDF_HIP(hipMalloc((void **)&v->occ_rim_bits, n), "occupancy scratch alloc");
        TSDF_HIP(hipMemsetAsync(v->occ_rim_bits, 0, n, stream), "occupancy scratch reset");
    }
    dim3 grid((v->occ.nbx + 63) / 64, (v->occ.nby + kScanRows - 1) / kScanRows, v->occ.nbz);
    const bool incremental = !tuning().occ_scan_all && !v->occ_scan_all && v->touched && (v->g.z_store_begin % kBrick) == 0;
    const uint32_t n_touched = v->touched ? v->touched_nx * v->touched_ny * v->touched_nz : 0u;
    TSDF_REQUIRE(n_touched <= n || !v->touched, "occupancy rebuild: more integrate bricks than occupancy bricks");
    hipLaunchKernelGGL(occupancy_scan_kernel, grid, dim3(256), 0, stream, v->dist, v->g, v->occ, v->occ_bits, v->occ_rim_bits,
                       incremental ? (const uint8_t *)v->touched : (const uint8_t *)nullptr, v->touched_nx, v->touched_ny, v->touched_nz);
    TSDF_HIP(hipGetLastError(), "occupancy scan");
    hipLaunchKernelGGL(occupancy_flags_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, stream, v->occ_bits, v->occ_rim_bits, v->occ,
                       v->g.X, v->g.Y, v->g.Z, v->touched, n_touched, incremental);
    TSDF_HIP(hipGetLastError(), "occupancy rebuild");
    v->occ_scan_all = 0;
    v->occ_dirty = 0;
    v->occ_tighten_due = 0;
    v->reach_dirty = 1;
    v->integrations_since_rebuild = 0;
    return TSDF_OK;
}

int occupancy_rebuild(tsdf_volume *v) {
    int rc = occupancy_join(v);
    if (rc != TSDF_OK) return rc;
    return occupancy_rebuild_on(v, v->stream);
}

// The periodic tightening on `stream`, which the caller has ordered behind the integrate that made it due.  What runs on the
// volume's stream meanwhile may READ the flags (a ray cast: whichever byte it sees, old or new, is true of the distances, which
// do not change); the next writer joins first.  Not taken (0 returned through *enqueued) when the flags are invalid rather than
// loose -- then the ray cast itself must rebuild first -- or when nothing is due.
int occupancy_tighten_on(tsdf_volume *v, hipStream_t stream) {
    if (!v->occ_tighten_due || v->occ_dirty || v->occ_tighten_pending) return TSDF_OK;
    if (!v->occ_tightened) TSDF_HIP(hipEventCreateWithFlags(&v->occ_tightened, stream_order_event_flags()), "occupancy event");
    int rc = occupancy_rebuild_on(v, stream);
    if (rc != TSDF_OK) return rc;
    TSDF_HIP(hipEventRecord(v->occ_tightened, stream), "occupancy event");
    v->occ_tighten_pending = 1;
    return TSDF_OK;
}

int occupancy_flags_refresh(tsdf_volume *v) {
    if (v->occ_dirty || v->occ_tighten_due) return occupancy_rebuild(v);
    return TSDF_OK;
}

int occupancy_refresh(tsdf_volume *v, const EntryParams *entry) {
    if (v->occ_dirty || v->occ_tighten_due) {
        int rc = occupancy_rebuild(v);
        if (rc != TSDF_OK) return rc;
    }
    EntryParams ep;
    memset(&ep, 0, sizeof(ep));   // (ztile == nullptr: no entry bound asked for)
    if (entry) ep = *entry;
    if (v->reach_dirty || entry) {   // (an entry bound is per view: the summary's launch is repeated for it even when `reach` is current)
        dim3 grid((v->occ.nbx + kSuper - 1) / kSuper, (v->occ.nby + kSuper - 1) / kSuper, (v->occ.nbz + kSuper - 1) / kSuper);
        const bool lds_variant = tuning().reach_lds != 0;   // (tuning aid: the workgroup variant always)
        const bool whole_blocks = v->occ.nbx % kSuper == 0 && v->occ.nby % kSuper == 0 && v->occ.nbz % kSuper == 0 &&
                                  (reinterpret_cast<uintptr_t>(v->occ.fine) & 15u) == 0 && (reinterpret_cast<uintptr_t>(v->occ.reach) & 15u) == 0;
        if (whole_blocks && !lds_variant)
            hipLaunchKernelGGL(reach_mip_wave_kernel, dim3(grid.z, grid.y, grid.x), dim3(64), 0, v->stream, v->occ, ep);
        else
            hipLaunchKernelGGL(reach_mip_kernel, grid, dim3(256), 0, v->stream, v->occ, ep);
        TSDF_HIP(hipGetLastError(), "occupancy summary");
        v->reach_dirty = 0;
    }
    return TSDF_OK;
}

// The ray caster's parameter table: the values t takes in the reference's loop (`t = t + step_size`,
// src/RayCaster/GPURaycaster.cu:324,361), produced by the same fp32 additions on the host.
int build_t_table(tsdf_volume *v) {
    const int len = 4402 + 2;
    std::vector<float> T(len);
    const float step_size = (float)((double)v->g.trunc * 0.05);
    volatile float t = 0.0f;  // volatile: one rounded fp32 addition per step, nothing folded
    for (int k = 0; k < len; k++) {
        T[k] = t;
        t = t + step_size;
    }
    if (!v->t_table) TSDF_HIP(hipMalloc((void **)&v->t_table, len * sizeof(float)), "ray table alloc");
    TSDF_HIP(hipMemcpy(v->t_table, T.data(), len * sizeof(float), hipMemcpyHostToDevice), "ray table upload");
    return TSDF_OK;
}

// Exhaustive proof for one denominator b: over all finite fp32 numerators a with |a| >= kFastDivMin,
// does the reciprocal sequence of raycast.hip (q0 = a*y, r = fma(-b,q0,a), q = fma(r,y,q0), y = RN(1/b))
// return the bits of the IEEE quotient a / b?  (Numerators outside that set take the IEEE division there.)
__global__ __launch_bounds__(256) void fastdiv_check_kernel(float b, unsigned long long *__restrict__ mismatches) {
    const float y = 1.0f / b;
    unsigned int bad = 0;
    // 2^32 patterns = 65536 blocks x 256 threads x 256 iterations
    const unsigned int base = (blockIdx.x * 256u + threadIdx.x) << 8;
    for (unsigned int i = 0; i < 256u; i++) {
        const float a = __uint_as_float(base + i);
        const float mag = fabsf(a);
        const bool in_domain = (mag < INFINITY) && (mag >= kFastDivMin);
        const float ref = a / b;
        const float q0 = a * y;
        const float r = __builtin_fmaf(-b, q0, a);
        const float q = __builtin_fmaf(r, y, q0);
        if (in_domain && __float_as_uint(q) != __float_as_uint(ref)) bad++;
    }
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_down(bad, o);
    if ((threadIdx.x & 63u) == 0 && bad) atomicAdd(mismatches, (unsigned long long)bad);
}

int verify_fast_division(tsdf_volume *v) {
    v->fast_div = 0;
    const float b[3] = {v->g.vs.x, v->g.vs.y, v->g.vs.z};
    TSDF_HIP(hipMemsetAsync(v->counter_dev + 3, 0, sizeof(unsigned long long), v->stream), "fastdiv counter");
    for (int i = 0; i < 3; i++) {
        if (i > 0 && b[i] == b[0]) continue;  // cubic voxels: one proof covers all axes
        if (i == 2 && b[2] == b[1]) continue;
        hipLaunchKernelGGL(fastdiv_check_kernel, dim3(65536), dim3(256), 0, v->stream, b[i], v->counter_dev + 3);
    }
    TSDF_HIP(hipGetLastError(), "fast division check");
    unsigned long long bad = 1;
    TSDF_HIP(hipMemcpyAsync(&bad, v->counter_dev + 3, sizeof(bad), hipMemcpyDeviceToHost, v->stream), "fast division check");
    TSDF_HIP(hipStreamSynchronize(v->stream), "fast division check");
    v->fast_div = (bad == 0) ? 1 : 0;
    v->fast_div_mismatches = bad;
    if (bad && tuning().verbose) fprintf(stderr, "tsdf: fast division not verified (%llu mismatches)\n", bad);
    return TSDF_OK;
}

static bool timing_brackets() { return tuning().timing_bracket != 0; }

// Start / stop events for the next launch of kernel `which` (filled by hipExtLaunchKernel with the dispatch's timestamps);
// false when this launch is not to be timed, or when the brackets are asked for.
bool timing_pair(tsdf_volume *v, int which, hipEvent_t *start, hipEvent_t *stop) {
    if (!v->timing || timing_brackets()) return false;
    if (v->timing_launches[which]++ % (uint32_t)v->timing != 0) return false;
    if (!v->tev[which]) v->tev[which] = new std::vector<hipEvent_t>();
    if (hipEventCreate(start) != hipSuccess) return false;
    if (hipEventCreate(stop) != hipSuccess) {
        (void)hipEventDestroy(*start);
        return false;
    }
    v->tev[which]->push_back(*start);
    v->tev[which]->push_back(*stop);
    return true;
}

void timing_begin(tsdf_volume *v, int which) {
    if (!v->timing || !timing_brackets()) return;
    if (v->timing_launches[which]++ % (uint32_t)v->timing != 0) return;   // every timing-th launch is bracketed
    if (!v->tev[which]) v->tev[which] = new std::vector<hipEvent_t>();
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, v->stream);
    v->tev[which]->push_back(e);
}

void timing_end(tsdf_volume *v, int which) {
    if (!v->timing || !timing_brackets() || !v->tev[which] || (v->tev[which]->size() & 1) == 0) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) {
        (void)hipEventDestroy(v->tev[which]->back());
        v->tev[which]->pop_back();
        return;
    }
    (void)hipEventRecord(e, v->stream);
    v->tev[which]->push_back(e);
}

static int init_nodes(tsdf_volume *v) {
    dim3 block(256, 1, 1);
    dim3 grid((v->g.X + 255) / 256, v->g.Y, v->g.z_store_end - v->g.z_store_begin);
    hipLaunchKernelGGL(init_nodes_kernel, grid, block, 0, v->stream, v->nodes, v->g);
    TSDF_HIP(hipGetLastError(), "initialise deformation nodes");
    return TSDF_OK;
}

// deformation_kernel + get_trilinear_elements + rotate (src/TSDF/TSDFVolume.cu:101-263): mesh points are pushed through
// the volume's deformation field -- the trilinear blend of the eight surrounding nodes' translations -- then through the
// global rotation and translation.  Kept as the reference has it, quirks included: the point is clamped with
// POINT_EPSILON = 0.001 (:22, :117-122), node positions carry no offset here (centre_of_voxel_at with its default, :136,
// :150), coefficients 6 and 7 belong to the other one's node (indices :164-171 vs coefficients :174-181).  The rotation
// matrix entries are the reference's float products of cos / sin of the three angles (:213-224); they are the same for
// every point, so the host forms them once (rot9, row-major).  Two cases the reference leaves undefined are defined here:
// a point outside the volume (it only prints, then blends uninitialised memory) is returned unchanged, and a neighbour
// index past the array (points in the last half voxel of an axis: lower + 1 is not clamped) is clamped to the last node.
// nodes == nullptr: the regular grid clear() leaves (voxel centre + the offset at clear time).
__global__ __launch_bounds__(256) void deform_points_kernel(const tsdf_deformation_node *__restrict__ nodes, const Geom g,
                                                            const Mat33 rot, const F3 translation, const int num_points,
                                                            float *__restrict__ points) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= num_points) return;
    const float kPointEpsilon = 0.001f;
    float px = points[idx * 3 + 0] - g.offset.x, py = points[idx * 3 + 1] - g.offset.y, pz = points[idx * 3 + 2] - g.offset.z;
    const float mx = g.X * g.vs.x, my = g.Y * g.vs.y, mz = g.Z * g.vs.z;
    float ax = px, ay = py, az = pz;
    if ((px > mx) && (px - mx < kPointEpsilon)) ax = mx - kPointEpsilon;
    if ((py > my) && (py - my < kPointEpsilon)) ay = my - kPointEpsilon;
    if ((pz > mz) && (pz - mz < kPointEpsilon)) az = mz - kPointEpsilon;
    if (px < -kPointEpsilon) ax = 0.0f;
    if (py < -kPointEpsilon) ay = 0.0f;
    if (pz < -kPointEpsilon) az = 0.0f;
    const int vx = f2i_sat(floorf(ax / g.vs.x)), vy = f2i_sat(floorf(ay / g.vs.y)), vz = f2i_sat(floorf(az / g.vs.z));
    if (!(vx >= 0 && vy >= 0 && vz >= 0 && (uint32_t)vx < g.X && (uint32_t)vy < g.Y && (uint32_t)vz < g.Z)) return;
    const float cx = (vx + 0.5f) * g.vs.x + 0.0f, cy = (vy + 0.5f) * g.vs.y + 0.0f, cz = (vz + 0.5f) * g.vs.z + 0.0f;
    int lx = (ax < cx) ? vx - 1 : vx, ly = (ay < cy) ? vy - 1 : vy, lz = (az < cz) ? vz - 1 : vz;
    lx = max(lx, 0); ly = max(ly, 0); lz = max(lz, 0);
    const float lcx = (lx + 0.5f) * g.vs.x + 0.0f, lcy = (ly + 0.5f) * g.vs.y + 0.0f, lcz = (lz + 0.5f) * g.vs.z + 0.0f;
    const float u = (ax - lcx) / g.vs.x, v = (ay - lcy) / g.vs.y, w = (az - lcz) / g.vs.z;
    const long long n_nodes = (long long)g.X * g.Y * g.Z;
    const long long dx = 1, dy = g.X, dz = (long long)g.X * g.Y;
    long long ind[8];
    ind[0] = lx + (ly * dy) + (lz * dz);
    ind[1] = ind[0] + dx;
    ind[2] = ind[1] + dz;
    ind[3] = ind[0] + dz;
    ind[4] = ind[0] + dy;
    ind[5] = ind[1] + dy;
    ind[6] = ind[2] + dy;
    ind[7] = ind[3] + dy;
    float co[8];
    co[0] = (1 - u) * (1 - v) * (1 - w);
    co[1] = u * (1 - v) * (1 - w);
    co[2] = u * (1 - v) * w;
    co[3] = (1 - u) * (1 - v) * w;
    co[4] = (1 - u) * v * (1 - w);
    co[5] = u * v * (1 - w);
    co[6] = (1 - u) * v * w;
    co[7] = u * v * w;
    float dxs = 0.0f, dys = 0.0f, dzs = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const long long i = min(ind[k], n_nodes - 1);
        float tx, ty, tz;
        if (nodes) {
            tx = nodes[i].translation[0]; ty = nodes[i].translation[1]; tz = nodes[i].translation[2];
        } else {
            const int nz = (int)(i / dz), ny = (int)((i - (long long)nz * dz) / dy), nx = (int)(i - (long long)nz * dz - (long long)ny * dy);
            tx = ((nx + 0.5f) * g.vs.x) + g.offset_clear.x;
            ty = ((ny + 0.5f) * g.vs.y) + g.offset_clear.y;
            tz = ((nz + 0.5f) * g.vs.z) + g.offset_clear.z;
        }
        dxs = (tx * co[k]) + dxs;
        dys = (ty * co[k]) + dys;
        dzs = (tz * co[k]) + dzs;
    }
    const float rx = (rot.m11 * dxs - rot.m12 * dys) + rot.m13 * dzs;   // rot holds the reference's nine products, signs as at :219-221
    const float ry = (rot.m21 * dxs + rot.m22 * dys) - rot.m23 * dzs;
    const float rz = (rot.m31 * dxs + rot.m32 * dys) + rot.m33 * dzs;
    points[idx * 3 + 0] = translation.x + rx;
    points[idx * 3 + 1] = translation.y + ry;
    points[idx * 3 + 2] = translation.z + rz;
}

}  // namespace tsdf

using namespace tsdf;

extern "C" {

const char *tsdf_last_error(void) { return g_err; }

const char *tsdf_build_arch(void) { return "gfx950"; }

int tsdf_device_count(int *count) {
    TSDF_REQUIRE(count, "tsdf_device_count: null argument");
    TSDF_HIP(hipGetDeviceCount(count), "hipGetDeviceCount");
    return TSDF_OK;
}

int tsdf_set_device(int device) {
    TSDF_HIP(hipSetDevice(device), "hipSetDevice");
    return TSDF_OK;
}

int tsdf_get_device(int *device) {
    TSDF_REQUIRE(device, "tsdf_get_device: null argument");
    TSDF_HIP(hipGetDevice(device), "hipGetDevice");
    return TSDF_OK;
}

// Device memory for callers that keep their frames and maps in HBM without touching the HIP headers (tools/kinfu_stream.cpp).
int tsdf_device_alloc(size_t bytes, void **device_ptr) {
    TSDF_REQUIRE(device_ptr && bytes > 0, "tsdf_device_alloc: bad argument");
    *device_ptr = nullptr;
    TSDF_HIP(hipMalloc(device_ptr, bytes), "device allocation");
    return TSDF_OK;
}

int tsdf_device_free(void *device_ptr) {
    if (device_ptr) TSDF_HIP(hipFree(device_ptr), "device free");
    return TSDF_OK;
}

int tsdf_device_upload(void *device_dst, const void *host_src, size_t bytes) {
    TSDF_REQUIRE(device_dst && host_src, "tsdf_device_upload: null argument");
    TSDF_HIP(hipMemcpy(device_dst, host_src, bytes, hipMemcpyHostToDevice), "upload");
    return TSDF_OK;
}

int tsdf_device_download(void *host_dst, const void *device_src, size_t bytes) {
    TSDF_REQUIRE(host_dst && device_src, "tsdf_device_download: null argument");
    TSDF_HIP(hipMemcpy(host_dst, device_src, bytes, hipMemcpyDeviceToHost), "download");
    return TSDF_OK;
}

int tsdf_stream_synchronize(void *hip_stream) {
    TSDF_HIP(hipStreamSynchronize((hipStream_t)hip_stream), "stream synchronize");
    return TSDF_OK;
}

int tsdf_volume_create_slab(uint32_t sx, uint32_t sy, uint32_t sz, float px, float py, float pz,
                            uint32_t z_begin, uint32_t z_end, tsdf_volume **out) {
    TSDF_REQUIRE(out, "tsdf_volume_create: null out pointer");
    *out = nullptr;
    // src/TSDF/TSDFVolume.cu:431-436: all sizes and physical sizes must be > 0
    TSDF_REQUIRE(sx > 0 && sy > 0 && sz > 0 && px > 0 && py > 0 && pz > 0,
                 "Attempt to construct TSDFVolume with zero or negative size");
    // the reference's set_size takes uint16_t dimensions (src/include/TSDFVolume.hpp:112)
    TSDF_REQUIRE(sx <= 65535 && sy <= 65535 && sz <= 65535, "TSDFVolume dimensions must fit in 16 bits");
    TSDF_REQUIRE(z_begin < z_end && z_end <= sz, "invalid slab [%u,%u) of %u planes", z_begin, z_end, sz);

    tsdf_volume *v = new (std::nothrow) tsdf_volume();
    if (!v) {
        set_error("out of host memory");
        return TSDF_ERR_NOMEM;
    }
    std::memset(v, 0, sizeof(*v));
    v->occ_scan_all = 1;
    Geom &g = v->g;
    g.X = sx; g.Y = sy; g.Z = sz;
    g.phys = {px, py, pz};
    // src/TSDF/TSDFVolume.cu:690 f3_div_elem(float3, dim3); :693 trunc = 1.1f * f3_norm(voxel_size)
    g.vs.x = px / (float)sx;
    g.vs.y = py / (float)sy;
    g.vs.z = pz / (float)sz;
    g.trunc = 1.1f * sqrtf(g.vs.x * g.vs.x + g.vs.y * g.vs.y + g.vs.z * g.vs.z);
    g.offset = {0.0f, 0.0f, 0.0f};
    g.offset_clear = {0.0f, 0.0f, 0.0f};
    v->z_begin = z_begin;
    v->z_end = z_end;
    g.z_store_begin = z_begin;
    g.z_store_end = (z_end < sz) ? z_end + 1 : sz;  // one halo plane above (trilinear reads lower.z+1)
    v->max_weight = 15.0f;                           // src/TSDF/TSDFVolume.cu:717
    v->stream = nullptr;

    v->occ.nbx = (sx + kBrick - 1) / kBrick;
    v->occ.nby = (sy + kBrick - 1) / kBrick;
    v->occ.nbz = (sz + kBrick - 1) / kBrick;
    set_occupancy_thresholds(v);
    if (v->occ.fine_count() >= ((size_t)1 << 32)) {   // the ray caster indexes bricks with 32 bits (grids beyond ~6500^3)
        delete v;
        set_error("tsdf_volume_create: grid of %u x %u x %u voxels is too large", (unsigned)sx, (unsigned)sy, (unsigned)sz);
        return TSDF_ERR_INVALID;
    }
    hipError_t e = hipGetDevice(&v->device);
    if (e == hipSuccess) e = hipMalloc((void **)&v->occ.fine, v->occ.fine_count());
    if (e == hipSuccess) e = hipMalloc((void **)&v->occ.cell, v->occ.fine_count());
    if (e == hipSuccess) e = hipMalloc((void **)&v->occ.reach, v->occ.fine_count());
    size_t bytes = v->resident_voxels() * sizeof(float);
    if (e == hipSuccess) e = hipMalloc((void **)&v->dist, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&v->counter_dev, 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(v->counter_dev, 0, 4 * sizeof(unsigned long long));
    if (e != hipSuccess) {
        int rc = hip_fail(e, "Couldn't allocate space for TSDF data");
        tsdf_volume_destroy(v);
        return rc;
    }
    int rc = weights_create(v);
    if (rc == TSDF_OK) rc = build_t_table(v);
    if (rc == TSDF_OK) rc = verify_fast_division(v);
    if (rc == TSDF_OK) rc = tsdf_volume_clear(v);
    if (rc == TSDF_OK) rc = tsdf_volume_synchronize(v);
    if (rc != TSDF_OK) {
        tsdf_volume_destroy(v);
        return rc;
    }
    *out = v;
    return TSDF_OK;
}

int tsdf_volume_create(uint32_t sx, uint32_t sy, uint32_t sz, float px, float py, float pz, tsdf_volume **out) {
    TSDF_REQUIRE(sz > 0, "Attempt to construct TSDFVolume with zero or negative size");
    return tsdf_volume_create_slab(sx, sy, sz, px, py, pz, 0, sz, out);
}

int tsdf_volume_destroy(tsdf_volume *v) {
    if (!v) return TSDF_OK;
    if (v->occ_tighten_pending) (void)hipEventSynchronize(v->occ_tightened);
    if (v->occ_tightened) (void)hipEventDestroy(v->occ_tightened);
    if (v->dist) (void)hipFree(v->dist);
    weights_destroy(v);
    if (v->nodes) (void)hipFree(v->nodes);
    if (v->depth_buf) (void)hipFree(v->depth_buf);
    if (v->vert_buf) (void)hipFree(v->vert_buf);
    if (v->norm_buf) (void)hipFree(v->norm_buf);
    if (v->counter_dev) (void)hipFree(v->counter_dev);
    if (v->occ.fine) (void)hipFree(v->occ.fine);
    if (v->occ.cell) (void)hipFree(v->occ.cell);
    if (v->occ.reach) (void)hipFree(v->occ.reach);
    if (v->occ_bits) (void)hipFree(v->occ_bits);
    if (v->occ_rim_bits) (void)hipFree(v->occ_rim_bits);
    if (v->touched) (void)hipFree(v->touched);
    if (v->plane_const) (void)hipFree(v->plane_const);
    if (v->depth_pad) (void)hipFree(v->depth_pad);
    if (v->tail_entries) (void)hipFree(v->tail_entries);
    if (v->tail_count) (void)hipFree(v->tail_count);
    if (v->cell_rays) (void)hipFree(v->cell_rays);
    if (v->cell_bricks) (void)hipFree(v->cell_bricks);
    if (v->cell_count_scratch) (void)hipFree(v->cell_count_scratch);
    for (int i = 0; i < 2; i++)
        if (v->chooser.ev[i]) (void)hipEventDestroy(v->chooser.ev[i]);
    if (v->cell_cast_host) (void)hipHostFree(v->cell_cast_host);
    if (v->ray_heavy) (void)hipFree(v->ray_heavy);
    if (v->ray_order) (void)hipFree(v->ray_order);
    if (v->ztile) (void)hipFree(v->ztile);
    if (v->t_table) (void)hipFree(v->t_table);
    if (v->ray_best) (void)hipFree(v->ray_best);
    for (int w = 0; w < 3; w++)
        if (v->tev[w]) {
            for (hipEvent_t e : *v->tev[w]) (void)hipEventDestroy(e);
            delete v->tev[w];
        }
    if (v->brick_list) (void)hipFree(v->brick_list);
    if (v->brick_boxes) (void)hipFree(v->brick_boxes);
    if (v->tile_max) (void)hipFree(v->tile_max);
    delete v;
    return TSDF_OK;
}

int tsdf_volume_set_stream(tsdf_volume *v, void *hip_stream) {
    TSDF_REQUIRE(v, "null volume");
    v->stream = (hipStream_t)hip_stream;
    return TSDF_OK;
}

int tsdf_volume_stream(const tsdf_volume *v, void **hip_stream) {
    TSDF_REQUIRE(v && hip_stream, "null argument");
    *hip_stream = v->stream;
    return TSDF_OK;
}

int tsdf_volume_synchronize(const tsdf_volume *v) {
    TSDF_REQUIRE(v, "null volume");
    TSDF_HIP(hipStreamSynchronize(v->stream), "stream synchronize");
    return TSDF_OK;
}

int tsdf_volume_clear(tsdf_volume *v) {
    TSDF_REQUIRE(v, "null volume");
    {
        int rcj = occupancy_join(v);
        if (rcj != TSDF_OK) return rcj;
    }
    size_t n = v->resident_voxels();
    {
        const int rcw = weights_clear(v);
        if (rcw != TSDF_OK) return rcw;
    }
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, v->stream, v->dist, n, v->g.trunc);
    TSDF_HIP(hipGetLastError(), "Couldn't clear TSDF data");
    // every distance is +trunc again: only the permanent boundary marks remain
    int rc0 = occupancy_reset(v);
    if (rc0 != TSDF_OK) return rc0;
    v->occ_scan_all = 1;
    v->occ_dirty = 0;
    v->occ_tighten_due = 0;
    if (v->cell_cast_host) *v->cell_cast_host = 0;   // (the cell-parallel cast's list of an empty volume; a cast in flight may still write the old count: it only costs a march)
    v->chooser.seen[0] = v->chooser.seen[1] = 0;   // (what was measured on the old contents says nothing about the new: choose_cast starts over)
    v->chooser.gap = 0;
    v->chooser.blocked = false;
    v->chooser.trial_left = 0;
    v->prepared_valid = 0;   // (a brick list prepared ahead bakes in the offset at clear time)
    v->integrations_since_rebuild = v->integrations_total = 0;
    // initialise_deformation bakes the CURRENT offset into the node translations (Q1)
    v->g.offset_clear = v->g.offset;
    if (v->nodes) return init_nodes(v);
    return TSDF_OK;
}

int tsdf_volume_get_info(const tsdf_volume *v, tsdf_volume_info *info) {
    TSDF_REQUIRE(v && info, "null argument");
    const Geom &g = v->g;
    info->size[0] = g.X; info->size[1] = g.Y; info->size[2] = g.Z;
    info->z_begin = v->z_begin; info->z_end = v->z_end;
    info->z_store_begin = g.z_store_begin; info->z_store_end = g.z_store_end;
    const F3 *src[5] = {&g.phys, &g.vs, &g.offset, &g.offset_clear, nullptr};
    float *dst[4] = {info->physical_size, info->voxel_size, info->offset, info->offset_at_clear};
    for (int i = 0; i < 4; i++) {
        dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z;
    }
    info->truncation_distance = g.trunc;
    info->max_weight = v->max_weight;
    for (int i = 0; i < 3; i++) {
        info->global_translation[i] = v->global_translation[i];
        info->global_rotation[i] = v->global_rotation[i];
    }
    info->deformation_materialised = v->nodes ? 1 : 0;
    info->fast_division_verified = v->fast_div;
    return TSDF_OK;
}

int tsdf_volume_set_offset(tsdf_volume *v, float ox, float oy, float oz) {
    TSDF_REQUIRE(v, "null volume");
    v->g.offset = {ox, oy, oz};
    v->prepared_valid = 0;   // (a brick list prepared ahead was culled with the old offset)
    return TSDF_OK;
}

int tsdf_volume_set_header(tsdf_volume *v, const float offset[3], float trunc, float max_weight,
                           const float gt[3], const float gr[3]) {
    TSDF_REQUIRE(v && offset && gt && gr, "null argument");
    v->g.offset = {offset[0], offset[1], offset[2]};
    v->g.trunc = trunc;
    v->prepared_valid = 0;
    set_occupancy_thresholds(v);
    v->occ_dirty = 1;
    v->occ_scan_all = 1;
    int rc = build_t_table(v);
    if (rc != TSDF_OK) return rc;
    v->max_weight = max_weight;
    for (int i = 0; i < 3; i++) {
        v->global_translation[i] = gt[i];
        v->global_rotation[i] = gr[i];
    }
    return TSDF_OK;
}

int tsdf_volume_occupancy(const tsdf_volume *v, uint64_t *occupied_bricks, uint64_t *total_bricks) {
    TSDF_REQUIRE(v && occupied_bricks && total_bricks, "null argument");
    {
        int rc = occupancy_join(const_cast<tsdf_volume *>(v));
        if (rc == TSDF_OK) rc = occupancy_refresh(const_cast<tsdf_volume *>(v));
        if (rc != TSDF_OK) return rc;
    }
    size_t n = v->occ.fine_count();
    uint8_t *h = new (std::nothrow) uint8_t[n];
    if (!h) {
        set_error("out of host memory");
        return TSDF_ERR_NOMEM;
    }
    hipError_t e = hipMemcpyAsync(h, v->occ.fine, n, hipMemcpyDeviceToHost, v->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(v->stream);
    uint64_t c = 0;
    for (size_t i = 0; i < n; i++) c += h[i] ? 1 : 0;
    delete[] h;
    if (e != hipSuccess) return hip_fail(e, "read occupancy");
    *occupied_bricks = c;
    *total_bricks = n;
    return TSDF_OK;
}

int tsdf_volume_get_occupancy_data(const tsdf_volume *cv, int force_rebuild, uint8_t *host_fine, uint8_t *host_cell,
                                   uint8_t *host_reach) {
    TSDF_REQUIRE(cv, "null volume");
    tsdf_volume *v = const_cast<tsdf_volume *>(cv);
    if (force_rebuild) v->occ_dirty = 1;
    int rc = occupancy_join(v);
    if (rc == TSDF_OK) rc = occupancy_refresh(v);
    if (rc != TSDF_OK) return rc;
    const size_t n = v->occ.fine_count();
    hipError_t e = hipSuccess;
    if (host_fine) e = hipMemcpyAsync(host_fine, v->occ.fine, n, hipMemcpyDeviceToHost, v->stream);
    if (e == hipSuccess && host_cell) e = hipMemcpyAsync(host_cell, v->occ.cell, n, hipMemcpyDeviceToHost, v->stream);
    if (e == hipSuccess && host_reach) e = hipMemcpyAsync(host_reach, v->occ.reach, n, hipMemcpyDeviceToHost, v->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(v->stream);
    if (e != hipSuccess) return hip_fail(e, "read occupancy");
    return TSDF_OK;
}

// rotate() of the reference (src/TSDF/TSDFVolume.cu:213-224): the nine float products of the cosines / sines of the three
// angles, as the kernel's per-point code forms them; the kernel applies the signs.
static void rotation_products(const float r[3], float m[9]) {
    const float c1 = cosf(r[0]), c2 = cosf(r[1]), c3 = cosf(r[2]);
    const float s1 = sinf(r[0]), s2 = sinf(r[1]), s3 = sinf(r[2]);
    m[0] = (c2 * c3);            m[1] = (c2 * s3);             m[2] = s2;
    m[3] = (c1 * s3 + s1 * s2 * c3); m[4] = (c1 * c3 - s1 * s2 * s3); m[5] = (s1 * c2);
    m[6] = (s1 * s3 - c1 * s2 * c3); m[7] = (s1 * c3 + c1 * s2 * s3); m[8] = (c1 * c2);
}

int tsdf_volume_deform_points_device(const tsdf_volume *v, int num_points, float *device_points) {
    TSDF_REQUIRE(v && device_points && num_points >= 0, "tsdf_volume_deform_points: bad argument");
    TSDF_REQUIRE(v->z_begin == 0 && v->z_end == v->g.Z, "tsdf_volume_deform_points needs a whole volume");
    if (num_points == 0) return TSDF_OK;
    float m[9];
    rotation_products(v->global_rotation, m);
    Mat33 rot;   // column-major members: mRC
    rot.m11 = m[0]; rot.m12 = m[1]; rot.m13 = m[2];
    rot.m21 = m[3]; rot.m22 = m[4]; rot.m23 = m[5];
    rot.m31 = m[6]; rot.m32 = m[7]; rot.m33 = m[8];
    const F3 t = {v->global_translation[0], v->global_translation[1], v->global_translation[2]};
    hipLaunchKernelGGL(deform_points_kernel, dim3((unsigned)((num_points + 255) / 256)), dim3(256), 0, v->stream, v->nodes, v->g, rot, t,
                       num_points, device_points);
    TSDF_HIP(hipGetLastError(), "Deformation kernel failed");
    return TSDF_OK;
}

int tsdf_volume_deform_points(const tsdf_volume *v, int num_points, float *host_points) {
    TSDF_REQUIRE(v && host_points && num_points >= 0, "tsdf_volume_deform_points: bad argument");
    if (num_points == 0) return TSDF_OK;
    float *d = nullptr;
    const size_t bytes = (size_t)num_points * 3 * sizeof(float);
    TSDF_HIP(hipMalloc((void **)&d, bytes), "d_points");
    hipError_t e = hipMemcpyAsync(d, host_points, bytes, hipMemcpyHostToDevice, v->stream);
    int rc = e == hipSuccess ? tsdf_volume_deform_points_device(v, num_points, d) : hip_fail(e, "Failed to copy points to device for deformation");
    if (rc == TSDF_OK) {
        e = hipMemcpyAsync(host_points, d, bytes, hipMemcpyDeviceToHost, v->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(v->stream);
        if (e != hipSuccess) rc = hip_fail(e, "Failed to copy points from device after deformation");
    }
    (void)hipFree(d);
    return rc;
}

int tsdf_volume_mark_dirty(tsdf_volume *v) {
    TSDF_REQUIRE(v, "null volume");
    // (a tightening of the flags still running beside the last ray cast reads the distances: what follows on the volume's stream --
    // the caller's next writes, ordered on it -- comes after that scan)
    const int rcj = occupancy_join(v);
    if (rcj != TSDF_OK) return rcj;
    v->occ_dirty = 1;
    v->occ_scan_all = 1;   // (written from outside: anywhere)
    return TSDF_OK;
}

int tsdf_volume_distances(const tsdf_volume *v, float **p) {
    TSDF_REQUIRE(v && p, "null argument");
    // The pointer may be written through.  A tightening of the ray caster's flags enqueued on another stream (tsdf_pipeline_step)
    // reads the distances: the volume's stream waits for it here, so that work the caller orders on that stream comes after the scan.
    const int rcj = occupancy_join(const_cast<tsdf_volume *>(v));
    if (rcj != TSDF_OK) return rcj;
    *p = v->dist;
    return TSDF_OK;
}

int tsdf_volume_weights(const tsdf_volume *v, float **p) {
    TSDF_REQUIRE(v && p, "null argument");
    const int rcj = occupancy_join(const_cast<tsdf_volume *>(v));
    if (rcj != TSDF_OK) return rcj;
    // the reference's weight_data(): a device pointer to fp32 weights.  The volume takes the reference's layout and keeps it
    // (weights.hip): the caller may hold the pointer and write through it.
    const int rcw = weights_require_f32(const_cast<tsdf_volume *>(v));
    if (rcw != TSDF_OK) return rcw;
    const_cast<tsdf_volume *>(v)->weight_pinned = 1;
    *p = v->weight;
    return TSDF_OK;
}

int tsdf_volume_deformation(tsdf_volume *v, tsdf_deformation_node **p) {
    TSDF_REQUIRE(v && p, "null argument");
    if (!v->nodes) {
        v->prepared_valid = 0;   // (custom nodes: every brick is walked)
        TSDF_HIP(hipMalloc((void **)&v->nodes, v->resident_voxels() * sizeof(tsdf_deformation_node)),
                 "Couldn't allocate space for deformation nodes for TSDF");
        // materialise what clear() would have written: centres + the offset at the time of clear()
        Geom saved = v->g;
        v->g.offset = v->g.offset_clear;
        int rc = init_nodes(v);
        v->g = saved;
        if (rc != TSDF_OK) return rc;
        TSDF_HIP(hipStreamSynchronize(v->stream), "initialise deformation nodes");
    }
    *p = v->nodes;
    return TSDF_OK;
}

static int copy_in(tsdf_volume *v, void *dst, const void *src, size_t bytes, const char *what) {
    TSDF_REQUIRE(v && src, "null argument");
    TSDF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, v->stream), what);
    TSDF_HIP(hipStreamSynchronize(v->stream), what);
    return TSDF_OK;
}

int tsdf_volume_set_distance_data(tsdf_volume *v, const float *host) {
    TSDF_REQUIRE(v, "null volume");
    {
        int rcj = occupancy_join(v);
        if (rcj != TSDF_OK) return rcj;
    }
    v->occ_dirty = 1;
    v->occ_scan_all = 1;
    return copy_in(v, v->dist, host, v->resident_voxels() * sizeof(float), "Couldn't set distance data");
}

int tsdf_volume_set_weight_data(tsdf_volume *v, const float *host) {
    TSDF_REQUIRE(v, "null volume");
    TSDF_REQUIRE(host, "null argument");
    v->prepared_valid = 0;   // (the storage of the weights may change, and with it what a brick list prepared ahead came with)
    return weights_upload(v, host);
}

int tsdf_volume_set_deformation(tsdf_volume *v, const tsdf_deformation_node *host) {
    TSDF_REQUIRE(v && host, "null argument");
    tsdf_deformation_node *p;
    int rc = tsdf_volume_deformation(v, &p);
    if (rc != TSDF_OK) return rc;
    return copy_in(v, p, host, v->resident_voxels() * sizeof(tsdf_deformation_node), "Couldn't set deformation");
}

static int copy_out(const tsdf_volume *v, void *dst, const void *src, size_t bytes, const char *what) {
    TSDF_REQUIRE(v && dst, "null argument");
    TSDF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, v->stream), what);
    TSDF_HIP(hipStreamSynchronize(v->stream), what);
    return TSDF_OK;
}

int tsdf_volume_get_distance_data(const tsdf_volume *v, float *host) {
    TSDF_REQUIRE(v, "null volume");
    return copy_out(v, host, v->dist, v->resident_voxels() * sizeof(float), "Couldn't read distance data");
}

int tsdf_volume_get_weight_data(const tsdf_volume *v, float *host) {
    TSDF_REQUIRE(v, "null volume");
    TSDF_REQUIRE(host, "null argument");
    return weights_download(v, host);
}

int tsdf_volume_get_deformation_planes(const tsdf_volume *v, uint32_t plane_begin, uint32_t plane_count,
                                       tsdf_deformation_node *host) {
    TSDF_REQUIRE(v && host, "null argument");
    const Geom &g = v->g;
    const uint32_t planes = g.z_store_end - g.z_store_begin;
    TSDF_REQUIRE(plane_begin <= planes && plane_count <= planes - plane_begin, "planes [%u, +%u) are not resident (%u planes)",
                 plane_begin, plane_count, planes);
    const size_t per_plane = (size_t)g.X * g.Y;
    if (v->nodes)
        return copy_out(v, host, v->nodes + per_plane * plane_begin, per_plane * plane_count * sizeof(tsdf_deformation_node),
                        "Couldn't read deformation data");
    // implicit grid: the expression of init_nodes_kernel / initialise_deformation with the offset of the last clear()
    for (uint32_t p = 0; p < plane_count; p++) {
        const uint32_t z = g.z_store_begin + plane_begin + p;
        tsdf_deformation_node *out = host + per_plane * p;
        for (uint32_t y = 0; y < g.Y; y++)
            for (uint32_t x = 0; x < g.X; x++, out++) {
                out->translation[0] = (((int)x + 0.5f) * g.vs.x) + g.offset_clear.x;
                out->translation[1] = (((int)y + 0.5f) * g.vs.y) + g.offset_clear.y;
                out->translation[2] = (((int)z + 0.5f) * g.vs.z) + g.offset_clear.z;
                out->rotation[0] = out->rotation[1] = out->rotation[2] = 0.0f;
            }
    }
    return TSDF_OK;
}

int tsdf_volume_set_offset_at_clear(tsdf_volume *v, const float oc[3]) {
    TSDF_REQUIRE(v && oc, "null argument");
    TSDF_REQUIRE(!v->nodes, "the deformation nodes are materialised: their translations are what they are");
    v->g.offset_clear = {oc[0], oc[1], oc[2]};
    v->prepared_valid = 0;
    return TSDF_OK;
}

int tsdf_measure_copy_bandwidth(size_t bytes, int reps, void *hip_stream, double *gb_per_s) {
    TSDF_REQUIRE(gb_per_s && bytes >= 16 && bytes % 16 == 0 && reps >= 1, "tsdf_measure_copy_bandwidth: bad argument");
    hipStream_t s = (hipStream_t)hip_stream;
    float4 *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc((void **)&a, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&b, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(a, 0x3c, bytes, s);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    double best = 0.0;
    const size_t n = bytes / 16;
    // several launch shapes, the best one counts (a ceiling is asked for): one element per thread, and grid-stride loops of
    // 8 ... 64 workgroups per compute unit
    const size_t shapes[5] = {(n + 255) / 256, (size_t)256 * 8, (size_t)256 * 16, (size_t)256 * 32, (size_t)256 * 64};
    for (int sh = 0; sh < 5 && e == hipSuccess; sh++) {
        const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, shapes[sh]);
        for (int r = 0; r <= reps && e == hipSuccess; r++) {          // (the first round warms up)
            e = hipEventRecord(e0, s);
            hipLaunchKernelGGL(copy4_kernel, dim3(grid), dim3(256), 0, s, a, b, n);
            if (e == hipSuccess) e = hipEventRecord(e1, s);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0.0f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e == hipSuccess && r > 0 && ms > 0.0f) best = std::max(best, 2.0 * (double)bytes / (ms * 1e-3) / 1e9);
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (e != hipSuccess) return hip_fail(e, "copy bandwidth measurement");
    *gb_per_s = best;
    return TSDF_OK;
}

int tsdf_measure_update_bandwidth(int reps, void *hip_stream, double *gb_per_s) {
    TSDF_REQUIRE(gb_per_s && reps >= 1, "tsdf_measure_update_bandwidth: bad argument");
    hipStream_t s = (hipStream_t)hip_stream;
    const size_t bytes = (size_t)512 * 512 * 512 * sizeof(float);
    float *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc((void **)&a, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&b, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(a, 0, bytes, s);
    if (e == hipSuccess) e = hipMemsetAsync(b, 0, bytes, s);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    double best = 0.0;
    for (int r = 0; r <= reps && e == hipSuccess; r++) {          // (the first round warms up)
        e = hipEventRecord(e0, s);
        hipLaunchKernelGGL(update_walk_kernel, dim3(8 * 128 * 16), dim3(64, 4), 0, s, a, b);
        if (e == hipSuccess) e = hipEventRecord(e1, s);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && r > 0 && ms > 0.0f) best = std::max(best, 4.0 * (double)bytes / (ms * 1e-3) / 1e9);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (e != hipSuccess) return hip_fail(e, "update bandwidth measurement");
    *gb_per_s = best;
    return TSDF_OK;
}

int tsdf_volume_set_timing(tsdf_volume *v, int enabled) {
    TSDF_REQUIRE(v, "null volume");
    v->timing = enabled > 0 ? enabled : 0;
    for (int w = 0; w < 3; w++) v->timing_launches[w] = 0;
    for (int w = 0; w < 3; w++)
        if (v->tev[w]) {
            for (hipEvent_t e : *v->tev[w]) (void)hipEventDestroy(e);
            v->tev[w]->clear();
        }
    return TSDF_OK;
}

int tsdf_volume_kernel_time(tsdf_volume *v, int which, uint32_t *launches, float *average_ms) {
    TSDF_REQUIRE(v && launches && average_ms && which >= 0 && which <= 2, "bad argument");
    *launches = 0;
    *average_ms = 0.0f;
    if (!v->tev[which] || v->tev[which]->size() < 2) return TSDF_OK;
    TSDF_HIP(hipStreamSynchronize(v->stream), "kernel timing");
    double total = 0.0;
    uint32_t n = 0;
    for (size_t i = 0; i + 1 < v->tev[which]->size(); i += 2) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, (*v->tev[which])[i], (*v->tev[which])[i + 1]) == hipSuccess) {
            total += ms;
            n++;
        }
    }
    *launches = n;
    *average_ms = n ? (float)(total / n) : 0.0f;
    return TSDF_OK;
}

int tsdf_volume_set_counting(tsdf_volume *v, int enabled) {
    TSDF_REQUIRE(v, "null volume");
    v->counting = enabled ? 1 : 0;
    return TSDF_OK;
}

int tsdf_volume_last_updated_voxels(const tsdf_volume *v, uint64_t *count) {
    TSDF_REQUIRE(v && count, "null argument");
    unsigned long long c = 0;
    TSDF_HIP(hipMemcpyAsync(&c, v->counter_dev, sizeof(c), hipMemcpyDeviceToHost, v->stream), "read counter");
    TSDF_HIP(hipStreamSynchronize(v->stream), "read counter");
    *count = c;
    return TSDF_OK;
}

}  // extern "C"
