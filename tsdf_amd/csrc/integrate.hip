// Depth-map integration for gfx950 (wave64).  Replaces TSDFVolume::integrate and
// integrate_kernel of the reference (src/TSDF/TSDFVolume.cu:861-902, 308-392).
//
// Mapping (the reference runs one thread per (y,z) with a serial x loop, so neighbouring
// lanes are 4*X bytes apart): here lane <-> x, so one wave touches 64 consecutive voxels
// = 256 contiguous bytes of the distance array and of the weight array; a 256-thread
// workgroup owns a brick of 64(x) x 4(y) x 16(z) voxels and walks it plane by plane.
//
// Three launches per frame:
//   1. depth_tile_max_kernel: max depth of every 16x16 pixel tile (1200 tiles at 640x480).
//   2. brick_cull_kernel (eight lanes per brick, one per corner): exact, conservative culling.  The 8 corner voxel centres are
//      projected with running error bounds.  A projective map sends the convex brick into the convex hull of
//      the projected corners as long as the homogeneous divisor keeps one sign over the brick, so
//        (a) if all corners fall off the same side of the depth image no voxel passes the reference's
//            frustum test (bricks that straddle the camera plane are never culled: the reference projects
//            voxels behind the camera too, Q2);
//        (b) if every depth tile the hull can touch is all-invalid, no voxel finds a depth > 0;
//        (c) for rigid poses / standard intrinsics (surface z == depth, w == 1 exactly): if the nearest corner
//            lies more than trunc behind the largest depth of those tiles, every sdf is < -trunc.
//      Surviving bricks are appended to a compact list.
//   3. integrate_kernel: one workgroup per listed brick.  Per voxel, the reference's arithmetic in its
//      operation order (fp contraction is off): world_to_pixel -> depth gather -> pixel_to_camera.z ->
//      world_to_camera.z -> sdf -> running weighted mean.  Distance and weight are loaded only under the
//      update predicate and stored with the same mask, so the algorithmic traffic is 16 B per updated voxel +
//      the depth pixels gathered (L2 resident: 614 KB).  Camera matrices, intrinsics and grid geometry are
//      kernel arguments: wave-uniform, they live in SGPRs.
#include <algorithm>
#include <cmath>

#include <cstdlib>

#include "common.hpp"
#include "integrate_grid.hpp"

namespace tsdf {


struct Projected {
    float ix, iy, iz;  // K * cam
    float cam_z;       // row 3 of inv_pose applied to the point
    float ecz;         // absolute error bound of cam_z
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
    r.ecz = ecz;
    r.ix = k.m11 * cx + k.m12 * cy + k.m13 * cz;
    r.iy = k.m21 * cx + k.m22 * cy + k.m23 * cz;
    r.iz = k.m31 * cx + k.m32 * cy + k.m33 * cz;
    r.ex = u * (fabsf(k.m11 * cx) + fabsf(k.m12 * cy) + fabsf(k.m13 * cz)) + fabsf(k.m11) * ecx + fabsf(k.m12) * ecy + fabsf(k.m13) * ecz;
    r.ey = u * (fabsf(k.m21 * cx) + fabsf(k.m22 * cy) + fabsf(k.m23 * cz)) + fabsf(k.m21) * ecx + fabsf(k.m22) * ecy + fabsf(k.m23) * ecz;
    r.ez = u * (fabsf(k.m31 * cx) + fabsf(k.m32 * cy) + fabsf(k.m33 * cz)) + fabsf(k.m31) * ecx + fabsf(k.m32) * ecy + fabsf(k.m33) * ecz;
    return r;
}

// A voxel whose new distance is not safely positive flags every brick whose grown region
// (brick +- kBrickGrow voxels) contains it: the bricks holding voxel v-2 .. v+2 on each axis.
// (The definition, voxel by voxel: what integrate_kernel did up to round 2.  It now makes the same marks once per brick,
// mark_low_voxels below; this function is kept as the statement of which bricks a low voxel marks.  "Low" is "not safely positive";
// for the voxels that bricks at the grid boundary depend on it is "not flat", the stricter test of those bricks -- OccGrid.)
__device__ inline void mark_occupied(const OccGrid &occ, uint32_t vx, uint32_t vy, uint32_t vz) {
    const uint32_t bx0 = (max(vx, (uint32_t)kBrickGrow) - kBrickGrow) >> kBrickShift;
    const uint32_t by0 = (max(vy, (uint32_t)kBrickGrow) - kBrickGrow) >> kBrickShift;
    const uint32_t bz0 = (max(vz, (uint32_t)kBrickGrow) - kBrickGrow) >> kBrickShift;
    const uint32_t bx1 = min((vx + kBrickGrow) >> kBrickShift, occ.nbx - 1);
    const uint32_t by1 = min((vy + kBrickGrow) >> kBrickShift, occ.nby - 1);
    const uint32_t bz1 = min((vz + kBrickGrow) >> kBrickShift, occ.nbz - 1);
    for (uint32_t bz = bz0; bz <= bz1; bz++)
        for (uint32_t by = by0; by <= by1; by++)
            for (uint32_t bx = bx0; bx <= bx1; bx++) occ.fine[((size_t)bz * occ.nby + by) * occ.nbx + bx] = 1;
    // cell bricks whose voxel range [4b, 4b+4] contains the voxel: b = v/4, and b-1 when v is a multiple of 4
    const uint32_t cx1 = vx >> kBrickShift, cy1 = vy >> kBrickShift, cz1 = vz >> kBrickShift;
    const uint32_t cx0 = ((vx & (kBrick - 1)) == 0 && cx1 > 0) ? cx1 - 1 : cx1;
    const uint32_t cy0 = ((vy & (kBrick - 1)) == 0 && cy1 > 0) ? cy1 - 1 : cy1;
    const uint32_t cz0 = ((vz & (kBrick - 1)) == 0 && cz1 > 0) ? cz1 - 1 : cz1;
    for (uint32_t bz = cz0; bz <= cz1; bz++)
        for (uint32_t by = cy0; by <= cy1; by++)
            for (uint32_t bx = cx0; bx <= cx1; bx++) occ.cell[((size_t)bz * occ.nby + by) * occ.nbx + bx] = 1;
}

// integrate_packed.hip
int launch_integrate_packed_kernel(tsdf_volume *v, dim3 grid, const BrickGrid &bg, const Mat44 &ip, const Mat33 &mk, uint32_t width,
                                   uint32_t height, const uint16_t *d_depth, unsigned long long *counter_arg, const uint4 *boxes,
                                   const uint2 *coords, const uint32_t *count, const float4 *plane_const);

constexpr int kDepthTile = TSDF_DEPTH_TILE;  // pixels per side of a depth tile (16)
constexpr int kCullTilesLds = 4096;  // tile maxima brick_cull_kernel keeps in LDS (1200 at 640x480)

static uint32_t occupancy_rebuild_period() { return (uint32_t)tuning().occ_rebuild_period; }

// Max depth per 16x16 pixel tile (0 = the tile holds no valid depth).  One wave per tile.
__global__ __launch_bounds__(64) void depth_tile_max_kernel(const uint16_t *__restrict__ depth, uint32_t width,
                                                            uint32_t height, uint32_t tiles_x,
                                                            uint16_t *__restrict__ tile_max) {
    const uint32_t tx = blockIdx.x, ty = blockIdx.y;
    uint32_t m = 0;
    for (uint32_t i = threadIdx.x; i < kDepthTile * kDepthTile; i += 64) {
        uint32_t x = tx * kDepthTile + (i & (kDepthTile - 1)), y = ty * kDepthTile + (i / kDepthTile);
        if (x < width && y < height) m = max(m, (uint32_t)depth[(size_t)y * width + x]);
    }
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_down(m, o));
    if (threadIdx.x == 0) tile_max[ty * tiles_x + tx] = (uint16_t)m;
}

// Eight lanes per 64x4x32 brick: decide whether any voxel of it can be updated by this frame (see the header).
__global__ __launch_bounds__(256) void brick_cull_kernel(const Geom g, const BrickGrid bg, const Mat44 ip, const Mat33 k,
                                                         const uint32_t width, const uint32_t height,
                                                         const uint16_t *__restrict__ tile_max, const uint32_t tiles_x,
                                                         const int depth_test, uint32_t *__restrict__ list,
                                                         uint4 *__restrict__ boxes, uint32_t *__restrict__ count,
                                                         uint32_t *__restrict__ count_next,
                                                         float4 *__restrict__ plane_const, const uint32_t n_plane_const,
                                                         const float cone_mx, const float cone_my,
                                                         const uint16_t *__restrict__ depth, uint16_t *__restrict__ depth_pad,
                                                         uint2 *__restrict__ coords) {
    // One lane per corner: 8 consecutive lanes share a brick and combine their corners with 3 butterfly steps (a thread per brick
    // walked its 8 corners one after the other on a quarter of the chip's compute units: 12 us of dependent arithmetic).
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    // The list's length lives in two words used alternately: this launch appends behind *count and zeroes the word of the NEXT
    // integration (nothing reads that one before the next brick_cull_kernel, which comes after this frame's integrate_kernel in
    // stream order).  Up to round 2h depth_tile_max_kernel reset the one word; that launch is skipped when the caller brings the
    // tile maxima (tsdf_integrate_device_tiles).
    if (t == 0) *count_next = 0;
    // side job: the per-plane constants of integrate_kernel (see there)
    for (uint32_t p = t; p < n_plane_const; p += gridDim.x * 256) {
        const uint32_t vz = g.z_store_begin + p;
        const float cz = ((((int)vz + 0.5f) * g.vs.z) + g.offset_clear.z) + g.offset.z;
        plane_const[p] = make_float4(cz, ip.m13 * cz, ip.m23 * cz, ip.m33 * cz);
    }
    // side job (volumes with packed weights): the image inside a ring of zeros, for integrate_packed_kernel's bricks without a tile
    // (the ring was zeroed when the buffer was allocated)
    if (depth_pad)
        for (uint32_t p = t; p < width * height; p += gridDim.x * 256) {
            const uint32_t y = p / width, x = p - y * width;
            depth_pad[(size_t)(y + 1u) * (width + 2u) + (x + 1u)] = depth[p];
        }
    // the tile maxima in LDS (when they fit): the depth test below reads up to 256 of them per brick
    __shared__ uint16_t tmax_lds[kCullTilesLds];
    __shared__ uint32_t wave_count[4], wg_base;   // survivors per wave of this workgroup, the workgroup's first slot in the list
    const uint32_t n_tiles = tiles_x * ((height + kDepthTile - 1) / kDepthTile);
    const bool tiles_in_lds = depth_test && n_tiles <= (uint32_t)kCullTilesLds;
    if (tiles_in_lds)
        for (uint32_t i = threadIdx.x; i < n_tiles; i += 256) tmax_lds[i] = tile_max[i];
    __syncthreads();
    // Groups of 8 lanes take the bricks column by column: all rows (y, then z) of x-brick 0, then of x-brick 1, ...  The list
    // keeps roughly that order, and workgroup i of integrate_kernel (on XCD i % 8) takes entry i: bricks in flight together are
    // then rows apart, never neighbours along x.  Neighbours along x share every 2 KiB row of the volume they touch, and with it
    // the memory channel: listed next to each other (index order) they cost integrate_kernel 0.130 ms against 0.121 ms in this
    // order and 0.133 ms in a scattered one (same box, host-sorted lists, round 2).
    const uint32_t g8 = t >> 3, c = t & 7u;
    const uint32_t n_bricks = bg.nx * bg.ny * bg.nz, n_rows = bg.ny * bg.nz;
    const bool live = g8 < n_bricks;   // (whole groups of 8 lanes; dead groups compute on brick 0 and append nothing)
    const uint32_t bx = live ? g8 / n_rows : 0u, row = live ? g8 % n_rows : 0u;
    const uint32_t by = row % bg.ny, bz = row / bg.ny;
    const uint32_t b = bx + bg.nx * row;
    const uint32_t x0 = bx * kTileX, x1 = min(x0 + kTileX, g.X) - 1;
    const uint32_t y0 = by * kTileY, y1 = min(y0 + kTileY, g.Y) - 1;
    const uint32_t z0 = g.z_store_begin + bz * kChunkZ, z1 = min(z0 + kChunkZ + (bz + 1 == bg.nz ? bg.z_extra : 0u), g.z_store_end) - 1;

    const uint32_t vx = (c & 1) ? x1 : x0, vy = (c & 2) ? y1 : y0, vz = (c & 4) ? z1 : z0;
    const float px = ((((int)vx + 0.5f) * g.vs.x) + g.offset_clear.x) + g.offset.x;
    const float py = ((((int)vy + 0.5f) * g.vs.y) + g.offset_clear.y) + g.offset.y;
    const float pz = ((((int)vz + 0.5f) * g.vs.z) + g.offset_clear.z) + g.offset.z;
    // voxel centres inside the brick deviate from the exact lattice spanned by the corners by a few ulps of
    // the coordinate; that is folded into the error bounds
    const Projected p = project_with_bounds(px, py, pz, ip, k);
    const float aiz = fabsf(p.iz);
    const bool sign_ok = aiz > 8.0f * p.ez + 1.0e-3f;  // divisor reliably away from zero
    // (bounds only: the hardware reciprocal instead of four IEEE divisions per corner; its error, a few 1e-7 relative,
    // goes into the margins)
    const float riz = __builtin_amdgcn_rcpf(p.iz), raiz = fabsf(riz);
    const float qx = p.ix * riz, qy = p.iy * riz;
    // error of the quotient (first order, doubled)
    const float mqx = 2.0f * (p.ex + fabsf(qx) * p.ez) * raiz + 1.0e-3f + 1.0e-6f * fabsf(qx);
    const float mqy = 2.0f * (p.ey + fabsf(qy) * p.ez) * raiz + 1.0e-3f + 1.0e-6f * fabsf(qy);
    // a voxel passes the frustum test iff round(q) in [0, W-1]  <=>  q in [-0.5, W-0.5)
    // bits that must hold for ALL corners: 0 divisor positive, 1 divisor negative, 2 left of the image, 3 right, 4 above, 5 below
    uint32_t all = ((sign_ok && p.iz > 0.0f) ? 1u : 0u) | ((sign_ok && p.iz < 0.0f) ? 2u : 0u) | ((qx + mqx < -1.0f) ? 4u : 0u) |
                   ((qx - mqx > (float)width) ? 8u : 0u) | ((qy + mqy < -1.0f) ? 16u : 0u) | ((qy - mqy > (float)height) ? 32u : 0u);
    // Bricks that straddle the camera plane (the camera is inside the volume: a whole layer of them) have no hull to bound, but the
    // reference's frustum test (:349) still is a statement about signs.  With a = image.x, b = image.z, c = image.y as the reference
    // computes them, a pixel column in [0, W-1] needs -1 < a / b < W (the rounded IEEE quotient cannot differ from a / b by 1/2), i.e.
    //   b > 0:  L1 = a + b > 0  and  L2 = W b - a > 0        b < 0:  L1 < 0  and  L2 < 0        b == 0:  a == 0 (NaN -> pixel 0, Q3)
    // and the same for rows with L3 = c + b, L4 = H b - c: a voxel is in the image only inside the double cone
    // {all L > 0} u {all L < 0} u {all L == 0}.  Each L is an affine function of the position, so over the brick it stays under the
    // largest corner value: if some L_i is < 0 at all 8 corners (beyond the margin: cone_mx / cone_my, the host's bound on what the
    // fp32 evaluation of a, b, c can be off by anywhere in the grid, both at the corner and at the voxel) no voxel of the brick has all
    // L > 0 or all L == 0, and if some L_j is > 0 at all corners none has all L < 0: the brick updates nothing.
    {
        const float fw = (float)width, fh = (float)height;
        const float l1 = p.ix + p.iz, l2 = fw * p.iz - p.ix, l3 = p.iy + p.iz, l4 = fh * p.iz - p.iy;   // (NaN: no bit)
        all |= ((l1 < -cone_mx) ? 64u : 0u) | ((l2 < -cone_mx) ? 128u : 0u) | ((l3 < -cone_my) ? 256u : 0u) | ((l4 < -cone_my) ? 512u : 0u) |
               ((l1 > cone_mx) ? 1024u : 0u) | ((l2 > cone_mx) ? 2048u : 0u) | ((l3 > cone_my) ? 4096u : 0u) | ((l4 > cone_my) ? 8192u : 0u);
    }
    float qx_lo = qx - mqx, qx_hi = qx + mqx, qy_lo = qy - mqy, qy_hi = qy + mqy;
    float camz_lo = p.cam_z - p.ecz, ecz_max = p.ecz;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {   // (fminf / fmaxf drop a NaN, as the serial loop's did)
        all &= (uint32_t)__shfl_xor((int)all, o);
        qx_lo = fminf(qx_lo, __shfl_xor(qx_lo, o)); qx_hi = fmaxf(qx_hi, __shfl_xor(qx_hi, o));
        qy_lo = fminf(qy_lo, __shfl_xor(qy_lo, o)); qy_hi = fmaxf(qy_hi, __shfl_xor(qy_hi, o));
        camz_lo = fminf(camz_lo, __shfl_xor(camz_lo, o));
        ecz_max = fmaxf(ecz_max, __shfl_xor(ecz_max, o));
    }
    bool keep = true;
    // pixel box that contains the pixel of every voxel of the brick that passes the frustum test; 0 x 0 = unknown
    uint4 box = make_uint4(0, 0, 0, 0);  // x0, y0, width, height
    if (all & 3u) {   // the divisor keeps one sign over the brick
        if (all & 60u) keep = false;
        // pixels any voxel of the brick can round to: the hull's bounding box grown by 1 px
        const float fx0 = fmaxf(qx_lo - 1.0f, 0.0f), fx1 = fminf(qx_hi + 1.0f, (float)(width - 1));
        const float fy0 = fmaxf(qy_lo - 1.0f, 0.0f), fy1 = fminf(qy_hi + 1.0f, (float)(height - 1));
        const bool boxed = keep && fx0 <= fx1 && fy0 <= fy1;  // (false for NaN: box stays unknown)
        if (boxed) {
            box = make_uint4((uint32_t)fx0, (uint32_t)fy0, (uint32_t)fx1 - (uint32_t)fx0 + 1, (uint32_t)fy1 - (uint32_t)fy0 + 1);
            if (bg.pair_loads) {   // grown to even columns (any superset inside the image will do; the width is even)
                const uint32_t xa = box.x & ~1u, xe = (box.x + box.z + 1u) & ~1u;
                box.x = xa;
                box.z = xe - xa;
            }
        }
        // (every lane of the group holds the same box; the group's 8 lanes share the tiles)
        const uint32_t tx0 = boxed ? (uint32_t)fx0 / kDepthTile : 0u, tx1 = boxed ? (uint32_t)fx1 / kDepthTile : 0u;
        const uint32_t ty0 = boxed ? (uint32_t)fy0 / kDepthTile : 0u, ty1 = boxed ? (uint32_t)fy1 / kDepthTile : 0u;
        const uint32_t tw = tx1 - tx0 + 1, n_box_tiles = tw * (ty1 - ty0 + 1);
        const bool tested = boxed && depth_test && n_box_tiles <= 256u;
        uint32_t dmax = 0;
        if (tested)
            for (uint32_t i = c; i < n_box_tiles; i += 8) {
                const uint32_t ty = ty0 + i / tw, tx = tx0 + i % tw;
                dmax = max(dmax, (uint32_t)(tiles_in_lds ? tmax_lds[ty * tiles_x + tx] : tile_max[ty * tiles_x + tx]));
            }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, o));
        if (tested) {
            // (b) nothing but invalid depth in reach; (c) the whole brick lies more than trunc behind the
            // farthest surface in reach: sdf = depth - cam_z < -trunc for every voxel
            if (dmax == 0) keep = false;
            // (a voxel's own cam_z carries the same kind of rounding error as a corner's: 2 * ecz_max more)
            else if (camz_lo - (float)dmax > g.trunc * 1.0001f + 1.0e-3f + 1.0e-5f * fabsf(camz_lo) + 2.0f * ecz_max)
                keep = false;
        }
    }
    else if ((all & 960u) && (all & 15360u)) keep = false;   // straddles the camera plane, but lies outside both halves of the cone
    // One atomic on the list's length per workgroup (a few thousand of them on one address, from 8 XCDs, were most of this
    // kernel's time), the workgroup's survivors in brick order behind it: neighbouring bricks stay neighbours in the list, so
    // that they are in flight together in integrate_kernel (rows of the volume they share stay open in memory).
    const bool append = keep && live && c == 0;
    const uint64_t mask = __ballot(append);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (lane == 0) wave_count[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t n = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        wg_base = n ? atomicAdd(count, n) : 0u;
    }
    __syncthreads();
    if (append) {
        uint32_t slot = wg_base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < wave; w++) slot += wave_count[w];
        list[slot] = b;
        boxes[slot] = box;
        coords[slot] = make_uint2(bx | (by << 16), bz);   // (bx < 2^10, by < 2^14: the grid's sides fit 16 bits)
    }
}

// STD: the camera has the standard shape -- K = [fx 0 cx; 0 fy cy; 0 0 1], K^-1 with last row (0,0,1), inverse
// pose with last row (0,0,0,1), all entries finite (checked on the host).  Then, for the finite voxel centres of
// a grid, the reference's terms `0 * x` are +-0 and drop out of its sums without changing a bit that matters:
// image.z == cam.z, surface z == depth, w == 1 (adding -0 is the identity, adding +0 only turns a -0 into +0, and
// a zero's sign reaches neither the rounded pixel nor the sdf).  The kernel then skips those multiplications and
// the divisions by 1.
// rx = roundf(a1 / b), ry = roundf(a2 / b) exactly as the reference's IEEE divisions + round() give them
// (src/Utilities/cuda_coordinate_transforms.cu:25-26), at a fraction of the cost.  The quotients are first formed with
// the hardware reciprocal (relative error < 2.4e-7 against the correctly rounded quotient) and rounded to the nearest
// integer r (v_rndne; round 1 used floor(q + 1/2), the same number wherever the fast path is kept);
// h = 1/2 - |q - r| is the distance of q to the nearest rounding boundary (x.5).
//   * |q| <= L = max(width, height) + 2 and h > thr = 4e-7 * L: the IEEE quotient lies on the same side of the same
//     boundaries, and away from a boundary floor(q + 1/2) == roundf(q) for either sign (the addition q + 1/2 is exact
//     or errs by less than thr), so the result is the reference's;
//   * |q| > L: the IEEE quotient is beyond the image as well (> max(width, height) + 1 in magnitude), whatever the
//     two round to fails the reference's frustum test alike;
//   * otherwise (also NaN / infinite quotients, zero or denormal divisors: h is NaN or <= thr) the lane redoes the IEEE
//     division and roundf, and maps NaN to 0 as the target's float -> int conversion does.
__device__ inline void round_quotients(float a1, float a2, float b, float near_half, float &rx, float &ry) {
    const float rc = __builtin_amdgcn_rcpf(b);
    const float q1 = a1 * rc, q2 = a2 * rc;
    // nearest integer (ties to even): equals floor(q + 1/2) wherever the test below lets the fast path stand -- the two differ
    // only on ties and where q + 1/2 itself rounds across an integer, and both leave |q - r| within thr of 1/2
    rx = __builtin_rintf(q1);
    ry = __builtin_rintf(q2);
    // h > thr  <=>  |q - r| < 1/2 - thr; near_half is the float just BELOW fl(1/2 - thr) (host), so the comparison can only
    // send more lanes to the exact path than the bound in the comment above asks for.  NaN compares false -> exact path.
    if (!(fabsf(q1 - rx) < near_half) || !(fabsf(q2 - ry) < near_half)) {
        rx = roundf(a1 / b);
        ry = roundf(a2 / b);
        if (rx != rx) rx = 0.0f;
        if (ry != ry) ry = 0.0f;
    }
}

// Per z plane, the terms of the projection that depend on z only (the same fp32 products the reference forms per voxel):
// {cz, inv_pose.m13 * cz, inv_pose.m23 * cz, inv_pose.m33 * cz}, cz = the voxel-centre z of the plane (:343, :783-785).
// Written by brick_cull_kernel once per frame; wave-uniform in integrate_kernel, so read with scalar loads.
template <bool DEFORM, bool COUNT, bool STD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void integrate_kernel(float *__restrict__ dist, float *__restrict__ weight,
                                                        const tsdf_deformation_node *__restrict__ nodes,
                                                        const Geom g, const BrickGrid bg, const Mat44 ip, const Mat33 k,
                                                        const Mat33 kinv, const uint32_t width,
                                                        const uint32_t height,
                                                        const uint16_t *__restrict__ depth,
                                                        unsigned long long *__restrict__ counter,
                                                        const OccGrid occ, const uint32_t *__restrict__ list,
                                                        const uint4 *__restrict__ boxes,
                                                        const uint32_t *__restrict__ count,
                                                        const float4 *__restrict__ plane_const,
                                                        uint8_t *__restrict__ touched) {
    // Depth tile of the current brick: the pixel box the cull kernel derived for it, staged once per brick with
    // coalesced row loads; the per-voxel depth look-ups then read LDS instead of gathering from L2.
    __shared__ uint16_t tile[kTilePixels + 2];  // [kTilePixels] stays 0: where look-ups that miss the box are pointed
    __shared__ float4 plane_lds[kChunkZ + kBatchZ];  // this brick's rows of plane_const
    const uint32_t tid = threadIdx.y * kTileX + threadIdx.x;
    const uint32_t n_active = DEFORM ? bg.nx * bg.ny * bg.nz : *count;  // custom nodes: every brick
    const size_t plane = (size_t)g.X * g.Y;
    const float neg_trunc = -g.trunc;
    const float fwidth = (float)width, fheight = (float)height;
    // see round_quotients: thr = 4e-7 * (max(width, height) + 2); the float just below 1/2 - thr
    const float round_near_half = __uint_as_float(__float_as_uint(0.5f - 4.0e-7f * ((float)max(width, height) + 2.0f)) - 1u);   // (positive: one ulp down)
    uint32_t updated = 0;

    for (uint32_t i = blockIdx.x; i < n_active; i += gridDim.x) {
        const unsigned long long dbg_t0 = (!COUNT && counter) ? wall_clock64() : 0ull;   // (diagnostics, TSDF_DEBUG_BRICKS=3: per-brick clocks)
        const uint32_t b = DEFORM ? i : list[i];
        const uint32_t bx = b % bg.nx, by = (b / bg.nx) % bg.ny, bz = b / (bg.nx * bg.ny);
        if (tid == 0) touched[b] = 1;   // for the next occupancy rebuild: this brick's distances may change (volume.hip)
        const uint32_t vx = bx * kTileX + threadIdx.x;
        const uint32_t vy = by * kTileY + threadIdx.y;
        const uint32_t z0 = g.z_store_begin + bz * kChunkZ;
        const uint32_t z_extra = bz + 1 == bg.nz ? bg.z_extra : 0u;
        const uint32_t z1 = min(z0 + kChunkZ + z_extra, g.z_store_end);  // exclusive
        // stage the brick's pixel box (whole workgroup; falls back to global gathers when it is unknown or too big)
        uint4 box = make_uint4(0, 0, 0, 0);
        if (!DEFORM) box = boxes[i];
        const uint32_t pitch = (box.z + 1u) & ~1u;  // even, so a row starts on a 4-byte boundary
        const bool staged = box.z != 0 && pitch * box.w <= (uint32_t)kTilePixels;
        __syncthreads();  // the previous brick's look-ups are done
        if (!DEFORM && tid < (uint32_t)(kChunkZ + kBatchZ)) {
            const uint32_t p = z0 - g.z_store_begin + tid;   // (plane_const is padded by kBatchZ rows)
            if (p < g.z_store_end - g.z_store_begin + kBatchZ) plane_lds[tid] = plane_const[p];
        }
        if (tid == 0) tile[kTilePixels] = 0;
        if (staged) {
            // kStageBatch look-ups are requested before the first is waited for (a loop of single look-ups is one memory round
            // trip after the other: ~16 of them per brick).  No branch inside a batch: slots past the end re-read the last pixel
            // and are not written.
            constexpr uint32_t kStageBatch = 8;
            const uint32_t total = pitch * box.w;
            if (bg.pair_loads) {
                // (even image width, 4-byte aligned image: the cull kernel has made box.x and box.z even, a lane takes two pixels)
                const uint32_t half = pitch >> 1, total2 = half * box.w;
                const uint32_t *depth2 = reinterpret_cast<const uint32_t *>(depth);
                uint32_t *tile2 = reinterpret_cast<uint32_t *>(tile);
                for (uint32_t p0 = tid; p0 < total2; p0 += kTileX * kTileY * kStageBatch) {
                    uint32_t px[kStageBatch];
#pragma unroll
                    for (uint32_t u = 0; u < kStageBatch; u++) {
                        const uint32_t p = min(p0 + u * (kTileX * kTileY), total2 - 1u);
                        const uint32_t ty = p / half, tx2 = p - ty * half;
                        px[u] = depth2[(((size_t)(box.y + ty) * width + box.x) >> 1) + tx2];
                    }
#pragma unroll
                    for (uint32_t u = 0; u < kStageBatch; u++) {
                        const uint32_t p = p0 + u * (kTileX * kTileY);
                        if (p < total2) tile2[p] = px[u];
                    }
                }
            } else
            for (uint32_t p0 = tid; p0 < total; p0 += kTileX * kTileY * kStageBatch) {
                uint16_t px[kStageBatch];
#pragma unroll
                for (uint32_t u = 0; u < kStageBatch; u++) {
                    const uint32_t p = min(p0 + u * (kTileX * kTileY), total - 1u);
                    const uint32_t ty = p / pitch, tx = min(p - ty * pitch, box.z - 1u);
                    px[u] = depth[(size_t)(box.y + ty) * width + (box.x + tx)];
                }
#pragma unroll
                for (uint32_t u = 0; u < kStageBatch; u++) {
                    const uint32_t p = p0 + u * (kTileX * kTileY);
                    const uint32_t ty = p / pitch, tx = p - ty * pitch;
                    if (p < total) tile[p] = (tx < box.z) ? px[u] : (uint16_t)0;
                }
            }
        }
        __syncthreads();
        if (vy >= g.Y) continue;   // (a whole wave)
        const bool lane_ok = vx < g.X;   // lanes past the grid's x edge stay in (the marks at the end of the brick are made by the wave's first lanes): they update nothing

        size_t idx = plane * (z0 - g.z_store_begin) + (size_t)g.X * vy + vx;  // (custom nodes only)
        // distance / weight addressing: a wave-uniform base per plane (scalar registers) + one 32-bit lane offset that
        // is the same for every plane, so the per-voxel loads and stores need no address arithmetic on the vector unit
        const size_t brick_base = plane * (z0 - g.z_store_begin) + (size_t)g.X * (by * kTileY) + (size_t)bx * kTileX;
        const uint32_t lane_off = threadIdx.y * g.X + threadIdx.x;

        // voxel centre, x and y parts: initialise_deformation (src/TSDF/TSDFVolume.cu:783-784) then
        // integrate_kernel's offset + translation (:343)
        float cx = 0.f, cy = 0.f;
        // partial row sums of inv_pose * (c,1): the reference evaluates ((m_i1*x + m_i2*y) + m_i3*z) + m_i4
        float r1 = 0.f, r2 = 0.f, r3 = 0.f, r4 = 0.f;
        float r4_[kBatchZ] = {};
        uint32_t low_lo = 0, low_hi = 0;   // bit o: my voxel of plane z0 + o got a distance that is not safely positive
        // Voxels a brick at the grid boundary depends on are held to the stricter test of those bricks (flat, not just positive:
        // OccGrid).  Both tests are "not (d > lo) or d > hi" with per-lane bounds: (tau, +inf) inside, (the float below flat_lo,
        // flat_hi) for lanes in the x / y part of the rim zone; the planes of the z part are picked per batch of planes below.
        const bool rim_xy = occ.in_rim_zone(vx, occ.nbx) || occ.in_rim_zone(vy, occ.nby);
        const float flat_lo_open = __uint_as_float(__float_as_uint(occ.flat_lo) - 1u);   // d >= flat_lo  <=>  d > this  (flat_lo > 0, normal)
        const float mark_lo = rim_xy ? flat_lo_open : occ.tau, mark_hi = rim_xy ? occ.flat_hi : INFINITY;
        if (!DEFORM) {
            cx = ((((int)vx + 0.5f) * g.vs.x) + g.offset_clear.x) + g.offset.x;
            cy = ((((int)vy + 0.5f) * g.vs.y) + g.offset_clear.y) + g.offset.y;
            r1 = ip.m11 * cx + ip.m12 * cy;
            r2 = ip.m21 * cx + ip.m22 * cy;
            r3 = ip.m31 * cx + ip.m32 * cy;
            r4 = ip.m41 * cx + ip.m42 * cy;
        }

        // The planes of the brick are processed kBatchZ at a time in three passes -- project + gather depth,
        // decide + load distance/weight, blend + store -- so that the depth gathers of a batch, and then its
        // HBM loads, are all in flight together instead of one dependent chain per plane.  The batches are software
        // pipelined: the loads of batch b+1 are issued before batch b is blended and stored, so that there is always a
        // batch of loads in flight (two register sets, used alternately).
        auto project_and_load = [&](const uint32_t zb, float (&tsdf_)[kBatchZ], float (&pw_)[kBatchZ], float (&pd_)[kBatchZ]) {
            const size_t idx_b = idx + plane * (size_t)(zb - z0);
            float camz_[kBatchZ], cz_[kBatchZ];
            int px_[kBatchZ], py_[kBatchZ];
            uint32_t d_[kBatchZ];  // depth of the voxel's pixel, 0 = none
            bool act[kBatchZ];
#pragma unroll
            for (int j = 0; j < kBatchZ; j++) {
                const uint32_t vz = zb + j;
                act[j] = vz < z1 && lane_ok;
                float cz = 0.f;
                if (DEFORM) {
                    // (custom nodes: x/y parts differ per voxel)
                    cz = 0.f;
                    if (act[j]) {
                        const tsdf_deformation_node &nd = nodes[idx_b + plane * j];
                        cx = nd.translation[0] + g.offset.x;
                        cy = nd.translation[1] + g.offset.y;
                        cz = nd.translation[2] + g.offset.z;
                    }
                    r1 = ip.m11 * cx + ip.m12 * cy;
                    r2 = ip.m21 * cx + ip.m22 * cy;
                    r3 = ip.m31 * cx + ip.m32 * cy;
                    r4 = ip.m41 * cx + ip.m42 * cy;
                }
                // world_to_pixel (src/Utilities/cuda_coordinate_transforms.cu:10-30)
                float camx, camy, camz;
                if (DEFORM) {
                    camx = (r1 + ip.m13 * cz) + ip.m14;
                    camy = (r2 + ip.m23 * cz) + ip.m24;
                    camz = (r3 + ip.m33 * cz) + ip.m34;
                } else {
                    const float4 pc = plane_lds[vz - z0];
                    cz = pc.x;
                    camx = (r1 + pc.y) + ip.m14;
                    camy = (r2 + pc.z) + ip.m24;
                    camz = (r3 + pc.w) + ip.m34;
                }
                cz_[j] = cz;
                camz_[j] = camz;
                const float imx = STD ? k.m11 * camx + k.m13 * camz : k.m11 * camx + k.m12 * camy + k.m13 * camz;
                const float imy = STD ? k.m22 * camy + k.m23 * camz : k.m21 * camx + k.m22 * camy + k.m23 * camz;
                const float imz = STD ? camz : k.m31 * camx + k.m32 * camy + k.m33 * camz;
                // pixel = (int)round(q) with the target's conversion (NaN -> 0, saturating); the frustum test (:349) is
                // done on the rounded floats, which order exactly like the saturated ints
                float rx, ry;
                round_quotients(imx, imy, imz, round_near_half, rx, ry);
                // (the hardware conversion saturates: anything beyond the int range is off the image either way)
                px_[j] = cvt_i32_sat(rx);
                py_[j] = cvt_i32_sat(ry);
                // The brick's pixel box (cull kernel) holds every pixel a voxel of this brick can map to, and it lies
                // inside the image: a pixel in the box passes the frustum test.  The LDS read is unconditional (a slot
                // holding 0 when outside) and the global gather a separate, rare branch, so that neither turns into a generic
                // load that would have to be waited for plane by plane.
                const uint32_t tx = (uint32_t)px_[j] - box.x, ty = (uint32_t)py_[j] - box.y;
                const bool in_box = act[j] && staged && tx < box.z && ty < box.w;
                d_[j] = tile[in_box ? __umul24(ty, pitch) + tx : (uint32_t)kTilePixels];   // (in the box both factors are < 2^13: 24-bit multiply, full rate)
                if (act[j] && !in_box && rx >= 0.0f && rx < fwidth && ry >= 0.0f && ry < fheight)
                    d_[j] = depth[(uint32_t)py_[j] * width + (uint32_t)px_[j]];
                if (DEFORM) {  // keep the per-voxel row sums for pass 2
                    r4_[j] = r4;
                }
            }
            // tsdf_[j] is NaN for a voxel this frame does not update
#pragma unroll
            for (int j = 0; j < kBatchZ; j++) {
                // pixel_to_camera(...).z (cuda_coordinate_transforms.cu:132-146)
                float surf_z, voxel_cam_z;
                if (STD) {
                    surf_z = (float)d_[j];
                    voxel_cam_z = camz_[j];
                } else {
                    const float ipz = kinv.m31 * px_[j] + kinv.m32 * py_[j] + kinv.m33;
                    const float scale = (float)d_[j] / ipz;
                    surf_z = ipz * scale;
                    // world_to_camera(...).z (cuda_coordinate_transforms.cu:108-121): same numerator as camz
                    const float w = ((DEFORM ? r4_[j] : r4) + ip.m43 * cz_[j]) + ip.m44;
                    voxel_cam_z = camz_[j] / w;
                }
                const float sdf = surf_z - voxel_cam_z;
                // depth > 0 (:355; also false for planes past the brick and pixels off the image) and sdf >= -trunc (:366)
                const bool update = d_[j] != 0 && sdf >= neg_trunc;
                // (sdf > 0) ? min(sdf, trunc) : sdf  ==  sdf < trunc ? sdf : trunc   (trunc > 0)
                tsdf_[j] = update ? (sdf < g.trunc ? sdf : g.trunc) : NAN;
                pw_[j] = pd_[j] = 0.f;
                if (update) {
                    const size_t pb = brick_base + plane * (size_t)(zb - z0 + j);
                    pw_[j] = (weight + pb)[lane_off];
                    pd_[j] = (dist + pb)[lane_off];
                }
            }
        };
        auto blend_and_store = [&](const uint32_t zb, const float (&tsdf_)[kBatchZ], const float (&pw_)[kBatchZ], const float (&pd_)[kBatchZ]) {
            // (uniform) a batch with a plane in the z part of the rim zone -- z < 6 or z >= 4 (nbz - 1) - 2 -- takes the flat test on every lane
            const bool z_rim = zb < (uint32_t)(kBrick + kBrickGrow) || zb + (uint32_t)kBatchZ - 1u + (uint32_t)kBrickGrow >= (uint32_t)kBrick * (occ.nbz - 1u);
            const float lo = z_rim ? flat_lo_open : mark_lo, hi = z_rim ? occ.flat_hi : mark_hi;
#pragma unroll
            for (int j = 0; j < kBatchZ; j++) {
                if (tsdf_[j] == tsdf_[j]) {
                    const float new_weight = pw_[j] + 1.0f;
                    const float new_distance = ((pd_[j] * pw_[j]) + (tsdf_[j] * 1.0f)) / new_weight;
                    const size_t pb = brick_base + plane * (size_t)(zb - z0 + j);
                    (weight + pb)[lane_off] = new_weight;
                    (dist + pb)[lane_off] = new_distance;
                    if (!(new_distance > lo) || new_distance > hi) {   // not safely positive (rim zone: not flat): remember the plane, the bricks are marked when this one is done
                        const uint32_t o_ = zb + j - z0;
                        if (o_ < 32u) low_lo |= 1u << o_; else low_hi |= 1u << (o_ - 32u);
                    }
                    if (COUNT) updated++;
                }
            }
        };
        static_assert(kChunkZ % (2 * kBatchZ) == 0, "the pipeline alternates two register sets");
        float tsdf_a[kBatchZ], pw_a[kBatchZ], pd_a[kBatchZ], tsdf_b[kBatchZ], pw_b[kBatchZ], pd_b[kBatchZ];
        project_and_load(z0, tsdf_a, pw_a, pd_a);
#pragma unroll
        for (uint32_t o = 0; o < (uint32_t)kChunkZ; o += 2 * kBatchZ) {
            // (batches past z1 -- the last bricks of a grid whose depth is not a multiple of kChunkZ -- are all inactive)
            project_and_load(z0 + o + kBatchZ, tsdf_b, pw_b, pd_b);
            blend_and_store(z0 + o, tsdf_a, pw_a, pd_a);
            if (o + 2 * kBatchZ < (uint32_t)kChunkZ) project_and_load(z0 + o + 2 * kBatchZ, tsdf_a, pw_a, pd_a);
            blend_and_store(z0 + o + kBatchZ, tsdf_b, pw_b, pd_b);
        }
        if (z_extra != 0) {   // (uniform; after the pipeline, not inside it)
            project_and_load(z0 + kChunkZ, tsdf_a, pw_a, pd_a);
            blend_and_store(z0 + kChunkZ, tsdf_a, pw_a, pd_a);
        }
        if (__any((low_lo | low_hi) != 0u))
            mark_low_voxels(occ, low_lo, low_hi, (bx * kTileX) >> kBrickShift, __builtin_amdgcn_readfirstlane(vy), z0, z0, z1 - 1u, threadIdx.x);
        if (!COUNT && counter && tid == 0) { counter[2 * i] = dbg_t0; counter[2 * i + 1] = wall_clock64(); }
    }
    if (COUNT) {
        // wave reduction then one atomic per wave
        for (int o = 32; o > 0; o >>= 1) updated += __shfl_down(updated, o);
        if ((threadIdx.x & 63u) == 0 && updated) atomicAdd(counter, (unsigned long long)updated);
    }
}

// phase: kIntBoth = culling + integrate_kernel on the volume's stream; kIntPrepare = the culling only, on `prepare_stream`
// (tsdf_integrate_prepare_device_tiles: the brick list of a frame built ahead, e.g. on a lower-priority stream while the
// previous frame's ray cast runs -- the culling needs the depth image's tile maxima and the pose, not the volume); kIntBoth
// then finds the list prepared (same image, pose, intrinsics, tile maxima) and launches integrate_kernel alone.
enum IntegratePhase { kIntBoth = 0, kIntPrepare = 1 };
static int launch_integrate(tsdf_volume *v, const uint16_t *d_depth, uint32_t width, uint32_t height,
                            const float inv_pose[16], const float k[9], const float kinv[9], const uint16_t *caller_tile_max = nullptr,
                            IntegratePhase phase = kIntBoth, hipStream_t prepare_stream = nullptr) {
    Mat44 ip;
    Mat33 mk, mkinv;
    memcpy(&ip, inv_pose, sizeof(ip));
    memcpy(&mk, k, sizeof(mk));
    memcpy(&mkinv, kinv, sizeof(mkinv));
    const Geom &g = v->g;
    BrickGrid bg;
    bg.nx = (g.X + kTileX - 1) / kTileX;
    bg.ny = (g.Y + kTileY - 1) / kTileY;
    {
        const uint32_t planes = g.z_store_end - g.z_store_begin, full = planes / kChunkZ, rest = planes % kChunkZ;
        const bool append = full >= 1 && rest >= 1 && rest <= (uint32_t)kBatchZ;
        bg.nz = append ? full : (planes + kChunkZ - 1) / kChunkZ;
        bg.z_extra = append ? rest : 0;
    }
    bg.pair_loads = (width % 2 == 0 && (reinterpret_cast<uintptr_t>(d_depth) & 3u) == 0) ? 1u : 0u;
    const size_t n_bricks = (size_t)bg.nx * bg.ny * bg.nz;
    TSDF_REQUIRE(n_bricks < 0xFFFFFFFFull, "volume too large for the brick list");

    bool fresh_scratch = false;   // scratch allocated (and zeroed, on the volume's stream) by this very call
    if (!v->touched || v->touched_nx != bg.nx || v->touched_ny != bg.ny || v->touched_nz != bg.nz) {
        fresh_scratch = true;
        if (v->touched) (void)hipFree(v->touched);
        v->touched = nullptr;
        TSDF_HIP(hipMalloc((void **)&v->touched, n_bricks), "touched bricks alloc");
        TSDF_HIP(hipMemsetAsync(v->touched, 0, n_bricks, v->stream), "touched bricks alloc");
        v->touched_nx = bg.nx; v->touched_ny = bg.ny; v->touched_nz = bg.nz;
        v->occ_scan_all = 1;
    }
    // scratch: brick list + counter, depth tile maxima
    const uint32_t tiles_x = (width + kDepthTile - 1) / kDepthTile, tiles_y = (height + kDepthTile - 1) / kDepthTile;
    if (v->brick_list_cap < n_bricks + 2) {
        fresh_scratch = true;
        if (v->brick_list) (void)hipFree(v->brick_list);
        v->brick_list = nullptr;
        v->brick_list_cap = 0;
        TSDF_HIP(hipMalloc((void **)&v->brick_list, (n_bricks + 2) * sizeof(uint32_t)), "brick list alloc");
        TSDF_HIP(hipMemsetAsync(v->brick_list + n_bricks, 0, 2 * sizeof(uint32_t), v->stream), "brick list alloc");   // both length words
        v->brick_list_cap = n_bricks + 2;
        v->brick_count_side = 0;
    }
    if (v->tile_max_cap < (size_t)tiles_x * tiles_y) {
        if (v->tile_max) (void)hipFree(v->tile_max);
        v->tile_max = nullptr;
        v->tile_max_cap = 0;
        TSDF_HIP(hipMalloc((void **)&v->tile_max, (size_t)tiles_x * tiles_y * sizeof(uint16_t)), "depth tile alloc");
        v->tile_max_cap = (size_t)tiles_x * tiles_y;
    }
    if (v->wmode != 0 && !v->nodes && (!v->depth_pad || v->depth_pad_w != width || v->depth_pad_h != height)) {
        // (re)allocated between frames only: the culling that fills it and the integrate_packed_kernel that reads it belong to one frame
        if (v->depth_pad) (void)hipFree(v->depth_pad);
        v->depth_pad = nullptr;
        v->prepared_valid = 0;
        const size_t pad_bytes = (size_t)(width + 2u) * (height + 2u) * sizeof(uint16_t);
        TSDF_HIP(hipMalloc((void **)&v->depth_pad, pad_bytes), "padded depth alloc");
        // (zeroed on the stream the culling that fills it runs on, and waited for: the ring must be there before, not after, the interior)
        const hipStream_t pad_stream = phase == kIntPrepare ? prepare_stream : v->stream;
        TSDF_HIP(hipMemsetAsync(v->depth_pad, 0, pad_bytes, pad_stream), "padded depth alloc");
        TSDF_HIP(hipStreamSynchronize(pad_stream), "padded depth alloc");
        v->depth_pad_w = width; v->depth_pad_h = height;
    }
    // the list's length: the last two slots, used alternately (see brick_cull_kernel)
    // (a prepared list: its length is behind the word the culling of the prepare call used)
    PreparedCull sig;
    memset(&sig, 0, sizeof(sig));
    sig.depth = d_depth; sig.tile_max = caller_tile_max; sig.width = width; sig.height = height;
    memcpy(sig.inv_pose, inv_pose, sizeof(sig.inv_pose)); memcpy(sig.k, k, sizeof(sig.k)); memcpy(sig.kinv, kinv, sizeof(sig.kinv));
    const bool prepared = phase == kIntBoth && v->prepared_valid && !v->nodes && memcmp(&sig, &v->prepared, sizeof(sig)) == 0;
    if (phase == kIntBoth) v->prepared_valid = 0;     // (used up, or stale)
    if (prepared) v->brick_count_side = 1u - v->brick_count_side;   // back to the side the prepare call appended behind (toggled again below)
    uint32_t *count = v->brick_list + n_bricks + v->brick_count_side, *count_next = v->brick_list + n_bricks + (1u - v->brick_count_side);
    if (v->brick_box_cap < n_bricks) {
        if (v->brick_boxes) (void)hipFree(v->brick_boxes);
        v->brick_boxes = nullptr;
        v->brick_box_cap = 0;
        TSDF_HIP(hipMalloc((void **)&v->brick_boxes, n_bricks * 6 * sizeof(uint32_t)), "brick box alloc");   // (boxes, then the bricks' coordinates)
        v->brick_box_cap = n_bricks;
    }
    uint4 *boxes = reinterpret_cast<uint4 *>(v->brick_boxes);
    uint2 *coords = reinterpret_cast<uint2 *>(v->brick_boxes + 4 * v->brick_box_cap);   // per listed brick: {bx | by << 16, bz}, for integrate_packed_kernel
    const uint32_t n_plane_const = g.z_store_end - g.z_store_begin + kBatchZ;  // padded: a batch may run past the last plane
    if (!v->plane_const) TSDF_HIP(hipMalloc((void **)&v->plane_const, n_plane_const * 4 * sizeof(float)), "plane constants alloc");
    float4 *plane_const = reinterpret_cast<float4 *>(v->plane_const);

    if (v->counting && phase == kIntBoth) TSDF_HIP(hipMemsetAsync(v->counter_dev, 0, sizeof(unsigned long long), v->stream), "reset counter");
    if (prepared) {
        v->brick_count_side = 1u - v->brick_count_side;
    } else if (!v->nodes) {
        const hipStream_t cull_stream = phase == kIntPrepare ? prepare_stream : v->stream;
        if (phase == kIntPrepare && fresh_scratch) TSDF_HIP(hipStreamSynchronize(v->stream), "integrate scratch");   // (zeroed on the volume's stream)
        // the depth tests need surface z == depth and w == 1 exactly (rigid pose, standard intrinsics)
        const int depth_test = (ip.m41 == 0.0f && ip.m42 == 0.0f && ip.m43 == 0.0f && ip.m44 == 1.0f &&
                                mkinv.m31 == 0.0f && mkinv.m32 == 0.0f && mkinv.m33 == 1.0f) ? 1 : 0;
        // brick_cull_kernel's margins for the bricks that straddle the camera plane: what the fp32 values of image.x / .y / .z can be
        // off by anywhere in the resident grid (the kernel's own corner bounds, with every coordinate at its largest magnitude)
        float cone_mx, cone_my;
        {
            const double u = 6.0e-7;
            auto centre = [](uint32_t i, float vs, float oc, float o) { return (double)((((int)i + 0.5f) * vs + oc) + o); };
            const double PX = std::max(std::fabs(centre(0, g.vs.x, g.offset_clear.x, g.offset.x)), std::fabs(centre(g.X - 1, g.vs.x, g.offset_clear.x, g.offset.x)));
            const double PY = std::max(std::fabs(centre(0, g.vs.y, g.offset_clear.y, g.offset.y)), std::fabs(centre(g.Y - 1, g.vs.y, g.offset_clear.y, g.offset.y)));
            const double PZ = std::max(std::fabs(centre(g.z_store_begin, g.vs.z, g.offset_clear.z, g.offset.z)), std::fabs(centre(g.z_store_end - 1, g.vs.z, g.offset_clear.z, g.offset.z)));
            auto row = [&](float a, float b, float c, float d) { return std::fabs((double)a) * PX + std::fabs((double)b) * PY + std::fabs((double)c) * PZ + std::fabs((double)d); };
            const double CX = row(ip.m11, ip.m12, ip.m13, ip.m14), CY = row(ip.m21, ip.m22, ip.m23, ip.m24), CZ = row(ip.m31, ip.m32, ip.m33, ip.m34);
            auto img = [&](float a, float b, float c) { return 2.0 * u * (std::fabs((double)a) * CX + std::fabs((double)b) * CY + std::fabs((double)c) * CZ); };
            const double ex = img(mk.m11, mk.m12, mk.m13), ey = img(mk.m21, mk.m22, mk.m23), ez = img(mk.m31, mk.m32, mk.m33);
            // at the corner + at the voxel + forming L itself, doubled; not finite (wild matrices): NaN margins switch the test off
            cone_mx = (float)(8.0 * (ex + (double)width * ez) + 1.0e-3);
            cone_my = (float)(8.0 * (ey + (double)height * ez) + 1.0e-3);
        }
        if (!caller_tile_max)
            hipLaunchKernelGGL(depth_tile_max_kernel, dim3(tiles_x, tiles_y), dim3(64), 0, cull_stream, d_depth, width, height,
                               tiles_x, v->tile_max);
        hipLaunchKernelGGL(brick_cull_kernel, dim3((unsigned)((8 * n_bricks + 255) / 256)), dim3(256), 0, cull_stream, g, bg, ip, mk,
                           width, height, caller_tile_max ? caller_tile_max : v->tile_max, tiles_x, depth_test, v->brick_list, boxes, count,
                           count_next, plane_const, n_plane_const, cone_mx, cone_my, d_depth, v->wmode != 0 ? v->depth_pad : nullptr, coords);
        v->brick_count_side = 1u - v->brick_count_side;
        if (phase == kIntPrepare) {
            TSDF_HIP(hipGetLastError(), "Integrate culling failed");
            v->prepared = sig;
            v->prepared_valid = 1;
            return TSDF_OK;
        }
    }
    if (phase == kIntPrepare) return TSDF_OK;   // (custom nodes: every brick is walked, nothing to prepare)
    {   // integrate_kernel sets occupancy flags: a tightening still running on another stream comes first
        const int rcj = occupancy_join(v);
        if (rcj != TSDF_OK) return rcj;
    }
#ifdef TSDF_DIAGNOSTICS
    diag_sort_brick_list(v, bg, count, boxes);   // TSDF_DEBUG_SORT: the list re-ordered on the host (synchronises), diagnostics.hip
#endif
    dim3 block(kTileX, kTileY, 1);
    // One workgroup per brick of the grid; those beyond the list's length leave at once.  The dispatcher hands the next brick to
    // whichever compute unit frees a slot, which balances the uneven bricks better than a resident grid walking the list
    // with a fixed stride did (0.136 vs 0.144 ms: 5 300 bricks over 1 536 resident workgroups are 3.3 rounds).
    // TSDF_INT_GRID_PER_CU = n > 0 restores a resident grid of n workgroups per compute unit (tuning aid).
    const int grid_per_cu = tuning().int_grid_per_cu;
    dim3 grid((unsigned)(grid_per_cu > 0 ? std::min<size_t>(n_bricks, (size_t)256 * grid_per_cu) : n_bricks));
    bool finite = true;
    for (int i = 0; i < 16; i++) finite = finite && std::isfinite(inv_pose[i]);
    for (int i = 0; i < 9; i++) finite = finite && std::isfinite(k[i]) && std::isfinite(kinv[i]);
    const bool std_camera = finite && ip.m41 == 0.0f && ip.m42 == 0.0f && ip.m43 == 0.0f && ip.m44 == 1.0f &&
                            mk.m21 == 0.0f && mk.m31 == 0.0f && mk.m12 == 0.0f && mk.m32 == 0.0f && mk.m33 == 1.0f &&
                            mkinv.m31 == 0.0f && mkinv.m32 == 0.0f && mkinv.m33 == 1.0f;
    unsigned long long *brick_log = nullptr;
#ifdef TSDF_DIAGNOSTICS
    brick_log = diag_brick_log_alloc(v, n_bricks);   // TSDF_DEBUG_BRICKS=3: per-brick clocks of the launch (diagnostics.hip)
#endif
    unsigned long long *counter_arg = brick_log ? brick_log : (v->counting ? v->counter_dev : nullptr);
    // Weights kept as packed counts (weights.hip) go with integrate_packed_kernel, which is written for the standard camera and
    // implicit nodes; anything else takes the reference's fp32 layout first (and keeps it until clear()).
    if (v->wmode != 0) {
        // (integrate_packed_kernel addresses a brick's planes with 32-bit byte offsets)
        const bool planes_fit = (size_t)g.X * g.Y * sizeof(float) * (kChunkZ + kBatchZ) < ((size_t)1 << 31);
        int rcw = (v->nodes || !std_camera || brick_log || !planes_fit) ? weights_require_f32(v) : weights_make_room(v);
        if (rcw != TSDF_OK) return rcw;
    }
#define LAUNCH(DEF, CNT, STDC)                                                                                       \
    TSDF_LAUNCH_TIMED(v, 0, (integrate_kernel<DEF, CNT, STDC>), grid, block, v->dist, v->weight, v->nodes,          \
                      g, bg, ip, mk, mkinv, width, height, d_depth, counter_arg, v->occ, v->brick_list, boxes, count, plane_const, v->touched)
    if (v->wmode != 0) {
        const int rcp = launch_integrate_packed_kernel(v, dim3((unsigned)n_bricks), bg,   // (one brick per workgroup, always)
                                                        ip, mk, width, height, d_depth, counter_arg, boxes, coords, count, plane_const);
        if (rcp != TSDF_OK) return rcp;
        v->weight_bound++;
    } else if (v->nodes) {
        if (v->counting) LAUNCH(true, true, false); else LAUNCH(true, false, false);
    } else if (std_camera) {
        if (v->counting) LAUNCH(false, true, true); else LAUNCH(false, false, true);
    } else {
        if (v->counting) LAUNCH(false, true, false); else LAUNCH(false, false, false);
    }
#undef LAUNCH
    TSDF_HIP(hipGetLastError(), "Integrate kernel failed");
#ifdef TSDF_DIAGNOSTICS
    diag_brick_report(v, bg, n_bricks, count, boxes, brick_log);   // TSDF_DEBUG_BRICKS: survivors, list order, per-brick clocks
#endif
    v->reach_dirty = 1;  // bricks may have been flagged
    // The kernel only sets occupancy flags.  A voxel that was low when first seen (sensor dropouts smeared by the
    // bilateral filter put phantom surfaces into free space) and has since been averaged back up keeps its bricks
    // flagged, which fragments the empty regions the ray caster jumps over; so the flags are recomputed from the
    // distances (volume.hip: occupancy_rebuild, one streaming read) after 2, 4, 8, 16 frames, then every period.
    v->integrations_total++;
    v->integrations_since_rebuild++;
    const uint32_t period = occupancy_rebuild_period();
    if (period && (v->integrations_since_rebuild >= period ||
                   (v->integrations_total <= period && (v->integrations_total & (v->integrations_total - 1)) == 0 &&
                    v->integrations_total >= 2)))
        v->occ_tighten_due = 1;
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

int tsdf_integrate_device_tiles(tsdf_volume *v, const uint16_t *device_depth, uint32_t width, uint32_t height,
                                const float pose[16], const float inv_pose[16], const float k[9], const float kinv[9],
                                const uint16_t *device_tile_max) {
    TSDF_REQUIRE(v && device_depth && inv_pose && k && kinv && device_tile_max, "tsdf_integrate: null argument");
    TSDF_REQUIRE(width > 0 && height > 0, "tsdf_integrate: empty depth map");
    (void)pose;
    return launch_integrate(v, device_depth, width, height, inv_pose, k, kinv, device_tile_max);
}

int tsdf_integrate_prepare_device_tiles(tsdf_volume *v, const uint16_t *device_depth, uint32_t width, uint32_t height,
                                        const float pose[16], const float inv_pose[16], const float k[9], const float kinv[9],
                                        const uint16_t *device_tile_max, void *hip_stream) {
    TSDF_REQUIRE(v && device_depth && inv_pose && k && kinv && device_tile_max, "tsdf_integrate: null argument");
    TSDF_REQUIRE(width > 0 && height > 0, "tsdf_integrate: empty depth map");
    (void)pose;
    return launch_integrate(v, device_depth, width, height, inv_pose, k, kinv, device_tile_max, kIntPrepare, (hipStream_t)hip_stream);
}

int tsdf_integrate_discard_prepared(tsdf_volume *v) {
    TSDF_REQUIRE(v, "null volume");
    v->prepared_valid = 0;
    return TSDF_OK;
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
