// integrate's brick grid (integrate.hip; shared with the optional host diagnostics in diagnostics.hip).
#pragma once

#include "common.hpp"

namespace tsdf {

constexpr int kTileX = kIntBrickX;  // one wave along x
constexpr int kTileY = kIntBrickY;  // waves per workgroup
constexpr int kChunkZ = kIntBrickZ; // planes walked by one workgroup
#ifndef TSDF_BATCH_Z
#define TSDF_BATCH_Z 4
#endif
constexpr int kBatchZ = TSDF_BATCH_Z;  // planes whose loads are issued together
#ifndef TSDF_TILE_PIXELS
#define TSDF_TILE_PIXELS 8192
#endif
constexpr int kTilePixels = TSDF_TILE_PIXELS;  // LDS depth tile of a brick: 16 KiB

struct BrickGrid {
    uint32_t nx, ny, nz;  // bricks per axis over the resident planes
    uint32_t z_extra;     // planes (<= kBatchZ) appended to the bricks of the last z layer: a slab's halo plane, which would
                          // otherwise cost a whole layer of bricks that project 32 planes to update one
    uint32_t pair_loads;  // 1 = the image has an even width and a 4-byte aligned base: pixel boxes are made even and staged two pixels per lane
};

#ifdef __HIPCC__
// Occupancy marks of integrate_kernel.  Each lane keeps one bit per plane of its brick ("my voxel of that plane got a new
// distance that is not safely positive"); when the brick is done, mark_low_voxels() turns the bits of a wave row (64 voxels along x
// from voxel 4 * bx0, at vy, planes z0 .. z0 + 63 at most) into exactly the bricks mark_occupied() flags voxel by voxel: per layer
// of bricks along z, the ballot of the lanes with a bit among the planes that reach the layer, grown along x with scalar mask
// arithmetic, written by at most 18 lanes.  Called by the whole wave.
__device__ inline uint64_t plane_range_mask(int lo, int hi) {   // bits lo .. hi (any ints) of a 64-bit mask
    lo = max(lo, 0);
    hi = min(hi, 63);
    return hi < lo ? 0ull : ((~0ull >> (63 - hi)) & (~0ull << lo));
}
__device__ inline void mark_low_voxels(const OccGrid &occ, const uint32_t bits_lo, const uint32_t bits_hi, const uint32_t bx0,
                                       const uint32_t vy, const uint32_t z0, const uint32_t z_first, const uint32_t z_last, const uint32_t lane) {
    const int bxl = (int)bx0 - 1 + (int)lane;   // lane l looks after brick bx0 - 1 + l
    const bool in_grid = bxl >= 0 && bxl < (int)occ.nbx && lane < 18u;
    const uint32_t sh = 4u * ((lane - 1u) & 15u);
    // y: the row's voxels reach the bricks holding vy-2 .. vy+2 (fine) and vy / 4, plus the one before when vy is a multiple of 4 (cell)
    const uint32_t by0 = (max(vy, (uint32_t)kBrickGrow) - kBrickGrow) >> kBrickShift, by1 = min((vy + kBrickGrow) >> kBrickShift, occ.nby - 1);
    const uint32_t cy1 = vy >> kBrickShift, cy0 = ((vy & (kBrick - 1)) == 0 && cy1 > 0) ? cy1 - 1 : cy1;
    const uint32_t zb_first = (max(z_first, (uint32_t)kBrickGrow) - kBrickGrow) >> kBrickShift, zb_last = min((z_last + kBrickGrow) >> kBrickShift, occ.nbz - 1);
#pragma unroll 1
    for (uint32_t zb = zb_first; zb <= zb_last; zb++) {
        // fine[zb] takes voxels with (z-2)/4 <= zb <= (z+2)/4, i.e. z in [4 zb - 2, 4 zb + 5]; cell[zb] those with z / 4 == zb or
        // z == 4 (zb + 1), i.e. z in [4 zb, 4 zb + 4]  (mark_occupied's ranges, solved for the brick)
        const int rel = (int)(zb << kBrickShift) - (int)z0;
        const uint64_t zf = plane_range_mask(rel - 2, rel + 5), zc = plane_range_mask(rel, rel + 4);
        const uint64_t low = __ballot(((bits_lo & (uint32_t)zf) | (bits_hi & (uint32_t)(zf >> 32))) != 0u);
        if (low) {
            // x, fine: the bricks holding v-2 .. v+2.  Inside the row that is the mask grown by two lanes either way, nibble by
            // nibble; lanes 0, 1 also reach the brick before the row, lanes 62, 63 the one after it
            const uint64_t grown = low | (low << 1) | (low << 2) | (low >> 1) | (low >> 2);
            const bool f_fine = in_grid && (lane == 0u ? (low & 3ull) != 0 : lane == 17u ? (low >> 62) != 0 : ((grown >> sh) & 15ull) != 0);
            if (f_fine) {   // (by1 - by0 is 0 or 1: two stores, possibly to the same byte)
                occ.fine[((size_t)zb * occ.nby + by0) * occ.nbx + bxl] = 1;
                occ.fine[((size_t)zb * occ.nby + by1) * occ.nbx + bxl] = 1;
            }
        }
        const uint64_t lowc = __ballot(((bits_lo & (uint32_t)zc) | (bits_hi & (uint32_t)(zc >> 32))) != 0u);
        if (lowc) {
            // x, cell: brick v / 4, and the one before it when v is a multiple of 4
            const uint64_t with_prev = lowc | ((lowc & 0x1111111111111111ull) >> 1);
            const bool f_cell = in_grid && lane < 17u && (lane == 0u ? (lowc & 1ull) != 0 : ((with_prev >> sh) & 15ull) != 0);
            if (f_cell) {
                occ.cell[((size_t)zb * occ.nby + cy0) * occ.nbx + bxl] = 1;
                occ.cell[((size_t)zb * occ.nby + cy1) * occ.nbx + bxl] = 1;
            }
        }
    }
}

// float -> int as the hardware converts: saturating, NaN -> 0 (defined for every float, unlike the C cast)
__device__ inline int cvt_i32_sat(float f) {
    int i;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(i) : "v"(f));
    return i;
}

#endif  // __HIPCC__

#ifdef TSDF_DIAGNOSTICS
// diagnostics.hip (built only with `make DIAG=1`): host-side experiments on integrate's brick list; every one synchronises.
void diag_sort_brick_list(struct ::tsdf_volume *v, const BrickGrid &bg, uint32_t *count, uint4 *boxes);
unsigned long long *diag_brick_log_alloc(struct ::tsdf_volume *v, size_t n_bricks);
void diag_brick_report(struct ::tsdf_volume *v, const BrickGrid &bg, size_t n_bricks, uint32_t *count, uint4 *boxes, unsigned long long *brick_log);
#endif

}  // namespace tsdf
