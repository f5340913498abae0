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

#ifdef TSDF_DIAGNOSTICS
// diagnostics.hip (built only with `make DIAG=1`): host-side experiments on integrate's brick list; every one synchronises.
void diag_sort_brick_list(struct ::tsdf_volume *v, const BrickGrid &bg, uint32_t *count, uint4 *boxes);
unsigned long long *diag_brick_log_alloc(struct ::tsdf_volume *v, size_t n_bricks);
void diag_brick_report(struct ::tsdf_volume *v, const BrickGrid &bg, size_t n_bricks, uint32_t *count, uint4 *boxes, unsigned long long *brick_log);
#endif

}  // namespace tsdf
