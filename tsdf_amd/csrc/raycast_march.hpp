// The march: process_ray (src/RayCaster/GPURaycaster.cu:265-377) ray by ray, for the views the cell-parallel cast (raycast_cells.hpp) is not
// taken for -- a camera inside the volume, voxels of many pixels, fields in which every brick is flagged -- and for the instrumented
// casts (tsdf_raycast_stats).  (Included by raycast.hip inside namespace tsdf, behind the helpers it uses: trilinear, SkipCtx /
// samples_until / lipschitz_lookahead, RayState / setup_ray, TailQueue / hit_word / lower_best.)
//   process_sample, process_sample_eager   one sample of one ray: jump over proven-empty space, or the reference's value
//   march_bulk / process_ray_kernel        one lane per pixel, sample ranges, a pass budget, the queue of unfinished stretches
//   order_ray_tiles                        the dispatch order the next cast learns from this one
//   march_tail / process_ray_tail_kernel   the queue: several lanes per stretch
// What was measured on the march and dropped (both kernels in one launch with queue workers polling behind the marching workgroups, the
// tail's lanes spaced by cells, per-wave clocks as a diagnostics build): tools/experiments/raycast_march_fused_and_tail_chain.patch.txt.

// Locates the sample at voxel coordinate f = p / vs in the brick grid.  Returns true when the samples from this
// one up to the exit of an empty region may be skipped; n = their number (>= 1).  The region is the largest clear
// aligned block of bricks around the sample (reach[]).  When false (flagged or boundary brick, or a position off
// the grid) n = samples that stay inside the brick.
template <bool SLAB>
__device__ inline bool locate(float fx, float fy, float fz, const SkipCtx &c, const Geom &g, const OccGrid &occ,
                              const RayParams &rp, int &n) {
    const int vx = (int)floorf(fx), vy = (int)floorf(fy), vz = (int)floorf(fz);
    n = 1;
    if ((uint32_t)vx >= g.X || (uint32_t)vy >= g.Y || (uint32_t)vz >= g.Z) return false;
    if (SLAB) {
        // A slab evaluates only samples whose lower tap plane lz (= voxel z or voxel z - 1) it owns.  Samples
        // located (to within one voxel) in z planes [z0, z1) have lz in [z0 - 2, z1]; if that misses the owned
        // range entirely the whole run is passed unevaluated, like an empty region -- 32 voxels at a time if possible.
        const int own_lo = (int)rp.own_lo, own_hi = (int)rp.own_hi;
        const int cz0 = (vz >> kSlabSkipShift) << kSlabSkipShift, bz0 = (vz >> kBrickShift) << kBrickShift;
        if (cz0 + kSlabSkip < own_lo || cz0 - 2 >= own_hi) {
            const int cx0 = (vx >> kSlabSkipShift) << kSlabSkipShift, cy0 = (vy >> kSlabSkipShift) << kSlabSkipShift;
            n = samples_to_exit<true>(fx, fy, fz, c, cx0, cy0, cz0, (float)kSlabSkip);
            return true;
        }
        if (bz0 + kBrick < own_lo || bz0 - 2 >= own_hi) {
            const int bx0 = (vx >> kBrickShift) << kBrickShift, by0 = (vy >> kBrickShift) << kBrickShift;
            n = samples_to_exit<true>(fx, fy, fz, c, bx0, by0, bz0, (float)kBrick);
            return true;
        }
    }
    const int bx = vx >> kBrickShift, by = vy >> kBrickShift, bz = vz >> kBrickShift;
    const int reach = occ.reach[(__umul24((uint32_t)bz, occ.nby) + (uint32_t)by) * occ.nbx + (uint32_t)bx];   // (fewer than 2^32 bricks: volume.hip)
    // aligned block of 4 * 2^(reach-1) voxels per side (the brick itself when reach is 0)
    const int shift = kBrickShift + max(reach, 1) - 1, size = 1 << shift;
    const int x0 = (vx >> shift) << shift, y0 = (vy >> shift) << shift, z0 = (vz >> shift) << shift;
    n = samples_to_exit<true>(fx, fy, fz, c, x0, y0, z0, (float)size);
    return reach != 0;
}

__device__ inline int wave_min(int v) {
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}

// Classification that stays valid while the ray remains in the same brick / cell brick (single-ray marching only).
struct BrickCache {
    int k_brick_end;      // the brick classification holds while k < k_brick_end
    int k_cellbrick_end;  // ... and the cell-brick classification while k < k_cellbrick_end
    bool cellbrick_clear;
};
struct SampleWork {  // diagnostics (STATS)
    uint32_t samples, hops, cell_tests, trips;
};

// Sample k of ray r, at parameter t = T[k].  Either proves that samples k .. k+jump-1 cannot be <= 0 (jump > 0),
// or returns the sample's value, computed with the reference's arithmetic (jump == 0; NaN for a sample another slab owns):
//   1. (when a brick boundary was crossed) read the brick's reach; a clear interior region is jumped over;
//   2. otherwise gather the 8 voxels of the sample's dual cell; if they are all safely positive the samples up to the
//      cell's (shrunk) exit are jumped over;
//   3. otherwise the sample is interpolated from those 8 values.
// Samples in the outer half-voxel shell of the grid, within eps of a cell face, or with skipping disabled take the
// reference's full trilinearly_interpolate instead (rare).
template <bool SLAB, bool STATS, bool FASTDIV>
__device__ inline float process_sample(float t, int k, const RayState &r, const SkipCtx &sc, BrickCache &bc,
                                       const float *__restrict__ dist, const Geom &g, const TriConst &tc, const RayParams &rp,
                                       const OccGrid &occ, unsigned int *__restrict__ touched, SampleWork &work, int &jump, int &ahead) {
    const float px = (t * r.dx) + r.sx, py = (t * r.dy) + r.sy, pz = (t * r.dz) + r.sz;
    const float cell_lo = sc.eps, cell_hi = 1.0f - sc.eps, cell_half = 0.5f - sc.eps;
    jump = 0;
    ahead = 0;   // samples after this one proven positive, valid when the returned value is > 0
    if (sc.skip_ok) {
        // position in voxel units (approximate)
        const float fx = px * sc.inv_vx, fy = py * sc.inv_vy, fz = pz * sc.inv_vz;
        if (k >= bc.k_brick_end) {
            int n;
            const bool empty = locate<SLAB>(fx, fy, fz, sc, g, occ, rp, n);
            bc.k_brick_end = k + n;
            if (empty) {
                if (STATS) work.hops++;
                RAY_MIX(0);
                jump = n;
                return 1.0f;
            }
        }
        // dual cell of the sample: lower = floor(p/vs - 1/2), position inside it in [0,1)
        const float cx = fx - 0.5f, cy = fy - 0.5f, cz = fz - 0.5f;
        const float lfx = floorf(cx), lfy = floorf(cy), lfz = floorf(cz);
        const float rx = cx - lfx, ry = cy - lfy, rz = cz - lfz;
        const int lx = (int)lfx, ly = (int)lfy, lz = (int)lfz;
        // at least eps away from the cell faces, and all 8 voxels of the cell exist (then no tap is clamped, the
        // weights lie in [0,1], and p is inside the grid so nothing is clamped either; the outer half-voxel shell of
        // the grid, where the reference extrapolates (Q10), fails this)
        const bool safe = fabsf(rx - 0.5f) < cell_half && fabsf(ry - 0.5f) < cell_half && fabsf(rz - 0.5f) < cell_half &&
                          (uint32_t)lx < g.X - 1 && (uint32_t)ly < g.Y - 1 && (uint32_t)lz < g.Z - 1;
        if (safe) {
            if (k >= bc.k_cellbrick_end) {
                // the sample's cell is known exactly: is its whole cell brick (4^3 cells) clear?  If so the samples
                // up to the exit of that brick (shrunk by eps, in cell coordinates) cannot hit.
                const int qx = lx >> kBrickShift, qy = ly >> kBrickShift, qz = lz >> kBrickShift;
                const float e = sc.eps;
                const int n_cb = samples_until<false>(((float)(qx << kBrickShift) + __builtin_fmaf(kBrick - 2.0f * e, sc.posx, e)) - cx,
                                                      ((float)(qy << kBrickShift) + __builtin_fmaf(kBrick - 2.0f * e, sc.posy, e)) - cy,
                                                      ((float)(qz << kBrickShift) + __builtin_fmaf(kBrick - 2.0f * e, sc.posz, e)) - cz, sc);
                bc.k_cellbrick_end = k + n_cb;
                bc.cellbrick_clear = occ.cell[(__umul24((uint32_t)qz, occ.nby) + (uint32_t)qy) * occ.nbx + (uint32_t)qx] == 0;
            }
            if (bc.cellbrick_clear && k < bc.k_cellbrick_end) {
                RAY_MIX(1);
                jump = bc.k_cellbrick_end - k;
                return 1.0f;
            }
            if (STATS) work.cell_tests++;
            // samples until the ray leaves the cell shrunk by eps
            const int n_cell = samples_until<false>(__builtin_fmaf(cell_hi - cell_lo, sc.posx, cell_lo) - rx, __builtin_fmaf(cell_hi - cell_lo, sc.posy, cell_lo) - ry,
                                                    __builtin_fmaf(cell_hi - cell_lo, sc.posz, cell_lo) - rz, sc);
            if (SLAB && !((uint32_t)lz >= rp.own_lo && (uint32_t)lz < rp.own_hi)) {
                RAY_MIX(2);
                jump = n_cell;  // not this rank's samples (and possibly not its planes): passed unevaluated
                return 1.0f;
            }
            const float *b = dist + ((size_t)tc.plane * ((uint32_t)lz - g.z_store_begin) + (__umul24(tc.row, (uint32_t)ly) + (uint32_t)lx));   // (X, Y < 2^16: 24-bit multiply, X * Y < 2^32)
            // (pairs along x through one pointer each, so that they can be fetched as 64-bit loads)
            const float *b_y = b + tc.row, *b_z = b + tc.plane, *b_yz = b_z + tc.row;
            const float c000 = b[0], c100 = b[1], c010 = b_y[0], c110 = b_y[1];
            const float c001 = b_z[0], c101 = b_z[1], c011 = b_yz[0], c111 = b_yz[1];
            // all corners above the threshold.  (fminf drops a NaN corner; that is fine here: inside the cell every weight is
            // non-zero, so with a NaN corner every sample of the cell is NaN, which the reference steps over as well)
            const bool positive = fminf(fminf(fminf(c000, c100), fminf(c010, c110)), fminf(fminf(c001, c101), fminf(c011, c111))) > kCellPositive;
            if (positive) {
                RAY_MIX(3);
                jump = n_cell;
                return 1.0f;
            }
            // trilinearly_interpolate (:84-121) for lower = (lx,ly,lz), which is what the reference derives for a
            // sample this far from the cell faces
            // (lower + 0.5f) * vs + 0.0f: lf* are the lower indices as floats already (exact integers), and adding 0.0f to
            // a positive product is the identity
            const float lcx = (lfx + 0.5f) * g.vs.x;
            const float lcy = (lfy + 0.5f) * g.vs.y;
            const float lcz = (lfz + 0.5f) * g.vs.z;
            const float u = div_by<FASTDIV>(px - lcx, tc.dx);
            const float v = div_by<FASTDIV>(py - lcy, tc.dy);
            const float w = div_by<FASTDIV>(pz - lcz, tc.dz);
            if (STATS) work.samples++;
            const float val = c000 * (1 - u) * (1 - v) * (1 - w) +
                              c001 * (1 - u) * (1 - v) * w +
                              c010 * (1 - u) * v * (1 - w) +
                              c011 * (1 - u) * v * w +
                              c100 * u * (1 - v) * (1 - w) +
                              c101 * u * (1 - v) * w +
                              c110 * u * v * (1 - w) +
                              c111 * u * v * w;
            ahead = lipschitz_lookahead(val, c000, c100, c010, c110, c001, c101, c011, c111, sc, n_cell - 1);
            RAY_MIX(val <= 0 ? 4 : (ahead == n_cell - 1 ? 5 : 6));
            return val;
        }
    }
    RAY_MIX(7);
    bool owned;
    const float tsdf = trilinear<SLAB, STATS, FASTDIV>(px, py, pz, dist, g, tc, rp, owned, touched);
    if (STATS && owned) work.samples++;
    return tsdf;
}

// The value lane q of this lane's group of `lanes` lanes holds (groups of 4: one DPP move)
template <int LANES>
__device__ inline int group_lane(int x, int q, uint32_t lanes) {
    if (LANES == 4) {
        switch (q) {
        case 0: return __builtin_amdgcn_mov_dpp(x, 0x00, 0xf, 0xf, true);   // quad_perm [0,0,0,0]
        case 1: return __builtin_amdgcn_mov_dpp(x, 0x55, 0xf, 0xf, true);   // [1,1,1,1]
        case 2: return __builtin_amdgcn_mov_dpp(x, 0xaa, 0xf, 0xf, true);   // [2,2,2,2]
        default: return __builtin_amdgcn_mov_dpp(x, 0xff, 0xf, 0xf, true);  // [3,3,3,3]
        }
    }
    return __shfl(x, (int)((threadIdx.x & 63u & ~(lanes - 1u)) + (uint32_t)q));
}

// process_sample for the tail kernel: the same decisions in the same order, but without the per-brick memory (a lane
// takes a different ray's sample every time) and with every load issued before the first decision -- the brick's reach,
// the cell brick's flag and the 8 voxels have addresses that depend only on the sample position, so one memory round
// trip serves all three instead of three dependent ones.  (Only worth it where rays are few: the speculative gathers
// cost bandwidth in the bulk kernel.)
template <bool SLAB, bool FASTDIV>
__device__ inline float process_sample_eager(float t, const RayState &r, const SkipCtx &sc, const float *__restrict__ dist,
                                             const Geom &g, const TriConst &tc, const RayParams &rp, const OccGrid &occ, int &jump, int &ahead, bool mix0 = false) {
    const float px = (t * r.dx) + r.sx, py = (t * r.dy) + r.sy, pz = (t * r.dz) + r.sz;
    const float cell_lo = sc.eps, cell_hi = 1.0f - sc.eps, cell_half = 0.5f - sc.eps;
    jump = 0;
    ahead = 0;
    if (sc.skip_ok) {
        const float fx = px * sc.inv_vx, fy = py * sc.inv_vy, fz = pz * sc.inv_vz;
        int n_brick;
        const bool empty = locate<SLAB>(fx, fy, fz, sc, g, occ, rp, n_brick);
        const float cx = fx - 0.5f, cy = fy - 0.5f, cz = fz - 0.5f;
        const float lfx = floorf(cx), lfy = floorf(cy), lfz = floorf(cz);
        const float rx = cx - lfx, ry = cy - lfy, rz = cz - lfz;
        const int lx = (int)lfx, ly = (int)lfy, lz = (int)lfz;
        const bool safe = fabsf(rx - 0.5f) < cell_half && fabsf(ry - 0.5f) < cell_half && fabsf(rz - 0.5f) < cell_half &&
                          (uint32_t)lx < g.X - 1 && (uint32_t)ly < g.Y - 1 && (uint32_t)lz < g.Z - 1;
        const int qx = lx >> kBrickShift, qy = ly >> kBrickShift, qz = lz >> kBrickShift;
        const float e = sc.eps;
        const int n_cb = samples_until<false>(((float)(qx << kBrickShift) + __builtin_fmaf(kBrick - 2.0f * e, sc.posx, e)) - cx,
                                              ((float)(qy << kBrickShift) + __builtin_fmaf(kBrick - 2.0f * e, sc.posy, e)) - cy,
                                              ((float)(qz << kBrickShift) + __builtin_fmaf(kBrick - 2.0f * e, sc.posz, e)) - cz, sc);
        const int n_cell = samples_until<false>(__builtin_fmaf(cell_hi - cell_lo, sc.posx, cell_lo) - rx, __builtin_fmaf(cell_hi - cell_lo, sc.posy, cell_lo) - ry,
                                                    __builtin_fmaf(cell_hi - cell_lo, sc.posz, cell_lo) - rz, sc);
        const bool owned = !SLAB || ((uint32_t)lz >= rp.own_lo && (uint32_t)lz < rp.own_hi);
        unsigned char cell_flag = 1;
        float c000 = 0, c100 = 0, c010 = 0, c110 = 0, c001 = 0, c101 = 0, c011 = 0, c111 = 0;
        if (safe) {
            cell_flag = occ.cell[(__umul24((uint32_t)qz, occ.nby) + (uint32_t)qy) * occ.nbx + (uint32_t)qx];
            if (owned) {  // (a slab holds the planes of the samples it owns, and only those for certain)
                const float *b = dist + ((size_t)tc.plane * ((uint32_t)lz - g.z_store_begin) + (__umul24(tc.row, (uint32_t)ly) + (uint32_t)lx));   // (X, Y < 2^16: 24-bit multiply, X * Y < 2^32)
                const float *b_y = b + tc.row, *b_z = b + tc.plane, *b_yz = b_z + tc.row;
                c000 = b[0]; c100 = b[1]; c010 = b_y[0]; c110 = b_y[1];
                c001 = b_z[0]; c101 = b_z[1]; c011 = b_yz[0]; c111 = b_yz[1];
            }
        }
        if (empty) {
            RAY_MIX(8); if (mix0) RAY_MIX(16);
            jump = n_brick;
            return 1.0f;
        }
        if (safe) {
            if (cell_flag == 0) {
                RAY_MIX(9); if (mix0) RAY_MIX(17);
                jump = n_cb;
                return 1.0f;
            }
            // all corners above the threshold.  (fminf drops a NaN corner; that is fine here: inside the cell every weight is
            // non-zero, so with a NaN corner every sample of the cell is NaN, which the reference steps over as well)
            const bool positive = fminf(fminf(fminf(c000, c100), fminf(c010, c110)), fminf(fminf(c001, c101), fminf(c011, c111))) > kCellPositive;
            if (!owned || positive) {
                RAY_MIX(owned ? 11 : 10); if (mix0) RAY_MIX(owned ? 19 : 18);
                jump = n_cell;
                return 1.0f;
            }
            // (lower + 0.5f) * vs + 0.0f: lf* are the lower indices as floats already (exact integers), and adding 0.0f to
            // a positive product is the identity
            const float lcx = (lfx + 0.5f) * g.vs.x;
            const float lcy = (lfy + 0.5f) * g.vs.y;
            const float lcz = (lfz + 0.5f) * g.vs.z;
            const float u = div_by<FASTDIV>(px - lcx, tc.dx);
            const float v = div_by<FASTDIV>(py - lcy, tc.dy);
            const float w = div_by<FASTDIV>(pz - lcz, tc.dz);
            const float val = c000 * (1 - u) * (1 - v) * (1 - w) +
                              c001 * (1 - u) * (1 - v) * w +
                              c010 * (1 - u) * v * (1 - w) +
                              c011 * (1 - u) * v * w +
                              c100 * u * (1 - v) * (1 - w) +
                              c101 * u * (1 - v) * w +
                              c110 * u * v * (1 - w) +
                              c111 * u * v * w;
            ahead = lipschitz_lookahead(val, c000, c100, c010, c110, c001, c101, c011, c111, sc, n_cell - 1);
            RAY_MIX(val <= 0 ? 12 : (ahead == n_cell - 1 ? 13 : 14)); if (mix0) RAY_MIX(val <= 0 ? 20 : (ahead == n_cell - 1 ? 21 : 22));
            return val;
        }
    }
    RAY_MIX(15); if (mix0) RAY_MIX(23);
    bool owned;
    return trilinear<SLAB, false, FASTDIV>(px, py, pz, dist, g, tc, rp, owned, nullptr);
}

// One lane per pixel, a wave is an 8x8 pixel tile of coherent rays, a workgroup a 16x16 tile.  Every pass of the loop
// does the same straight-line work for all lanes (process_sample), so lanes do not serialise on divergent code paths.
// Sample-range splitting (SEG): with rp.seg_len > 0 the z index of the workgroup selects a contiguous range of sample
// indices; every range is marched independently and lowers tail.best[pixel] to the index of its first sample <= 0.  The
// smallest index over all ranges is the sample the reference's serial loop stops at -- every sample before it was
// evaluated positive or proven positive by the range it belongs to -- and resolve_hits_kernel recomputes that one sample
// to form the vertex.  A range that starts at or after an index already in best[] has nothing to contribute and leaves
// (ranges are dispatched in ascending order, so later ones usually find the earlier ones' hits).
// TAIL: most rays finish within a few dozen passes, a few (those grazing a surface, e.g. the skirts the bilateral filter
// leaves at depth discontinuities) need hundreds of evaluated samples.  After tail.trip_budget passes a wave appends what
// is left of its unfinished rays to a queue, in pieces, and leaves; process_ray_tail_kernel finishes them with 16 lanes per
// piece.
//   SEG: nothing is written but best[]; otherwise out = packed float3 vertices (diagnostic variants).
//   STATS: counters[1] += samples evaluated, counters[2] += hits, touched bitmap marked per tap.  With
//          SKIP=false the counts are those of the reference's march.
template <bool SLAB, bool STATS, bool SKIP, bool FASTDIV, bool SEG, bool TAIL>
__device__ inline void march_bulk(const float *__restrict__ dist, const Geom &g, const RayParams &rp, float *__restrict__ out,
                                  unsigned long long *__restrict__ counters, unsigned int *__restrict__ touched, const OccGrid &occ,
                                  const float *__restrict__ t_table, const TailQueue &tail, float *Ts, const uint32_t nz) {
    // (nz: slabs of marching workgroups in the launch: gridDim.z)
    // Whole volume: range z of the workgroup grid is the fixed sample interval [z * seg_len, (z+1) * seg_len).
    // Slab (rp.slab_ranges > 0): a slab owns only a short stretch of every ray, a different one per ray, so the ranges
    // are cut per ray out of ITS stretch (below); any sample index may be needed and the whole table is staged.
    const bool per_ray_ranges = SLAB && rp.slab_ranges > 0;
    // Workgroups are dispatched in blockIdx order, z slowest, and the launch is as long as its last-dispatched long waves.  The
    // long waves are those of the ranges that hold surfaces and of the last and the first range, where every ray crosses the
    // permanently flagged rim bricks of the grid (Q10) sample by sample; ranges of free space take no pass at all.  A view from
    // outside has its surfaces and the exit rim in the far ranges, so the ranges are dispatched from far to near: on the bench
    // scene 0.097 ms instead of 0.112 (ascending) or 0.105 (last, first, then descending).  Per-wave clocks (a diagnostics build of round 3, profiles/r03i_ray_waves.txt):
    // range 5 -- 11.5 passes a wave, 672 waves using the whole budget, up to 53 us each -- ended the launch at 99 us when
    // dispatched last and is done at 56 us when dispatched first; ranges 4 and 3 have 424 and 329 such waves, range 0 (10.5
    // passes through the entry rim, none over 23 us) ends last at 89 us.  Any order gives the same picture: the ranges meet in
    // an atomicMin, and the early exit below only ever drops work.
    const uint32_t bz_ = blockIdx.z;
    // Which (sample range, tile) this workgroup takes: its own in launch order, or -- tail.order, scheduling only -- what the order
    // learnt from the previous cast gives its slot: the pairs in which a wave used its whole pass budget first.  The launch is as long
    // as its long waves (40-60 us each on the bench scene, 1 400 of 24 000) started late: the chip holds a quarter of the launch,
    // and the long waves of the second range used to start when the first range's short ones had gone (20 us in), the few of the
    // near ranges after 40 us.  An entry keeps the workgroup on its XCD (same tile index modulo 8), so locality and balance stay as
    // the tile map made them.
    // The order also MERGES ranges: consecutive ranges of a tile in which no wave made a single pass in the previous cast (free space: a
    // third of the launch's wave time went into setting their workgroups up) are given to one workgroup, [range, range_hi); any cut of
    // a ray's samples gives the same picture.  The slots this frees are empty (kNoSlot) and leave at once.
    uint32_t lin = blockIdx.y * gridDim.x + blockIdx.x;
    uint32_t range = ray_range_of_slot(bz_, nz, rp.range_order), range_hi = range + 1u;
    if (TAIL && tail.order) {
        const uint32_t e = tail.order[bz_ * (gridDim.x * gridDim.y) + lin];
        if (e == kNoSlot) return;
        lin = e & 0xffffu;
        range = (e >> 16) & 0xffu;
        range_hi = e >> 24;
    }
    const int k_lo = per_ray_ranges ? 0 : (int)(range * rp.seg_len);
    const int k_hi = (rp.seg_len && !per_ray_ranges) ? min(kMaxSamples, (int)(range_hi * rp.seg_len)) : kMaxSamples;
    // T[0], T[1] (the step) and the part of the table this range reads, T[k_lo .. k_hi], at Ts[2 ..]: 3.5 KB of LDS for a fifth of
    // the table (dynamic allocation, ray_table_lds_bytes) instead of 17.6 KB for all of it.  (Workgroup residency is not what
    // limits this kernel: 9 -> 16 workgroups per compute unit by LDS left the launch at 0.091 ms.)
    for (int i = (int)threadIdx.x; i <= k_hi - k_lo; i += 256) Ts[2 + i] = t_table[k_lo + i];
    if (threadIdx.x < 2) Ts[threadIdx.x] = t_table[threadIdx.x];
    __syncthreads();
    const int t_off = 2 - k_lo;
    auto T = [&](int k_) { return Ts[k_ + t_off]; };

    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // Workgroups are dealt to the 8 XCDs round robin in launch order.  Which tiles an XCD gets decides two things: how much of what
    // its rays read stays in ITS L2 (neighbouring tiles read neighbouring voxels) and how even the XCDs' shares of the work are (the
    // long waves sit where the surfaces and silhouettes are).  Every eighth tile (tile_map 0): even, no locality, bulk kernel 93.5 us on
    // the bench scene; one contiguous eighth of the image per XCD (1, up to round 2): local, uneven, 91 us; the image cut into 8 x 6
    // blocks of 5 x 5 tiles and each XCD given one block of every block row, in a different column each time (2, when the tile
    // counts divide that way -- 640 x 480 does): 87.5 us.
    uint32_t tile_x = blockIdx.x, tile_y = blockIdx.y;
    tile_y = lin / gridDim.x;
    tile_x = lin - tile_y * gridDim.x;
    if (rp.tile_map != 0u) {
        const uint32_t n_tiles = gridDim.x * gridDim.y;
        const uint32_t per_xcd = n_tiles / 8;
        if (lin < per_xcd * 8) {   // (the last n_tiles % 8 tiles keep their place)
            const uint32_t xcd = lin & 7u, j = lin >> 3;
            uint32_t remapped = xcd * per_xcd + j;
            if (rp.tile_map == 2u && gridDim.x % 8u == 0u && gridDim.y % 5u == 0u) {
                // blocks of (gridDim.x / 8) x 5 tiles, 8 across: XCD x takes one block of every block row, a different column each
                const uint32_t bw = gridDim.x / 8u, per_block = bw * 5u, jb = j / per_block, w = j - jb * per_block;
                const uint32_t col = (xcd + 3u * jb) & 7u, wy = w / bw, wx = w - wy * bw;
                remapped = (jb * 5u + wy) * gridDim.x + col * bw + wx;
            }
            tile_y = remapped / gridDim.x;
            tile_x = remapped - tile_y * gridDim.x;
        }
    }
    const int imx = tile_x * 16 + (wave & 1u) * 8 + (lane & 7u);
    const int imy = tile_y * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool in_image = imx < (int)rp.width && imy < (int)rp.height;

    static_assert(SEG || !(SLAB || TAIL), "slabs and the tail queue go with sample ranges");
    float ix = NAN, iy = NAN, iz = NAN;
    const float previous_tsdf = g.trunc;  // Q7
    const float step_size = Ts[1];         // = (float)((double)trunc * 0.05), :324

    RayState ray;
    int k_first, k_end;
    float near_t = 0.f;
    setup_ray<SLAB>(imx, imy, in_image, k_lo, k_hi, Ts, t_off, rp, g, step_size, ray, k_first, k_end, &near_t);
    if (per_ray_ranges && k_end > k_first) {
        // part blockIdx.z of rp.slab_ranges equal parts of this ray's stretch [k_first, k_end) through the slab
        const int len = k_end - k_first, a = k_first + (int)(((long long)len * range) / rp.slab_ranges);
        k_end = k_first + (int)(((long long)len * range_hi) / rp.slab_ranges);
        k_first = a;
    }
    const TriConst &tc = rp.tc;
    SkipCtx sc = make_skip_ctx(g, step_size);
    set_ray<SKIP>(sc, ray, step_size, g);

    const size_t idx = (size_t)imy * rp.width + imx;
    int k = (k_end <= k_first) ? kDone : k_first;  // next sample of this lane's ray (kDone when finished)
    if (SEG && range > 0 && k != kDone && load_best(&tail.best[idx]) <= (uint32_t)k_first) k = kDone;
    // Entry bound (EntryParams, common.hpp): no flagged brick holds a sample of this tile's rays whose camera depth -- near + T[k] for
    // the views the bound is made for -- is below the tile's word, so those samples are passed as the hops below would pass them, in one go:
    // the ray goes on with the first sample at or beyond the bound (or is done with this range).
    if (SKIP && !SLAB && !STATS && rp.ztile && k != kDone && sc.skip_ok && rp.ztile[rp.ztile_count] != 0u) {
        const float t_safe = (__uint_as_float(rp.ztile[tile_y * rp.ztile_pitch + tile_x]) - near_t) * 0.9999f;   // (the tile's word: uniform)
        // the last sample of [k, k_end) with T[ks] < t_safe: T[k] is k * step up to the rounding of its k additions (under a sample
        // over the whole table), so the walk down from the estimate is a few entries
        // (clamped BEFORE the + 2: a tile that sees no flagged unit has the word kEntryFar, the estimate saturates at INT_MAX, and
        // INT_MAX + 2 wrapped negative -- the tiles with nothing in view, of all, kept hopping through the whole volume)
        int ks = min(f2i_sat(t_safe * sc.inv_step), k_end - 3) + 2;
        while (ks >= k && !(T(ks) < t_safe)) ks--;
        if (ks >= k) k = ks + 1 >= k_end ? kDone : ks + 1;
    }
    BrickCache bc = {0, 0, false};
    SampleWork work = {0, 0, 0, 0};

    // Lead-in: most (tile, range) pairs start in empty space and many never leave it.  Jumping from block to block needs
    // only the brick look-up, so it gets a loop of its own -- a fraction of the instructions of the full pass below --
    // that runs until every lane has either finished or arrived in a flagged brick (its classification is kept in bc).
    // Entry: a ray that starts on a face of the grid spends its first samples in the outer half-voxel shell, where the
    // reference extrapolates (Q10); a sample there whose brick is flagged (its voxels are not flat, OccGrid) takes the reference's
    // full interpolation.  The rays of a tile do that together, so these samples get a loop of their own too -- a third of the
    // instructions of a full pass -- and after each of them the hops go on: behind a shell sample the next brick may be clear.
    if (SKIP && !STATS) {
        while (true) {
            while (true) {
                const bool hopping = k != kDone && sc.skip_ok && k >= bc.k_brick_end;
                if (__ballot(hopping) == 0ull) break;
                if (hopping) {
                    const float t = T(k);
                    const float fx = ((t * ray.dx) + ray.sx) * sc.inv_vx, fy = ((t * ray.dy) + ray.sy) * sc.inv_vy, fz = ((t * ray.dz) + ray.sz) * sc.inv_vz;
                    int n;
                    const bool empty = locate<SLAB>(fx, fy, fz, sc, g, occ, rp, n);
                    bc.k_brick_end = k + n;
                    if (empty) {
                        k += n;
                        if (k >= k_end) k = kDone;
                    }
                }
            }
            bool shell = false;
            float t = 0.f, px = 0.f, py = 0.f, pz = 0.f;
            if (k != kDone) {
                t = T(k);
                px = (t * ray.dx) + ray.sx; py = (t * ray.dy) + ray.sy; pz = (t * ray.dz) + ray.sz;
                // lower tap index of the sample's dual cell, as process_sample derives it: off the lattice on some axis?
                const int lx = (int)floorf(px * sc.inv_vx - 0.5f), ly = (int)floorf(py * sc.inv_vy - 0.5f), lz = (int)floorf(pz * sc.inv_vz - 0.5f);
                shell = !((uint32_t)lx < g.X - 1 && (uint32_t)ly < g.Y - 1 && (uint32_t)lz < g.Z - 1);
            }
            if (__ballot(shell) == 0ull) break;
            if (shell) {
                bool owned;
                const float tsdf = trilinear<SLAB, false, FASTDIV>(px, py, pz, dist, g, tc, rp, owned, nullptr);
                if (tsdf <= 0) {
                    if (SEG) lower_best(&tail.best[idx], k, tsdf);
                    else refine_hit(t, tsdf, previous_tsdf, step_size, ray, rp, ix, iy, iz);
                    k = kDone;
                } else {
                    k += 1;
                    if (k >= k_end) k = kDone;
                }
            }
        }
    }

    uint32_t dbg_trips = 0;
    for (uint32_t trip = 0; __ballot(k != kDone) != 0ull; trip++) {
        if (TAIL && trip >= tail.trip_budget) break;
        dbg_trips = trip + 1;
        if (STATS) work.trips++;
#ifdef TSDF_DIAG_RAY_MIX
        {
            const int act = __popcll(__ballot(k != kDone)), bucket = act <= 4 ? 0 : act <= 16 ? 1 : act <= 32 ? 2 : 3;
            if ((threadIdx.x & 63u) == 0) {
                RAY_MIX(24 + bucket);
                if (trip >= 12) RAY_MIX(28 + bucket);
            }
        }
#endif
        if (k != kDone) {
            const float t = T(k);
            int jump, ahead;
            const float tsdf = process_sample<SLAB, STATS, FASTDIV>(t, k, ray, sc, bc, dist, g, tc, rp, occ, touched, work, jump, ahead);
            if (jump > 0) {
                k += jump;
            } else if (tsdf <= 0) {
                if (SEG) lower_best(&tail.best[idx], k, tsdf);
                else refine_hit(t, tsdf, previous_tsdf, step_size, ray, rp, ix, iy, iz);
                k = kDone;
            } else {
                // positive (or NaN) sample: the reference steps on; `previous_tsdf < 0` never holds (Q7).  The samples the
                // look-ahead proved positive are passed with it (ahead is 0 for a NaN).
                k += 1 + (tsdf > 0 ? ahead : 0);
            }
            if (k != kDone && k >= k_end) k = kDone;
        }
    }

    if (TAIL) {
        // (for the next cast's dispatch order: this tile holds a long wave in this range)
        // ([0]: a pass at all -- the range is not free space for this tile; [1]: a long wave.  A merged workgroup speaks for all its ranges)
        if (tail.heavy && dbg_trips >= 1u && lane < range_hi - range) {
            const uint32_t n_tiles = gridDim.x * gridDim.y, at = (range + lane) * n_tiles + lin;
            tail.heavy[at] = 1;
            if (dbg_trips >= tail.heavy_passes) tail.heavy[nz * n_tiles + at] = 1;
        }
        // hand over what is left of the unfinished rays, in pieces (one atomic per wave); a ray whose hit is already known
        // to lie at or before its next sample is dropped
        int len = 0;
        uint32_t n_sub = 0;
        if (k != kDone && load_best(&tail.best[idx]) > (uint32_t)k) {
            len = k_end - k;
            n_sub = (uint32_t)min(kTailPieces, (len + tail.piece_min - 1) / tail.piece_min);
        }
        if (__ballot(n_sub != 0) != 0ull) {
            uint32_t incl = n_sub;   // inclusive prefix sum over the wave
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(incl, o);
                if ((int)lane >= o) incl += up;
            }
            const uint32_t total = __shfl(incl, 63);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&tail.count[0], total);
            base = __shfl(base, 0) + incl - n_sub;
            for (uint32_t s_ = 0; s_ < n_sub; s_++) {
                const int a = k + (int)((uint32_t)len * s_ / n_sub), b = k + (int)((uint32_t)len * (s_ + 1) / n_sub);
                const uint2 entry = make_uint2((uint32_t)idx, ((uint32_t)b << 13) | (uint32_t)a);
                tail.entries[base + s_] = entry;
            }
        }
    }
    if (!SEG && in_image) {
        if (STATS && SKIP) {  // diagnostics: per-ray work instead of the vertex
            out[idx * 3 + 0] = (float)work.samples;
            out[idx * 3 + 1] = (float)work.hops;
            out[idx * 3 + 2] = (float)work.cell_tests;
        } else {
            out[idx * 3 + 0] = ix;
            out[idx * 3 + 1] = iy;
            out[idx * 3 + 2] = iz;
        }
    }
    if (STATS) {
        uint32_t h = (in_image && ix == ix) ? 1u : 0u;
        uint32_t samples = work.samples;
        for (int o = 32; o > 0; o >>= 1) {
            samples += __shfl_down(samples, o);
            h += __shfl_down(h, o);
        }
        if (lane == 0) {
            atomicAdd(&counters[1], (unsigned long long)samples);
            atomicAdd(&counters[2], (unsigned long long)h);
        }
    }
}

template <bool SLAB, bool STATS, bool SKIP, bool FASTDIV, bool SEG, bool TAIL>
__global__ __launch_bounds__(256) void process_ray_kernel(const float *__restrict__ dist, const Geom g,
                                                          const RayParams rp, float *__restrict__ out,
                                                          unsigned long long *__restrict__ counters,
                                                          unsigned int *__restrict__ touched,
                                                          const OccGrid occ, const float *__restrict__ t_table,
                                                          const TailQueue tail) {
    extern __shared__ float Ts[];
    march_bulk<SLAB, STATS, SKIP, FASTDIV, SEG, TAIL>(dist, g, rp, out, counters, touched, occ, t_table, tail, Ts, gridDim.z);
}

// The dispatch order of the next cast's first kernel (TailQueue::order), from the marks this cast's waves left: heavy[range][tile
// slot] = some wave made a pass there, and behind those n_ranges * n_tiles bytes the same for long waves.  Workgroup i of a launch
// runs on XCD i % 8, so each XCD orders its own tile slots (8 j + xcd) and they stay on it.  Per tile the ranges become entries: one
// per range that saw a pass, one per RUN of consecutive ranges that saw none (free space: one workgroup marches the run).  Per XCD
// the entries with a long wave come first, then those with passes, then the runs of free space; what is left of the XCD's slots is
// empty.  One wave per XCD; the marks are cleared for the next cast.  Scheduling only: any order and any cut give the same picture.
struct OrderJob {
    uint8_t *heavy;
    uint32_t *order;
    uint32_t n_tiles, n_ranges, range_order;   // n_ranges == 0: nothing to do
};
constexpr uint32_t kOrderWorkgroups = 2;   // workgroups of 4 waves appended to the tail kernel's launch for the 8 XCDs
constexpr uint32_t kOrderMaxRanges = 16;   // (a range index in 8 bits of an entry, a mask of ranges in 32 bits: more ranges, no learnt order)
__device__ inline void order_ray_tiles(uint32_t xcd, const OrderJob &job) {
    if (xcd >= 8u || job.n_ranges == 0u) return;
    const uint32_t lane = threadIdx.x & 63u, n_tiles = job.n_tiles, n = job.n_ranges, per_xcd = n_tiles / 8;
    const bool descending = job.range_order != 0u;   // the far ranges first within a class, as the launch dispatches them
    const uint8_t *any_ = job.heavy, *long_ = job.heavy + (size_t)n * n_tiles;
    uint32_t next = 0;   // slots of this XCD filled so far: slot i is workgroup 8 (i % per_xcd) + xcd of slab i / per_xcd of the launch
    auto slot_address = [&](uint32_t i) { return (size_t)(i / per_xcd) * n_tiles + 8u * (i % per_xcd) + xcd; };
    for (int cls = 2; cls >= 0; cls--)
        for (uint32_t j0 = 0; j0 < per_xcd; j0 += 64) {
            const uint32_t j = j0 + lane, tile = 8u * j + xcd;
            uint32_t any_m = 0, long_m = 0;
            if (j < per_xcd)
                for (uint32_t r = 0; r < n; r++) {
                    const uint8_t a_ = any_[(size_t)r * n_tiles + tile], l_ = long_[(size_t)r * n_tiles + tile];
                    any_m |= (a_ ? 1u : 0u) << r;
                    long_m |= (l_ ? 1u : 0u) << r;
                }
            // this tile's entries of the class, in dispatch order: twice the same walk, first to count, then to write
            auto walk = [&](uint32_t at, bool write) {
                uint32_t count = 0;
                if (j < per_xcd)
                    for (uint32_t i = 0; i < n;) {
                        const uint32_t r = descending ? n - 1u - i : i;
                        const int c = ((long_m >> r) & 1u) ? 2 : ((any_m >> r) & 1u) ? 1 : 0;
                        uint32_t len = 1;
                        if (c == 0)   // the run of free space that starts here
                            while (i + len < n && !((any_m >> (descending ? n - 1u - (i + len) : i + len)) & 1u)) len++;
                        if (c == cls) {
                            const uint32_t lo = descending ? r + 1u - len : r, hi = lo + len;
                            if (write) job.order[slot_address(at + count)] = (hi << 24) | (lo << 16) | tile;
                            count++;
                        }
                        i += len;
                    }
                return count;
            };
            const uint32_t mine = walk(0, false);
            uint32_t incl = mine;   // inclusive prefix sum over the wave
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(incl, o);
                if ((int)lane >= o) incl += up;
            }
            (void)walk(next + incl - mine, true);
            next += __shfl(incl, 63);
        }
    for (uint32_t i = next + lane; i < n * per_xcd; i += 64) job.order[slot_address(i)] = kNoSlot;
    for (uint32_t r = 0; r < n; r++) {
        for (uint32_t j = lane; j < per_xcd; j += 64) {
            job.heavy[(size_t)r * n_tiles + 8 * j + xcd] = 0;
            job.heavy[(size_t)(n + r) * n_tiles + 8 * j + xcd] = 0;
        }
        if (xcd == 0)   // the n_tiles % 8 workgroups past the last full round of a slab keep their place and their one range
            for (uint32_t i = per_xcd * 8 + lane; i < n_tiles; i += 64) {
                const uint32_t range = ray_range_of_slot(r, n, job.range_order);
                job.order[(size_t)r * n_tiles + i] = ((range + 1u) << 24) | (range << 16) | i;
                job.heavy[(size_t)r * n_tiles + i] = 0;
                job.heavy[(size_t)(n + r) * n_tiles + i] = 0;
            }
    }
}

// The stretches of rays process_ray_kernel did not finish.  lanes_per_ray lanes per queue entry: the group's lanes take
// the next samples k .. k+15 of the entry's ray, each classifying / evaluating its own (process_sample without the
// per-brick memory).  The first lane of the group with a value <= 0 has the stretch's first hit -- everything before it
// was evaluated positive or proven positive -- and lowers best[pixel] to it; otherwise the group advances past everything
// it has dealt with, and gives up once best[pixel] shows a hit at or before its next sample (the pieces of one ray are
// worked on side by side; when an early one hits, the later ones stop).  Sample k of a ray is computed by the same
// expressions whichever lane does it, so the result does not depend on the schedule.  Groups take queue entries round
// robin until none is left (persistent workgroups); every pass of the loop is uniform across the wave.
//   LANES: lanes per queue entry fixed at compile time (the group reductions become DPP operations), 0 = tail.lanes.
template <bool SLAB, bool FASTDIV, int LANES>
__device__ inline void march_tail(const float *__restrict__ dist, const Geom &g, const RayParams &rp, const OccGrid &occ,
                                  const float *__restrict__ t_table, const TailQueue &tail, float *T, const uint32_t block,
                                  const uint32_t n_blocks) {
    const uint32_t lanes_per_ray = LANES ? (uint32_t)LANES : tail.lanes;
    const uint32_t n_entries = tail.count[0];
    if ((size_t)block * 4u * (64u / lanes_per_ray) >= n_entries) return;   // nothing for this workgroup: skip the staging too
    for (int i = (int)threadIdx.x; i < kTableLen; i += 256) T[i] = t_table[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const int j = (int)(lane & (lanes_per_ray - 1));
    const float step_size = T[1];
    const TriConst &tc = rp.tc;
    SkipCtx sc = make_skip_ctx(g, step_size);
    // A wave takes as many consecutive queue entries as it has groups (pieces of one ray, or of rays of one tile and one
    // sample range: alike in length), works on them until all are finished, then takes the next batch: waves round robin.
    const uint32_t groups_per_wave = 64 / lanes_per_ray;
    const uint32_t n_waves = n_blocks * 4, wave_id = block * 4 + (threadIdx.x >> 6);
    uint32_t dbg_rounds = 0;   // rounds of this wave so far
    for (uint32_t batch = wave_id * groups_per_wave;; batch += n_waves * groups_per_wave) {
        const uint32_t e = batch + (lane / lanes_per_ray);
        if (batch >= n_entries) break;
        RayState ray = {0, 0, 0, 0, 0, 0};
        int k = kDone, k_end = 0;   // the group's stretch (all its lanes hold the same values); kDone: none
        uint64_t *best = tail.best;
        if (e < n_entries) {
            const uint2 q = tail.entries[e];
            float max_t;
            (void)ray_geometry((int)(q.x % rp.width), (int)(q.x / rp.width), true, rp, ray, max_t);
            set_ray<true>(sc, ray, step_size, g);
            k = (int)(q.y & 0x1fffu);
            k_end = (int)((q.y >> 13) & 0x1fffu);
            best += q.x;
        }
        while (true) {
            if (__ballot(k != kDone) == 0ull) break;
            dbg_rounds++;
            // Lane j of the group takes sample k + j.  (Lanes spaced by the cells the ray crosses instead -- a round then deals with up to
            // `lanes` cells -- gave the same picture and a slower kernel, 0.060 -> 0.087 ms: tools/experiments/raycast_march_fused_and_tail_chain.patch.txt.)
            const int kk = k + j;
            int end_ = kk;   // samples in [kk, end_) are dealt with by this lane (a hit: sample kk itself)
            bool hit = false;
            float hit_value = 0.0f;
            uint32_t known = kNoHit;
            // the pixel's word is read past this XCD's L2 (other pieces of the ray run on other XCDs): every fourth round -- a hit found
            // elsewhere is then noticed at most three rounds late, which only costs those rounds -- instead of every round (61 -> 58 us)
            if (k != kDone && j == 0 && (dbg_rounds & 3u) == 1u) known = load_best(best);   // in flight together with the sample's loads
            if (k != kDone && kk < k_end) {
                const float t = T[kk];
                int jump, ahead;
                const float tsdf = process_sample_eager<SLAB, FASTDIV>(t, ray, sc, dist, g, tc, rp, occ, jump, ahead, j == 0);
                if (jump > 0) {
                    end_ = kk + jump;
                } else if (tsdf <= 0) {
                    hit = true;
                    hit_value = tsdf;
                } else {
                    end_ = kk + 1 + (tsdf > 0 ? ahead : 0);
                }
            }
            // over the group, in lane order: `cover` = every sample before it is dealt with.  A lane joins while its start is not
            // beyond the cover; the first one that joins with a hit has the stretch's first sample <= 0.
            int cover = k, hit_lane = -1;
            bool open = true;
            for (int q = 0; q < (int)lanes_per_ray; q++) {
                const int kq = group_lane<LANES>(kk, q, lanes_per_ray), eq = group_lane<LANES>(end_, q, lanes_per_ray);
                const bool hq = group_lane<LANES>((int)hit, q, lanes_per_ray) != 0;
                open = open && kq <= cover && kq < k_end;
                if (open) {
                    if (hq) {
                        hit_lane = q;
                        open = false;
                    } else {
                        cover = max(cover, eq);
                    }
                }
            }
            for (int o = 1; o < (int)lanes_per_ray; o <<= 1) known = min(known, (uint32_t)__shfl_xor((int)known, o));
            if (k != kDone) {
                if (hit_lane >= 0) {
                    if (j == hit_lane) lower_best(best, kk, hit_value);
                    k = kDone;
                } else {
                    k = cover;
                    if (k >= k_end || known <= (uint32_t)k) k = kDone;
                }
            }
        }
    }
}

template <bool SLAB, bool FASTDIV, int LANES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void process_ray_tail_kernel(const float *__restrict__ dist, const Geom g, const RayParams rp,
                                                               const OccGrid occ, const float *__restrict__ t_table,
                                                               const TailQueue tail, const OrderJob order_job) {
    __shared__ float T[kTableLen];
    if (blockIdx.x < kOrderWorkgroups) {   // the first workgroups: the next cast's dispatch order, one wave per XCD (beside the march, from its start)
        if (order_job.n_ranges) order_ray_tiles(blockIdx.x * 4u + (threadIdx.x >> 6), order_job);
        return;
    }
    march_tail<SLAB, FASTDIV, LANES>(dist, g, rp, occ, t_table, tail, T, blockIdx.x - kOrderWorkgroups, gridDim.x - kOrderWorkgroups);
}
