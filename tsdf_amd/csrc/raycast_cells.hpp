// The cell-parallel cast (round 5): the reference's first sample <= 0 of every ray WITHOUT marching the rays.
// (Included by raycast.hip inside namespace tsdf, behind the helpers it uses: ray_geometry / setup_ray, trilinear, the evaluation
// expressions of process_sample, lipschitz_lookahead, hit_word / lower_best.)
//
// process_ray (src/RayCaster/GPURaycaster.cu:265-377) stops at the first sample whose interpolated value is <= 0.  The march kernels
// above find it ray by ray, each ray a chain of dependent look-ups whose length is the number of cells it crosses near surfaces; a wave
// is as slow as its slowest lane and the launches as their slowest waves (22 passes of 2.5 us, then the queue: 0.12 ms for 1.07 M
// evaluated samples).  But WHICH samples can be <= 0 is a property of the volume, not of the ray:
//   * a sample strictly inside the lattice of voxel centres interpolates the 8 voxels of its dual cell with weights in [0, 1]: it can
//     only be <= 0 when one of the 8 is not safely positive (kCellPositive, process_sample) -- a MIXED cell.  Mixed cells lie in
//     bricks whose `cell` flag is set (OccGrid: a superset at all times), and those in bricks whose `fine` flag is set too;
//   * a sample in the outer half-voxel shell of the grid (where the reference extrapolates, Q10) or off the grid by rounding can only
//     be <= 0 when the boundary brick of its voxel is flagged (`fine`: otherwise every voxel in reach is flat, OccGrid): the
//     first workgroups of the launch take those bricks (cast_shell_bricks);
//   * a sample FAR off the grid exists only on rays for which the reference's ray_box loses an exit (a direction component of exactly 0,
//     the camera inside): those rays are walked where the ray records are formed (cell_cast_prepare_kernel).
// So the work is turned round: one wave per flagged brick.  It loads the brick's 5^3 voxels once, finds its mixed cells, projects each
// (grown by the guard band eps) into the image -- the view's projection is only used to bound the PIXELS worth looking at -- and for
// every such pixel intersects that pixel's ray (the same start point and direction as ever: cell_cast_prepare_kernel runs setup_ray) with
// the cell, walks the few samples inside -- the reference's expressions for the value, the look-ahead of process_sample to pass the
// ones proven positive -- and lowers the pixel's word to the first one <= 0.  A sample within eps of a cell face, where the cheap
// arithmetic and the reference's may disagree about the cell, takes the reference's full trilinearly_interpolate, which picks its own
// cell: it is evaluated by the task of every mixed cell it could belong to, and two tasks that evaluate the same sample compute the
// same value.  The minimum over all tasks (atomicMin on {k, value}) is the sample the reference's loop stops at: every sample <= 0 is
// found by some task, and every sample evaluated is one the reference would evaluate with the same result.  resolve_*_kernel is unchanged.
// No ray is marched through free space, no wave waits for a long ray: the cast is the number of (mixed cell, pixel) pairs, a few per ray.
// Needs a view whose projection exists (camera depth == ray parameter: view_projection); a mixed cell that reaches across the camera
// plane is bounded from the side it lies on, one that holds the camera is offered to every pixel (project_box).  Volumes whose flagged
// bricks are so many that marching is cheaper (arbitrary fields: every cell mixed), views from inside the volume (surface behind
// surface: every mixed cell is looked at, hidden or not) and voxels of more than ten pixels keep the march kernels unless
// TSDF_RAY_CELLS=2 (choose_cell_cast; scheduling only -- both give the same bits).

struct RayRecord {      // 8 bytes per pixel: what of setup_ray's result cannot be formed again in a dozen instructions
    float near_t;       // the ray's start point is origin + near_t * direction (ray_from_near); the direction follows from the pixel (ray_direction)
    uint32_t k_range;   // first sample | one past the last << 16 (k_end <= 4402)
};
static_assert(sizeof(RayRecord) == 8, "one 8-byte load");
// (32-byte records with direction and start point spelt out were 9.8 MB read in no order at all; the direction is a dozen instructions from the pixel)

struct CellCast {
    RayRecord *rays;        // width * height
    uint2 *bricks;          // the cast's tasks: {kCellTasks | kShellTasks | part << 10 | parts - 1, brick coordinates 3 x 10 bits}
    uint32_t *n_bricks;     // entries appended: TailQueue::count[3], reset by the resolve kernel of the previous cast
    uint32_t *n_bricks_host;  // pinned mirror of the count (the next cast's choice of kernels), may be null
    uint32_t pairs_per_task;  // a brick whose (cell, pixel) pairs are estimated above this is listed in several parts (TSDF_RAY_CELLS_PAIRS)
    // front-to-back order of the list (views from inside the volume, round 6): depth_hist != null -- kDepthBins counters the list's builder
    // fills, kDepthBins more that cell_list_sort_kernel takes its places from -- and the list as it is built (`unsorted`), which that
    // kernel scatters into `bricks` by the bin of each brick's camera depth, (depth - depth0) * depth_scale
    uint32_t *depth_hist;
    uint2 *unsorted;
    float depth0, depth_scale;
    const float *dist;        // the distances and the pixels' words: for the rays whose samples leave the grid (cell_cast_prepare_kernel)
    uint64_t *best;
    uint32_t *release_word;   // null, or where cell_cast_prepare_kernel stores release_value as it starts: everything in front of it on the
    uint32_t release_value;   // stream (this frame's integrate) is done -- the pipeline's second stream waits for that word, not for an event
};
constexpr uint32_t kCellTasks = 1u << 31, kShellTasks = 1u << 30;
constexpr uint32_t kDepthBins = 256;
// the bin of a listed brick: by the camera depth of its centre (the same expression where the list is built and where it is sorted)
__device__ inline uint32_t depth_bin(const CellCast &cc, const EntryParams &ep, const Geom &g, uint32_t coords) {
    const float mid = 0.5f * ((float)kBrick + 1.5f);
    const float wx = ((float)((coords & 1023u) * kBrick) + mid) * g.vs.x + ep.offset.x, wy = ((float)(((coords >> 10) & 1023u) * kBrick) + mid) * g.vs.y + ep.offset.y,
                wz = ((float)((coords >> 20) * kBrick) + mid) * g.vs.z + ep.offset.z;
    const float d = ep.r[2][0] * wx + ep.r[2][1] * wy + ep.r[2][2] * wz + ep.r[2][3];
    const float b = (d - cc.depth0) * cc.depth_scale;
    return b > 0.0f ? (b < (float)(kDepthBins - 1u) ? (uint32_t)b : kDepthBins - 1u) : 0u;   // (anything not a number: the first bin)
}
// A brick seen from close by -- a camera inside the volume, a coarse grid -- is thousands of pairs, one wave's work for a long time while
// the chip idles: it is listed in up to kMaxParts parts, each a wave's task (the wave loads the brick, finds its cells and boxes as
// ever, and takes its share of the packed pairs).  The list has room for kPartsRoom extra entries per 1 024 bricks scanned (a workgroup's turn); a
// workgroup whose bricks ask for more scales all of them down.
constexpr uint32_t kMaxParts = 1024, kPartsRoom = 1024;
__host__ __device__ inline size_t cell_list_capacity(size_t n_bricks) { return n_bricks + (size_t)kPartsRoom * ((n_bricks + 1023) / 1024); }

// pixel box of the axis-aligned box [lo, hi] (grid millimetres) under the view's projection: false = no pixel can see it.
// A sample's camera depth is its ray parameter t >= z_clip >= 0 (the last row of kinv is (0, 0, 1): view_projection; z_clip > 0 when
// the camera is outside the volume, choose_cell_cast): only the part of the box with depth D >= z_clip can hold samples.
//   * wholly behind that plane: none;
//   * wholly in front of z_near > 0: the integers inside the hull of its 8 projected corners;
//   * otherwise (the box straddles the plane, or comes within a quarter voxel of the camera plane): u = N(P) / D(P), N and D affine in
//     P, so for any u*, u - u* = F(P) / D(P) with F = N - u* D affine too: over the box F lies in [F_c - h, F_c + h] (its value at the
//     centre -+ half the sum of its coefficients along the box's edges) and D in (max(z_min, z_clip), z_max].  F_c - h > 0: the box is
//     to the right of u* for every sample in it, by at least (F_c - h) / z_max; F_c + h < 0: to the left; the far side is bounded by
//     the nearest depth when that is positive, not at all otherwise (a box that holds the camera asks every pixel).  u* = the image's
//     centre: a box beside or behind the camera that reaches across the camera plane lies far off the image and is dropped here.
struct PixelBox {
    int u0, v0, w, h;
};
template <bool HULL = true>   // (false: the bound from the centre only -- wider in front of z_near, where the cells' kernel has its own)
__device__ inline bool project_box(const EntryParams &ep, float lox, float loy, float loz, float hix, float hiy, float hiz, PixelBox &pb) {
    const float mx = 0.5f * (lox + hix) + ep.offset.x, my = 0.5f * (loy + hiy) + ep.offset.y, mz = 0.5f * (loz + hiz) + ep.offset.z;
    const float ccx = ep.r[0][0] * mx + ep.r[0][1] * my + ep.r[0][2] * mz + ep.r[0][3];
    const float ccy = ep.r[1][0] * mx + ep.r[1][1] * my + ep.r[1][2] * mz + ep.r[1][3];
    const float ccz = ep.r[2][0] * mx + ep.r[2][1] * my + ep.r[2][2] * mz + ep.r[2][3];
    const float ext[3] = {hix - lox, hiy - loy, hiz - loz};
    const float hz = 0.5f * ((fabsf(ep.r[2][0] * ext[0]) + fabsf(ep.r[2][1] * ext[1])) + fabsf(ep.r[2][2] * ext[2]));
    // (the centre's coordinates are good to a few ulps of the largest term: 1e-5 of the camera's distance covers it with room)
    const float zslack = 1.0e-5f * (((fabsf(ep.r[2][0] * mx) + fabsf(ep.r[2][1] * my)) + fabsf(ep.r[2][2] * mz)) + fabsf(ep.r[2][3])) + 1.0e-5f * hz;
    const float zmin = ccz - hz - zslack, zmax = ccz + hz + zslack;
    if (!(ccx == ccx && ccy == ccy && zmin == zmin && zmax == zmax) || fabsf(ccx) == INFINITY || fabsf(ccy) == INFINITY || zmax == INFINITY || zmin == -INFINITY) {
        pb.u0 = 0; pb.v0 = 0; pb.w = (int)ep.width; pb.h = (int)ep.height;   // (something not finite: every pixel is asked)
        return true;
    }
    if (zmax < ep.z_clip) return false;
    float a0, a1, b0, b1;
    const float kMargin = 0.05f;   // (the projection is the double-precision inverse of the matrices the rays are formed with, evaluated in fp32: 1e-2 px at most)
    if (HULL && zmin > ep.z_near) {
        float umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const float wx = ((c & 1) ? hix : lox) + ep.offset.x, wy = ((c & 2) ? hiy : loy) + ep.offset.y, wz = ((c & 4) ? hiz : loz) + ep.offset.z;
            const float cx = ep.r[0][0] * wx + ep.r[0][1] * wy + ep.r[0][2] * wz + ep.r[0][3];
            const float cy = ep.r[1][0] * wx + ep.r[1][1] * wy + ep.r[1][2] * wz + ep.r[1][3];
            const float cz = fmaxf(ep.r[2][0] * wx + ep.r[2][1] * wy + ep.r[2][2] * wz + ep.r[2][3], 0.5f * ep.z_near);
            const float rz = __builtin_amdgcn_rcpf(cz);   // (a bound, not a result: the margin below covers the last bits)
            const float u = (ep.k[0][0] * cx + ep.k[0][1] * cy + ep.k[0][2] * cz) * rz, w = (ep.k[1][0] * cx + ep.k[1][1] * cy + ep.k[1][2] * cz) * rz;
            umin = fminf(umin, u); umax = fmaxf(umax, u);
            vmin = fminf(vmin, w); vmax = fmaxf(vmax, w);
        }
        // a ray exists per INTEGER pixel: the integers inside the hull's box
        a0 = ceilf(umin - kMargin); a1 = floorf(umax + kMargin); b0 = ceilf(vmin - kMargin); b1 = floorf(vmax + kMargin);
    } else {
        const float us = 0.5f * (float)ep.width, vs_ = 0.5f * (float)ep.height;
        const float dlo = fmaxf(zmin, ep.z_clip), rhi = 1.0f / (zmax * 1.0001f), rlo = dlo > 0.0f ? 1.0001f / dlo : INFINITY;
        auto side = [&](const float *kr, float centre, float &lo_, float &hi_) {
            const float t0 = kr[0] * ccx, t1 = kr[1] * ccy, t2 = kr[2] * ccz, t3 = centre * ccz;
            const float fc = ((t0 + t1) + t2) - t3;
            float h = 0.0f;
#pragma unroll
            for (int a_ = 0; a_ < 3; a_++) {
                const float ex_ = ep.r[0][a_] * ext[a_], ey_ = ep.r[1][a_] * ext[a_], ez_ = ep.r[2][a_] * ext[a_];
                h += 0.5f * fabsf(((kr[0] * ex_ + kr[1] * ey_) + kr[2] * ez_) - centre * ez_);
            }
            const float slack = 1.0e-5f * ((((fabsf(t0) + fabsf(t1)) + fabsf(t2)) + fabsf(t3)) + h) + (fabsf(kr[0]) + fabsf(kr[1]) + fabsf(kr[2]) + centre) * zslack;
            const float fmin_ = fc - h - slack, fmax_ = fc + h + slack;
            lo_ = fmin_ > 0.0f ? fmin_ * rhi : (dlo > 0.0f ? fmin_ * rlo : -INFINITY);
            hi_ = fmax_ < 0.0f ? fmax_ * rhi : (dlo > 0.0f ? fmax_ * rlo : INFINITY);
        };
        float ul, uh, vl, vh;
        side(ep.k[0], us, ul, uh);
        side(ep.k[1], vs_, vl, vh);
        a0 = ceilf(us + ul - kMargin); a1 = floorf(us + uh + kMargin); b0 = ceilf(vs_ + vl - kMargin); b1 = floorf(vs_ + vh + kMargin);
    }
    const float wmax = (float)(ep.width - 1u), hmax = (float)(ep.height - 1u);
    if (!(a0 <= a1 && b0 <= b1) || a1 < 0.0f || b1 < 0.0f || a0 > wmax || b0 > hmax) return false;
    pb.u0 = (int)fmaxf(a0, 0.0f); pb.v0 = (int)fmaxf(b0, 0.0f);
    pb.w = (int)fminf(a1, wmax) - pb.u0 + 1; pb.h = (int)fminf(b1, hmax) - pb.v0 + 1;
    return true;
}
// inclusive scans over a wave's 64 lanes with data-parallel-primitive operands: four shifts inside a row of 16, then lane 15 of a
// row into the row behind it and lane 31 into the upper half (row_bcast: a GFX9 form).  A lane without a source keeps the identity 0.
template <int CTRL, int ROW_MASK>
__device__ inline uint32_t dpp_or_zero(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ inline uint32_t wave_scan_add(uint32_t v) {
    v += dpp_or_zero<0x111, 0xf>(v);
    v += dpp_or_zero<0x112, 0xf>(v);
    v += dpp_or_zero<0x114, 0xf>(v);
    v += dpp_or_zero<0x118, 0xf>(v);
    v += dpp_or_zero<0x142, 0xa>(v);
    v += dpp_or_zero<0x143, 0xc>(v);
    return v;
}
__device__ inline uint32_t wave_scan_max(uint32_t v) {
    v = max(v, dpp_or_zero<0x111, 0xf>(v));
    v = max(v, dpp_or_zero<0x112, 0xf>(v));
    v = max(v, dpp_or_zero<0x114, 0xf>(v));
    v = max(v, dpp_or_zero<0x118, 0xf>(v));
    v = max(v, dpp_or_zero<0x142, 0xa>(v));
    v = max(v, dpp_or_zero<0x143, 0xc>(v));
    return v;
}
// One launch in front of the cast, two kinds of workgroup.
// All but the first n_list_blocks: per pixel the direction, start point and sample range of its ray, exactly as the march kernels set a ray up
// (setup_ray with the whole table: samples [k_first, k_end) are the ones the reference evaluates unless it stops earlier; a slab's range
// is clipped to a superset of its own stretch).
// The first n_list_blocks (their chain of flag loads, prefix sums and one atomic is the launch's longest: they start first): the bricks the cast has to look at -- `cell` and `fine` set (mixed cells), or a brick touching the grid boundary with
// `fine` set (shell samples); a slab lists the bricks that hold a cell whose lower plane it owns.  Four bricks a thread (one word of
// each flag array), the brick's coordinates only for a flagged one, and ONE atomic per workgroup of 1 024 bricks: returning atomics
// on one address are a round trip each, one after the other -- one per wave and word made this 37 us for 20 000 listed bricks, one per
// 256 bricks (a brick a thread) 17 us against 10.  The counter (TailQueue::count[3]) is reset by the resolve kernel of the previous
// cast.  With cc.pairs_per_task != 0 (the default) every flagged brick is projected: one that no pixel sees is dropped, a large one is
// listed in parts (+ 0.3 us: these workgroups start first and end with the ray records').
template <bool SLAB>
__global__ __launch_bounds__(256) void cell_cast_prepare_kernel(const Geom g, const RayParams rp, const EntryParams ep, const float *__restrict__ t_table,
                                                                const OccGrid occ, const CellCast cc, const uint32_t n_list_blocks) {
    extern __shared__ float Ts[];   // T[0 .. kMaxSamples] (ray workgroups); the list workgroups use its first words
    if (cc.release_word && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(cc.release_word, cc.release_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (blockIdx.x >= n_list_blocks) {
        for (int i = (int)threadIdx.x; i <= kMaxSamples; i += 256) Ts[i] = t_table[i];
        __syncthreads();
        const uint32_t i = (blockIdx.x - n_list_blocks) * 256 + threadIdx.x;
        if (i >= rp.width * rp.height) return;
        RayState ray;
        int k_first, k_end;
        float near_t = 0.f;
        setup_ray<SLAB>((int)(i % rp.width), (int)(i / rp.width), true, 0, kMaxSamples, Ts, 0, rp, g, Ts[1], ray, k_first, k_end, &near_t);
        if (k_end <= k_first) k_first = k_end = 0;
        cc.rays[i] = {near_t, (uint32_t)k_first | ((uint32_t)k_end << 16)};
        // Samples far off the grid.  The cells' and the shell's tasks cover the grid grown by eps; the reference's range can reach beyond
        // it: for a ray that starts inside the volume ray_box (GPURaycaster.cu:197-251) takes the smallest of three exit parameters with
        // `<` on values that are NaN for a zero direction component -- d.y == 0 leaves the z exit alone, d.z == 0 leaves NaN and the march
        // goes on to sample 4402 -- and every sample out there interpolates the clamped boundary voxels with unclamped weights, often to
        // something <= 0.  Such rays (a row or a column of pixels of a camera that looks along an axis plane) are walked here, sample by
        // sample from where they really leave the grown grid, with the reference's interpolation; their word takes the minimum like
        // everyone's.  An ordinary ray's last sample lies in front of that exit and costs one comparison.
        if (k_end > k_first) {
            const float e = make_skip_ctx(g, Ts[1]).eps;
            float tout = INFINITY;
            bool never = false;
            auto axis = [&](float s_, float d_, float lo, float hi) {
                if (d_ != 0.0f) tout = fminf(tout, fmaxf((lo - s_) / d_, (hi - s_) / d_));
                else if (s_ < lo || s_ > hi) never = true;
            };
            axis(ray.sx, ray.dx, -e * g.vs.x, ((float)g.X + e) * g.vs.x);
            axis(ray.sy, ray.dy, -e * g.vs.y, ((float)g.Y + e) * g.vs.y);
            axis(ray.sz, ray.dz, -e * g.vs.z, ((float)g.Z + e) * g.vs.z);
            if (never) tout = -INFINITY;
            if (!(Ts[k_end - 1] <= tout)) {   // (also for anything not a number)
                const float step = Ts[1], kf = floorf(tout / step) - 2.0f;
                int k = kf > (float)k_first ? (kf < 8192.0f ? (int)kf : 8192) : k_first;
                uint64_t *word = cc.best + i;
                for (; k < k_end; k++) {
                    const float t = Ts[k];
                    const float px = (t * ray.dx) + ray.sx, py = (t * ray.dy) + ray.sy, pz = (t * ray.dz) + ray.sz;
                    const float fx = px / g.vs.x, fy = py / g.vs.y, fz = pz / g.vs.z;
                    if (fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && fx <= (float)g.X && fy <= (float)g.Y && fz <= (float)g.Z) continue;   // on the grid: the cells' / the shell's
                    bool owned;
                    const float tsdf = trilinear<SLAB, false, false>(px, py, pz, cc.dist, g, rp.tc, rp, owned, nullptr);
                    if (tsdf <= 0) {
                        lower_best(word, k, tsdf);
                        break;
                    }
                }
            }
        }
        return;
    }
    uint32_t *wave_count = reinterpret_cast<uint32_t *>(Ts);   // [0..3] the waves' entries, [4] the workgroup's base in the list, [8..11] the waves' extra parts
    uint32_t *bin_count = wave_count + 16;                     // [0 .. kDepthBins): this workgroup's entries per depth bin (cc.depth_hist != null)
    uint2 *const list_out = cc.depth_hist ? cc.unsorted : cc.bricks;
    if (cc.depth_hist) {
        for (uint32_t i = threadIdx.x; i < kDepthBins; i += 256) bin_count[i] = 0u;
        __syncthreads();
    }
    const uint32_t n = (uint32_t)occ.fine_count(), n_words = (n + 3u) / 4u, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t *fine4 = reinterpret_cast<const uint32_t *>(occ.fine), *cell4 = reinterpret_cast<const uint32_t *>(occ.cell);
    const bool look = cc.pairs_per_task != 0u;   // (uniform) project the bricks: drop the ones no pixel sees, list the large ones in parts
    const float eps = make_skip_ctx(g, t_table[1]).eps;
    const float side_ = (float)kBrick + 1.5f + 2.0f * eps;   // the listed box of a brick, in voxels: its half-diagonal in millimetres
    const float us = 0.5f * (float)ep.width, vs_ = 0.5f * (float)ep.height;
    const float gu0 = sqrtf(ep.k[0][0] * ep.k[0][0] + ep.k[0][1] * ep.k[0][1] + (ep.k[0][2] - us) * (ep.k[0][2] - us)), gv0 = sqrtf(ep.k[1][0] * ep.k[1][0] + ep.k[1][1] * ep.k[1][1] + (ep.k[1][2] - vs_) * (ep.k[1][2] - vs_));
    // (in the camera's frame: the world's half-diagonal times the bound of the pose's stretch -- a pose need not be rigid)
    const float radius = 0.5f * sqrtf((side_ * g.vs.x) * (side_ * g.vs.x) + (side_ * g.vs.y) * (side_ * g.vs.y) + (side_ * g.vs.z) * (side_ * g.vs.z)) * ep.r_scale;
    for (uint32_t w0 = blockIdx.x * 256u; w0 < n_words; w0 += n_list_blocks * 256u) {   // (uniform over the workgroup)
        const uint32_t w = w0 + threadIdx.x;
        uint32_t f = 0, c = 0;
        if (w < n_words) {
            if (4u * w + 3u < n) {
                f = fine4[w];
                if (f) c = cell4[w];
            } else {   // (the array's last, partial word)
                for (uint32_t j = 0; 4u * w + j < n; j++) {
                    f |= (occ.fine[4u * w + j] ? 1u : 0u) << (8u * j);
                    c |= (occ.cell[4u * w + j] ? 1u : 0u) << (8u * j);
                }
            }
        }
        if (__syncthreads_or(f != 0u) == 0) continue;
        uint32_t entry[4], coords[4], parts[4], extra = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) {
            entry[j] = 0;
            coords[j] = 0;
            parts[j] = 0;
            if ((f >> (8u * j)) & 0xffu) {
                const uint32_t b = 4u * w + j;
                const uint32_t bz = b / (occ.nbx * occ.nby), r = b - bz * (occ.nbx * occ.nby), by = r / occ.nbx, bx = r - by * occ.nbx;
                bool mine = true;
                if (SLAB) mine = (bz + 1u) * kBrick > rp.own_lo && bz * kBrick < rp.own_hi + 1u;   // (a cell's lower plane, or a shell sample's lower tap, in [own_lo, own_hi))
                if (mine) {
                    if ((c >> (8u * j)) & 0xffu) entry[j] |= kCellTasks;
                    if (bx == 0 || by == 0 || bz == 0 || bx + 1 == occ.nbx || by + 1 == occ.nby || bz + 1 == occ.nbz) entry[j] |= kShellTasks;
                }
                bool seen = true;
                float bw = 0.0f, bh = 0.0f;   // the sides of the brick's box in pixels
                if (look && entry[j]) {
                    // Can any pixel see it?  Its voxels and its cells (voxel centres x0 + 1/2 .. x0 + kBrick + 3/2), grown by eps -- the union
                    // of what the cells' and the shell's tasks project -- inside a sphere around its centre: in front of the camera by more
                    // than its radius R, every point of it projects within |grad F| R / (depth - R) of the centre (F = N - u_c D, zero at
                    // the centre, D >= depth - R; project_box has the symbols).
                    const float x0 = (float)(bx * kBrick), y0 = (float)(by * kBrick), z0 = (float)(bz * kBrick), mid = 0.5f * ((float)kBrick + 1.5f);
                    const float wx = (x0 + mid) * g.vs.x + ep.offset.x, wy = (y0 + mid) * g.vs.y + ep.offset.y, wz = (z0 + mid) * g.vs.z + ep.offset.z;
                    const float ccx = ep.r[0][0] * wx + ep.r[0][1] * wy + ep.r[0][2] * wz + ep.r[0][3];
                    const float ccy = ep.r[1][0] * wx + ep.r[1][1] * wy + ep.r[1][2] * wz + ep.r[1][3];
                    const float ccz = ep.r[2][0] * wx + ep.r[2][1] * wy + ep.r[2][2] * wz + ep.r[2][3];
                    const float zfront = ccz - radius;
                    if (zfront > ep.z_near && zfront > 0.01f * radius) {
                        const float rz = __builtin_amdgcn_rcpf(ccz), rf = radius * __builtin_amdgcn_rcpf(zfront) * 1.001f;
                        const float uc = (ep.k[0][0] * ccx + ep.k[0][1] * ccy + ep.k[0][2] * ccz) * rz, vc = (ep.k[1][0] * ccx + ep.k[1][1] * ccy + ep.k[1][2] * ccz) * rz;
                        const float gu = sqrtf(ep.k[0][0] * ep.k[0][0] + ep.k[0][1] * ep.k[0][1] + (ep.k[0][2] - uc) * (ep.k[0][2] - uc)) * rf + 0.5f;
                        const float gv = sqrtf(ep.k[1][0] * ep.k[1][0] + ep.k[1][1] * ep.k[1][1] + (ep.k[1][2] - vc) * (ep.k[1][2] - vc)) * rf + 0.5f;
                        // (anything not a number: seen, and as large as the image)
                        seen = !(uc + gu < 0.0f || vc + gv < 0.0f || uc - gu > (float)(ep.width - 1u) || vc - gv > (float)(ep.height - 1u));
                        bw = fminf(1.2f * gu, (float)ep.width);    // (the sphere is wider than the box it holds: 2 r / sqrt 3)
                        bh = fminf(1.2f * gv, (float)ep.height);
                        if (!(bw == bw && bh == bh)) { bw = (float)ep.width; bh = (float)ep.height; }
                    } else {
                        // at the camera plane, or across it: beside the camera it lies off the image, on the side it is on (project_box:
                        // F over the sphere is F_c -+ |grad F| R, D at most depth + R); otherwise it is as large as the image
                        const float rhi = 1.0f / ((ccz + radius) * 1.001f);
                        const float fu = (ep.k[0][0] * ccx + ep.k[0][1] * ccy + ep.k[0][2] * ccz) - us * ccz, fv = (ep.k[1][0] * ccx + ep.k[1][1] * ccy + ep.k[1][2] * ccz) - vs_ * ccz;
                        const float su = gu0 * radius * 1.001f + 1.0e-4f * (fabsf(fu) + us * fabsf(ccz)), sv = gv0 * radius * 1.001f + 1.0e-4f * (fabsf(fv) + vs_ * fabsf(ccz));
                        seen = !(ccz + radius < ep.z_clip || (fu - su > 0.0f && (fu - su) * rhi > us) || (fu + su < 0.0f && (fu + su) * rhi < -us - 1.0f) ||
                                 (fv - sv > 0.0f && (fv - sv) * rhi > vs_) || (fv + sv < 0.0f && (fv + sv) * rhi < -vs_ - 1.0f));
                        bw = (float)ep.width;
                        bh = (float)ep.height;
                    }
                }
                if (!seen) entry[j] = 0;   // a brick outside the view is not listed at all
                if (entry[j]) {
                    coords[j] = bx | (by << 10) | (bz << 20);
                    parts[j] = 1;
                    if (look) {
                        // all 64 cells mixed, each a quarter of the brick's box and a pixel of margin: an estimate from above (a boundary
                        // brick's shell samples are shared out by the same parts: every pixel of its box is looked at)
                        const float est = 64.0f * (0.25f * bw + 1.0f) * (0.25f * bh + 1.0f);
                        parts[j] = (uint32_t)fminf(fmaxf(ceilf(est / (float)cc.pairs_per_task), 1.0f), (float)kMaxParts);
                    }
                    extra += parts[j] - 1u;
                }
            }
        }
        if (look) {   // the workgroup's extra parts against the room it has
            const uint32_t extra_wave = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_add(extra), 63);
            if (lane == 0u) wave_count[8u + wave] = extra_wave;
            __syncthreads();
            const uint32_t extra_all = wave_count[8] + wave_count[9] + wave_count[10] + wave_count[11];
#pragma unroll
            for (uint32_t j = 0; j < 4u; j++)
                if (extra_all > kPartsRoom && parts[j] > 1u) parts[j] = 1u + (uint32_t)(((uint64_t)(parts[j] - 1u) * kPartsRoom) / extra_all);
        }
        const uint32_t mine_n = (parts[0] + parts[1]) + (parts[2] + parts[3]);
        const uint32_t incl = wave_scan_add(mine_n);   // inclusive prefix over the wave
        if (lane == 63u) wave_count[wave] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t total = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
            wave_count[4] = total ? atomicAdd(cc.n_bricks, total) : 0u;
        }
        __syncthreads();
        uint32_t at = wave_count[4] + incl - mine_n;
        for (uint32_t q = 0; q < wave; q++) at += wave_count[q];
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) {
            for (uint32_t part = 0; part < parts[j]; part++)
                list_out[at++] = make_uint2(entry[j] | (part << 10) | (parts[j] - 1u), coords[j]);
            if (cc.depth_hist && parts[j]) atomicAdd(&bin_count[depth_bin(cc, ep, g, coords[j])], parts[j]);
        }
        __syncthreads();   // (wave_count is written again in the next turn)
    }
    if (cc.depth_hist) {   // (uniform) this workgroup's counts into the launch's
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < kDepthBins; i += 256)
            if (bin_count[i]) atomicAdd(&cc.depth_hist[i], bin_count[i]);
    }
}

// The list in front-to-back order (round 6; views from inside the volume, where surface lies behind surface: every listed cell is
// looked at, hidden or not, unless its pixels' words already hold a nearer hit -- the waves take the list from its first entry on,
// so with the near bricks first the `known <= k` test of a pair ends most hidden ones before their samples are walked: 0.173 -> 0.133 ms
// for the cast's kernel from inside a 1024^3 volume, what the same launch costs with every word final (profiles/r06_cells_front_to_back.txt);
// from outside the order buys nothing and the list is taken as built).  A counting sort by the bin of each brick's camera depth: the
// counts per bin come with the list (cell_cast_prepare_kernel), a workgroup here takes 1 024 entries, ranks them within the bins in
// LDS, takes its places in each bin with one atomic per bin it holds, and scatters.  Within a bin the order is whatever it comes to.
__global__ __launch_bounds__(1024) void cell_list_sort_kernel(const Geom g, const EntryParams ep, const CellCast cc) {
    __shared__ uint32_t start[kDepthBins], mine[kDepthBins], base[kDepthBins];
    const uint32_t n = *cc.n_bricks, t = threadIdx.x;
    if (t < kDepthBins) start[t] = cc.depth_hist[t];
    __syncthreads();
    if (t < 64u) {   // exclusive prefix over the 256 bins, one wave: four bins a lane
        const uint32_t a0 = start[4u * t], a1 = start[4u * t + 1u], a2 = start[4u * t + 2u], a3 = start[4u * t + 3u];
        uint32_t incl = (a0 + a1) + (a2 + a3);
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if ((int)t >= o) incl += up;
        }
        const uint32_t ex = incl - ((a0 + a1) + (a2 + a3));
        start[4u * t] = ex; start[4u * t + 1u] = ex + a0; start[4u * t + 2u] = ex + a0 + a1; start[4u * t + 3u] = ex + a0 + a1 + a2;
    }
    uint32_t *fill = cc.depth_hist + kDepthBins;
    for (uint32_t c0 = blockIdx.x * 1024u; c0 < n; c0 += gridDim.x * 1024u) {   // (uniform)
        __syncthreads();
        if (t < kDepthBins) mine[t] = 0u;
        __syncthreads();
        const uint32_t i = c0 + t;
        uint2 e = make_uint2(0u, 0u);
        uint32_t bin = 0, rank = 0;
        if (i < n) {
            e = cc.unsorted[i];
            bin = depth_bin(cc, ep, g, e.y);
            rank = atomicAdd(&mine[bin], 1u);
        }
        __syncthreads();
        if (t < kDepthBins && mine[t]) base[t] = atomicAdd(&fill[t], mine[t]);
        __syncthreads();
        if (i < n) cc.bricks[start[bin] + base[bin] + rank] = e;
    }
}

// Samples of the ray (s, d) whose positions can lie in the box [lo, hi] (grid millimetres): [k_lo, k_hi], clipped to [k_first, k_end).
// Approximate on purpose -- the caller tests every candidate's own position -- and generous by kIntervalSlack of a sample either side.
// (Round 5 took a whole sample either side + the floor / ceil: two or three candidates outside the cell per pair, and a round of 64
// pairs walks as long as its longest lane -- 3.0 M candidate samples for 1.3 M inside.  What the slack has to cover: T[k] = k step +
// drift(k), the drift -- the rounding of k fp32 additions, under a sample over the whole table -- read at the entry and changing by at most
// half an ulp of T, 2e-4 of a sample, per sample from there; the parameters from hardware reciprocals, a few ulps of up to 4 402 samples:
// 2e-3; the box itself is the cell grown by eps.  An eighth of a sample is that fifty times over.)
constexpr float kIntervalSlack = 0.125f;
__device__ inline bool sample_interval(const RayState &r, float lox, float loy, float loz, float hix, float hiy, float hiz, float inv_step, float step,
                                       const float *T, int k_first, int k_end, int &k_lo, int &k_hi) {
    float tin = 0.0f, tout = INFINITY;
    bool miss = false;
    auto axis = [&](float s, float d, float lo, float hi) {
        if (d != 0.0f) {
            const float rd = __builtin_amdgcn_rcpf(d);
            const float t0 = (lo - s) * rd, t1 = (hi - s) * rd;
            tin = fmaxf(tin, fminf(t0, t1));
            tout = fminf(tout, fmaxf(t0, t1));
        } else if (s < lo || s > hi) {
            miss = true;
        }
    };
    axis(r.sx, r.dx, lox, hix);
    axis(r.sy, r.dy, loy, hiy);
    axis(r.sz, r.dz, loz, hiz);
    // (relative slack on the parameters: the reciprocal and the products are good to a few ulps, the table to a sample -- covered below)
    if (miss || !(tin <= tout * 1.00001f + 1.0e-3f)) return false;
    // T[k] = k * step + drift(k), the drift -- the rounding of k additions -- under a sample over the whole table and all but constant
    // over a cell's few samples: read where the ray enters
    const float kc = fminf(fmaxf(rintf(tin * inv_step), 0.0f), (float)kMaxSamples);
    const float drift = T[(int)kc] - kc * step;
    const float a = ceilf((tin - drift) * inv_step - kIntervalSlack), b = floorf((tout - drift) * inv_step + kIntervalSlack);
    k_lo = max(k_first, a > 0.0f ? (a < 8192.0f ? (int)a : 8192) : 0);
    k_hi = min(k_end - 1, b < 8192.0f ? (b > 0.0f ? (int)b : 0) : 8192);
    return k_lo <= k_hi;
}

// The shell samples of the flagged bricks that touch the grid boundary (kShellTasks): the work of the FIRST kShellWorkgroups workgroups
// of cast_cells_kernel.  A few hundred bricks at most, every sample the reference's full interpolation -- a chain of round trips per
// pixel (17 us as a launch of its own, profiles/r05n_*) that costs next to no issue slots: at the head of the cells' launch it runs
// beside the bricks' waves from their first microsecond to long before their last.  Inside the brick loop it cost every wave of the
// kernel a fifth of its registers (132 -> 96 a lane without it); per pixel in the resolve kernels, where no list is needed, it was
// 20 us on top of their 7 (every pixel pays the flag look-ups, tools/experiments/raycast_shell_in_resolve.patch.txt).
// Each of these workgroups looks at its share of the brick list -- entries blockIdx.x, + kShellWorkgroups, ... -- one entry a thread, at
// once (such bricks sit next to each other in the list: a stretch per workgroup left a handful with all of them), and then takes its
// boundary bricks one after the other: every pixel the brick's voxel box can be seen by, eight threads a pixel -- each every eighth
// sample of the pixel's stretch through a slab; the samples are independent, the pixel's word takes the minimum -- and per pixel only
// the samples in the half-voxel slabs along the grid faces the brick touches (a tenth of the samples that cross the brick).
#ifndef TSDF_SHELL_WGS
#define TSDF_SHELL_WGS 512   // (1 024 in round 5; 512 / 256 are 1-2 % faster on the bench scenes -- fewer workgroups that find nothing in front of the bricks' --, 128 twice as slow at 256^3: profiles/r06_cells_grid_sweep.txt)
#endif
constexpr uint32_t kShellWorkgroups = TSDF_SHELL_WGS;
template <bool SLAB, bool FASTDIV>
__device__ inline void cast_shell_bricks(const float *__restrict__ dist, const Geom &g, const RayParams &rp, const EntryParams &ep, const float *T,
                                         const CellCast &cc, uint64_t *__restrict__ best, uint32_t n_bricks, uint2 *mine, uint32_t *n_mine) {
    const TriConst &tc = rp.tc;
    const float step_size = T[1];
    const SkipCtx sc = make_skip_ctx(g, step_size);
    const float e = sc.eps;
    const float size_[3] = {(float)g.X, (float)g.Y, (float)g.Z}, vs_[3] = {g.vs.x, g.vs.y, g.vs.z};
    for (uint32_t j0 = 0; blockIdx.x + (size_t)j0 * kShellWorkgroups < n_bricks; j0 += 256) {   // (uniform; one turn unless the list is longer than 256 x these workgroups)
        __syncthreads();   // (the turn before has read `mine`)
        if (threadIdx.x == 0) *n_mine = 0;
        __syncthreads();
        const size_t ei = blockIdx.x + (size_t)(j0 + threadIdx.x) * kShellWorkgroups;
        if (ei < n_bricks) {
            const uint2 en = cc.bricks[ei];
            if (en.x & kShellTasks) mine[atomicAdd(n_mine, 1u)] = en;   // (at most one a thread: 256)
        }
        __syncthreads();
        const uint32_t n_shell = *n_mine;
        for (uint32_t si = 0; si < n_shell; si++) {
            const uint2 entry2 = mine[si];
            const uint32_t o_[3] = {(entry2.y & 1023u) * kBrick, ((entry2.y >> 10) & 1023u) * kBrick, (entry2.y >> 20) * kBrick};
            // the brick's voxels in voxel units, grown by eps (and so reaching off the grid where the brick touches it)
            float lo_[3], hi_[3];
#pragma unroll
            for (int a_ = 0; a_ < 3; a_++) {
                lo_[a_] = (float)o_[a_] - e;
                hi_[a_] = fminf((float)(o_[a_] + kBrick), size_[a_]) + e;
            }
            PixelBox sb;
            if (!project_box(ep, lo_[0] * vs_[0], lo_[1] * vs_[1], lo_[2] * vs_[2], hi_[0] * vs_[0], hi_[1] * vs_[1], hi_[2] * vs_[2], sb)) continue;
            // this task's share of the box's pixels (all of them, unless the brick was listed in parts: one seen from close by is
            // tens of thousands of pixels, 32 a turn)
            const uint32_t part = (entry2.x >> 10) & (kMaxParts - 1u), parts = (entry2.x & (kMaxParts - 1u)) + 1u;
            const uint32_t share = ((uint32_t)(sb.w * sb.h) + parts - 1u) / parts;
            const int pi_begin = (int)min(part * share, (uint32_t)(sb.w * sb.h)), n_pix = (int)min((uint32_t)pi_begin + share, (uint32_t)(sb.w * sb.h));
            if (threadIdx.x == 0) { RAY_MIX(42); }
            const int sub = (int)(threadIdx.x & 7u);
            for (int pi = pi_begin + (int)(threadIdx.x >> 3); pi < n_pix; pi += 32) {
                const int py = pi / sb.w, px = pi - py * sb.w;
                if (sub == 0) { RAY_MIX(43); }
                const uint32_t idx = (uint32_t)(sb.v0 + py) * rp.width + (uint32_t)(sb.u0 + px);
                const RayRecord rec_ = cc.rays[idx];
                const uint32_t known = (uint32_t)(best[idx] >> 32);
                const RayState ray = ray_from_near(ray_direction(sb.u0 + px, sb.v0 + py, rp), rec_.near_t, rp);
                const int k_first = (int)(rec_.k_range & 0xffffu), k_end = (int)(rec_.k_range >> 16);
                // the slabs of the brick in which a sample's lower tap is off the lattice of cells (or within eps of that): the first half
                // voxel behind a low face of the grid, the last half voxel in front of a high one
#pragma unroll
                for (int a_ = 0; a_ < 3; a_++) {
#pragma unroll
                    for (int side = 0; side < 2; side++) {
                        float slo[3] = {lo_[0], lo_[1], lo_[2]}, shi[3] = {hi_[0], hi_[1], hi_[2]};
                        if (side == 0) {
                            if (o_[a_] != 0u) continue;
                            shi[a_] = 0.5f + e;
                        } else {
                            if ((float)(o_[a_] + kBrick) < size_[a_]) continue;
                            slo[a_] = size_[a_] - 0.5f - e;
                        }
                        int k, k_hi;
                        if (!sample_interval(ray, slo[0] * vs_[0], slo[1] * vs_[1], slo[2] * vs_[2], shi[0] * vs_[0], shi[1] * vs_[1], shi[2] * vs_[2], sc.inv_step, step_size, T,
                                             k_first, k_end, k, k_hi))
                            continue;
                        if (known <= (uint32_t)k) continue;
                        for (k += sub; k <= k_hi; k += 8) {
                            RAY_MIX(44);
                            const float t = T[k];
                            const float ppx = (t * ray.dx) + ray.sx, ppy = (t * ray.dy) + ray.sy, ppz = (t * ray.dz) + ray.sz;
                            const float f_[3] = {ppx * sc.inv_vx, ppy * sc.inv_vy, ppz * sc.inv_vz};
                            // in this slab of this brick's (grown) voxel box
                            if (!(f_[0] >= slo[0] && f_[0] <= shi[0] && f_[1] >= slo[1] && f_[1] <= shi[1] && f_[2] >= slo[2] && f_[2] <= shi[2])) continue;
                            bool owned;
                            const float tsdf = trilinear<SLAB, false, FASTDIV>(ppx, ppy, ppz, dist, g, tc, rp, owned, nullptr);
                            if (tsdf <= 0) {
                                RAY_MIX(41);
                                lower_best(&best[idx], k, tsdf);
                                break;   // (of this thread's samples of the slab nothing behind it counts)
                            }
                        }
                    }
                }
            }
        }
    }
}

// One wave per flagged brick; the four waves of a workgroup share the table and nothing else.
// How many workgroups (TSDF_RAY_CELLS_GRID; round 6, profiles/r06_cells_*): a wave's preamble -- the table, the view's constants, the
// scalar registers that do not fit parked in vector lanes -- is 300 vector instructions, a brick's turn 630 and a round of 64 pairs
// 600; a wave per brick (8 192 workgroups, 36 864 waves for 20 000 bricks: round 5) spent a quarter of the kernel's 37.5 M vector
// instructions on preambles.  2 048 workgroups (each wave two or three bricks, the next one's voxels requested a turn ahead) is the
// measured optimum at 512^3 and 256^3: 0.083 -> 0.076 ms; as many waves as the chip holds (1 024) with every n-th entry dealt out in
// advance ends when its unluckiest wave does (0.080), and the same with the list taken a chunk at a time from eight counters is
// slower still (0.086; 256^3 0.102: tools/experiments/raycast_cells_dynamic_chunks.patch.txt).
// (Measured and dropped, profiles/r05m_*: the walk of a pair's samples deferred to a ring of the wave in LDS and done 64 records at a
// time, all lanes busy -- three lanes in four idle through the walk otherwise -- was no faster: the deferred walks find their hits
// later, so twice as many pairs survive the test against the pixel's word, and a record has to fetch its cell's voxels again.)
#ifndef TSDF_CELLS_WAVES
#define TSDF_CELLS_WAVES 4
#endif
// The cell of pair q of a brick.  1: every mixed cell leaves its lane at the place of its first pair in the round's window of 64
// (one LDS write), a lane reads its place and a max-scan over the lanes (DPP, six instructions) hands every pair the last cell that
// starts at or before it.  0: a binary search of the cells' prefix sums -- six dependent LDS reads a round, whose latency showed in the
// knock-out figures (13 us for 4.3 M instructions: profiles/r06_cells_knockouts.txt).
#ifndef TSDF_CELLS_SCAN_DECODE
#define TSDF_CELLS_SCAN_DECODE 1
#endif
template <bool SLAB, bool FASTDIV>
#if TSDF_CELLS_WAVES
__attribute__((amdgpu_waves_per_eu(TSDF_CELLS_WAVES, TSDF_CELLS_WAVES)))
#endif
__global__ __launch_bounds__(256) void cast_cells_kernel(const float *__restrict__ dist, const Geom g, const RayParams rp, const EntryParams ep,
                                                         const OccGrid occ, const float *__restrict__ t_table, const CellCast cc,
                                                         uint64_t *__restrict__ best) {
    __shared__ __attribute__((aligned(16))) float T[kTableLen];
    __shared__ float corner[4][128];       // the brick's 5^3 voxels, x fastest
    __shared__ __attribute__((aligned(16))) int box_of[4][64][4];   // per cell lane: its pixel box
    __shared__ int prefix[4][64];          // pairs of the cells before this one
#if TSDF_CELLS_SCAN_DECODE
    __shared__ uint32_t mark[4][64];       // per place in the round's window: (round << 6 | cell) of the cell whose pairs start there
    mark[threadIdx.x >> 6][threadIdx.x & 63u] = 0u;
    uint32_t round = 0u;                   // (a wave's rounds are numbered through: a place written in an earlier round loses the max)
#endif
    static_assert(kTableLen % 4 == 0, "the table in 16-byte pieces");
    for (int i = (int)threadIdx.x; i < kTableLen / 4; i += 256) reinterpret_cast<float4 *>(T)[i] = reinterpret_cast<const float4 *>(t_table)[i];
    const uint32_t n_bricks = *cc.n_bricks;
    if (cc.n_bricks_host && blockIdx.x == 0 && threadIdx.x == 0) *cc.n_bricks_host = n_bricks;
    static_assert(sizeof(box_of) >= 256 * sizeof(uint2), "a shell workgroup's list of bricks");
    if (blockIdx.x < kShellWorkgroups) {   // the boundary bricks' shell samples, beside the cells' waves from the start
        __syncthreads();
        cast_shell_bricks<SLAB, FASTDIV>(dist, g, rp, ep, T, cc, best, n_bricks, reinterpret_cast<uint2 *>(&box_of[0][0][0]), reinterpret_cast<uint32_t *>(&prefix[0][0]));
        return;
    }
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const float step_size = t_table[1];
    const TriConst &tc = rp.tc;
    const SkipCtx sc = make_skip_ctx(g, step_size);
    const float e = sc.eps;
    const size_t plane = (size_t)g.X * g.Y;
    // The pixel box of a cell without projecting its 8 corners.  u = N(P) / D(P), N and D affine in P, D the camera depth: for P in the
    // (grown) cell |u(P) - u(centre)| = |N(P) - u_c D(P)| / D(P) <= S / D_min, S = half the sum over the axes of |the coefficient of
    // N - u_c D along the cell's edge| -- the edges are the same for every cell: uniform registers.
    const float ex = (1.0f + 2.0f * e) * g.vs.x, ey = (1.0f + 2.0f * e) * g.vs.y, ez = (1.0f + 2.0f * e) * g.vs.z;
    float nu_[3], nv_[3], nz_[3];
#pragma unroll
    for (int a_ = 0; a_ < 3; a_++) {
        const float ext = a_ == 0 ? ex : a_ == 1 ? ey : ez;
        const float cxa = ep.r[0][a_] * ext, cya = ep.r[1][a_] * ext, cza = ep.r[2][a_] * ext;
        nu_[a_] = ep.k[0][0] * cxa + ep.k[0][1] * cya + ep.k[0][2] * cza;
        nv_[a_] = ep.k[1][0] * cxa + ep.k[1][1] * cya + ep.k[1][2] * cza;
        nz_[a_] = cza;
    }
    const float half_dz = 0.5f * ((fabsf(nz_[0]) + fabsf(nz_[1])) + fabsf(nz_[2]));
    __syncthreads();   // (the table; from here on the four waves go their own ways: every other array is a wave's own)
    // LDS traffic inside one wave is in program order; the fence keeps the compiler from moving a lane's read above another lane's write
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's LDS traffic so far is done
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // a brick's 5^3 voxels, lane i voxels i and i + 64: requested a turn ahead, while the brick before is worked on
    // (lane i's two voxels sit at the same offsets from the brick's first voxel in every brick that lies inside the resident planes)
    auto offset_of = [&](uint32_t i) { return (size_t)(i / 25u) * plane + (size_t)((i / 5u) % 5u) * g.X + (i % 5u); };
    const size_t off_a = offset_of(lane), off_b = offset_of(min(lane + 64u, 124u));
    auto brick_voxels = [&](uint2 entry_, float &va, float &vb) {
        va = vb = 0.0f;
        if (!(entry_.x & kCellTasks)) return;
        const uint32_t bx_ = entry_.y & 1023u, by_ = (entry_.y >> 10) & 1023u, bz_ = entry_.y >> 20;
        if (bx_ * kBrick + kBrick < g.X && by_ * kBrick + kBrick < g.Y && bz_ * kBrick >= g.z_store_begin && bz_ * kBrick + kBrick < g.z_store_end) {   // (uniform)
            const float *first = dist + (plane * (bz_ * kBrick - g.z_store_begin) + (size_t)g.X * (by_ * kBrick) + bx_ * kBrick);
            va = first[off_a];
            vb = first[off_b];
            return;
        }
        auto voxel = [&](uint32_t i) {
            const uint32_t cx = i % 5u, cy = (i / 5u) % 5u, cz = i / 25u;
            const uint32_t vx = min(bx_ * kBrick + cx, g.X - 1u), vy = min(by_ * kBrick + cy, g.Y - 1u);
            const uint32_t vz = min(max(bz_ * kBrick + cz, g.z_store_begin), g.z_store_end - 1u);   // (a cell that needs a plane this object does not hold is not valid below)
            return dist[plane * (vz - g.z_store_begin) + (size_t)g.X * vy + vx];
        };
        va = voxel(lane);
        if (lane + 64u < 125u) vb = voxel(lane + 64u);
    };
    const uint32_t n_waves = (gridDim.x - kShellWorkgroups) * 4u;
    uint32_t ei = (blockIdx.x - kShellWorkgroups) * 4u + wave;
    uint2 entry_next = ei < n_bricks ? cc.bricks[ei] : make_uint2(0u, 0u);
    float va_next, vb_next;
    brick_voxels(entry_next, va_next, vb_next);
    for (; ei < n_bricks; ei += n_waves) {
        const uint2 entry2 = entry_next;
        const uint32_t entry = entry2.x;
        const float va = va_next, vb = vb_next;
        entry_next = ei + n_waves < n_bricks ? cc.bricks[ei + n_waves] : make_uint2(0u, 0u);
        brick_voxels(entry_next, va_next, vb_next);
        const uint32_t bx = entry2.y & 1023u, by = (entry2.y >> 10) & 1023u, bz = entry2.y >> 20;
        const uint32_t x0 = bx * kBrick, y0 = by * kBrick, z0 = bz * kBrick;
        wave_sync();   // (the previous turn's readers of corner / box_of / prefix)
        corner[wave][lane] = va;
        corner[wave][lane + 64u] = vb;
        wave_sync();
        // ---- the brick's mixed cells, one per lane ----
        const uint32_t cx = lane & 3u, cy = (lane >> 2) & 3u, cz = lane >> 4;
        const uint32_t lx = x0 + cx, ly = y0 + cy, lz = z0 + cz;
        bool mixed = false;
        if (entry & kCellTasks) {
            bool valid = lx + 1u < g.X && ly + 1u < g.Y && lz + 1u < g.Z && lz >= g.z_store_begin && lz + 1u < g.z_store_end;
            if (SLAB) valid = valid && lz >= rp.own_lo && lz < rp.own_hi;
            const float *c = &corner[wave][cx + 5u * cy + 25u * cz];
            const bool positive = fminf(fminf(fminf(c[0], c[1]), fminf(c[5], c[6])), fminf(fminf(c[25], c[26]), fminf(c[30], c[31]))) > kCellPositive;
            mixed = valid && !positive;
        }
        PixelBox pb = {0, 0, 0, 0};
        if (mixed) {
            // the cell's centre (voxel centres lx + 1/2 .. lx + 3/2) in the camera's frame
            const float wx = ((float)lx + 1.0f) * g.vs.x + ep.offset.x, wy = ((float)ly + 1.0f) * g.vs.y + ep.offset.y, wz = ((float)lz + 1.0f) * g.vs.z + ep.offset.z;
            const float ccx = ep.r[0][0] * wx + ep.r[0][1] * wy + ep.r[0][2] * wz + ep.r[0][3];
            const float ccy = ep.r[1][0] * wx + ep.r[1][1] * wy + ep.r[1][2] * wz + ep.r[1][3];
            const float ccz = ep.r[2][0] * wx + ep.r[2][1] * wy + ep.r[2][2] * wz + ep.r[2][3];
            const float zmin = ccz - half_dz;
            if (zmin > ep.z_near) {
                const float rz = __builtin_amdgcn_rcpf(ccz), rm = __builtin_amdgcn_rcpf(zmin);
                const float uc = (ep.k[0][0] * ccx + ep.k[0][1] * ccy + ep.k[0][2] * ccz) * rz, vc = (ep.k[1][0] * ccx + ep.k[1][1] * ccy + ep.k[1][2] * ccz) * rz;
                const float su = 0.5f * ((fabsf(nu_[0] - uc * nz_[0]) + fabsf(nu_[1] - uc * nz_[1])) + fabsf(nu_[2] - uc * nz_[2]));
                const float sv = 0.5f * ((fabsf(nv_[0] - vc * nz_[0]) + fabsf(nv_[1] - vc * nz_[1])) + fabsf(nv_[2] - vc * nz_[2]));
                // (a ray exists per INTEGER pixel; 0.05 px over the fp32 evaluation: coordinates < 2^16, reciprocals to an ulp)
                const float hu = su * rm * 1.0001f + 0.05f, hv = sv * rm * 1.0001f + 0.05f;
                const float a0 = fmaxf(ceilf(uc - hu), 0.0f), a1 = fminf(floorf(uc + hu), (float)(ep.width - 1u));
                const float b0 = fmaxf(ceilf(vc - hv), 0.0f), b1 = fminf(floorf(vc + hv), (float)(ep.height - 1u));
                mixed = a0 <= a1 && b0 <= b1;   // (false also for anything not a number)
                if (mixed) { pb.u0 = (int)a0; pb.v0 = (int)b0; pb.w = (int)a1 - (int)a0 + 1; pb.h = (int)b1 - (int)b0 + 1; }
            } else {
                // at or behind the plane in front of which the view's samples lie: the 8 corners, the part in front of the plane
                const float clx = ((float)lx + 0.5f - e) * g.vs.x, chx = ((float)lx + 1.5f + e) * g.vs.x;
                const float cly = ((float)ly + 0.5f - e) * g.vs.y, chy = ((float)ly + 1.5f + e) * g.vs.y;
                const float clz = ((float)lz + 0.5f - e) * g.vs.z, chz = ((float)lz + 1.5f + e) * g.vs.z;
                mixed = project_box<false>(ep, clx, cly, clz, chx, chy, chz, pb);
            }
        }
        if (lane == 0) { RAY_MIX(32); }
        if (mixed) { RAY_MIX(33); }
        // ---- (cell, pixel) pairs, packed: pair q of the brick belongs to the last cell whose exclusive prefix is <= q ----
#if TSDF_CELLS_SCAN_DECODE
        const int n_mine = mixed ? pb.w * pb.h : 0, incl = (int)wave_scan_add((uint32_t)n_mine);
        const int n_pairs = __builtin_amdgcn_readlane(incl, 63);
        const int first_mine = incl - n_mine;
        if (mixed) {   // (the fourth word: the cell's first pair instead of the box's height, which no pair asks for)
            box_of[wave][lane][0] = pb.u0; box_of[wave][lane][1] = pb.v0; box_of[wave][lane][2] = pb.w; box_of[wave][lane][3] = first_mine;
        }
#else
        int n_mine = mixed ? pb.w * pb.h : 0, incl = n_mine;
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if ((int)lane >= o) incl += up;
        }
        const int n_pairs = __shfl(incl, 63);
        prefix[wave][lane] = incl - n_mine;
        if (mixed) {
            box_of[wave][lane][0] = pb.u0; box_of[wave][lane][1] = pb.v0; box_of[wave][lane][2] = pb.w; box_of[wave][lane][3] = pb.h;
        }
        wave_sync();
#endif
        // this task's share of the brick's pairs (all of them, unless the brick was listed in parts)
        int q_begin = 0, q_end = n_pairs;
        if (entry & (kMaxParts - 1u)) {   // (uniform)
            const uint32_t part = (entry >> 10) & (kMaxParts - 1u), parts = (entry & (kMaxParts - 1u)) + 1u;
            const uint32_t share = ((uint32_t)n_pairs + parts - 1u) / parts;
            q_begin = (int)min(part * share, (uint32_t)n_pairs);
            q_end = (int)min((uint32_t)q_begin + share, (uint32_t)n_pairs);
        }
        for (int q0 = q_begin; q0 < q_end; q0 += 64) {
            const int q = q0 + (int)lane;
#if TSDF_CELLS_SCAN_DECODE
            // the cells with a pair in [q0, q0 + 64): each at the place of its first pair there (the one that began before q0 at place 0)
            round++;
            const int place = first_mine - q0;
            if (n_mine > 0 && place < 64 && place + n_mine > 0) mark[wave][max(place, 0)] = round << 6 | lane;
            wave_sync();
            const int cl = (int)(wave_scan_max(mark[wave][lane]) & 63u);
            if (q >= q_end) continue;
            const int4 box = *reinterpret_cast<const int4 *>(&box_of[wave][cl][0]);
            const int pi = q - box.w;
            const int u0 = box.x, v0 = box.y, bw = box.z;
#else
            if (q >= q_end) continue;
            int cl = 0, hi_ = 63;
#pragma unroll
            for (int it = 0; it < 6; it++) {
                const int mid = (cl + hi_ + 1) >> 1;
                if (prefix[wave][mid] <= q) cl = mid; else hi_ = mid - 1;
            }
            const int pi = q - prefix[wave][cl];
            const int u0 = box_of[wave][cl][0], v0 = box_of[wave][cl][1], bw = box_of[wave][cl][2];
#endif
            const uint32_t qx = (uint32_t)cl & 3u, qy = ((uint32_t)cl >> 2) & 3u, qz = (uint32_t)cl >> 4;
            const float lfx = (float)(x0 + qx), lfy = (float)(y0 + qy), lfz = (float)(z0 + qz);
            RAY_MIX(34);
            // (pi / bw in floats: exact below 2^22, with the half that keeps the quotient off the integers)
            const int py = pi < (1 << 22) ? (int)(((float)pi + 0.5f) * __builtin_amdgcn_rcpf((float)bw)) : pi / bw, px = pi - py * bw;
            const int imx = u0 + px, imy = v0 + py;
            const uint32_t idx = (uint32_t)imy * rp.width + (uint32_t)imx;
            const RayRecord rec_ = cc.rays[idx];
            // (requested with the ray's record, not behind it; past the compute unit's own cache, so that what this wave's earlier pairs lowered is seen)
            const uint32_t known = __hip_atomic_load(reinterpret_cast<const uint32_t *>(best + idx) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const RayState ray = ray_from_near(ray_direction(imx, imy, rp), rec_.near_t, rp);
            int k, k_hi;
            if (!sample_interval(ray, (lfx + 0.5f - e) * g.vs.x, (lfy + 0.5f - e) * g.vs.y, (lfz + 0.5f - e) * g.vs.z, (lfx + 1.5f + e) * g.vs.x,
                                 (lfy + 1.5f + e) * g.vs.y, (lfz + 1.5f + e) * g.vs.z, sc.inv_step, step_size, T, (int)(rec_.k_range & 0xffffu),
                                 (int)(rec_.k_range >> 16), k, k_hi))
                continue;
            RAY_MIX(35);
            if (known <= (uint32_t)k) continue;   // (a hit in front of this cell is known already: a stale word only costs the work)
            RAY_MIX(36);
            const float *c = &corner[wave][qx + 5u * qy + 25u * qz];
            const float d000 = c[0], d100 = c[1], d010 = c[5], d110 = c[6], d001 = c[25], d101 = c[26], d011 = c[30], d111 = c[31];
            // (lower + 0.5f) * vs: the centre of the lower voxel, as process_sample forms it
            const float lcx = (lfx + 0.5f) * g.vs.x, lcy = (lfy + 0.5f) * g.vs.y, lcz = (lfz + 0.5f) * g.vs.z;
            SkipCtx rs = sc;
            set_ray<true>(rs, ray, step_size, g);
            const CellBound cb = cell_bound(d000, d100, d010, d110, d001, d101, d011, d111, rs);   // (the look-ahead's constants: per cell and ray)
            bool entered = false;
            while (k <= k_hi) {
                RAY_MIX(37);
                const float t = T[k];
                const float ppx = (t * ray.dx) + ray.sx, ppy = (t * ray.dy) + ray.sy, ppz = (t * ray.dz) + ray.sz;
                // position inside the cell, in cell units (approximate)
                const float rx = (ppx * rs.inv_vx - 0.5f) - lfx, ry = (ppy * rs.inv_vy - 0.5f) - lfy, rz = (ppz * rs.inv_vz - 0.5f) - lfz;
                const float far_ = fmaxf(fmaxf(fabsf(rx - 0.5f), fabsf(ry - 0.5f)), fabsf(rz - 0.5f));
                if (!(far_ <= 0.5f + e)) {   // outside the grown cell (or NaN): not yet in, or through (a straight line does not come back)
                    if (entered) break;
                    k++;
                    continue;
                }
                entered = true;
                if (far_ < 0.5f - e) {
                    RAY_MIX(38);
                    // the sample's dual cell is this one, to the reference's arithmetic too: its value from the cell's 8 voxels
                    const float u = div_by<FASTDIV>(ppx - lcx, tc.dx);
                    const float v = div_by<FASTDIV>(ppy - lcy, tc.dy);
                    const float w = div_by<FASTDIV>(ppz - lcz, tc.dz);
                    const float val = d000 * (1 - u) * (1 - v) * (1 - w) +
                                      d001 * (1 - u) * (1 - v) * w +
                                      d010 * (1 - u) * v * (1 - w) +
                                      d011 * (1 - u) * v * w +
                                      d100 * u * (1 - v) * (1 - w) +
                                      d101 * u * (1 - v) * w +
                                      d110 * u * v * (1 - w) +
                                      d111 * u * v * w;
                    if (val <= 0) {
                        RAY_MIX(40);
                        lower_best(&best[idx], k, val);
                        break;
                    }
                    int ahead = 0;
                    if (rs.skip_ok && val > 0) {
                        const float cell_lo = e, cell_hi = 1.0f - e;
                        const int n_cell = samples_until<false>(__builtin_fmaf(cell_hi - cell_lo, rs.posx, cell_lo) - rx, __builtin_fmaf(cell_hi - cell_lo, rs.posy, cell_lo) - ry,
                                                                __builtin_fmaf(cell_hi - cell_lo, rs.posz, cell_lo) - rz, rs);
                        ahead = lookahead_in_cell(val, cb, n_cell - 1);
                    }
                    k += 1 + ahead;
                } else {
                    // within eps of a face of the cell: the reference's own choice of cell and taps
                    RAY_MIX(39);
                    bool owned;
                    const float tsdf = trilinear<SLAB, false, FASTDIV>(ppx, ppy, ppz, dist, g, tc, rp, owned, nullptr);
                    if (tsdf <= 0) {
                        lower_best(&best[idx], k, tsdf);
                        break;
                    }
                    k++;
                }
            }
        }
    }
}
