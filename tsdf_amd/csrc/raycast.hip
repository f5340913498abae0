// TSDF ray casting + normals for gfx950 (wave64).  Replaces GPURaycaster::raycast, get_vertices /
// process_ray and compute_normals of the reference (src/RayCaster/GPURaycaster.cu:14-547).
//
// The march is the reference's, sample for sample, including its quirks (direction not normalised: Q6,
// previous_tsdf == trunc: Q7, 4402-sample cap: Q8, t accumulated by repeated float adds: Q9, unclamped
// point in the interpolation weights: Q10); fp contraction is off so every evaluated sample is
// bit-identical.  How the loop is reorganised for wave64 is described at process_ray_kernel below.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.hpp"

namespace tsdf {

struct InvDiv {
    float b, y;
};

// Loop-invariant pieces of trilinearly_interpolate (:60-71) and of tsdf_value_at, same float expressions.
struct TriConst {
    float max_x, max_y, max_z;        // voxel_grid_size * voxel_size
    float clamp_x, clamp_y, clamp_z;  // max - voxel_size / 10
    InvDiv dx, dy, dz;
    uint32_t row, plane;              // X, X*Y
};
__host__ __device__ inline TriConst make_tri_const(const Geom &g) {
    TriConst c;
    c.max_x = g.X * g.vs.x;
    c.max_y = g.Y * g.vs.y;
    c.max_z = g.Z * g.vs.z;
    c.clamp_x = c.max_x - (g.vs.x / 10.0f);
    c.clamp_y = c.max_y - (g.vs.y / 10.0f);
    c.clamp_z = c.max_z - (g.vs.z / 10.0f);
    c.dx = {g.vs.x, 1.0f / g.vs.x};
    c.dy = {g.vs.y, 1.0f / g.vs.y};
    c.dz = {g.vs.z, 1.0f / g.vs.z};
    c.row = g.X;
    c.plane = g.X * g.Y;  // X, Y <= 65535
    return c;
}

struct RayParams {
    F3 origin;
    Mat33 rot;
    Mat33 kinv;
    F3 space_min;
    F3 space_max;
    uint32_t width, height;
    uint32_t own_lo, own_hi;  // slab ownership (planes of the lower trilinear tap)
    uint32_t seg_len;         // > 0: blockIdx.z handles samples [z*seg_len, (z+1)*seg_len) and writes records
    uint32_t slab_ranges;     // > 0 (slabs): blockIdx.z handles that part of each ray's own stretch through the slab
    uint32_t tile_map;        // which image tiles an XCD gets (see process_ray_kernel): 0 every eighth tile, 1 one contiguous eighth of the image, 2 one 5x5-tile block per block row
    uint32_t range_order;     // order in which the sample ranges are dispatched (see process_ray_kernel): 0 ascending, 1 descending (default), 2 last, first, then descending
    TriConst tc;              // loop-invariant pieces of the interpolation, formed once on the host (same IEEE operations)
    // entry bound of this view (EntryParams, common.hpp): one word per 16 x 16 tile + the on/off word at [ztile_count]; nullptr = none
    const uint32_t *ztile;
    uint32_t ztile_pitch, ztile_count;
};

// Division by a loop-invariant voxel edge.  The reference divides (IEEE, correctly rounded); when FASTDIV is
// set the quotient is formed as q0 = a*y, r = fma(-b, q0, a), q = fma(r, y, q0) with y = RN(1/b).  That
// sequence is used ONLY after volume.hip has checked it against the IEEE quotient for EVERY finite fp32
// numerator with |a| >= kFastDivMin for this very b (verify_fast_division, ~2^32 cases per voxel
// edge, a few ms once per volume), so on that domain it is the same function in three instructions instead
// of the ~14 of a full fp32 division; numerators outside the domain take the IEEE division.
template <bool FASTDIV>
__device__ inline float div_by(float a, const InvDiv &d) {
    if (FASTDIV) {
        const float mag = fabsf(a);
        const bool verified = mag >= kFastDivMin && mag < INFINITY;  // the verified domain
        const float q0 = a * d.y;
        const float r = __builtin_fmaf(-d.b, q0, a);
        float q = __builtin_fmaf(r, d.y, q0);
        // the IEEE sequence sits behind a wave-uniform branch: left to itself the compiler computes both and selects
        if (__builtin_expect(__ballot(!verified) != 0ull, 0)) {
            if (!verified) q = a / d.b;
        }
        return q;
    }
    return a / d.b;
}

// trilinearly_interpolate (src/RayCaster/GPURaycaster.cu:53-124) with voxel_for_point, centre_of_voxel_at and
// tsdf_value_at (src/TSDF/TSDF_utilities.cu:10-53) inlined.  For SLAB, samples whose lower tap plane is not
// owned are not evaluated (owned=false, result NaN).
template <bool SLAB, bool STATS, bool FASTDIV>
__device__ inline float trilinear(float px, float py, float pz, const float *__restrict__ dist, const Geom &g,
                                  const TriConst &tc, const RayParams &rp, bool &owned,
                                  unsigned int *__restrict__ touched) {
    float ax = px, ay = py, az = pz;
    if (px >= tc.max_x) ax = tc.clamp_x;
    if (py >= tc.max_y) ay = tc.clamp_y;
    if (pz >= tc.max_z) az = tc.clamp_z;
    if (px < 0.0f) ax = 0.0f;
    if (py < 0.0f) ay = 0.0f;
    if (pz < 0.0f) az = 0.0f;

    // voxel_for_point (src/TSDF/TSDF_utilities.cu:45-53)
    int vx = f2i_sat(floorf(div_by<FASTDIV>(ax, tc.dx)));
    int vy = f2i_sat(floorf(div_by<FASTDIV>(ay, tc.dy)));
    int vz = f2i_sat(floorf(div_by<FASTDIV>(az, tc.dz)));

    owned = true;
    if (vx < 0 || vy < 0 || vz < 0 || (uint32_t)vx >= g.X || (uint32_t)vy >= g.Y || (uint32_t)vz >= g.Z) {
        return NAN;  // the reference also printf's here (:78)
    }

    // centre_of_voxel_at with its default zero offset (src/TSDF/TSDF_utilities.cu:10-17)
    float ccx = (vx + 0.5f) * g.vs.x + 0.0f;
    float ccy = (vy + 0.5f) * g.vs.y + 0.0f;
    float ccz = (vz + 0.5f) * g.vs.z + 0.0f;

    int lx = (px < ccx) ? vx - 1 : vx;
    int ly = (py < ccy) ? vy - 1 : vy;
    int lz = (pz < ccz) ? vz - 1 : vz;
    lx = max(lx, 0);
    ly = max(ly, 0);
    lz = max(lz, 0);

    if (SLAB) {
        if (!((uint32_t)lz >= rp.own_lo && (uint32_t)lz < rp.own_hi)) {
            owned = false;
            return NAN;
        }
    }

    float lcx = (lx + 0.5f) * g.vs.x + 0.0f;
    float lcy = (ly + 0.5f) * g.vs.y + 0.0f;
    float lcz = (lz + 0.5f) * g.vs.z + 0.0f;
    float u = div_by<FASTDIV>(px - lcx, tc.dx);
    float v = div_by<FASTDIV>(py - lcy, tc.dy);
    float w = div_by<FASTDIV>(pz - lcz, tc.dz);

    // tsdf_value_at clamps each tap to the grid (:31-33): lower is in range, so only lower+1 can be
    // clamped, to lower itself (0 <= lower <= size-1 <= 65534: the uint16_t parameters never wrap).
    const uint32_t ox = ((uint32_t)lx + 1 < g.X) ? 1u : 0u;
    const uint32_t oy = ((uint32_t)ly + 1 < g.Y) ? tc.row : 0u;
    const uint32_t oz = ((uint32_t)lz + 1 < g.Z) ? tc.plane : 0u;
    const float *b000 = dist + ((size_t)tc.plane * ((uint32_t)lz - g.z_store_begin) + (__umul24(tc.row, (uint32_t)ly) + (uint32_t)lx));   // (X, Y < 2^16: 24-bit multiply, X * Y < 2^32)
    if (STATS) {
        const size_t gi = (size_t)tc.plane * (uint32_t)lz + (size_t)tc.row * (uint32_t)ly + (uint32_t)lx;
        const uint32_t offs[8] = {0, oz, oy, oy + oz, ox, ox + oz, ox + oy, ox + oy + oz};
        for (int i = 0; i < 8; i++) {
            size_t q = gi + offs[i];
            atomicOr(&touched[q >> 5], 1u << (q & 31));
        }
    }
    float c000 = b000[0];
    float c001 = b000[oz];
    float c010 = b000[oy];
    float c011 = b000[oy + oz];
    float c100 = b000[ox];
    float c101 = b000[ox + oz];
    float c110 = b000[ox + oy];
    float c111 = b000[ox + oy + oz];

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

// can_intersect_in_dimension (src/RayCaster/GPURaycaster.cu:138-181)
__device__ inline bool can_intersect_in_dimension(float space_min, float space_max, float origin, float direction,
                                                  float &near_t, float &far_t) {
    bool can_intersect = true;
    if (direction == 0) {
        if (origin < space_min || origin > space_max) can_intersect = false;
    } else {
        float dmin = (space_min - origin) / direction;
        float dmax = (space_max - origin) / direction;
        if (dmin > dmax) {
            float t = dmin;
            dmin = dmax;
            dmax = t;
        }
        if (dmin > near_t) near_t = dmin;
        if (dmax < far_t) far_t = dmax;
        if (near_t > far_t) can_intersect = false;
        else if (far_t < 0) can_intersect = false;
    }
    return can_intersect;
}

// compute_near_and_far_t (src/RayCaster/GPURaycaster.cu:197-251)
__device__ inline bool compute_near_and_far_t(const F3 &o, const F3 &d, const F3 &smin, const F3 &smax,
                                              float &near_t, float &far_t) {
    bool intersects = false;
    if (o.x >= smin.x && o.x <= smax.x && o.y >= smin.y && o.y <= smax.y && o.z >= smin.z && o.z <= smax.z) {
        near_t = 0;
        float x_t = NAN, y_t = NAN, z_t = NAN;
        if (d.x > 0) x_t = (smax.x - o.x) / d.x; else if (d.x < 0) x_t = (smin.x - o.x) / d.x;
        if (d.y > 0) y_t = (smax.y - o.y) / d.y; else if (d.y < 0) y_t = (smin.y - o.y) / d.y;
        if (d.z > 0) z_t = (smax.z - o.z) / d.z; else if (d.z < 0) z_t = (smin.z - o.z) / d.z;
        if (x_t < y_t) {
            if (x_t < z_t) far_t = x_t; else far_t = z_t;
        } else {
            if (y_t < z_t) far_t = y_t; else far_t = z_t;
        }
        intersects = true;
    } else {
        near_t = -INFINITY;
        far_t = INFINITY;
        if (can_intersect_in_dimension(smin.x, smax.x, o.x, d.x, near_t, far_t) &&
            can_intersect_in_dimension(smin.y, smax.y, o.y, d.y, near_t, far_t) &&
            can_intersect_in_dimension(smin.z, smax.z, o.z, d.z, near_t, far_t)) {
            intersects = true;
        }
    }
    return intersects;
}

// ---- process_ray (src/RayCaster/GPURaycaster.cu:265-377) ---------------------------------------------------
//
// What the reference's loop does, restated so that it parallelises without changing a bit:
//
//  * Its parameter t only ever takes the values T[0] = 0, T[k+1] = T[k] + step (float additions), the same for
//    every ray of a launch, until a sample is <= 0.  The host builds that table with the same additions
//    (volume.hip: build_t_table) and the kernel stages it in LDS, so sample k of any ray is at T[k] without
//    replaying k additions (Q9 is kept: the values ARE the repeated float sums).
//  * Sample k is evaluated iff k < k_end, k_end = min(4402, first k >= 1 with T[k] >= max_t): that is the
//    `t >= max_t` test and the `count++ > 4400` cap of the reference (Q8), hoisted out of the loop.
//  * The first k with tsdf <= 0 ends the ray; t is refined from T[k] exactly as the reference does (Q7).
//
// Exact empty-space skipping (SKIP).  A sample whose 8 taps are all safely positive (> occ.tau) and whose
// weights lie in [0,1] up to rounding cannot be <= 0, so the reference passes it doing nothing.  When sample k
// lies in an interior brick whose occupancy flag is clear, k jumps to the brick's exit.  The flag covers the
// brick grown by kBrickGrow = 2 voxels: one voxel for the reach of the taps, one for the slack of the
// approximate arithmetic that locates the sample and counts the samples to the exit (errors of a few 1e-3
// voxel; the jump may overshoot the face by < 1 step, i.e. < 0.25 voxel).  Bricks touching the grid boundary
// are never skipped (there the weights can leave [0,1]: Q10).
//
// The same argument one level down, inside occupied interior bricks: the sample at p interpolates the 8 voxels
// of its dual cell (lower = floor(p/vs - 1/2)).  If p is at least eps away from every face of that cell the
// cheap arithmetic here and the reference's exact arithmetic agree on the cell; if its 8 values are all > tau
// neither this sample nor the following ones that stay inside the cell shrunk by eps can be <= 0, so k jumps
// to the cell's exit.  Otherwise the sample is interpolated exactly -- from the 8 values just gathered.
//
// How the march is scheduled on the machine is described at process_ray_kernel / process_ray_tail_kernel below.
constexpr int kMaxSamples = 4402;          // src/RayCaster/GPURaycaster.cu:369
constexpr int kTableLen = kMaxSamples + 2;  // T[0..4402] is read
constexpr int kDone = 0x7fffffff;
// Cell-level positivity test.  In a cell the sample is located in EXACTLY (>= eps from its faces, all 8 voxels exist) the
// reference's weights u, v, w lie strictly inside (0,1), so every product c_i * w_i of its sum is >= 0 in fp32 and the
// term with the largest weight (>= 1/8) is >= c_i / 8: with all eight values above this (normal-range) threshold the
// value the reference computes is > 0 whatever the rounding.  No margin like occ.tau is needed here -- that one covers
// the APPROXIMATE location used at brick level.
constexpr float kCellPositive = 1.0e-30f;
constexpr int kTailLanesDefault = 4;        // lanes per queue entry in the tail kernel (the variant compiled with the group width fixed)
// Defaults of the schedule (common.hpp: Tuning, read from the environment in volume.hip): 6 sample ranges / 22 passes before a wave hands
// over (round 3, re-tuned with the long waves dispatched first and the runs of free space merged; round 2: 5 / 24 with the ranges
// dispatched far to near; round 1: 6 / 18); 2 560 workgroups of the tail kernel; pieces of at least 64 samples.
// What is left of a ray's range when the pass budget runs out is queued in up to kTailPieces pieces of at least
// tail_piece_min() samples, so that a long stretch is marched by several groups of the tail kernel at once.
constexpr int kTailPieces = 16;
static int ray_segments() { return tuning().ray_segments; }
static int tail_lanes() { return tuning().ray_tail_lanes; }
static int tail_grid() { return tuning().ray_tail_grid; }
// Parts a ray's stretch through a slab is cut into: in proportion to the slab's share of the grid (a whole volume
// uses ray_segments() ranges), at least 2.
static int slab_ray_ranges(const tsdf_volume *v) {
    if (tuning().ray_slab_ranges) return tuning().ray_slab_ranges;
    const uint32_t planes = v->z_end - v->z_begin, Z = v->g.Z ? v->g.Z : 1;
    const int n = (int)(((uint64_t)ray_segments() * planes + Z - 1) / Z);
    return n < 2 ? 2 : (n > 64 ? 64 : n);
}
static int tail_piece_min() { return tuning().ray_tail_piece; }
static int trip_budget() { return tuning().ray_trip_budget; }

// Per-ray constants of the skipping arithmetic.  Everything here is APPROXIMATE on purpose (hardware
// reciprocals, no care for rounding): it only decides how far k may jump, and the slack of the occupancy
// argument (one voxel at brick level, eps at cell level) covers its error.
struct SkipCtx {
    float inv_vx, inv_vy, inv_vz;  // ~ 1 / voxel size
    float inv_step;                // ~ 1 / step
    float tx, ty, tz;              // ~ voxel size / dir, signed: t needed to cross one voxel (+inf for a zero component)
    float posx, posy, posz;        // 1 when the ray moves towards +axis (or not at all on that axis), else 0
    float su, sv, sw;              // ~ |step * dir / voxel size| : movement per sample in cell units, per axis
    float eps;                     // guard band at the dual-cell faces, in voxels
    bool skip_ok;                  // skipping allowed for this ray (short enough steps)
};

// Samples (>= 1) from the one at voxel coordinate (fx,fy,fz) until the ray leaves the axis-aligned box
// [lo, hi) given in voxel units.  x = (exit parameter) / step: samples k .. k + ceil(x) - 1 are inside the box,
// sample k + ceil(x) is the first one beyond it.  ALL = true returns ceil(x) (every inside sample; the estimate of x is
// good to ~1e-2 voxel, which the one-voxel slack of the brick-level flags absorbs), ALL = false returns floor(x) and so
// keeps up to one step of margin (cell-level boxes, whose only slack is eps).
template <bool ALL>
__device__ inline int samples_until(float ax, float ay, float az, const SkipCtx &c) {
    // a = signed distance (voxels) from the sample to the face of the box the ray leaves through on each axis: it has the
    // sign of the direction component, like t per voxel, so the products are the (non-negative) exit parameters
    const float ex = ax * c.tx, ey = ay * c.ty, ez = az * c.tz;
    // fminf ignores a NaN (0 * inf when the ray runs inside a face of the box)
    const float x = fminf(ex, fminf(ey, ez)) * c.inv_step;
    return (int)fminf(fmaxf(ALL ? ceilf(x) : x, 1.0f), 8192.0f);
}
// ... for the axis-aligned cube [lo, lo + size)^3 given by its integer corner: the exit face is lo + size on the axes the
// ray ascends, lo on the others.
template <bool ALL>
__device__ inline int samples_to_exit(float fx, float fy, float fz, const SkipCtx &c, int lox, int loy, int loz, float size) {
    return samples_until<ALL>(__builtin_fmaf(size, c.posx, (float)lox) - fx, __builtin_fmaf(size, c.posy, (float)loy) - fy,
                              __builtin_fmaf(size, c.posz, (float)loz) - fz, c);
}



// Look-ahead after an evaluated sample.  Inside one dual cell the interpolant f is trilinear in the cell coordinates
// (u,v,w), so df/du lies between the smallest and the largest difference along the cell's four x edges (it is a convex combination
// of them), same for v and w.  A sample moves the ray by (du, dv, dw) -- signed -- so one sample further on, still inside the cell, f has
// changed by at least  L = min_e(du * Dx_e) + min_e(dv * Dy_e) + min_e(dw * Dz_e):  n samples on it is >= val + n * L.  (Round 1-2
// used |du| * max_e |Dx_e| + ...: the same number for a ray that descends onto a surface, but a ray that skims past a silhouette
// spends half its way RECEDING from it -- f grows -- and was still evaluated sample by sample.)  The sample just evaluated is `val` > 0
// (the reference's fp32 value); the values the reference would compute for the next samples differ from f at the exact positions by
// its rounding (a few 1e-6 * largest |corner|; 2e-5 is allowed) and the positions are known to 2*eps in each coordinate, the step
// to 1 % (the approximate reciprocals; T[k+1] - T[k] is the step to 5e-4).  So while  val - margin - n * drop > 0 (drop: the bound with that 1 % applied term by term, below)
// sample k+n cannot be <= 0.  Returns that n, at most `limit` (the samples known to stay inside the cell); 0 when anything is NaN
// or infinite.
struct CellBound {   // what the look-ahead needs of a cell and a ray: the same for every sample evaluated inside the cell
    float margin, inv_drop;
    bool finite;
};
__device__ inline CellBound cell_bound(float c000, float c100, float c010, float c110, float c001, float c101, float c011, float c111,
                                       const SkipCtx &sc) {
    const float x0 = c100 - c000, x1 = c110 - c010, x2 = c101 - c001, x3 = c111 - c011;
    const float y0 = c010 - c000, y1 = c110 - c100, y2 = c011 - c001, y3 = c111 - c101;
    const float z0 = c001 - c000, z1 = c101 - c100, z2 = c011 - c010, z3 = c111 - c110;
    const float rx = fmaxf(fmaxf(fabsf(x0), fabsf(x1)), fmaxf(fabsf(x2), fabsf(x3)));
    const float ry = fmaxf(fmaxf(fabsf(y0), fabsf(y1)), fmaxf(fabsf(y2), fabsf(y3)));
    const float rz = fmaxf(fmaxf(fabsf(z0), fabsf(z1)), fmaxf(fabsf(z2), fabsf(z3)));
    const float rsum = (rx + ry) + rz;
    CellBound cb;
    cb.margin = 4.0f * sc.eps * rsum + 2.0e-5f * (fabsf(c000) + rsum);   // |corner| <= |c000| + rsum
    // the ray's movement per sample in cell units, signed (posx = 1 when it moves towards +x)
    const float du = sc.posx != 0.0f ? sc.su : -sc.su, dv = sc.posy != 0.0f ? sc.sv : -sc.sv, dw = sc.posz != 0.0f ? sc.sw : -sc.sw;
    const float lx = fminf(fminf(du * x0, du * x1), fminf(du * x2, du * x3));
    const float ly = fminf(fminf(dv * y0, dv * y1), fminf(dv * y2, dv * y3));
    const float lz = fminf(fminf(dw * z0, dw * z1), fminf(dw * z2, dw * z3));
    // the most f can fall per sample.  The per-sample movements su, sv, sw are known to 1 %: a falling term may really be 1.01 x
    // as large, a rising one only 0.99 x -- -(1.01 N + 0.99 P) with N / P the sums of the negative / positive terms, written as
    // 1.01 (-(N + P)) + 0.02 P so that rising terms cannot cancel more of the fall than they are sure to
    const float rising = (fmaxf(lx, 0.0f) + fmaxf(ly, 0.0f)) + fmaxf(lz, 0.0f);
    const float drop = fmaxf(__builtin_fmaf(1.01f, -((lx + ly) + lz), 0.02f * rising), 0.0f);
    cb.inv_drop = __builtin_amdgcn_rcpf(drop);   // (inf when f cannot fall: the limit applies)
    cb.finite = rsum < INFINITY;                 // (an infinite corner makes rsum infinite or NaN)
    return cb;
}
__device__ inline int lookahead_in_cell(float val, const CellBound &cb, int limit) {
    const float x = 0.99f * (val - cb.margin) * cb.inv_drop;
    // (a NaN corner makes val, hence x, NaN -- fmaxf / fminf would drop it from the slopes)
    if (!cb.finite || !(x > 0.0f)) return 0;
    return (int)fminf(x, (float)limit);
}
__device__ inline int lipschitz_lookahead(float val, float c000, float c100, float c010, float c110, float c001, float c101, float c011,
                                          float c111, const SkipCtx &sc, int limit) {
    return lookahead_in_cell(val, cell_bound(c000, c100, c010, c110, c001, c101, c011, c111, sc), limit);
}

// Diagnostics build (-DTSDF_DIAG_RAY_MIX, tools/dbg_ray_mix.py): what the passes of the two march kernels are spent on.
// [0..7] bulk kernel, per lane and pass: 0 block jump (reach), 1 cell-brick jump, 2 another slab's cell, 3 cell of positive voxels,
// 4 evaluated: hit, 5 evaluated: look-ahead to the cell's exit, 6 evaluated: look-ahead short of it, 7 the reference's full path;
// [8..15] the same for the tail kernel's lanes, [16..23] for lane 0 of its groups; [24..27] bulk passes by lanes active (1-4, 5-16,
// 17-32, 33-64), [28..31] the same counting only the waves' passes 12 and later.
#ifdef TSDF_DIAG_RAY_MIX
__device__ unsigned long long g_ray_mix[64];
#define RAY_MIX(slot) atomicAdd(&g_ray_mix[(slot)], 1ull)
#else
#define RAY_MIX(slot) ((void)0)
#endif
struct MixSlot { int base; };

// ---- one sample of one ray ------------------------------------------------------------------------------------
struct RayState {
    float dx, dy, dz;  // direction (not normalised: Q6)
    float sx, sy, sz;  // start point in grid coordinates (:306)
};
// The per-ray part of the skipping arithmetic.
template <bool SKIP>
__device__ inline void set_ray(SkipCtx &sc, const RayState &r, float step_size, const Geom &g) {
    sc.tx = r.dx != 0 ? g.vs.x * __builtin_amdgcn_rcpf(r.dx) : INFINITY;
    sc.ty = r.dy != 0 ? g.vs.y * __builtin_amdgcn_rcpf(r.dy) : INFINITY;
    sc.tz = r.dz != 0 ? g.vs.z * __builtin_amdgcn_rcpf(r.dz) : INFINITY;
    sc.posx = r.dx < 0 ? 0.0f : 1.0f; sc.posy = r.dy < 0 ? 0.0f : 1.0f; sc.posz = r.dz < 0 ? 0.0f : 1.0f;
    sc.su = fabsf(r.dx) * step_size * sc.inv_vx; sc.sv = fabsf(r.dy) * step_size * sc.inv_vy; sc.sw = fabsf(r.dz) * step_size * sc.inv_vz;
    // one step must stay well inside the one-voxel slack on every axis
    sc.skip_ok = SKIP && fabsf(r.dx) * step_size < 0.25f * g.vs.x && fabsf(r.dy) * step_size < 0.25f * g.vs.y &&
                 fabsf(r.dz) * step_size < 0.25f * g.vs.z;
}

// The hit point for a sample with tsdf <= 0 at parameter t (:336-350), previous_tsdf == trunc (Q7): the refined parameter
// (refine_t, :338-341), then the point on the ray (hit_point, :344-347).  A slab ships the refined parameter; the merge forms the
// point with the same expressions on the same values (tsdf_merge_hits_device).
__device__ inline float refine_t(float t, float tsdf, float previous_tsdf, float step_size) {
    float th = t;
    if (tsdf < 0) {
        th = th - step_size;
        th = th + (previous_tsdf / (previous_tsdf - tsdf)) * step_size;
    }
    return th;
}
__device__ inline void hit_point(float th, const RayState &r, const RayParams &rp, float &ix, float &iy, float &iz) {
    ix = ((th * r.dx) + r.sx) + rp.space_min.x;
    iy = ((th * r.dy) + r.sy) + rp.space_min.y;
    iz = ((th * r.dz) + r.sz) + rp.space_min.z;
}
__device__ inline void refine_hit(float t, float tsdf, float previous_tsdf, float step_size, const RayState &r,
                                  const RayParams &rp, float &ix, float &iy, float &iz) {
    hit_point(refine_t(t, tsdf, previous_tsdf, step_size), r, rp, ix, iy, iz);
}

// Ray set-up: direction and start point (ray_geometry), and the range of sample indices [k_first, k_end) of the
// sample range [k_lo, k_hi] that the reference's loop evaluates unless it hits earlier (setup_ray).  T = the staged table.
// compute_ray_direction_at_pixel (:24-44); f3_normalise is a no-op (by-value argument): Q6
__device__ inline F3 ray_direction(int imx, int imy, const RayParams &rp) {
    uint16_t pix_x = (uint16_t)imx, pix_y = (uint16_t)imy;
    float rcx = pix_x * rp.kinv.m11 + pix_y * rp.kinv.m12 + rp.kinv.m13;
    float rcy = pix_x * rp.kinv.m21 + pix_y * rp.kinv.m22 + rp.kinv.m23;
    float rcz = pix_x * rp.kinv.m31 + pix_y * rp.kinv.m32 + rp.kinv.m33;
    F3 dir;
    dir.x = rp.rot.m11 * rcx + rp.rot.m12 * rcy + rp.rot.m13 * rcz;
    dir.y = rp.rot.m21 * rcx + rp.rot.m22 * rcy + rp.rot.m23 * rcz;
    dir.z = rp.rot.m31 * rcx + rp.rot.m32 * rcy + rp.rot.m33 * rcz;
    return dir;
}
// the start point in grid coordinates (:306) from the ray's near parameter
__device__ inline RayState ray_from_near(const F3 &dir, float near_t, const RayParams &rp) {
    const float sx = ((near_t * dir.x) + rp.origin.x) - rp.space_min.x;
    const float sy = ((near_t * dir.y) + rp.origin.y) - rp.space_min.y;
    const float sz = ((near_t * dir.z) + rp.origin.z) - rp.space_min.z;
    return {dir.x, dir.y, dir.z, sx, sy, sz};
}
__device__ inline bool ray_geometry(int imx, int imy, bool in_image, const RayParams &rp, RayState &ray, float &max_t, float *near_out = nullptr) {
    const F3 dir = ray_direction(imx, imy, rp);

    float near_t = 0.f, far_t = 0.f;
    bool intersects = in_image && compute_near_and_far_t(rp.origin, dir, rp.space_min, rp.space_max, near_t, far_t);

    ray = ray_from_near(dir, near_t, rp);
    max_t = far_t - near_t;
    if (near_out) *near_out = near_t;
    return intersects;
}

template <bool SLAB>
__device__ inline void setup_ray(int imx, int imy, bool in_image, int k_lo, int k_hi, const float *Ts, const int t_off, const RayParams &rp,
                                 const Geom &g, float step_size, RayState &ray, int &k_first, int &k_end, float *near_out = nullptr) {
    // Ts[k + t_off] = T[k] for k_lo <= k <= k_hi (the staged part of the table)
    auto T = [&](int k) { return Ts[k + t_off]; };
    float max_t;
    const bool intersects = ray_geometry(imx, imy, in_image, rp, ray, max_t, near_out);
    const float sz = ray.sz;
    const F3 dir = {ray.dx, ray.dy, ray.dz};

    // samples 0 .. k_end-1 are evaluated unless one of them is <= 0
    k_end = 0;
    if (intersects) {
        // smallest k in [max(k_lo,1), k_hi] with T[k] >= max_t (k_hi when there is none): only this range's
        // part of the table is staged, and only this range's samples are marched.  T[k] is k * step up to the
        // rounding of its k additions (under one sample over the whole table), so the answer lies within a few
        // entries of max_t / step: the bisection starts from that window when it brackets the answer, from the whole
        // range otherwise (NaN / infinite max_t included).
        int lo = max(k_lo, 1), hi = k_hi;
        {
            const int c = f2i_sat(max_t * __builtin_amdgcn_rcpf(step_size));
            const int a = max(lo, min(hi, c - 4)), b = min(hi, max(lo, c + 4));
            const bool below = a == lo || T(a - 1) < max_t, above = b == hi || T(b) >= max_t;
            if (below && above) {
                lo = a;
                hi = b;
            }
        }
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (T(mid) >= max_t) hi = mid; else lo = mid + 1;
        }
        k_end = lo;
        // the range starts beyond the ray's last sample (k_lo >= 1 and T[k_lo] >= max_t): nothing to do
        if (k_lo >= 1 && T(k_lo) >= max_t) k_end = 0;
    }

    // A slab only ever evaluates samples whose lower tap plane it owns; along a ray those occupy one interval
    // of sample indices (z is linear in t).  Clip the range to a conservative superset of that interval
    // -- two voxels and three samples of slack -- instead of hopping through the rest of the volume.
    k_first = k_lo;
    if (SLAB && k_end > 0) {
        const float za = ((float)rp.own_lo - 2.0f) * g.vs.z, zb = ((float)rp.own_hi + 2.0f) * g.vs.z;  // grid mm
        if (dir.z != 0.0f) {
            const float r = __builtin_amdgcn_rcpf(dir.z);
            const float t0 = (za - sz) * r, t1 = (zb - sz) * r;
            const float ta = fminf(t0, t1), tb = fmaxf(t0, t1);
            const float ka = floorf(ta * __builtin_amdgcn_rcpf(step_size)) - 3.0f;
            const float kb = ceilf(tb * __builtin_amdgcn_rcpf(step_size)) + 3.0f;
            if (ka > (float)k_first) k_first = (int)fminf(ka, (float)kMaxSamples);
            if (kb < (float)k_end) k_end = (int)fmaxf(kb, 0.0f);
        } else if (sz < za || sz > zb) {
            k_end = 0;  // the ray runs parallel to the slab, outside it
        }
    }
}

__device__ inline SkipCtx make_skip_ctx(const Geom &g, float step_size) {
    SkipCtx sc;
    sc.inv_vx = __builtin_amdgcn_rcpf(g.vs.x); sc.inv_vy = __builtin_amdgcn_rcpf(g.vs.y); sc.inv_vz = __builtin_amdgcn_rcpf(g.vs.z);
    sc.inv_step = __builtin_amdgcn_rcpf(step_size);
    // guard band at the cell faces: the dual-cell index computed approximately (coordinates up to max(X,Y,Z) voxels, a
    // handful of roundings of 2^-24 relative each) must agree with the reference's exact one
    sc.eps = fmaxf(1.0e-3f, 2.0e-6f * (float)max(g.X, max(g.Y, g.Z)));
    sc.tx = sc.ty = sc.tz = INFINITY;
    sc.su = sc.sv = sc.sw = INFINITY;
    sc.posx = sc.posy = sc.posz = 1.0f;
    sc.skip_ok = false;
    return sc;
}

// Unfinished rays handed from process_ray_kernel to process_ray_tail_kernel, and the per-pixel result of the march.
struct TailQueue {
    uint2 *entries;        // {pixel index, k_end << 13 | next sample k}: samples [k, k_end) of that pixel's ray
    uint32_t *count;       // [0] entries appended
    uint32_t trip_budget;  // passes of the marching loop before a wave hands its unfinished rays over
    uint32_t lanes;        // lanes per ray in the tail kernel (power of two, 4..64)
    uint64_t *best;        // per pixel: {smallest sample index found <= 0 so far, that sample's value}: hit_word (kNoHitWord = none)
    int piece_min;         // shortest piece a handed-over stretch is cut into
    const uint32_t *order; // dispatch order learnt from the previous cast (nullptr: none): [slab of the launch][workgroup] -> last range + 1 << 24 | first range << 16 | workgroup whose tile to take, or kNoSlot
    uint8_t *heavy;        // for the next cast's order (nullptr: off): [range][workgroup] set by a wave that made a pass at all, and behind those the
                           // same again for waves that took heavy_passes passes or more
    uint32_t heavy_passes;
};
constexpr uint32_t kTailSignals = 4 + 2 * 256;   // TailQueue::count: [0] entries appended, [1], [2] unused, [3] the cell-parallel cast's listed bricks, [4 ..) its list's depth bins: counts, places taken (raycast_cells.hpp)
constexpr uint32_t kNoHit = 0xffffffffu;
constexpr uint32_t kNoSlot = 0xffffffffu;   // TailQueue::order: a slot of the launch with nothing to do
// the sample range the z-th slab of workgroups marches (rp.range_order: 0 near to far, 1 far to near, 2 last, first, then far to near)
__host__ __device__ inline uint32_t ray_range_of_slot(uint32_t z, uint32_t nz, uint32_t order) {
    return order == 0 ? z : order == 1 ? nz - 1u - z : (z == 0 ? nz - 1u : z == 1 ? 0u : nz - z);
}
constexpr uint64_t kNoHitWord = ~0ull;
// The per-pixel word of the march: the index of the sample in the high half, the bits of its value (<= 0) in the low half.  An
// unsigned 64-bit minimum orders by the index; two writers of the same index computed the same sample with the same expressions, so
// their low halves are equal.  resolve_*_kernel refine the hit from the stored value instead of gathering the sample's 8 voxels again
// (up to round 2 the word held the index only: resolve_normals_kernel fetched 20.9 MB to write 7.4 MB, 9.4 us).
__device__ inline uint64_t hit_word(int k, float tsdf) { return ((uint64_t)(uint32_t)k << 32) | (uint64_t)__float_as_uint(tsdf); }

// best[] is only ever lowered (atomicMin at device scope) during a march; a reader that sees an old, larger value merely
// misses a shortcut.  Read past this XCD's L2 so that hits found on the other XCDs show up.
__device__ inline uint32_t load_best(const uint64_t *p) {   // (the index half: little endian, the high word is the second one)
    return __hip_atomic_load(reinterpret_cast<const uint32_t *>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void lower_best(uint64_t *p, int k, float tsdf) { atomicMin(reinterpret_cast<unsigned long long *>(p), (unsigned long long)hit_word(k, tsdf)); }

#include "raycast_march.hpp"

// render_to_depth_image, per pixel (src/RayCaster/GPURaycaster.cu:575-579 with Camera::world_to_camera, src/Camera.cpp:287-294):
// camera-space z of the vertex (homogeneous product, divided by w), rounded half away from zero; no hit (NaN) -> 0.
__device__ inline uint16_t vertex_depth(float x, float y, float z, const Mat44 &ip) {
    const float cz = ((ip.m31 * x + ip.m32 * y) + ip.m33 * z) + ip.m34 * 1.0f;
    const float cw = ((ip.m41 * x + ip.m42 * y) + ip.m43 * z) + ip.m44 * 1.0f;
    const float r = roundf(cz / cw);
    return (r == r && r > 0.0f && r < 65536.0f) ? (uint16_t)r : (uint16_t)0;
}

// The vertex of a pixel from best[]: the ray's first sample <= 0 -- its index and the value the march computed for it -- is refined
// into the hit point as the reference does (process_ray :336-350); no hit -> NaN.
__device__ inline uint32_t resolve_pixel(uint32_t i, const Geom &g, const RayParams &rp, const float *__restrict__ t_table,
                                         const uint64_t *__restrict__ best, float &ix, float &iy, float &iz, float &th) {
    const uint64_t word = best[i];
    const uint32_t kb = (uint32_t)(word >> 32);
    ix = iy = iz = th = NAN;
    if (kb != kNoHit) {
        RayState ray;
        float max_t;
        (void)ray_geometry((int)(i % rp.width), (int)(i / rp.width), true, rp, ray, max_t);
        const float t = t_table[kb], step_size = t_table[1];
        th = refine_t(t, __uint_as_float((uint32_t)word), g.trunc, step_size);   // previous_tsdf == trunc (Q7)
        hit_point(th, ray, rp, ix, iy, iz);
    }
    return kb;
}

// The vertices of all pixels.  best[] is double buffered: a march lowers one copy, this kernel reads it and resets the OTHER
// one (consumed by the previous march's resolve) for the next march, together with the tail queue's counter -- so a
// pixel's word may be read by several workgroups (resolve_normals_kernel) without racing against its reset.
//   SLAB: out = 8-byte records {k, t} (tsdf_hit_record) for the min-k merge across slabs; otherwise packed float3 vertices.
//   depth != nullptr (whole volume): also, or instead (out == nullptr), the pixel's depth as render_to_depth_image forms it from the vertex
//   (vertex_depth: src/RayCaster/GPURaycaster.cu:575-579) -- the model image of the tracked loop without a vertex map in between.
template <bool SLAB>
__global__ __launch_bounds__(256) void resolve_hits_kernel(const Geom g, const RayParams rp,
                                                           const float *__restrict__ t_table, const uint64_t *__restrict__ best,
                                                           uint64_t *__restrict__ best_next, float *__restrict__ out,
                                                           uint32_t *__restrict__ reset, const Mat44 ip, uint16_t *__restrict__ depth) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // the queue's entries appended, the cell-parallel cast's listed bricks and its list's depth bins (TailQueue::count): every word, also by
    // the one workgroup of an image of a few pixels (kTailSignals is 516 words: an image of 7 x 8 pixels left the bins' upper half
    // and the places taken standing, and the next sorted list lost entries -- found by the extended fuzz, round 6)
    for (uint32_t j = i; j < kTailSignals; j += gridDim.x * blockDim.x) reset[j] = 0;
    if (i >= rp.width * rp.height) return;
    best_next[i] = kNoHitWord;
    float ix, iy, iz, th;
    const uint32_t kb = resolve_pixel(i, g, rp, t_table, best, ix, iy, iz, th);
    if (SLAB) {
        reinterpret_cast<uint2 *>(out)[i] = make_uint2(kb, __float_as_uint(th));
    } else {
        if (out) {
            out[(size_t)i * 3 + 0] = ix;
            out[(size_t)i * 3 + 1] = iy;
            out[(size_t)i * 3 + 2] = iz;
        }
        if (depth) depth[i] = vertex_depth(ix, iy, iz, ip);
    }
}

// Vertices and normals in one launch (whole volume): a workgroup resolves a 16x16 pixel tile plus the column to its right
// and the row below -- 289 pixels, one per thread of its five waves -- into LDS, then forms the normals as normals_kernel
// does (compute_normals, Q11) from those.
constexpr int kResolveThreads = 320;
__global__ __launch_bounds__(kResolveThreads) void resolve_normals_kernel(const Geom g, const RayParams rp,
                                                                          const float *__restrict__ t_table, const uint64_t *__restrict__ best,
                                                                          uint64_t *__restrict__ best_next, float *__restrict__ V,
                                                                          float *__restrict__ N, uint32_t *__restrict__ reset) {
    constexpr int kT = 16, kS = kT + 1;

    static_assert(kS * kS <= kResolveThreads, "one thread per pixel of the tile and its halo");
    __shared__ float vx[kS * kS], vy[kS * kS], vz[kS * kS];
    {   // the queue's entries appended, the cell-parallel cast's listed bricks (TailQueue::count)
        const uint32_t t_ = (blockIdx.y * gridDim.x + blockIdx.x) * kResolveThreads + threadIdx.x;
        for (uint32_t j = t_; j < kTailSignals; j += gridDim.x * gridDim.y * kResolveThreads) reset[j] = 0;   // (every word, however few workgroups: see resolve_hits_kernel)
    }
    const uint32_t x0 = blockIdx.x * kT, y0 = blockIdx.y * kT;
    // own pixels first (threads 0..255, row-major in the tile), then the halo column and row
    const uint32_t s_ = threadIdx.x;
    if (s_ < (uint32_t)(kS * kS)) {
        uint32_t lx, ly;
        if (s_ < (uint32_t)(kT * kT)) {
            lx = s_ & 15u; ly = s_ >> 4;
        } else if (s_ < (uint32_t)(kT * kT + kT)) {
            lx = kT; ly = s_ - kT * kT;          // right column
        } else {
            lx = s_ - (kT * kT + kT); ly = kT;   // bottom row, corner included
        }
        const uint32_t x = x0 + lx, y = y0 + ly;
        float ix = NAN, iy = NAN, iz = NAN;
        if (x < rp.width && y < rp.height) {
            const uint32_t i = y * rp.width + x;
            float th;
            (void)resolve_pixel(i, g, rp, t_table, best, ix, iy, iz, th);
            if (lx < (uint32_t)kT && ly < (uint32_t)kT) {   // this workgroup's own pixel
                best_next[i] = kNoHitWord;
                V[(size_t)i * 3 + 0] = ix;
                V[(size_t)i * 3 + 1] = iy;
                V[(size_t)i * 3 + 2] = iz;
            }
        }
        const uint32_t slot = ly * kS + lx;
        vx[slot] = ix; vy[slot] = iy; vz[slot] = iz;
    }
    __syncthreads();
    if (threadIdx.x >= (uint32_t)(kT * kT)) return;
    const uint32_t lx = threadIdx.x & 15u, ly = threadIdx.x >> 4, x = x0 + lx, y = y0 + ly;
    if (x >= rp.width || y >= rp.height) return;
    float nx = 0, ny = 0, nz = 0;
    if (y != rp.height - 1 && x != rp.width - 1) {
        const uint32_t a = ly * kS + lx, r = a + 1, b = a + kS;
        float v2x = vx[r] - vx[a], v2y = vy[r] - vy[a], v2z = vz[r] - vz[a];
        float v1x = vx[b] - vx[a], v1y = vy[b] - vy[a], v1z = vz[b] - vz[a];
        float cx = v1y * v2z - v1z * v2y;
        float cy = v1z * v2x - v1x * v2z;
        float cz = v1x * v2y - v1y * v2x;
        float l = sqrtf(cx * cx + cy * cy + cz * cz);
        nx = cx / l;
        ny = cy / l;
        nz = cz / l;
    }
    const size_t idx = (size_t)y * rp.width + x;
    N[idx * 3 + 0] = nx;
    N[idx * 3 + 1] = ny;
    N[idx * 3 + 2] = nz;
}

// compute_normals (src/RayCaster/GPURaycaster.cu:393-427): Q11
__global__ __launch_bounds__(256) void normals_kernel(uint32_t width, uint32_t height,
                                                      const float *__restrict__ V, float *__restrict__ N) {
    const uint32_t imx = blockIdx.x * 64 + (threadIdx.x & 63u);
    const uint32_t imy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (imx >= width || imy >= height) return;
    size_t idx = (size_t)imy * width + imx;
    float nx = 0, ny = 0, nz = 0;
    if (imy != height - 1 && imx != width - 1) {
        const float *a = V + idx * 3, *r = V + (idx + 1) * 3, *b = V + (idx + width) * 3;
        float v2x = r[0] - a[0], v2y = r[1] - a[1], v2z = r[2] - a[2];
        float v1x = b[0] - a[0], v1y = b[1] - a[1], v1z = b[2] - a[2];
        float cx = v1y * v2z - v1z * v2y;
        float cy = v1z * v2x - v1x * v2z;
        float cz = v1x * v2y - v1y * v2x;
        float l = sqrtf(cx * cx + cy * cy + cz * cz);
        nx = cx / l;
        ny = cy / l;
        nz = cz / l;
    }
    N[idx * 3 + 0] = nx;
    N[idx * 3 + 1] = ny;
    N[idx * 3 + 2] = nz;
}

// render_to_depth_image, per pixel (src/RayCaster/GPURaycaster.cu:575-579 with Camera::world_to_camera, src/Camera.cpp:287-294):
// camera-space z of the vertex (homogeneous product, divided by w), rounded half away from zero.
__global__ __launch_bounds__(256) void vertices_to_depth_kernel(uint32_t n_pixels, const float *__restrict__ V, const Mat44 ip,
                                                                uint16_t *__restrict__ depth) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    depth[i] = vertex_depth(V[(size_t)i * 3 + 0], V[(size_t)i * 3 + 1], V[(size_t)i * 3 + 2], ip);
}

// Per pixel, the record {k, t} with the smallest k among n_slabs gathered buffers (layout [slab][pixel]); ties cannot occur: a
// sample has exactly one owner.  The vertex is formed here, from the winner's refined parameter t, with the expressions of
// process_ray (:306, :344-347) on the pixel's own start point and direction -- which depend on the pixel and the pose only, so
// every rank computes the bits the owning rank would have (ray_geometry / hit_point are the functions the march itself uses).
// Why not t alone (4 bytes, min-t): with previous_tsdf == trunc (Q7) a hit at sample k refines to t in (T[k] - step, T[k]] in exact
// arithmetic, so smaller k <=> smaller t as long as |tsdf| is moderate; but set_distance_data can hold anything, a hugely negative
// sample refines to fl(T[k] - step), which may equal the T[k-1] of another slab's exact-zero hit: the order by t is not strict,
// and "no hit" would need a sentinel value of t.  The index k is the reference's own stopping criterion; it stays in the record.
__device__ inline void merged_vertex(const uint2 *__restrict__ hits, uint32_t n_slabs, uint32_t n_pixels, uint32_t i,
                                     const RayParams &rp, float &ix, float &iy, float &iz) {
    uint2 best = hits[i];
    for (uint32_t s = 1; s < n_slabs; s++) {
        const uint2 h = hits[(size_t)s * n_pixels + i];
        if (h.x < best.x) best = h;
    }
    ix = iy = iz = NAN;
    if (best.x != kNoHit) {
        RayState ray;
        float max_t;
        (void)ray_geometry((int)(i % rp.width), (int)(i / rp.width), true, rp, ray, max_t);
        hit_point(__uint_as_float(best.y), ray, rp, ix, iy, iz);
    }
}

__global__ __launch_bounds__(256) void merge_hits_kernel(const uint2 *__restrict__ hits, uint32_t n_slabs, const RayParams rp,
                                                         float *__restrict__ V) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, n_pixels = rp.width * rp.height;
    if (i >= n_pixels) return;
    float ix, iy, iz;
    merged_vertex(hits, n_slabs, n_pixels, i, rp, ix, iy, iz);
    V[(size_t)i * 3 + 0] = ix;
    V[(size_t)i * 3 + 1] = iy;
    V[(size_t)i * 3 + 2] = iz;
}

// The same select together with the normals (compute_normals, Q11) in one launch: a workgroup merges a 16x16 pixel tile
// plus the column to its right and the row below -- one pixel per thread of its five waves -- into LDS, like
// resolve_normals_kernel.
__global__ __launch_bounds__(kResolveThreads) void merge_hits_normals_kernel(const uint2 *__restrict__ hits, uint32_t n_slabs,
                                                                             const RayParams rp, float *__restrict__ V,
                                                                             float *__restrict__ N) {
    constexpr int kT = 16, kS = kT + 1;
    __shared__ float vx[kS * kS], vy[kS * kS], vz[kS * kS];
    const uint32_t width = rp.width, height = rp.height;
    const uint32_t x0 = blockIdx.x * kT, y0 = blockIdx.y * kT, n_pixels = width * height;
    const uint32_t s_ = threadIdx.x;
    if (s_ < (uint32_t)(kS * kS)) {
        uint32_t lx, ly;
        if (s_ < (uint32_t)(kT * kT)) {
            lx = s_ & 15u; ly = s_ >> 4;
        } else if (s_ < (uint32_t)(kT * kT + kT)) {
            lx = kT; ly = s_ - kT * kT;
        } else {
            lx = s_ - (kT * kT + kT); ly = kT;
        }
        const uint32_t x = x0 + lx, y = y0 + ly;
        float ix = NAN, iy = NAN, iz = NAN;
        if (x < width && y < height) {
            const uint32_t i = y * width + x;
            merged_vertex(hits, n_slabs, n_pixels, i, rp, ix, iy, iz);
            if (lx < (uint32_t)kT && ly < (uint32_t)kT) {
                V[(size_t)i * 3 + 0] = ix;
                V[(size_t)i * 3 + 1] = iy;
                V[(size_t)i * 3 + 2] = iz;
            }
        }
        const uint32_t slot = ly * kS + lx;
        vx[slot] = ix; vy[slot] = iy; vz[slot] = iz;
    }
    __syncthreads();
    if (threadIdx.x >= (uint32_t)(kT * kT)) return;
    const uint32_t lx = threadIdx.x & 15u, ly = threadIdx.x >> 4, x = x0 + lx, y = y0 + ly;
    if (x >= width || y >= height) return;
    float nx = 0, ny = 0, nz = 0;
    if (y != height - 1 && x != width - 1) {
        const uint32_t a = ly * kS + lx, r = a + 1, b = a + kS;
        float v2x = vx[r] - vx[a], v2y = vy[r] - vy[a], v2z = vz[r] - vz[a];
        float v1x = vx[b] - vx[a], v1y = vy[b] - vy[a], v1z = vz[b] - vz[a];
        float cx = v1y * v2z - v1z * v2y;
        float cy = v1z * v2x - v1x * v2z;
        float cz = v1x * v2y - v1y * v2x;
        float l = sqrtf(cx * cx + cy * cy + cz * cz);
        nx = cx / l;
        ny = cy / l;
        nz = cz / l;
    }
    const size_t idx = (size_t)y * width + x;
    N[idx * 3 + 0] = nx;
    N[idx * 3 + 1] = ny;
    N[idx * 3 + 2] = nz;
}

__global__ __launch_bounds__(256) void popcount_kernel(const unsigned int *__restrict__ words, size_t n,
                                                       unsigned long long *__restrict__ counter) {
    unsigned long long c = 0;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) c += __popc(words[i]);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63u) == 0 && c) atomicAdd(counter, c);
}

// LDS of a process_ray_kernel workgroup: T[0], T[1] and the table entries of one sample range (the whole table when the ranges
// are cut per ray: slabs, seg_len == 0)
static size_t ray_table_lds_bytes(const RayParams &rp, bool per_ray_ranges) {
    const size_t entries = (rp.seg_len && !per_ray_ranges) ? std::min<size_t>(kMaxSamples, rp.seg_len) + 1 : (size_t)kMaxSamples + 1;
    return (entries + 2 + 2) * sizeof(float);   // (+ the word at Ts[kTableLen + 1]: the fused launch's waves-left counter, whole table only)
}

static RayParams make_params(const tsdf_volume *v, uint32_t width, uint32_t height, const float pose[16],
                             const float kinv[9]) {
    RayParams rp;
    // get_vertices (src/RayCaster/GPURaycaster.cu:432-464): origin = camera.position(), rot = pose 3x3
    rp.origin = {pose[12], pose[13], pose[14]};
    rp.rot = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
    memcpy(&rp.kinv, kinv, sizeof(Mat33));
    const Geom &g = v->g;
    rp.space_min = g.offset;
    rp.space_max = {g.offset.x + g.phys.x, g.offset.y + g.phys.y, g.offset.z + g.phys.z};
    rp.width = width;
    rp.height = height;
    rp.own_lo = v->z_begin;
    rp.own_hi = v->z_end;
    rp.seg_len = 0;
    rp.slab_ranges = 0;
    rp.range_order = 0;
    rp.tile_map = 1;
    rp.tc = make_tri_const(g);
    rp.ztile = nullptr;
    rp.ztile_pitch = rp.ztile_count = 0;
    return rp;
}

static int launch_normals(uint32_t width, uint32_t height, const float *V, float *N, hipStream_t s) {
    dim3 grid((width + 63) / 64, (height + 3) / 4);
    hipLaunchKernelGGL(normals_kernel, grid, dim3(256), 0, s, width, height, V, N);
    TSDF_HIP(hipGetLastError(), "compute_normals failed");
    return TSDF_OK;
}

static int check_ray_args(const tsdf_volume *v, uint32_t width, uint32_t height, const float *pose, const float *kinv) {
    TSDF_REQUIRE(v && pose && kinv, "tsdf_raycast: null argument");
    // the reference's raycaster stores width/height as uint16_t (src/include/Raycaster.hpp:35-36)
    TSDF_REQUIRE(width > 0 && height > 0 && width <= 65535 && height <= 65535, "tsdf_raycast: bad image size");
    return TSDF_OK;
}

// The entry bound of a whole-volume ray cast (EntryParams, common.hpp): usable when the camera depth of a sample is near + t --
// K^-1 with last row (0, 0, 1) -- and the pose's 3 x 3 block has an inverse.  Fills `ep` and rp's fields and returns true, or leaves
// rp.ztile null.  The launch that fills the words is the reach summary's (occupancy_refresh(v, &ep)).
// The view's projection (world -> pixel) from the pose and K^-1 a cast is given: exists when the camera depth of a sample is its ray
// parameter (K^-1 with last row (0, 0, 1)) and both 3 x 3 blocks have inverses.  Used only to BOUND what can be seen where (the entry
// bound's tiles, the cell-parallel cast's pixel boxes): no result is computed with it.
static bool view_projection(const tsdf_volume *v, const RayParams &rp, EntryParams &ep) {
    const Geom &g = v->g;
    const Mat33 &ki = rp.kinv;
    // The ray's camera z must be 1 per unit of t.  (An inverse formed in fp32 -- the reference's Eigen .inverse(), the host Camera's -- leaves
    // 0.99999994 there for focal lengths like 525 / 400: a depth off by an ulp is far inside every margin taken below -- two voxels for
    // the entry bound, 0.05 px and 1e-5 of the depth for the cells' boxes -- and the projection carries the factor exactly.)
    const double w33 = ki.m33;
    if (ki.m31 != 0.0f || ki.m32 != 0.0f || !(std::fabs(w33 - 1.0) <= 1.0e-6)) return false;
    // pose = [R t; 0 0 0 1] (rp.rot, rp.origin), kinv = [a s c; e b d; 0 0 1]: the inverses in double
    const double R[3][3] = {{rp.rot.m11, rp.rot.m12, rp.rot.m13}, {rp.rot.m21, rp.rot.m22, rp.rot.m23}, {rp.rot.m31, rp.rot.m32, rp.rot.m33}};
    const double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                       R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
    if (!(std::fabs(det) > 1e-9) || !std::isfinite(det)) return false;
    double Ri[3][3];
    Ri[0][0] = (R[1][1] * R[2][2] - R[1][2] * R[2][1]) / det; Ri[0][1] = (R[0][2] * R[2][1] - R[0][1] * R[2][2]) / det; Ri[0][2] = (R[0][1] * R[1][2] - R[0][2] * R[1][1]) / det;
    Ri[1][0] = (R[1][2] * R[2][0] - R[1][0] * R[2][2]) / det; Ri[1][1] = (R[0][0] * R[2][2] - R[0][2] * R[2][0]) / det; Ri[1][2] = (R[0][2] * R[1][0] - R[0][0] * R[1][2]) / det;
    Ri[2][0] = (R[1][0] * R[2][1] - R[1][1] * R[2][0]) / det; Ri[2][1] = (R[0][1] * R[2][0] - R[0][0] * R[2][1]) / det; Ri[2][2] = (R[0][0] * R[1][1] - R[0][1] * R[1][0]) / det;
    const double t[3] = {rp.origin.x, rp.origin.y, rp.origin.z};
    memset(&ep, 0, sizeof(ep));
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) ep.r[i][j] = (float)Ri[i][j];
        ep.r[i][3] = (float)-(Ri[i][0] * t[0] + Ri[i][1] * t[1] + Ri[i][2] * t[2]);
    }
    {   // the largest absolute row sum of Ri' Ri bounds its largest eigenvalue, the square of Ri's 2-norm (exactly 1 for a rotation):
        // poses need not be rigid -- anything with an invertible block is cast -- and a brick's radius in the world is a distance
        // in the camera's frame only up to that factor (cell_cast_prepare_kernel)
        double bound = 0.0;
        for (int i = 0; i < 3; i++) {
            double row = 0.0;
            for (int j = 0; j < 3; j++) row += std::fabs(Ri[0][i] * Ri[0][j] + Ri[1][i] * Ri[1][j] + Ri[2][i] * Ri[2][j]);
            bound = std::max(bound, row);
        }
        ep.r_scale = (float)(std::sqrt(bound) * (1.0 + 1.0e-5));
        if (!std::isfinite(ep.r_scale)) return false;
    }
    const double kd = (double)ki.m11 * ki.m22 - (double)ki.m12 * ki.m21;
    if (!(std::fabs(kd) > 1e-12) || !std::isfinite(kd)) return false;
    // inverse of [a s c; e b d; 0 0 1], rows 1-2; a direction's camera z is w33: camera x and y count w33-fold (pixel = k * camera / camera.z)
    ep.k[0][0] = (float)(ki.m22 / kd * w33); ep.k[0][1] = (float)(-ki.m12 / kd * w33); ep.k[0][2] = (float)((ki.m12 * (double)ki.m23 - ki.m22 * (double)ki.m13) / kd);
    ep.k[1][0] = (float)(-ki.m21 / kd * w33); ep.k[1][1] = (float)(ki.m11 / kd * w33); ep.k[1][2] = (float)((ki.m21 * (double)ki.m13 - ki.m11 * (double)ki.m23) / kd);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++)
            if (!std::isfinite(ep.r[i][j])) return false;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++)
            if (!std::isfinite(ep.k[i][j])) return false;
    ep.vs = g.vs;
    ep.offset = g.offset;
    ep.width = rp.width; ep.height = rp.height;
    ep.tiles_x = (rp.width + kEntryTile - 1) / kEntryTile; ep.tiles_y = (rp.height + kEntryTile - 1) / kEntryTile;
    ep.slack_z = 2.0f * std::max(g.vs.x, std::max(g.vs.y, g.vs.z));
    ep.units_x = (v->occ.nbx + 3) / 4; ep.units_y = (v->occ.nby + 3) / 4; ep.units_z = (v->occ.nbz + 3) / 4;
    return true;
}
static bool prepare_entry_bound(tsdf_volume *v, RayParams &rp, EntryParams &ep, int &rc) {
    rc = TSDF_OK;
    if (!tuning().ray_entry_bound) return false;
    if (!view_projection(v, rp, ep)) return false;
    const size_t n_words = (size_t)ep.tiles_x * ep.tiles_y + 1;
    if (v->ztile_words != n_words) {   // (first use, or another image size: both copies start reset)
        if (v->ztile) (void)hipFree(v->ztile);
        v->ztile = nullptr;
        v->ztile_words = 0;
        if (hipMalloc((void **)&v->ztile, 2 * n_words * sizeof(uint32_t)) != hipSuccess) { rc = hip_fail(hipErrorOutOfMemory, "entry bound alloc"); return false; }
        std::vector<uint32_t> init(2 * n_words, kEntryFar);
        init[n_words - 1] = init[2 * n_words - 1] = 1u;
        if (hipMemcpyAsync(v->ztile, init.data(), init.size() * sizeof(uint32_t), hipMemcpyHostToDevice, v->stream) != hipSuccess ||
            hipStreamSynchronize(v->stream) != hipSuccess) { rc = hip_fail(hipErrorUnknown, "entry bound reset"); return false; }
        v->ztile_words = n_words;
        v->ztile_side = 0;
    }
    ep.ztile = v->ztile + (size_t)v->ztile_side * n_words;
    ep.ztile_next = v->ztile + (size_t)(1 - v->ztile_side) * n_words;
    v->ztile_side = 1 - v->ztile_side;
    rp.ztile = ep.ztile;
    rp.ztile_pitch = ep.tiles_x;
    rp.ztile_count = ep.tiles_x * ep.tiles_y;
    return true;
}
// occupancy_refresh for a whole-volume cast, with the view's entry bound when the camera allows one
static int refresh_for_cast(tsdf_volume *v, RayParams &rp) {
    EntryParams ep;
    int rc;
    const bool bound = prepare_entry_bound(v, rp, ep, rc);
    if (rc != TSDF_OK) return rc;
    return occupancy_refresh(v, bound ? &ep : nullptr);
}

#include "raycast_cells.hpp"

// Which kernels a cast takes (scheduling only: the same bits either way).  The cell-parallel cast needs the view's projection; it is
// left to the march kernels when the previous cast listed so many flagged bricks that marching is cheaper (arbitrary fields in which
// every cell is mixed; TSDF_RAY_CELLS = 0 never / 1 by that count (default) / 2 whenever the view allows).
// After a bulk change of the distances (set_distance_data, a loaded file: occ_dirty) the count the choice below goes by -- the bricks
// the previous cell-parallel cast listed -- says nothing about the new field, and an arbitrary field at 1024^3 can list four million
// bricks of 64 mixed cells each: minutes of pairs.  The flags are rebuilt at once (the cast would do that anyway), one workgroup counts
// the bricks with `fine` and `cell` set into the mirror, and the host waits for it: once per bulk change, never in a stream of frames.
// (scratch: [0] the sum so far, [1] workgroups done; the last one to finish writes the mirror and leaves both words zero again)
__global__ __launch_bounds__(1024) void count_cell_bricks_kernel(const OccGrid occ, uint32_t *__restrict__ mirror, uint32_t *__restrict__ scratch) {
    __shared__ uint32_t total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    uint32_t mine = 0;
    const size_t n = occ.fine_count();
    for (size_t b = (size_t)blockIdx.x * 1024 + threadIdx.x; b < n; b += (size_t)gridDim.x * 1024) mine += (occ.fine[b] && occ.cell[b]) ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63u) == 0) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (total) atomicAdd(&scratch[0], total);
        __threadfence();
        if (atomicAdd(&scratch[1], 1u) + 1u == gridDim.x) {
            __threadfence();
            *mirror = atomicExch(&scratch[0], 0u);
            scratch[1] = 0u;
        }
    }
}
static int launch_brick_count(tsdf_volume *v) {
    if (!v->cell_cast_host) {
        TSDF_HIP(hipHostMalloc((void **)&v->cell_cast_host, sizeof(uint32_t), hipHostMallocDefault), "brick count mirror alloc");
        *v->cell_cast_host = 0;
    }
    if (!v->cell_count_scratch) {
        TSDF_HIP(hipMalloc((void **)&v->cell_count_scratch, 2 * sizeof(uint32_t)), "brick count scratch alloc");
        TSDF_HIP(hipMemsetAsync(v->cell_count_scratch, 0, 2 * sizeof(uint32_t), v->stream), "brick count scratch reset");
    }
    const unsigned blocks = (unsigned)std::min<size_t>((v->occ.fine_count() + 65535) / 65536, 256);   // (64 bricks a thread)
    hipLaunchKernelGGL(count_cell_bricks_kernel, dim3(std::max(blocks, 1u)), dim3(1024), 0, v->stream, v->occ, v->cell_cast_host, v->cell_count_scratch);
    TSDF_HIP(hipGetLastError(), "brick count failed");
    return TSDF_OK;
}
static int count_after_bulk_change(tsdf_volume *v) {
    if (tuning().ray_cells == 0 || !v->occ_dirty) return TSDF_OK;
    int rc = occupancy_flags_refresh(v);
    if (rc != TSDF_OK) return rc;
    rc = launch_brick_count(v);
    if (rc != TSDF_OK) return rc;
    TSDF_HIP(hipStreamSynchronize(v->stream), "brick count failed");
    return TSDF_OK;
}

// (can: the view allows the cell-parallel cast at all -- a projection, sizes within the formats; returns whether the STATIC rules
// prefer it: the choice where nothing has been measured, and the bound on what a trial may cost, choose_cast)
static bool choose_cell_cast(const tsdf_volume *v, const RayParams &rp, EntryParams &ep, bool *can = nullptr, bool *trial_ok = nullptr) {
    const int mode = tuning().ray_cells;
    if (can) *can = false;
    if (trial_ok) *trial_ok = false;
    // (a list entry: 10 bits of each brick coordinate; a record of the cast: 13 bits of sample index;
    // ... and a brick's pairs are counted in 31 bits: 64 cells that each ask every pixel)
    if (mode == 0 || v->occ.fine_count() >= ((size_t)1 << 30) || v->occ.nbx > 1024u || v->occ.nby > 1024u || v->occ.nbz > 1024u ||
        (uint64_t)rp.width * rp.height > ((uint64_t)1 << 24) || !view_projection(v, rp, ep))
        return false;
    // A camera OUTSIDE the volume: every sample has a camera depth (= its ray parameter) of at least z_clip > 0, a cell at or behind the
    // camera plane holds none, and one that straddles the plane z_clip is bounded from its part in front.  Inside (or within a voxel of)
    // the volume z_clip = 0: samples start at the camera, a box that holds it asks every pixel, one beside it that reaches across the
    // camera plane is bounded from the side it lies on (project_box).
    {
        const Geom &g_ = v->g;
        const float o[3] = {rp.origin.x, rp.origin.y, rp.origin.z}, lo[3] = {g_.offset.x, g_.offset.y, g_.offset.z};
        const float hi[3] = {g_.offset.x + g_.phys.x, g_.offset.y + g_.phys.y, g_.offset.z + g_.phys.z};
        double d2 = 0.0;
        for (int a = 0; a < 3; a++) {
            const double d = std::max(0.0, std::max((double)lo[a] - o[a], (double)o[a] - hi[a]));
            d2 += d * d;
        }
        const float outside = (float)std::sqrt(d2), vs_max_ = std::max(g_.vs.x, std::max(g_.vs.y, g_.vs.z)), vs_min_ = std::min(g_.vs.x, std::min(g_.vs.y, g_.vs.z));
        // the longest direction vector of the image (a ray's parameter to the volume is at least outside / |direction|)
        double dmax = 0.0;
        const float px[5] = {0.0f, (float)(rp.width - 1), 0.0f, (float)(rp.width - 1), 0.5f * rp.width}, py[5] = {0.0f, 0.0f, (float)(rp.height - 1), (float)(rp.height - 1), 0.5f * rp.height};
        for (int c = 0; c < 5; c++) {
            const double rcx = px[c] * rp.kinv.m11 + py[c] * rp.kinv.m12 + rp.kinv.m13, rcy = px[c] * rp.kinv.m21 + py[c] * rp.kinv.m22 + rp.kinv.m23, rcz = 1.0;
            const double dx = rp.rot.m11 * rcx + rp.rot.m12 * rcy + rp.rot.m13 * rcz, dy = rp.rot.m21 * rcx + rp.rot.m22 * rcy + rp.rot.m23 * rcz,
                         dz = rp.rot.m31 * rcx + rp.rot.m32 * rcy + rp.rot.m33 * rcz;
            dmax = std::max(dmax, std::sqrt(dx * dx + dy * dy + dz * dz));
        }
        if (!(dmax > 0.0) || !std::isfinite(dmax) || !std::isfinite(outside)) return false;
        ep.z_clip = outside >= 4.0f * vs_max_ ? (float)(0.5 * outside / dmax) : 0.0f;   // (a depth is the parameter to within an ulp: view_projection)   // (half of it: room for the fp32 evaluation on either side)
        if (!(ep.z_clip >= 0.0f)) return false;
        ep.z_near = std::max(ep.z_clip, 0.25f * vs_min_);
        if (!(ep.z_near > 0.0f)) return false;
    }
    const Geom &g = v->g;
    const float cxw = g.offset.x + 0.5f * g.phys.x, cyw = g.offset.y + 0.5f * g.phys.y, czw = g.offset.z + 0.5f * g.phys.z;
    const float depth = ep.r[2][0] * cxw + ep.r[2][1] * cyw + ep.r[2][2] * czw + ep.r[2][3];
    const float vs_max = std::max(g.vs.x, std::max(g.vs.y, g.vs.z));
    const float reach = 0.25f * std::max(g.phys.x, std::max(g.phys.y, g.phys.z));   // (a camera inside the volume: surfaces a quarter of it away)
    const float footprint = vs_max * std::max(std::fabs(ep.k[0][0]), std::fabs(ep.k[1][1])) / std::max(depth, reach);
    // The list's builder projects the bricks -- those no pixel sees are dropped, those that are large on the screen listed in parts.
    // Always: the footprint above is the volume centre's, and a surface just behind the face the camera looks through is hundreds of
    // pixels a cell -- one wave's work for milliseconds without the parts.  (Its cost hides behind the ray records since the list's
    // workgroups start first: 11.0 -> 11.3 us; TSDF_RAY_CELLS_LOOK=0 switches it off for study.)
    ep.cell_pairs = tuning().ray_cells_look ? (uint32_t)tuning().ray_cells_pairs : 0u;
    if (can) *can = true;
    if (mode == 2) return true;
    // What the cast costs is the number of (mixed cell, pixel) pairs: a voxel that covers several pixels makes every cell a dozen pairs
    // or more.  Measured over grid sizes on one scene (640x480, the camera 2 m from the centre of 3 m of volume; cast stage, march /
    // cells with the large bricks in parts, profiles/r05s_cells_footprint_sweep_parts.txt; one part a brick: r05q_*): 384^3, 2 px a
    // voxel, 0.194 / 0.113 ms; 256^3, 3 px, 0.170 / 0.106; 192^3, 4 px, 0.176 / 0.113; 128^3, 6 px, 0.170 / 0.132; 96^3, 8 px, 0.145 /
    // 0.131; 64^3, 12 px, 0.179 / 0.194.  The voxel's footprint at the depth of the volume's centre decides (10 px), and the previous
    // cell-parallel cast's list length (arbitrary fields: every brick flagged).  From inside the volume the view holds surface after
    // surface behind the first -- every mixed cell is looked at, hidden or not: 4.2 M pairs at 1024^3 against 2.0 M for the view from
    // outside, 0.29 ms against the march's 0.215 -- and the march kernels keep it (TSDF_RAY_CELLS=2 takes the cell-parallel cast there too).
    const uint32_t listed_ = v->cell_cast_host ? *v->cell_cast_host : 0u;
    // (a trial of the cast where these rules keep the march: its cost must be bounded -- the list within the limit, and from outside a
    // voxel of at most 64 pixels at the nearest depth a sample can have; from inside the volume the list with its parts is the bound)
    // (never from inside the volume: nothing bounds how close a surface is, and on the stream of BASELINE configs[3] single frames next to
    // a wall took the cells 10 ms -- 0.278 ms a cast over the stream against the march's 0.207, profiles/r06_cells_front_to_back.txt)
    if (trial_ok) *trial_ok = listed_ <= (uint32_t)tuning().ray_cells_limit && ep.z_clip > 0.0f &&
                              vs_max * std::max(std::fabs(ep.k[0][0]), std::fabs(ep.k[1][1])) <= 64.0f * (2.0f * ep.z_clip);
    if (ep.z_clip == 0.0f) return false;
    if (!(footprint <= tuning().ray_cells_footprint)) return false;
    // ... and a camera that could have a surface right in front of it: a wall six voxels behind the face it looks through, the camera
    // five voxels outside (cells of 48 pixels), is 2.0 ms against the march's 0.08; twenty voxels outside 0.18 against 0.105
    // (tools/dbg_near_wall.py, profiles/r05x_near_wall.txt).  The nearest depth a sample of the view can have is 2 z_clip: a voxel there
    // may cover 32 pixels (the camera some sixteen voxels from the volume; at 16 pixels the 256^3 bench stream, whose camera comes
    // within 35 voxels of the volume, went back to the march: 0.148 -> 0.193 ms per step).
    if (!(vs_max * std::max(std::fabs(ep.k[0][0]), std::fabs(ep.k[1][1])) <= 32.0f * (2.0f * ep.z_clip))) return false;
    const uint32_t listed = v->cell_cast_host ? *v->cell_cast_host : 0u;
    if (listed <= (uint32_t)tuning().ray_cells_limit) return true;
    // Over the limit: the march runs, and the march never writes that count -- left alone the volume would keep the march for good,
    // also after the periodic tightening has cleared most of the flags.  Every 16th such cast the flagged bricks are counted again
    // (asynchronously: a later cast reads the mirror; nothing waits).
    tsdf_volume *w = const_cast<tsdf_volume *>(v);
    if (++w->cell_recount_wait >= 16) {
        w->cell_recount_wait = 0;
        (void)launch_brick_count(w);
    }
    return false;
}

// Which cast a volume's stream of casts takes, from MEASURED times (round 6; TSDF_RAY_CELLS=1 with TSDF_RAY_CHOOSER=1 -- built, measured,
// OFF by default: see the end of this comment).  The static rules of
// choose_cell_cast have cliffs on both sides -- a flat wall in front of the camera is 0.094 ms marched and 0.122 with the cells, the
// bench's room 0.149 and 0.100; a view from inside a 1024^3 volume 0.197 and 0.175 now that the list is sorted front to back, a
// close-up of a wall 0.08 and 2.0 -- and no rule on footprints tells these apart.  So: one cast in sixteen is bracketed with two events
// on the volume's stream, from in front of its flag refresh to behind its resolve kernel (5-6 us each: 0.7 us a cast on average; read
// when a later cast finds them complete, nothing waits), the cast not taken is tried once every `gap` casts where the static rules
// bound what the trial can cost, and the faster of the two runs.  (The brackets, not the dominant launches' own timestamps: those
// miss what differs between the casts around them -- the march's reach summary is rebuilt whenever integrate has set a flag -- and in
// the pipelined step they called the march, 0.135 ms of its two kernels, equal to the cells' 0.115 + list + resolve while every step
// with it was 8 % slower.  Where the host enqueues slower than the GPU runs the brackets hold the host's gaps, the same for both casts.)  A trial that loses doubles the gap (64 ... 4096 casts), one that loses by 3 x is not repeated before the camera has moved
// a tenth of the volume or turned by 10 degrees.  Scheduling only: both casts give the same bits (tests/test_parity_raycast.py).
// Why it is off by default (profiles/r06_cast_chooser.txt): a trial has to be SHORT to be cheap and LONG to be fair.  The march reaches its
// steady state over several casts -- the reach summary, the entry bound it leaves for the next cast, the dispatch order it learns: on the
// wall scene 0.183 ms in a trial's first cast, 0.112 in the second, 0.121 / 0.111 in the third / fourth, 0.096 in a stream of its own -- so
// trials of up to four casts call the march lost or even where it wins by a fifth, while on the bench scene each trial is three or four steps
// at 1.5 x the median.  The static rules stay the default; the chooser is there for streams that sit in one of their cliffs.
constexpr int kTrialCasts = 4;
static void chooser_poll(tsdf_volume *v) {
    CastChooser &c = v->chooser;
    if (!c.pending || hipEventQuery(c.ev[1]) != hipSuccess) { (void)hipGetLastError(); return; }
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c.ev[0], c.ev[1]) == hipSuccess && ms > 0.0f) {
        const int k = c.pending_kind;
        if (tuning().debug_rays) fprintf(stderr, "tsdf: chooser: cast %llu measured %s %.4f ms%s (march %.4f, cells %.4f so far)\n", (unsigned long long)c.casts, k ? "cells" : "march", ms, c.pending_trial ? " [trial]" : "", c.ms[0], c.ms[1]);
        // (the first casts of a kind run cold -- flags just rebuilt, nothing in the caches: 0.34 ms for a cast that takes 0.10 -- so the
        // smaller of the first three counts, the mean of old and new after that)
        c.ms[k] = c.seen[k] == 0 ? ms : (c.seen[k] < 3 ? std::min(c.ms[k], ms) : 0.5f * (c.ms[k] + ms));
        c.seen[k]++;
        c.measured_at[k] = c.casts;
        if (c.pending_trial) {
            const int other = 1 - k;
            const bool lost = c.seen[other] && ms > 1.05f * c.ms[other];
            c.gap = tuning().ray_chooser == 2 ? 3u : (lost ? std::min(c.gap * 2u, 4096u) : 64u);
            c.blocked = lost && ms > 3.0f * c.ms[other] && tuning().ray_chooser != 2;   // (until the view has changed: chooser_view_moved)
            c.next_trial = c.casts + c.gap;
        }
    } else {
        (void)hipGetLastError();
    }
    c.pending = false;
}
static bool chooser_view_moved(const tsdf_volume *v, const RayParams &rp) {
    const CastChooser &c = v->chooser;
    const float dx = rp.origin.x - c.trial_origin[0], dy = rp.origin.y - c.trial_origin[1], dz = rp.origin.z - c.trial_origin[2];
    const float reach = 0.1f * std::max(v->g.phys.x, std::max(v->g.phys.y, v->g.phys.z));
    const float dot = rp.rot.m13 * c.trial_axis[0] + rp.rot.m23 * c.trial_axis[1] + rp.rot.m33 * c.trial_axis[2];
    const float n2 = rp.rot.m13 * rp.rot.m13 + rp.rot.m23 * rp.rot.m23 + rp.rot.m33 * rp.rot.m33;
    const float m2 = c.trial_axis[0] * c.trial_axis[0] + c.trial_axis[1] * c.trial_axis[1] + c.trial_axis[2] * c.trial_axis[2];
    return !(dx * dx + dy * dy + dz * dz <= reach * reach) || !(dot >= 0.985f * std::sqrt(n2 * m2));
}
// -> true: the cell-parallel cast (ep filled); *sample: bracket this cast with the chooser's events
static bool choose_cast(tsdf_volume *v, const RayParams &rp, EntryParams &ep, bool *sample) {
    *sample = false;
    bool can = false, trial_ok = false;
    const bool by_rules = choose_cell_cast(v, rp, ep, &can, &trial_ok);
    if (tuning().ray_cells != 1 || !tuning().ray_chooser) return by_rules;
    CastChooser &c = v->chooser;
    if (c.gap == 0u) {   // (a new volume, or one that was cleared: the first trial after 64 casts)
        c.gap = 64u;
        c.next_trial = c.casts + (tuning().ray_chooser == 2 ? 3u : 64u);   // (2: a test aid -- a trial every few casts, whatever the times say)
    }
    chooser_poll(v);
    c.casts++;
    if (!can) return false;
    int kind = by_rules ? 1 : 0;
    // both measured, and not long ago: the faster one (5 % of hysteresis towards the one that ran last)
    const bool fresh[2] = {c.seen[0] && c.casts - c.measured_at[0] <= 8192u, c.seen[1] && c.casts - c.measured_at[1] <= 8192u};
    if (fresh[0] && fresh[1]) {
        const float bias0 = c.last_kind == 0 ? 0.95f : 1.0f, bias1 = c.last_kind == 1 ? 0.95f : 1.0f;
        kind = c.ms[1] * bias1 <= c.ms[0] * bias0 ? 1 : 0;
    }
    bool trial = false;
    if (c.trial_left > 0) {
        // A trial is kTrialCasts casts of the other kind, the LAST of them measured: the first march after a run of cell casts rebuilds
        // the reach summary and has no dispatch order learnt, the first cell cast its list's length unknown -- 0.18 ms for a march that
        // takes 0.09 from its fourth cast on -- the entry bound it leaves for the next cast, the order it learns (the wall scene: trials of one, two and three casts called it lost: 0.183, 0.112, 0.121 ms).
        kind = c.trial_kind;
        if (kind == 1 && !can) { c.trial_left = 0; kind = 0; }
        else if (c.trial_left > 1) c.trial_left--;
        else if (!c.pending) { trial = true; c.trial_left = 0; }   // (its measured cast; waits a cast while an earlier sample is still out)
    } else if (!c.pending && c.casts >= c.next_trial && (!c.blocked || chooser_view_moved(v, rp))) {
        const int other = 1 - kind;
        // A trial is a slow step when it loses (the march on the bench scene: + 0.06 ms), so none is made that cannot win: the cast that
        // runs is measured anyway, and the other one's best case is known -- the march 0.092 ms for 640 x 480 rays on a wall in front of
        // the camera, the cells 0.098 on the bench's room (whole casts, per pixel; profiles/r06_cast_chooser.txt) -- a cast already within 15 % of
        // that stays.
        const float mpix = (float)rp.width * (float)rp.height * 1.0e-6f, best_other = (other == 0 ? 0.30f : 0.32f) * mpix;
        const bool could_win = !c.seen[kind] || c.ms[kind] > 1.15f * best_other || tuning().ray_chooser == 2;
        if (!could_win) {
            c.next_trial = c.casts + c.gap;
        } else if (other == 0 || trial_ok) {   // (the march is always affordable; the cells where the static bound says so)
            kind = other;
            c.trial_kind = other;
            c.trial_left = kTrialCasts - 1;   // (this cast is the first of them)
            c.next_trial = c.casts + c.gap;   // (replaced when the trial's measurement arrives)
            c.blocked = false;
            c.trial_origin[0] = rp.origin.x; c.trial_origin[1] = rp.origin.y; c.trial_origin[2] = rp.origin.z;
            c.trial_axis[0] = rp.rot.m13; c.trial_axis[1] = rp.rot.m23; c.trial_axis[2] = rp.rot.m33;
        } else {
            c.next_trial = c.casts + c.gap;
        }
    }
    if (kind == 1 && (v->cell_cast_host ? *v->cell_cast_host : 0u) > (uint32_t)tuning().ray_cells_limit) kind = 0;   // (never past the list's limit)
    if (!c.pending && c.casts > 1u && (trial || (c.trial_left == 0 && (c.seen[kind] < 3u || (c.casts & 15u) == 0u)))) {
        if (!c.ev[0]) {
            if (hipEventCreate(&c.ev[0]) != hipSuccess || hipEventCreate(&c.ev[1]) != hipSuccess) { (void)hipGetLastError(); c.ev[0] = nullptr; }
        }
        if (c.ev[0]) {
            *sample = true;
            c.pending_kind = kind;
            c.pending_trial = trial;
        }
    }
    c.last_kind = kind;
    return kind == 1;
}
// a whole-volume cast of the kind chosen, bracketed for the chooser when it asks
static int cast_whole(tsdf_volume *v, RayParams &rp, float *out, float *normals, const float *depth_inv_pose, uint16_t *depth_out);

// The production march: process_ray_kernel over the sample ranges of every ray with a pass budget, process_ray_tail_kernel
// for the stretches it handed over, resolve_hits_kernel for the vertices (packed float3, or {k, t} records for a slab).
template <bool SLAB>
static int march_and_resolve(tsdf_volume *v, RayParams &rp, float *out, float *normals = nullptr, const float *depth_inv_pose = nullptr,
                             uint16_t *depth_out = nullptr, const EntryParams *cells = nullptr) {
    const size_t n_pix = (size_t)rp.width * rp.height;
    const int n_segments = SLAB ? slab_ray_ranges(v) : ray_segments();
    // queue capacity: every range unfinished and cut into all the pieces its length allows (a range holds at most
    // ceil(kMaxSamples / n_segments) samples, whole volume or slab)
    const int max_len = (kMaxSamples + n_segments - 1) / n_segments;
    const int max_pieces = std::min(kTailPieces, (max_len + tail_piece_min() - 1) / tail_piece_min());
    const size_t n_entries = n_pix * n_segments * (size_t)std::max(max_pieces, 1);
    if (v->ray_best_cap < n_pix) {
        if (v->ray_best) (void)hipFree(v->ray_best);
        v->ray_best = nullptr;
        v->ray_best_cap = 0;
        TSDF_HIP(hipMalloc((void **)&v->ray_best, 2 * n_pix * sizeof(uint64_t)), "ray result alloc");   // (double buffered)
        v->ray_best_cap = n_pix;
        v->ray_best_dirty = 1;
    }
    if (cells) {
        const size_t n_bricks_max = cell_list_capacity(v->occ.fine_count());
        if (v->cell_rays_cap < n_pix) {
            if (v->cell_rays) (void)hipFree(v->cell_rays);
            v->cell_rays = nullptr;
            v->cell_rays_cap = 0;
            TSDF_HIP(hipMalloc(&v->cell_rays, n_pix * sizeof(RayRecord)), "ray record alloc");
            v->cell_rays_cap = n_pix;
        }
        if (v->cell_bricks_cap < n_bricks_max) {
            if (v->cell_bricks) (void)hipFree(v->cell_bricks);
            v->cell_bricks = nullptr;
            v->cell_bricks_cap = 0;
            TSDF_HIP(hipMalloc((void **)&v->cell_bricks, 2 * n_bricks_max * sizeof(uint2)), "brick list alloc");   // (the list, and the list as built when it is sorted)
            v->cell_bricks_cap = n_bricks_max;
        }
        if (!v->cell_cast_host) {
            TSDF_HIP(hipHostMalloc((void **)&v->cell_cast_host, sizeof(uint32_t), hipHostMallocDefault), "brick count mirror alloc");
            *v->cell_cast_host = 0;
        }
    }
    if (v->tail_cap < n_entries) {
        if (v->tail_entries) (void)hipFree(v->tail_entries);
        v->tail_entries = nullptr;
        v->tail_cap = 0;
        TSDF_HIP(hipMalloc(&v->tail_entries, n_entries * sizeof(uint2)), "ray tail queue alloc");
        v->tail_cap = n_entries;
    }
    if (!v->tail_count) {
        TSDF_HIP(hipMalloc((void **)&v->tail_count, kTailSignals * sizeof(uint32_t)), "ray tail counter alloc");
        v->ray_best_dirty = 1;
    }
    if (v->ray_best_pixels != n_pix) {   // the copy this march's resolve does not read was reset for another image size
        v->ray_best_pixels = n_pix;
        v->ray_best_dirty = 1;
    }
    if (v->ray_best_dirty) {   // otherwise the previous march's resolve kernel left both reset
        TSDF_HIP(hipMemsetAsync(v->ray_best, 0xff, 2 * v->ray_best_cap * sizeof(uint64_t), v->stream), "ray result reset");
        TSDF_HIP(hipMemsetAsync(v->tail_count, 0, kTailSignals * sizeof(uint32_t), v->stream), "ray tail counter reset");
    }
    v->ray_best_dirty = 1;
    TailQueue tail = {reinterpret_cast<uint2 *>(v->tail_entries), v->tail_count, (uint32_t)trip_budget(), (uint32_t)tail_lanes(), v->ray_best + (size_t)v->ray_best_side * v->ray_best_cap, tail_piece_min(), nullptr};
    uint64_t *best_next = v->ray_best + (size_t)(1 - v->ray_best_side) * v->ray_best_cap;
    // the dispatch order learnt from the previous cast (TSDF_RAY_LEARNED_ORDER=0: launch order, tuning aid)
    const bool learn_order = tuning().ray_learned_order != 0;
    const uint32_t n_tiles = ((rp.width + 15) / 16) * ((rp.height + 15) / 16);
    OrderJob order_job = {nullptr, nullptr, 0, 0, 0};
    if (learn_order && n_tiles <= 65535u && (uint32_t)n_segments <= kOrderMaxRanges) {
        if (!v->ray_heavy || v->ray_order_tiles != n_tiles || v->ray_order_ranges != (uint32_t)n_segments) {
            if (v->ray_heavy) (void)hipFree(v->ray_heavy);
            if (v->ray_order) (void)hipFree(v->ray_order);
            v->ray_heavy = nullptr;
            v->ray_order = nullptr;
            v->ray_order_valid = 0;
            TSDF_HIP(hipMalloc((void **)&v->ray_heavy, (size_t)2 * n_tiles * n_segments), "ray order alloc");
            TSDF_HIP(hipMalloc((void **)&v->ray_order, (size_t)n_tiles * n_segments * sizeof(uint32_t)), "ray order alloc");
            TSDF_HIP(hipMemsetAsync(v->ray_heavy, 0, (size_t)2 * n_tiles * n_segments, v->stream), "ray order reset");
            v->ray_order_tiles = n_tiles;
            v->ray_order_ranges = (uint32_t)n_segments;
        }
        // "long": three quarters of the pass budget or more.  (With the runs of free space merged and three classes -- long, some pass, none --
        // any threshold from 14 to 22 of 22 passes gives the same launch, 62-63 us; 12 and below 69-74 us: the waves of a few passes are
        // many, and listed first they push the long ones back.)
        tail.heavy_passes = tuning().ray_heavy_passes > 0 ? (uint32_t)tuning().ray_heavy_passes : std::max(1u, tail.trip_budget * 3u / 4u);
        tail.heavy = v->ray_heavy;
        tail.order = v->ray_order_valid ? v->ray_order : nullptr;
        order_job = {v->ray_heavy, v->ray_order, n_tiles, (uint32_t)n_segments, 0};
    }
    bool order_built = false;
    v->last_cast_cells = cells ? 1 : 0;
    if (cells) {
        // ---- the cell-parallel cast (raycast_cells.hpp): the rays' records, the flagged bricks, one wave per brick ----
        // the list front to back when the camera is inside (or within a few voxels of) the volume: TSDF_RAY_CELLS_SORT 0 never / 1 then (default) / 2 always
        const bool sort_list = tuning().ray_cells_sort == 2 || (tuning().ray_cells_sort == 1 && cells->z_clip == 0.0f);
        float depth0 = 0.0f, depth_scale = 0.0f;
        if (sort_list) {   // the depths of the volume's corners: the bins' range
            float dmin = INFINITY, dmax = -INFINITY;
            for (int c = 0; c < 8; c++) {
                const float wx = v->g.offset.x + ((c & 1) ? v->g.phys.x : 0.0f), wy = v->g.offset.y + ((c & 2) ? v->g.phys.y : 0.0f), wz = v->g.offset.z + ((c & 4) ? v->g.phys.z : 0.0f);
                const float d = cells->r[2][0] * wx + cells->r[2][1] * wy + cells->r[2][2] * wz + cells->r[2][3];
                dmin = std::min(dmin, d); dmax = std::max(dmax, d);
            }
            dmin = std::max(dmin, 0.0f);   // (nothing behind the camera is seen: one bin for all of it)
            depth0 = dmin;
            depth_scale = dmax > dmin ? (float)kDepthBins / (dmax - dmin) : 0.0f;
            if (!std::isfinite(depth0) || !std::isfinite(depth_scale)) { depth0 = 0.0f; depth_scale = 0.0f; }
        }
        CellCast cc = {reinterpret_cast<RayRecord *>(v->cell_rays), reinterpret_cast<uint2 *>(v->cell_bricks), v->tail_count + 3, v->cell_cast_host, cells->cell_pairs,
                       sort_list ? v->tail_count + 4 : nullptr, reinterpret_cast<uint2 *>(v->cell_bricks) + v->cell_bricks_cap, depth0, depth_scale,
                       v->dist, tail.best, v->release_word, v->release_value};
        v->release_word = nullptr;   // (taken)
        const size_t table_lds = ((size_t)kMaxSamples + 1) * sizeof(float);
        const uint32_t n_ray_blocks = (uint32_t)((n_pix + 255) / 256);
        const uint32_t n_list_blocks = (uint32_t)std::min<size_t>((v->occ.fine_count() / 4 + 255) / 256 + 1, 2048);
        hipLaunchKernelGGL((cell_cast_prepare_kernel<SLAB>), dim3(n_ray_blocks + n_list_blocks), dim3(256), (uint32_t)table_lds, v->stream, v->g, rp, *cells, v->t_table,
                           v->occ, cc, n_list_blocks);
        if (sort_list) {
            hipLaunchKernelGGL(cell_list_sort_kernel, dim3(128), dim3(1024), 0, v->stream, v->g, *cells, cc);
        }
        if (getenv("TSDF_DEBUG_CELLS_SORT")) {
            // (diagnostics, synchronises: the list put in front-to-back order on the HOST -- by the camera depth of each brick's centre --
            // before the cast: what that order buys the views from inside the volume, where every surface behind the first is looked at;
            // =2: back to front.  profiles/r06_cells_front_to_back.txt)
            TSDF_HIP(hipStreamSynchronize(v->stream), "cells sort");
            uint32_t n_listed = 0;
            TSDF_HIP(hipMemcpy(&n_listed, cc.n_bricks, sizeof(n_listed), hipMemcpyDeviceToHost), "cells sort");
            std::vector<uint2> list(n_listed);
            TSDF_HIP(hipMemcpy(list.data(), cc.bricks, n_listed * sizeof(uint2), hipMemcpyDeviceToHost), "cells sort");
            const bool back_first = atoi(getenv("TSDF_DEBUG_CELLS_SORT")) == 2;
            auto depth_of = [&](const uint2 &e_) {
                const float mid = 0.5f * ((float)kBrick + 1.5f);
                const float wx = ((float)((e_.y & 1023u) * kBrick) + mid) * v->g.vs.x + cells->offset.x, wy = ((float)(((e_.y >> 10) & 1023u) * kBrick) + mid) * v->g.vs.y + cells->offset.y,
                            wz = ((float)((e_.y >> 20) * kBrick) + mid) * v->g.vs.z + cells->offset.z;
                const float d = cells->r[2][0] * wx + cells->r[2][1] * wy + cells->r[2][2] * wz + cells->r[2][3];
                return back_first ? -d : d;
            };
            std::stable_sort(list.begin(), list.end(), [&](const uint2 &a_, const uint2 &b_) { return depth_of(a_) < depth_of(b_); });
            TSDF_HIP(hipMemcpy(cc.bricks, list.data(), n_listed * sizeof(uint2), hipMemcpyHostToDevice), "cells sort");
        }
        const dim3 cgrid((unsigned)tuning().ray_cells_grid + kShellWorkgroups);   // (the first kShellWorkgroups: the boundary bricks' shell samples)
        if (v->fast_div)
            TSDF_LAUNCH_TIMED(v, 1, (cast_cells_kernel<SLAB, true>), cgrid, dim3(256), v->dist, v->g, rp, *cells, v->occ, v->t_table, cc, tail.best);
        else
            TSDF_LAUNCH_TIMED(v, 1, (cast_cells_kernel<SLAB, false>), cgrid, dim3(256), v->dist, v->g, rp, *cells, v->occ, v->t_table, cc, tail.best);
        if (getenv("TSDF_DEBUG_CELLS_TWICE")) {   // (diagnostics: the same launch again, every pixel's word final: what perfect culling of hidden cells would cost)
            if (v->fast_div)
                TSDF_LAUNCH_TIMED(v, 2, (cast_cells_kernel<SLAB, true>), cgrid, dim3(256), v->dist, v->g, rp, *cells, v->occ, v->t_table, cc, tail.best);
            else
                TSDF_LAUNCH_TIMED(v, 2, (cast_cells_kernel<SLAB, false>), cgrid, dim3(256), v->dist, v->g, rp, *cells, v->occ, v->t_table, cc, tail.best);
        }
        TSDF_HIP(hipGetLastError(), "cell-parallel cast failed");
        if (v->after_bulk) TSDF_HIP(hipEventRecord(v->after_bulk, v->stream), "process_ray: bulk kernel event");
    } else {
    rp.seg_len = (kMaxSamples + n_segments - 1) / n_segments;
    rp.slab_ranges = SLAB ? (uint32_t)n_segments : 0u;
    rp.tile_map = (uint32_t)tuning().ray_tile_map;
    rp.range_order = (uint32_t)tuning().ray_range_order;
    dim3 grid((rp.width + 15) / 16, (rp.height + 15) / 16, n_segments);
    const size_t table_lds = ray_table_lds_bytes(rp, (SLAB && rp.slab_ranges > 0) || tail.order != nullptr);   // (a merged workgroup may read any part of the table)
    if (v->fast_div)
        TSDF_LAUNCH_TIMED_LDS(v, 1, (process_ray_kernel<SLAB, false, true, true, true, true>), grid, dim3(256), table_lds, v->dist, v->g, rp,
                              (float *)nullptr, (unsigned long long *)nullptr, (unsigned int *)nullptr, v->occ, v->t_table, tail);
    else
        TSDF_LAUNCH_TIMED_LDS(v, 1, (process_ray_kernel<SLAB, false, true, false, true, true>), grid, dim3(256), table_lds, v->dist, v->g, rp,
                              (float *)nullptr, (unsigned long long *)nullptr, (unsigned int *)nullptr, v->occ, v->t_table, tail);
    TSDF_HIP(hipGetLastError(), "process_ray failed");
    if (v->after_bulk) TSDF_HIP(hipEventRecord(v->after_bulk, v->stream), "process_ray: bulk kernel event");
    // persistent workgroups: groups of 16 lanes, fetching queue entries until none is left
    // (the default group width is compiled in; another one, a tuning aid, takes the variant that reads it at run time)
    const bool fixed_lanes = tail_lanes() == kTailLanesDefault;
    order_job.range_order = rp.range_order;
    dim3 tgrid_(tail_grid() + kOrderWorkgroups);
    order_built = order_job.n_ranges != 0;
    if (v->fast_div && fixed_lanes)
        TSDF_LAUNCH_TIMED(v, 2, (process_ray_tail_kernel<SLAB, true, kTailLanesDefault>), tgrid_, dim3(256), v->dist, v->g, rp, v->occ, v->t_table, tail, order_job);
    else if (v->fast_div)
        TSDF_LAUNCH_TIMED(v, 2, (process_ray_tail_kernel<SLAB, true, 0>), tgrid_, dim3(256), v->dist, v->g, rp, v->occ, v->t_table, tail, order_job);
    else if (fixed_lanes)
        TSDF_LAUNCH_TIMED(v, 2, (process_ray_tail_kernel<SLAB, false, kTailLanesDefault>), tgrid_, dim3(256), v->dist, v->g, rp, v->occ, v->t_table, tail, order_job);
    else
        TSDF_LAUNCH_TIMED(v, 2, (process_ray_tail_kernel<SLAB, false, 0>), tgrid_, dim3(256), v->dist, v->g, rp, v->occ, v->t_table, tail, order_job);
    TSDF_HIP(hipGetLastError(), "process_ray (tail) failed");
    if (tuning().debug_rays) {   // diagnostics: how much went through the tail queue (synchronises)
        uint32_t n_tail = 0;
        (void)hipMemcpy(&n_tail, v->tail_count, sizeof(n_tail), hipMemcpyDeviceToHost);
        fprintf(stderr, "tsdf: %u pieces of the %zu (ray, range) pairs finished by the tail kernel\n", n_tail, n_pix * n_segments);
    }
    }
    const dim3 rgrid((unsigned)((n_pix + 255) / 256)), tgrid((rp.width + 15) / 16, (rp.height + 15) / 16);
    if (!SLAB && normals) {
        hipLaunchKernelGGL(resolve_normals_kernel, tgrid, dim3(kResolveThreads), 0, v->stream, v->g, rp, v->t_table, tail.best, best_next, out, normals,
                           v->tail_count);
    } else {
        Mat44 ip;
        memset(&ip, 0, sizeof(ip));
        if (depth_inv_pose) memcpy(&ip, depth_inv_pose, sizeof(ip));
        hipLaunchKernelGGL((resolve_hits_kernel<SLAB>), rgrid, dim3(256), 0, v->stream, v->g, rp, v->t_table, tail.best, best_next, out, v->tail_count, ip,
                           SLAB ? (uint16_t *)nullptr : depth_out);
    }
    if (order_built) v->ray_order_valid = 1;
    TSDF_HIP(hipGetLastError(), "resolve ray hits failed");
    v->ray_best_side = 1 - v->ray_best_side;
    v->ray_best_dirty = 0;
    return TSDF_OK;
}

static int cast_whole(tsdf_volume *v, RayParams &rp, float *out, float *normals, const float *depth_inv_pose, uint16_t *depth_out) {
    int rc = count_after_bulk_change(v);
    if (rc != TSDF_OK) return rc;
    EntryParams view;
    bool sample = false;
    const bool cells = choose_cast(v, rp, view, &sample);
    CastChooser &c = v->chooser;
    if (sample && hipEventRecord(c.ev[0], v->stream) != hipSuccess) { (void)hipGetLastError(); sample = false; }
    if (cells) rc = occupancy_flags_refresh(v); else rc = refresh_for_cast(v, rp);
    if (rc != TSDF_OK) return rc;
    rc = cells ? march_and_resolve<false>(v, rp, out, normals, depth_inv_pose, depth_out, &view) : march_and_resolve<false>(v, rp, out, normals, depth_inv_pose, depth_out);
    if (sample && rc == TSDF_OK) {
        if (hipEventRecord(c.ev[1], v->stream) == hipSuccess) c.pending = true; else (void)hipGetLastError();
    }
    return rc;
}

// Would tsdf_raycast_device take the cell-parallel cast for this view now?  (tsdf_pipeline_step: how to release its second stream.)
bool raycast_takes_cells(const tsdf_volume *v, uint32_t width, uint32_t height, const float pose[16], const float kinv[9]) {
    if (!v || v->z_begin != 0 || v->z_end != v->g.Z || check_ray_args(v, width, height, pose, kinv) != TSDF_OK) return false;
    if (count_after_bulk_change(const_cast<tsdf_volume *>(v)) != TSDF_OK) return false;
    RayParams rp = make_params(v, width, height, pose, kinv);
    EntryParams view;
    return choose_cell_cast(v, rp, view);
}

}  // namespace tsdf

using namespace tsdf;

extern "C" {

#ifdef TSDF_DIAG_RAY_MIX
// diagnostics build only: read (and reset) the pass-type counters
int tsdf_debug_ray_mix(unsigned long long out[64]) {
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ray_mix), 64 * sizeof(unsigned long long)) != hipSuccess) return TSDF_ERR_INVALID;
    unsigned long long zero[64] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ray_mix), zero, sizeof(zero));
    return TSDF_OK;
}
#endif

int tsdf_raycast_device(const tsdf_volume *v, uint32_t width, uint32_t height, const float pose[16],
                        const float kinv[9], float *device_vertices, float *device_normals) {
    int rc = check_ray_args(v, width, height, pose, kinv);
    if (rc != TSDF_OK) return rc;
    TSDF_REQUIRE(device_vertices, "tsdf_raycast: null vertex buffer");
    TSDF_REQUIRE(v->z_begin == 0 && v->z_end == v->g.Z, "tsdf_raycast on a slab: use tsdf_raycast_slab_device");
    RayParams rp = make_params(v, width, height, pose, kinv);
    return cast_whole(const_cast<tsdf_volume *>(v), rp, device_vertices, device_normals, nullptr, nullptr);
}

int tsdf_raycast(const tsdf_volume *cv, uint32_t width, uint32_t height, const float pose[16], const float kinv[9],
                 float *host_vertices, float *host_normals) {
    int rc = check_ray_args(cv, width, height, pose, kinv);
    if (rc != TSDF_OK) return rc;
    TSDF_REQUIRE(host_vertices, "tsdf_raycast: null vertex buffer");
    tsdf_volume *v = const_cast<tsdf_volume *>(cv);  // per-call temporaries are cached in the handle
    size_t bytes = (size_t)width * height * 3 * sizeof(float);
    if (v->ray_cap < bytes) {
        if (v->vert_buf) (void)hipFree(v->vert_buf);
        if (v->norm_buf) (void)hipFree(v->norm_buf);
        v->vert_buf = v->norm_buf = nullptr;
        v->ray_cap = 0;
        TSDF_HIP(hipMalloc((void **)&v->vert_buf, bytes), "Vertices alloc failed");
        TSDF_HIP(hipMalloc((void **)&v->norm_buf, bytes), "Normals alloc failed");
        v->ray_cap = bytes;
    }
    rc = tsdf_raycast_device(v, width, height, pose, kinv, v->vert_buf, host_normals ? v->norm_buf : nullptr);
    if (rc != TSDF_OK) return rc;
    TSDF_HIP(hipMemcpyAsync(host_vertices, v->vert_buf, bytes, hipMemcpyDeviceToHost, v->stream), "Vertices Memcpy failed");
    if (host_normals)
        TSDF_HIP(hipMemcpyAsync(host_normals, v->norm_buf, bytes, hipMemcpyDeviceToHost, v->stream), "Normals Memcpy failed");
    TSDF_HIP(hipStreamSynchronize(v->stream), "process_ray failed");
    return TSDF_OK;
}

int tsdf_volume_last_raycast_kind(const tsdf_volume *v, int *cell_parallel) {
    TSDF_REQUIRE(v && cell_parallel, "null argument");
    *cell_parallel = v->last_cast_cells;
    return TSDF_OK;
}

int tsdf_volume_last_cell_list(const tsdf_volume *v, uint32_t *listed) {
    TSDF_REQUIRE(v && listed, "tsdf_volume_last_cell_list: null argument");
    TSDF_HIP(hipStreamSynchronize(v->stream), "tsdf_volume_last_cell_list");
    *listed = v->cell_cast_host ? *v->cell_cast_host : 0u;
    return TSDF_OK;
}

int tsdf_raycast_depth_device(const tsdf_volume *v, uint32_t width, uint32_t height, const float pose[16], const float inv_pose[16],
                              const float kinv[9], uint16_t *device_depth, float *device_vertices) {
    int rc = check_ray_args(v, width, height, pose, kinv);
    if (rc != TSDF_OK) return rc;
    TSDF_REQUIRE(device_depth && inv_pose, "tsdf_raycast_depth: null argument");
    TSDF_REQUIRE(v->z_begin == 0 && v->z_end == v->g.Z, "tsdf_raycast_depth needs a whole volume");
    RayParams rp = make_params(v, width, height, pose, kinv);
    return cast_whole(const_cast<tsdf_volume *>(v), rp, device_vertices, nullptr, inv_pose, device_depth);
}

int tsdf_normals_device(uint32_t width, uint32_t height, const float *device_vertices, float *device_normals,
                        void *hip_stream) {
    TSDF_REQUIRE(device_vertices && device_normals && width > 0 && height > 0, "tsdf_normals: bad argument");
    return launch_normals(width, height, device_vertices, device_normals, (hipStream_t)hip_stream);
}

int tsdf_vertices_to_depth_device(uint32_t width, uint32_t height, const float *device_vertices, const float inv_pose[16],
                                  uint16_t *device_depth, void *hip_stream) {
    TSDF_REQUIRE(device_vertices && inv_pose && device_depth && width > 0 && height > 0, "tsdf_vertices_to_depth: bad argument");
    Mat44 ip;
    memcpy(&ip, inv_pose, sizeof(ip));
    const uint32_t n = width * height;
    hipLaunchKernelGGL(vertices_to_depth_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, n, device_vertices, ip,
                       device_depth);
    TSDF_HIP(hipGetLastError(), "vertices to depth failed");
    return TSDF_OK;
}

int tsdf_raycast_stats(const tsdf_volume *v, uint32_t width, uint32_t height, const float pose[16],
                       const float kinv[9], uint64_t *samples, uint64_t *touched_voxels, uint64_t *hits) {
    int rc = check_ray_args(v, width, height, pose, kinv);
    if (rc != TSDF_OK) return rc;
    TSDF_REQUIRE(v->z_begin == 0 && v->z_end == v->g.Z, "tsdf_raycast_stats needs a whole volume");
    RayParams rp = make_params(v, width, height, pose, kinv);
    size_t n = (size_t)v->g.X * v->g.Y * v->g.Z;
    size_t words = (n + 31) / 32;
    unsigned int *bitmap = nullptr;
    float *verts = nullptr;
    TSDF_HIP(hipMalloc((void **)&bitmap, words * sizeof(unsigned int)), "stats bitmap alloc");
    hipError_t e = hipMalloc((void **)&verts, (size_t)width * height * 3 * sizeof(float));
    if (e != hipSuccess) {
        (void)hipFree(bitmap);
        return hip_fail(e, "stats vertex alloc");
    }
    (void)hipMemsetAsync(bitmap, 0, words * sizeof(unsigned int), v->stream);
    (void)hipMemsetAsync(v->counter_dev, 0, 4 * sizeof(unsigned long long), v->stream);
    dim3 grid((width + 15) / 16, (height + 15) / 16);
    hipLaunchKernelGGL((process_ray_kernel<false, true, false, false, false, false>), grid, dim3(256), ray_table_lds_bytes(rp, false), v->stream, v->dist, v->g, rp, verts,
                       v->counter_dev, bitmap, v->occ, v->t_table, TailQueue{nullptr, nullptr, 0, 0, nullptr, 0});
    hipLaunchKernelGGL(popcount_kernel, dim3(1024), dim3(256), 0, v->stream, bitmap, words, v->counter_dev + 3);
    unsigned long long c[4] = {0, 0, 0, 0};
    e = hipMemcpyAsync(c, v->counter_dev, sizeof(c), hipMemcpyDeviceToHost, v->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(v->stream);
    (void)hipFree(bitmap);
    (void)hipFree(verts);
    if (e != hipSuccess) return hip_fail(e, "raycast stats");
    if (samples) *samples = c[1];
    if (hits) *hits = c[2];
    if (touched_voxels) *touched_voxels = c[3];
    return TSDF_OK;
}

int tsdf_raycast_evaluated_samples(const tsdf_volume *v, uint32_t width, uint32_t height, const float pose[16],
                                   const float kinv[9], uint64_t *evaluated, float *host_per_ray) {
    int rc = check_ray_args(v, width, height, pose, kinv);
    if (rc != TSDF_OK) return rc;
    TSDF_REQUIRE(evaluated, "null argument");
    TSDF_REQUIRE(v->z_begin == 0 && v->z_end == v->g.Z, "needs a whole volume");
    rc = occupancy_refresh(const_cast<tsdf_volume *>(v));
    if (rc != TSDF_OK) return rc;
    RayParams rp = make_params(v, width, height, pose, kinv);
    size_t words = ((size_t)v->g.X * v->g.Y * v->g.Z + 31) / 32;
    unsigned int *bitmap = nullptr;
    float *verts = nullptr;
    TSDF_HIP(hipMalloc((void **)&bitmap, words * sizeof(unsigned int)), "stats bitmap alloc");
    hipError_t e = hipMalloc((void **)&verts, (size_t)width * height * 3 * sizeof(float));
    if (e != hipSuccess) {
        (void)hipFree(bitmap);
        return hip_fail(e, "stats vertex alloc");
    }
    (void)hipMemsetAsync(v->counter_dev, 0, 4 * sizeof(unsigned long long), v->stream);
    dim3 grid((width + 15) / 16, (height + 15) / 16);
    hipLaunchKernelGGL((process_ray_kernel<false, true, true, true, false, false>), grid, dim3(256), ray_table_lds_bytes(rp, false), v->stream, v->dist, v->g, rp, verts,
                       v->counter_dev, bitmap, v->occ, v->t_table, TailQueue{nullptr, nullptr, 0, 0, nullptr, 0});
    unsigned long long c[4] = {0, 0, 0, 0};
    e = hipMemcpyAsync(c, v->counter_dev, sizeof(c), hipMemcpyDeviceToHost, v->stream);
    if (e == hipSuccess && host_per_ray)
        e = hipMemcpyAsync(host_per_ray, verts, (size_t)width * height * 3 * sizeof(float), hipMemcpyDeviceToHost, v->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(v->stream);
    (void)hipFree(bitmap);
    (void)hipFree(verts);
    if (e != hipSuccess) return hip_fail(e, "raycast stats");
    *evaluated = c[1];
    return TSDF_OK;
}

int tsdf_raycast_slab_device(const tsdf_volume *v, uint32_t width, uint32_t height, const float pose[16],
                             const float kinv[9], tsdf_hit_record *device_hits) {
    int rc = check_ray_args(v, width, height, pose, kinv);
    if (rc != TSDF_OK) return rc;
    TSDF_REQUIRE(device_hits, "tsdf_raycast_slab: null hit buffer");
    RayParams rp = make_params(v, width, height, pose, kinv);
    EntryParams view;
    rc = count_after_bulk_change(const_cast<tsdf_volume *>(v));
    if (rc != TSDF_OK) return rc;
    if (choose_cell_cast(v, rp, view)) {
        rc = occupancy_flags_refresh(const_cast<tsdf_volume *>(v));
        if (rc != TSDF_OK) return rc;
        return march_and_resolve<true>(const_cast<tsdf_volume *>(v), rp, reinterpret_cast<float *>(device_hits), nullptr, nullptr, nullptr, &view);
    }
    rc = occupancy_refresh(const_cast<tsdf_volume *>(v));
    if (rc != TSDF_OK) return rc;
    // as on a single GPU the march is split into sample ranges; one {k, t} record per pixel leaves this rank
    return march_and_resolve<true>(const_cast<tsdf_volume *>(v), rp, reinterpret_cast<float *>(device_hits));
}

int tsdf_merge_hits_device(const tsdf_volume *v, const tsdf_hit_record *device_hits_all, uint32_t n_slabs, uint32_t width,
                           uint32_t height, const float pose[16], const float kinv[9], float *device_vertices, void *hip_stream) {
    int rc = check_ray_args(v, width, height, pose, kinv);
    if (rc != TSDF_OK) return rc;
    TSDF_REQUIRE(device_hits_all && device_vertices && n_slabs > 0, "tsdf_merge_hits: bad argument");
    const RayParams rp = make_params(v, width, height, pose, kinv);
    const uint32_t n = width * height;
    hipLaunchKernelGGL(merge_hits_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream,
                       reinterpret_cast<const uint2 *>(device_hits_all), n_slabs, rp, device_vertices);
    TSDF_HIP(hipGetLastError(), "merge hits failed");
    return TSDF_OK;
}

int tsdf_merge_hits_normals_device(const tsdf_volume *v, const tsdf_hit_record *device_hits_all, uint32_t n_slabs, uint32_t width,
                                   uint32_t height, const float pose[16], const float kinv[9], float *device_vertices,
                                   float *device_normals, void *hip_stream) {
    int rc = check_ray_args(v, width, height, pose, kinv);
    if (rc != TSDF_OK) return rc;
    TSDF_REQUIRE(device_hits_all && device_vertices && device_normals && n_slabs > 0, "tsdf_merge_hits_normals: bad argument");
    const RayParams rp = make_params(v, width, height, pose, kinv);
    hipLaunchKernelGGL(merge_hits_normals_kernel, dim3((width + 15) / 16, (height + 15) / 16), dim3(kResolveThreads), 0,
                       (hipStream_t)hip_stream, reinterpret_cast<const uint2 *>(device_hits_all), n_slabs, rp, device_vertices,
                       device_normals);
    TSDF_HIP(hipGetLastError(), "merge hits + normals failed");
    return TSDF_OK;
}

}  // extern "C"
