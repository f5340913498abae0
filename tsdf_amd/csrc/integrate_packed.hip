// integrate_kernel for volumes whose weights are stored as packed counts (weights.hip), standard cameras and implicit deformation
// nodes: the per-voxel arithmetic of the reference (src/TSDF/TSDFVolume.cu:337-390, src/Utilities/cuda_coordinate_transforms.cu:10-30,
// 108-146) in its operation order, bit for bit what integrate_kernel<false, *, true> of integrate.hip computes -- with roughly half
// the vector instructions and 10 (12) instead of 16 bytes moved per updated voxel.
//
// Why (round 4, knock-out builds of integrate_kernel, `profiles/r04n_*`): with every distance / weight access removed the kernel still
// took 0.097 of its 0.118 ms -- its vector instruction stream (80 issue slots per 64-voxel row x 6 waves per SIMD = the 31 us a brick
// takes) -- and the read-modify-write of 558 MB at the 5-6 TB/s an in-place walk reaches is 0.093-0.11 ms: two equal bounds, so that
// halving either alone changed nothing (what rounds 2 and 3 measured).  This kernel lowers both.
//   * weights: the reference only ever adds 1 to a weight (the clamp to max_weight is commented out, TSDFVolume.cu:377), so a
//     volume's weights are small integers unless a caller uploads something else.  They are kept as 8- (16-) bit counts, the four
//     (two) planes of a batch of one lane in ONE dword ("z-packed"): a wave still moves whole 256-byte rows, one per batch instead of
//     four.  Dense walks in this shape: 0.237 (0.274) ms against 0.360 for two fp32 arrays (tools/ubench_layout.hip).
//   * arithmetic: two planes at a time with packed fp32 (v_pk_add / v_pk_mul: separately rounded lanes, no contraction); one test
//     per pair of planes for "some quotient is near a rounding boundary" (v_maximum3, NaN-propagating) instead of a branch per
//     plane; (d w + tsdf) / (w + 1) as rcp + mul + two fmas (div_by_count, proven and checked exhaustively); the depth look-up addressed in fp32 into a tile with a ring of zeros (v_med3 clamps what misses the pixel box onto the
//     ring: no range tests, no select; bricks without a tile look up a copy of the whole image inside such a ring that
//     brick_cull_kernel leaves in memory); distance rows addressed as wave-uniform base + one 32-bit lane offset.
#include <type_traits>

#include "common.hpp"
#include "integrate_grid.hpp"

namespace tsdf {

#ifndef TSDF_PACKED_WAVES
#define TSDF_PACKED_WAVES 6   // waves per SIMD the kernel is compiled for (register budget)
#endif
#ifndef TSDF_PACKED_PIPELINE
#define TSDF_PACKED_PIPELINE 1   // two register sets, a batch of loads always in flight (0: one set, for the occupancy A/B of round 6: profiles/r06_integrate_occupancy_ab.txt)
#endif

// (knock-out builds for timing experiments only: -DTSDF_DIAG_NOLOAD / -DTSDF_DIAG_NOSTORE make the accesses depend on conditions that never hold)
#ifdef TSDF_DIAG_NOLOAD
#define DIAG_NOLOAD && r1 == 12345.678f
#else
#define DIAG_NOLOAD
#endif
#ifdef TSDF_DIAG_NOSTORE
#define DIAG_NOSTORE_D && new_distance == 12345.678f
#define DIAG_NOSTORE_W && r1 == 12345.678f
#else
#define DIAG_NOSTORE_D
#define DIAG_NOSTORE_W
#endif

typedef float f2 __attribute__((ext_vector_type(2)));   // two planes of one lane: arithmetic on it issues as v_pk_*_f32

// max(acc, |a|, |b|) that keeps a NaN (IEEE 754-2019 maximum; v_max3_f32 would drop it)
__device__ inline float max3_abs_keep_nan(float acc, float a, float b) {
    float r;
    asm("v_maximum3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc));
    return r;
}
__device__ inline float med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

// a / b as the IEEE division rounds it, for b an integer-valued float in [1, 2^17] (a weight count + 1) and any float a.
//   y = rcp(b) (relative error e, |e| <= 2^-22 and far less in practice), q0 = fl(a y), r = a - q0 b, q1 = fl(q0 + r y).
//   * |q0 - a / b| < 2.5 ulp of the quotient, so r = b (a / b - q0) is a multiple of ulp(q) / 2 below 2^24 of them: the fma forms it
//     exactly;
//   * q0 + r y = a / b + (a / b - q0) e, off the true quotient by less than 2.5 * 2^-22 ulp;
//   * a quotient by an integer is never a rounding midpoint m (b m would need a 25th significant bit) and a - b m is a multiple of
//     ulp(q) / 2, so a / b stays ulp / (2 b) >= 2^-18 ulp away from every midpoint: rounding the perturbed value rounds like the true
//     one, q1 = RN(a / b).
// That needs every intermediate normal: |a| in [2^-90, 2^90].  Anything else (zeros, denormals, huge values, infinities, NaN -- what a
// caller may have uploaded) takes the division instruction sequence.  tsdf_selftest_count_division compares the two for every
// mantissa of a and every b (tests/test_weight_storage.py).
__device__ inline float div_by_count(float a, float b) {
    const float y = __builtin_amdgcn_rcpf(b);
    const float q0 = a * y;
    const float r = __builtin_fmaf(-q0, b, a);
    float q = __builtin_fmaf(r, y, q0);
    const float aa = __builtin_fabsf(a);
    if (__builtin_expect(!(aa >= 0x1p-90f && aa <= 0x1p90f), 0)) q = a / b;
    return q;
}

constexpr int kPairFloats = 8;   // LDS per pair of planes: {cz, cz', m13 cz, m13 cz', m23 cz, m23 cz', m33 cz, m33 cz'}

// WBITS = 8 or 16: bits per weight; 32 / WBITS planes of one (x, y) share a dword, group g of planes at wpk + g * X * Y.
template <bool COUNT, int WBITS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TSDF_PACKED_WAVES, TSDF_PACKED_WAVES))) void integrate_packed_kernel(
    float *__restrict__ dist, uint32_t *__restrict__ wpk, const Geom g, const BrickGrid bg, const Mat44 ip, const Mat33 k, const uint32_t width,
    const uint32_t height, const uint16_t *__restrict__ depth, const uint16_t *__restrict__ depth_pad, unsigned long long *__restrict__ counter, const OccGrid occ,
    const uint32_t *__restrict__ list, const uint4 *__restrict__ boxes, const uint2 *__restrict__ coords, const uint32_t *__restrict__ count,
    const float4 *__restrict__ plane_const, uint8_t *__restrict__ touched) {
    constexpr int kPlanesPerWord = 32 / WBITS, kWords = kBatchZ / kPlanesPerWord;
    static_assert(kBatchZ == 4 && (WBITS == 8 || WBITS == 16), "a batch is two pairs of planes");
    __shared__ uint16_t tile[kTilePixels];                                   // the brick's pixel box inside a ring of zeros
    __shared__ __align__(16) float plane_lds[(kChunkZ + kBatchZ) / 2 * kPairFloats];
    const uint32_t tid = threadIdx.y * kTileX + threadIdx.x;
    const uint32_t n_active = *count;
    const size_t plane = (size_t)g.X * g.Y;
    const float neg_trunc = -g.trunc;
    // see round_quotients (integrate.hip): thr = 4e-7 * (max(width, height) + 2); the float just below 1/2 - thr
    const float round_near_half = __uint_as_float(__float_as_uint(0.5f - 4.0e-7f * ((float)max(width, height) + 2.0f)) - 1u);
    const uint32_t planes_resident = g.z_store_end - g.z_store_begin;
    uint32_t updated = 0;

    // One brick per workgroup (launch_integrate sizes the grid to the whole brick grid; workgroups beyond the list leave at once).  No loop
    // over bricks: what the prologue needs of the kernel's arguments is dead once the planes are walked, which the scalar register
    // file needs (with a loop around it the compiler kept it all live and spilled scalars into vector lanes: a v_readlane per use).
    {
        // The list's length first (the same word for every workgroup: a scalar-cache hit; the many workgroups beyond the list -- nine in ten
        // at 1024^3 -- leave here without touching memory), then the list entry, its box and the brick's coordinates in ONE scalar round
        // trip: all three are needed here, in scalar registers -- left alone the compiler spreads them over three dependent trips to L2
        // (the coordinates, the entry inside the `touched` store's branch, the box) in front of the tile's pixels.
        const uint32_t i = blockIdx.x;
        if (i >= n_active) return;
        const uint32_t b = list[i];
        const uint4 box = boxes[i];
        const uint2 co = coords[i];   // the brick's coordinates as the cull kernel had them (three divisions by run-time extents otherwise)
        asm volatile("" :: "s"(b), "s"(box.x), "s"(box.y), "s"(box.z), "s"(box.w), "s"(co.x), "s"(co.y));
#ifdef TSDF_DIAGNOSTICS
        const uint32_t bx = b % bg.nx, by = (b / bg.nx) % bg.ny, bz = b / (bg.nx * bg.ny);   // (the diagnostics may have re-sorted list and boxes on the host)
        (void)co;
#else
        const uint32_t bx = co.x & 0xffffu, by = co.x >> 16, bz = co.y;
#endif
        if (tid == 0) touched[b] = 1;   // for the next occupancy rebuild: this brick's distances may change (volume.hip)
        const uint32_t vx = bx * kTileX + threadIdx.x;
        const uint32_t vy = by * kTileY + threadIdx.y;
        const uint32_t z0 = g.z_store_begin + bz * kChunkZ;
        const uint32_t z_extra = bz + 1 == bg.nz ? bg.z_extra : 0u;
        const uint32_t z1 = min(z0 + kChunkZ + z_extra, g.z_store_end);  // exclusive
        // The tile: image columns box.x - lead .. box.x + box.z + lead - 1, rows box.y - 1 .. box.y + box.w; everything outside the
        // box itself is written as 0 (the ring).  The box holds every pixel a voxel of this brick can project to and lies inside the
        // image (brick_cull_kernel), so a voxel that misses the box fails the reference's frustum test (:349): its look-up, clamped
        // onto the ring, finds depth 0 = no update (:355).
        const uint32_t lead = bg.pair_loads ? 2u : 1u;
        const uint32_t pitch = box.z + 2u * lead, rows = box.w + 2u;
        const bool staged = box.z != 0 && pitch * rows <= (uint32_t)kTilePixels;
        if (tid < (uint32_t)(kChunkZ + kBatchZ)) {
            // z-only terms of the projection (brick_cull_kernel's side job), two planes side by side; planes this brick does not
            // hold get a NaN depth term: their voxels then compare false everywhere below
            const uint32_t p = z0 - g.z_store_begin + tid;
            float4 pc = make_float4(0.f, 0.f, 0.f, NAN);
            if (p < planes_resident + kBatchZ) pc = plane_const[p];
            if (z0 + tid >= z1) pc.w = NAN;
            float *dst = plane_lds + (tid >> 1) * kPairFloats + (tid & 1u);
            dst[0] = pc.x; dst[2] = pc.y; dst[4] = pc.z; dst[6] = pc.w;
        }
        if (staged) {
            constexpr uint32_t kStageBatch = 8;   // look-ups requested before the first is waited for
            if (bg.pair_loads) {
                // (even image width, 4-byte aligned image; the cull kernel has made box.x and box.z even: a lane takes two pixels)
                const uint32_t half = pitch >> 1, total2 = half * rows;
                const uint32_t *depth2 = reinterpret_cast<const uint32_t *>(depth);
                uint32_t *tile2 = reinterpret_cast<uint32_t *>(tile);
                // pair index of tile slot (0, 0) -- may lie before the image (box.y == 0 or box.x == 0); the clamped slots below do not
                const int64_t org2 = (((int64_t)box.y - 1) * (int64_t)width + (int64_t)box.x - 2) / 2;
                // slot p = tid + 256 u of the tile as (row, pair): one division per thread, then steps of 256 slots = (dq rows, dr pairs)
                // (eight divisions by `half` per batch, twice, were a quarter of the prologue's instructions)
                const uint32_t dq = (kTileX * kTileY) / half, dr = (kTileX * kTileY) - dq * half;
                uint32_t ty_ = tid / half, tx_ = tid - ty_ * half;
                for (uint32_t p0 = tid; p0 < total2; p0 += kTileX * kTileY * kStageBatch) {
                    uint32_t px[kStageBatch], row_[kStageBatch], col_[kStageBatch];
#pragma unroll
                    for (uint32_t u = 0; u < kStageBatch; u++) {
                        row_[u] = ty_; col_[u] = tx_;
                        const uint32_t cy = min(max(ty_, 1u), rows - 2u), cx2 = min(max(tx_, 1u), half - 2u);   // (inside the box, whatever the slot)
                        px[u] = depth2[org2 + (int64_t)cy * (int64_t)(width >> 1) + (int64_t)cx2];
                        tx_ += dr; ty_ += dq;
                        if (tx_ >= half) { tx_ -= half; ty_ += 1u; }
                    }
#pragma unroll
                    for (uint32_t u = 0; u < kStageBatch; u++) {
                        const uint32_t p = p0 + u * (kTileX * kTileY);
                        const bool ring = row_[u] == 0u || row_[u] == rows - 1u || col_[u] == 0u || col_[u] == half - 1u;
                        if (p < total2) tile2[p] = ring ? 0u : px[u];
                    }
                }
            } else {
                const uint32_t total = pitch * rows;
                for (uint32_t p0 = tid; p0 < total; p0 += kTileX * kTileY * kStageBatch) {
                    uint16_t px[kStageBatch];
#pragma unroll
                    for (uint32_t u = 0; u < kStageBatch; u++) {
                        const uint32_t p = min(p0 + u * (kTileX * kTileY), total - 1u);
                        const uint32_t ty = p / pitch, tx = p - ty * pitch;
                        const uint32_t cy = min(max(ty, 1u), rows - 2u), cx = min(max(tx, 1u), pitch - 2u);
                        px[u] = depth[(size_t)(box.y + cy - 1u) * width + (box.x + cx - 1u)];
                    }
#pragma unroll
                    for (uint32_t u = 0; u < kStageBatch; u++) {
                        const uint32_t p = p0 + u * (kTileX * kTileY);
                        const uint32_t ty = p / pitch, tx = p - ty * pitch;
                        const bool ring = ty == 0u || ty == rows - 1u || tx == 0u || tx == pitch - 1u;
                        if (p < total) tile[p] = ring ? (uint16_t)0 : px[u];
                    }
                }
            }
        }
        // (the per-lane constants of the walk are formed while the tile's pixels are on their way, in front of the barrier)

        // distance / weight addressing: one buffer descriptor per array based at this brick's first row (scalar registers), the plane as
        // the instruction's scalar byte offset, one 32-bit lane offset in bytes: no address arithmetic on the vector unit (the flat
        // form cost a 64-bit vector add per access).  launch_integrate keeps grids whose 36 planes pass 2^31 bytes on the fp32 kernel.
        const size_t brick_base = plane * (z0 - g.z_store_begin) + (size_t)g.X * (by * kTileY) + (size_t)bx * kTileX;
        const size_t wbrick_base = plane * ((z0 - g.z_store_begin) / kPlanesPerWord) + (size_t)g.X * (by * kTileY) + (size_t)bx * kTileX;
        const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(dist + brick_base, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(wpk + wbrick_base, 0, 0x7fffffff, 0x00020000);
        const uint32_t plane_bytes = (uint32_t)plane * 4u;
        const uint32_t lane_off4 = (threadIdx.y * g.X + threadIdx.x) * 4u;
        auto dsoff = [&](uint32_t zrel) { return plane_bytes * zrel; };                       // plane z0 + zrel of the distances
        auto wsoff = [&](uint32_t zrel) { return plane_bytes * (zrel / kPlanesPerWord); };   // the word holding plane z0 + zrel (z0 - z_store_begin is a multiple of 32)

        // voxel centre, x and y parts: initialise_deformation (src/TSDF/TSDFVolume.cu:783-784) then integrate_kernel's
        // offset + translation (:343); partial row sums of inv_pose * (c, 1): the reference evaluates ((m_i1 x + m_i2 y) + m_i3 z) + m_i4
        const float cx = ((((int)vx + 0.5f) * g.vs.x) + g.offset_clear.x) + g.offset.x;
        const float cy = ((((int)vy + 0.5f) * g.vs.y) + g.offset_clear.y) + g.offset.y;
        const float r1 = ip.m11 * cx + ip.m12 * cy;
        const float r2 = ip.m21 * cx + ip.m22 * cy;
        float r3 = ip.m31 * cx + ip.m32 * cy;
        if (vx >= g.X) r3 = NAN;   // lanes past the grid's x edge stay in (the marks at the end are made by the wave's first lanes): NaN depth, no update
        uint32_t low_lo = 0, low_hi = 0;   // bit o: my voxel of plane z0 + o got a distance that is not safely positive
        // (see integrate_kernel: voxels a boundary brick depends on are held to the stricter "flat" test)
        const bool rim_xy = occ.in_rim_zone(vx, occ.nbx) || occ.in_rim_zone(vy, occ.nby);
        const float flat_lo_open = __uint_as_float(__float_as_uint(occ.flat_lo) - 1u);
        const float mark_lo = rim_xy ? flat_lo_open : occ.tau, mark_hi = rim_xy ? occ.flat_hi : INFINITY;
        // look-up constants (exact small integers in fp32): byte address in the tile = fy * (2 pitch) + fx2,
        // fx2 = clamp(2 rx - 2 (box.x - lead)), fy = clamp(ry - (box.y - 1))
        // (depth_pad: image column x at padded column x + 1, row y at y + 1, pitch width + 2)
        const float x_org2 = staged ? -2.0f * (float)((int)box.x - (int)lead) : 2.0f, y_org = staged ? -(float)((int)box.y - 1) : 1.0f;
        const float fx2_lo = staged ? 2.0f * (float)(lead - 1u) : 0.0f, fx2_hi = staged ? 2.0f * (float)(lead + box.z) : 2.0f * (float)(width + 1u);
        const float fy_hi = staged ? (float)(box.w + 1u) : (float)(height + 1u);
        const float pitch2 = 2.0f * (float)(staged ? pitch : width + 2u);

        __syncthreads();
        if (vy >= g.Y) return;   // (a whole wave)
        // The walk over the brick's planes exists twice in the kernel, once per kind of look-up (a workgroup takes one): with both
        // kinds in one body every LDS read after the join waited for all outstanding memory loads -- the compiler cannot tell which
        // branch filled the destination registers -- and the batches stopped overlapping.
        auto walk = [&](auto staged_c) {
        constexpr bool kStaged = decltype(staged_c)::value;
        auto project_and_load = [&](const uint32_t o, bool (&upd_)[kBatchZ], float (&tsdf_)[kBatchZ], float (&pd_)[kBatchZ], uint32_t (&pw_)[kWords]) {
            // o = first plane of the batch relative to z0 (a multiple of 4)
            f2 rx_[2], ry_[2], camz_[2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const float *pl = plane_lds + ((o >> 1) + p) * kPairFloats;
                const f2 a2 = *reinterpret_cast<const f2 *>(pl + 2), b2 = *reinterpret_cast<const f2 *>(pl + 4), c2 = *reinterpret_cast<const f2 *>(pl + 6);
                // world_to_pixel (src/Utilities/cuda_coordinate_transforms.cu:10-30), standard camera: the 0 * x terms dropped
                const f2 camx = (f2{r1, r1} + a2) + f2{ip.m14, ip.m14};
                const f2 camy = (f2{r2, r2} + b2) + f2{ip.m24, ip.m24};
                const f2 camz = (f2{r3, r3} + c2) + f2{ip.m34, ip.m34};
                const f2 imx = f2{k.m11, k.m11} * camx + f2{k.m13, k.m13} * camz;
                const f2 imy = f2{k.m22, k.m22} * camy + f2{k.m23, k.m23} * camz;
                const f2 rc = f2{__builtin_amdgcn_rcpf(camz.x), __builtin_amdgcn_rcpf(camz.y)};
                const f2 qx = imx * rc, qy = imy * rc;
                f2 rx = f2{__builtin_rintf(qx.x), __builtin_rintf(qx.y)}, ry = f2{__builtin_rintf(qy.x), __builtin_rintf(qy.y)};
                const f2 dx = qx - rx, dy = qy - ry;
                // round_quotients' test (integrate.hip), once for the two planes: every |q - r| < 1/2 - thr, none NaN; otherwise this
                // lane redoes them with the IEEE divisions and roundf of the reference (:25-26), NaN -> 0 as the target's float -> int
                // conversion does
                const float near = max3_abs_keep_nan(max3_abs_keep_nan(0.0f, dx.x, dx.y), dy.x, dy.y);
                if (!(near < round_near_half)) {
                    rx.x = roundf(imx.x / camz.x); ry.x = roundf(imy.x / camz.x);
                    rx.y = roundf(imx.y / camz.y); ry.y = roundf(imy.y / camz.y);
                    if (rx.x != rx.x) rx.x = 0.0f;
                    if (ry.x != ry.x) ry.x = 0.0f;
                    if (rx.y != rx.y) rx.y = 0.0f;
                    if (ry.y != ry.y) ry.y = 0.0f;
                }
                rx_[p] = rx; ry_[p] = ry; camz_[p] = camz;
            }
            // look-up: the brick's tile in LDS, or -- no tile: the box is unknown (the brick straddles the camera plane) or too big --
            // the whole image inside its own ring of zeros in memory (depth_pad); same address arithmetic, other constants
            uint32_t d_[kBatchZ];   // depth of the voxel's pixel, 0 = none
#pragma unroll
            for (int p = 0; p < 2; p++) {
                f2 fx2 = __builtin_elementwise_fma(rx_[p], f2{2.0f, 2.0f}, f2{x_org2, x_org2});   // (exact: integers; huge or infinite values are clamped next)
                f2 fy = ry_[p] + f2{y_org, y_org};
                fx2.x = med3(fx2.x, fx2_lo, fx2_hi); fx2.y = med3(fx2.y, fx2_lo, fx2_hi);
                fy.x = med3(fy.x, 0.0f, fy_hi); fy.y = med3(fy.y, 0.0f, fy_hi);
                const f2 addr = __builtin_elementwise_fma(fy, f2{pitch2, pitch2}, fx2);           // (exact: < 2^24)
                if (kStaged) {
                    d_[2 * p] = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(tile) + (uint32_t)addr.x);
                    d_[2 * p + 1] = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(tile) + (uint32_t)addr.y);
                } else {
                    d_[2 * p] = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(depth_pad) + (uint32_t)addr.x);
                    d_[2 * p + 1] = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(depth_pad) + (uint32_t)addr.y);
                }
            }
            bool any = false;
#pragma unroll
            for (int j = 0; j < kBatchZ; j++) {
                // pixel_to_camera(...).z == depth, world_to_camera(...).z == camz for the standard camera (cuda_coordinate_transforms.cu:108-146)
                const float camz = (j & 1) ? camz_[j >> 1].y : camz_[j >> 1].x;
                const float sdf = (float)d_[j] - camz;
                // depth > 0 (:355) and sdf >= -trunc (:366); NaN camz (planes / lanes past the grid): false
                const bool update = d_[j] != 0 && sdf >= neg_trunc;
                // (sdf > 0) ? min(sdf, trunc) : sdf  ==  min(sdf, trunc)   (trunc > 0, sdf not NaN under `update`)
                upd_[j] = update;
                asm("v_min_f32 %0, %1, %2" : "=v"(tsdf_[j]) : "s"(g.trunc), "v"(sdf));   // (the plain instruction: fminf would first canonicalise both operands)
                if (update DIAG_NOLOAD) pd_[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(drsrc, lane_off4, dsoff(o + j), 0));
                any = any || update;
                if (kPlanesPerWord == 2 && (j & 1)) {
                    pw_[j >> 1] = 0u;
                    if (any DIAG_NOLOAD) pw_[j >> 1] = __builtin_amdgcn_raw_buffer_load_b32(wrsrc, lane_off4, wsoff(o + j - 1), 0);
                    any = false;
                }
            }
            if (kPlanesPerWord == 4) {
                pw_[0] = 0u;
                if (any DIAG_NOLOAD) pw_[0] = __builtin_amdgcn_raw_buffer_load_b32(wrsrc, lane_off4, wsoff(o), 0);
            }
        };
        auto blend_and_store = [&](const uint32_t o, const bool (&upd_)[kBatchZ], const float (&tsdf_)[kBatchZ], const float (&pd_)[kBatchZ], const uint32_t (&pw_)[kWords]) {
            const uint32_t zb = z0 + o;
            // (uniform) a batch with a plane in the z part of the rim zone takes the flat test on every lane
            const bool z_rim = zb < (uint32_t)(kBrick + kBrickGrow) || zb + (uint32_t)kBatchZ - 1u + (uint32_t)kBrickGrow >= (uint32_t)kBrick * (occ.nbz - 1u);
            const float lo = z_rim ? flat_lo_open : mark_lo, hi = z_rim ? occ.flat_hi : mark_hi;
            uint32_t nw_[kWords];
#pragma unroll
            for (int w = 0; w < kWords; w++) nw_[w] = pw_[w];
#pragma unroll
            for (int j = 0; j < kBatchZ; j++) {
                if (upd_[j]) {
                    constexpr uint32_t kMask = WBITS == 8 ? 0xffu : 0xffffu;
                    const int w = j / kPlanesPerWord, s = (j % kPlanesPerWord) * WBITS;
                    float prior_weight = (float)((pw_[w] >> s) & kMask);
                    asm("" : "+v"(prior_weight));   // (opaque: the compiler otherwise forms count + 1 in integers and converts a second time)
                    const float new_weight = prior_weight + 1.0f;                                                   // :375-376
                    const float new_distance = div_by_count((pd_[j] * prior_weight) + (tsdf_[j] * 1.0f), new_weight);   // :381
                    nw_[w] += 1u << s;   // (the caller has made room: weights.hip, weights_make_room)
                    if (true DIAG_NOSTORE_D) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, new_distance), drsrc, lane_off4, dsoff(o + j), 0);
                    if (!(new_distance > lo) || new_distance > hi) {   // not safely positive (rim zone: not flat): remember the plane
                        const uint32_t o_ = o + j;
                        if (o_ < 32u) low_lo |= 1u << o_; else low_hi |= 1u << (o_ - 32u);
                    }
                    if (COUNT) updated++;
                }
            }
#pragma unroll
            for (int w = 0; w < kWords; w++)
                if (nw_[w] != pw_[w] DIAG_NOSTORE_W) __builtin_amdgcn_raw_buffer_store_b32(nw_[w], wrsrc, lane_off4, wsoff(o + w * kPlanesPerWord), 0);
        };
        static_assert(kChunkZ % (2 * kBatchZ) == 0, "the pipeline alternates two register sets");
        float tsdf_a[kBatchZ], pd_a[kBatchZ], tsdf_b[kBatchZ], pd_b[kBatchZ];
        uint32_t pw_a[kWords], pw_b[kWords];
        bool upd_a[kBatchZ], upd_b[kBatchZ];   // (lane masks in scalar registers)
#if TSDF_PACKED_PIPELINE
        project_and_load(0, upd_a, tsdf_a, pd_a, pw_a);
#pragma unroll
        for (uint32_t o = 0; o < (uint32_t)kChunkZ; o += 2 * kBatchZ) {
            // (batches past z1 -- the last bricks of a grid whose depth is not a multiple of kChunkZ -- are all NaN planes)
            project_and_load(o + kBatchZ, upd_b, tsdf_b, pd_b, pw_b);
            blend_and_store(o, upd_a, tsdf_a, pd_a, pw_a);
            if (o + 2 * kBatchZ < (uint32_t)kChunkZ) project_and_load(o + 2 * kBatchZ, upd_a, tsdf_a, pd_a, pw_a);
            blend_and_store(o + kBatchZ, upd_b, tsdf_b, pd_b, pw_b);
        }
#else
        // (A/B of round 6: one register set, a batch's loads waited for before its blend: more waves per SIMD hide the round trip instead)
        (void)tsdf_b; (void)pd_b; (void)pw_b; (void)upd_b;
#pragma unroll
        for (uint32_t o = 0; o < (uint32_t)kChunkZ; o += kBatchZ) {
            project_and_load(o, upd_a, tsdf_a, pd_a, pw_a);
            blend_and_store(o, upd_a, tsdf_a, pd_a, pw_a);
        }
#endif
        if (z_extra != 0) {   // (uniform; after the pipeline, not inside it)
            project_and_load(kChunkZ, upd_a, tsdf_a, pd_a, pw_a);
            blend_and_store(kChunkZ, upd_a, tsdf_a, pd_a, pw_a);
        }
        };
        if (staged) walk(std::true_type{}); else walk(std::false_type{});
        if (__any((low_lo | low_hi) != 0u))
            mark_low_voxels(occ, low_lo, low_hi, (bx * kTileX) >> kBrickShift, __builtin_amdgcn_readfirstlane(vy), z0, z0, z1 - 1u, threadIdx.x);
    }
    if (COUNT) {   // (waves that left early counted nothing)
        for (int o = 32; o > 0; o >>= 1) updated += __shfl_down(updated, o);
        if ((threadIdx.x & 63u) == 0 && updated) atomicAdd(counter, (unsigned long long)updated);
    }
}

// Launched by launch_integrate (integrate.hip) in place of integrate_kernel<false, *, true> when the volume's weights are packed.
int launch_integrate_packed_kernel(tsdf_volume *v, dim3 grid, const BrickGrid &bg, const Mat44 &ip, const Mat33 &mk, uint32_t width,
                                   uint32_t height, const uint16_t *d_depth, unsigned long long *counter_arg, const uint4 *boxes,
                                   const uint2 *coords, const uint32_t *count, const float4 *plane_const) {
    const dim3 block(kTileX, kTileY, 1);
#define LAUNCH(CNT, BITS)                                                                                                       \
    TSDF_LAUNCH_TIMED(v, 0, (integrate_packed_kernel<CNT, BITS>), grid, block, v->dist, v->wpacked, v->g, bg, ip, mk, width, height, d_depth, v->depth_pad, \
                      counter_arg, v->occ, v->brick_list, boxes, coords, count, plane_const, v->touched)
    if (v->wmode == 8) {
        if (v->counting) LAUNCH(true, 8); else LAUNCH(false, 8);
    } else if (v->wmode == 16) {
        if (v->counting) LAUNCH(true, 16); else LAUNCH(false, 16);
    } else {
        set_error("integrate: packed kernel asked for with fp32 weights");
        return TSDF_ERR_INVALID;
    }
#undef LAUNCH
    return TSDF_OK;
}

// div_by_count against the division it replaces: one workgroup per divisor, every mantissa of the dividend, both signs
__global__ __launch_bounds__(256) void count_division_check_kernel(uint32_t b0, float scale, unsigned long long *__restrict__ bad) {
    const float b = (float)(b0 + blockIdx.x);
    uint32_t n = 0;
    for (uint32_t m = threadIdx.x; m < (1u << 23); m += 256) {
        const float a = __uint_as_float(0x3f800000u | m) * scale;   // (a power of two: exact)
        n += __float_as_uint(div_by_count(a, b)) != __float_as_uint(a / b);
        n += __float_as_uint(div_by_count(-a, b)) != __float_as_uint(-a / b);
    }
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
    if ((threadIdx.x & 63u) == 0 && n) atomicAdd(bad, (unsigned long long)n);
}
__global__ void count_division_special_kernel(const float *__restrict__ a, uint32_t n_a, uint32_t b_end, unsigned long long *__restrict__ bad) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_a * b_end) return;
    const float x = a[t % n_a], b = (float)(t / n_a + 1u);
    const float q = div_by_count(x, b), r = x / b;
    if (__float_as_uint(q) != __float_as_uint(r) && !(q != q && r != r)) atomicAdd(bad, 1ull);
}

}  // namespace tsdf

extern "C" int tsdf_selftest_count_division(uint32_t b_begin, uint32_t b_end, unsigned long long *mismatches) {
    using namespace tsdf;
    TSDF_REQUIRE(mismatches && b_begin >= 1 && b_begin < b_end && b_end <= (1u << 17) + 1u, "tsdf_selftest_count_division: divisors are 1 .. 2^17");
    unsigned long long *bad = nullptr;
    TSDF_HIP(hipMalloc((void **)&bad, sizeof(*bad)), "selftest alloc");
    hipError_t e = hipMemset(bad, 0, sizeof(*bad));
    // every mantissa at three scales: 1, the bottom and the top of the range the short sequence is taken in
    for (float scale : {1.0f, 0x1p-90f, 0x1p89f})
        if (e == hipSuccess) {
            hipLaunchKernelGGL(count_division_check_kernel, dim3(b_end - b_begin), dim3(256), 0, nullptr, b_begin, scale, bad);
            e = hipGetLastError();
        }
    // and the values outside it (the division's own instruction sequence is taken: equal by construction, checked all the same)
    const float special[] = {0.0f, -0.0f, 1.0e-45f, -1.0e-40f, 0x1p-126f, 0x1p-91f, 0x1.fffffep-91f, 0x1.000002p90f, 0x1p100f, -0x1p127f,
                             3.4028235e38f, INFINITY, -INFINITY, NAN};
    const uint32_t n_a = sizeof(special) / sizeof(special[0]);
    float *a_dev = nullptr;
    if (e == hipSuccess) e = hipMalloc((void **)&a_dev, sizeof(special));
    if (e == hipSuccess) e = hipMemcpy(a_dev, special, sizeof(special), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const uint32_t n = n_a * (b_end - 1u);
        hipLaunchKernelGGL(count_division_special_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, a_dev, n_a, b_end - 1u, bad);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(mismatches, bad, sizeof(*bad), hipMemcpyDeviceToHost);
    (void)hipFree(bad);
    if (a_dev) (void)hipFree(a_dev);
    if (e != hipSuccess) return hip_fail(e, "tsdf_selftest_count_division");
    return TSDF_OK;
}
