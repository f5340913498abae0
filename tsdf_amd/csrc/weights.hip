// How a volume stores its weights.
//
// The reference keeps one float per voxel (m_weights, src/include/TSDFVolume.hpp) and integrate_kernel only ever adds 1 to it (the
// clamp to max_weight is commented out, src/TSDF/TSDFVolume.cu:375-377): unless a caller uploads something else, a weight is the
// number of frames that have updated the voxel.  Integration is a read-modify-write of distance and weight of every updated voxel
// and is bound by that traffic, so the count is kept as narrow as it can be:
//   wmode 8   one byte per voxel, the bytes of planes 4g .. 4g + 3 (counted from the first resident plane) of one (x, y) in one dword
//             at wpacked[g * X * Y + y * X + x] -- the four planes a lane of integrate_packed_kernel walks as one batch;
//   wmode 16  two bytes per voxel, planes 2g, 2g + 1 in one dword;
//   wmode 0   the reference's layout: fp32, `weight`, index x + y X + z X Y.
// A volume starts at wmode 8 (TSDF_WEIGHT_PACK = 8 | 16 | 0 picks the starting mode), moves to 16 before the integration that could
// take a count past 255, and to fp32 before the one that could pass 65535 (`weight_bound` counts integrations since the weights were
// last known; when it reaches the mode's limit the counts themselves are looked at first, so a stream whose camera moves on keeps its
// bytes far beyond 255 frames) -- or at once when it needs the general kernel (custom deformation nodes, a camera that is not of the standard shape),
// when weights that are not such counts are uploaded, or when the caller asks for the device pointer (tsdf_volume_weights: the
// reference's weight_data(); from then on the volume keeps the reference's layout, "pinned").  clear() returns an unpinned volume to
// the starting mode.  Every accessor speaks fp32 whatever the mode; the arithmetic of integration is the same exact fp32 expression in
// all three ((float)count is exact), tests/test_weight_storage.py.
#include <algorithm>

#include "common.hpp"

namespace tsdf {

static size_t packed_words(const tsdf_volume *v, int bits) {
    const size_t per = 32 / bits, planes = v->g.z_store_end - v->g.z_store_begin;
    return (size_t)v->g.X * v->g.Y * ((planes + per - 1) / per);
}

// one thread per dword of the packed array: its planes out as floats
template <int BITS>
__global__ __launch_bounds__(256) void weights_expand_kernel(const uint32_t *__restrict__ wp, float *__restrict__ w, size_t xy, uint32_t planes, size_t n_words) {
    constexpr uint32_t kPer = 32 / BITS, kMask = BITS == 8 ? 0xffu : 0xffffu;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (size_t)gridDim.x * 256) {
        const size_t grp = i / xy, in_plane = i - grp * xy;
        const uint32_t word = wp[i];
#pragma unroll
        for (uint32_t s = 0; s < kPer; s++) {
            const size_t z = grp * kPer + s;
            if (z < planes) w[z * xy + in_plane] = (float)((word >> (BITS * s)) & kMask);
        }
    }
}
// 8 -> 16 bits: one thread per dword of the 8-bit array, two dwords out
__global__ __launch_bounds__(256) void weights_widen_kernel(const uint32_t *__restrict__ w8, uint32_t *__restrict__ w16, size_t xy, size_t n_words8, size_t n_words16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_words8; i += (size_t)gridDim.x * 256) {
        const size_t grp = i / xy, in_plane = i - grp * xy;
        const uint32_t word = w8[i];
        const size_t o = (2 * grp) * xy + in_plane;
        w16[o] = (word & 0xffu) | ((word & 0xff00u) << 8);
        if (o + xy < n_words16) w16[o + xy] = ((word >> 16) & 0xffu) | ((word >> 24) << 16);
    }
}
// fp32 -> packed: one thread per dword out.  stats[0] |= 1 when a weight is not an integer in [0, 65535]; stats[1] = the largest weight
// (as the bits of a non-negative float: they order like the values)
template <int BITS>
__global__ __launch_bounds__(256) void weights_pack_kernel(const float *__restrict__ w, uint32_t *__restrict__ wp, size_t xy, uint32_t planes, size_t n_words) {
    constexpr uint32_t kPer = 32 / BITS;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (size_t)gridDim.x * 256) {
        const size_t grp = i / xy, in_plane = i - grp * xy;
        uint32_t word = 0;
#pragma unroll
        for (uint32_t s = 0; s < kPer; s++) {
            const size_t z = grp * kPer + s;
            if (z < planes) word |= (uint32_t)w[z * xy + in_plane] << (BITS * s);
        }
        wp[i] = word;
    }
}
__global__ __launch_bounds__(256) void weights_survey_kernel(const float *__restrict__ w, size_t n, uint32_t *__restrict__ stats) {
    uint32_t bad = 0, top = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float x = w[i];
        // a count: an integer in [0, 65535] (-0 is not: its bits differ from +0's)
        const bool ok = x >= 0.0f && x <= 65535.0f && x == truncf(x) && __float_as_uint(x) != 0x80000000u;
        bad |= ok ? 0u : 1u;
        if (ok) top = max(top, (uint32_t)x);
    }
    for (int o = 32; o > 0; o >>= 1) {
        bad |= (uint32_t)__shfl_down((int)bad, o);
        top = max(top, (uint32_t)__shfl_down((int)top, o));
    }
    if ((threadIdx.x & 63u) == 0) {
        if (bad) atomicOr(&stats[0], 1u);
        if (top) atomicMax(&stats[1], top);
    }
}

static int weight_pack_start() { return tuning().weight_pack; }

// the packed array for `bits`, zeroed or not
static int alloc_packed(tsdf_volume *v, int bits, uint32_t **out) {
    const size_t words = packed_words(v, bits);
    TSDF_HIP(hipMalloc((void **)out, words * sizeof(uint32_t)), "Couldn't allocate space for TSDF weights");
    return TSDF_OK;
}

int weights_create(tsdf_volume *v) {
    v->wmode = weight_pack_start();
    v->weight_bound = 0;
    v->weight_pinned = 0;
    if (v->wmode == 0) {
        TSDF_HIP(hipMalloc((void **)&v->weight, v->resident_voxels() * sizeof(float)), "Couldn't allocate space for TSDF weights");
        return TSDF_OK;
    }
    return alloc_packed(v, v->wmode, &v->wpacked);
}

void weights_destroy(tsdf_volume *v) {
    if (v->weight) (void)hipFree(v->weight);
    if (v->wpacked) (void)hipFree(v->wpacked);
    v->weight = nullptr;
    v->wpacked = nullptr;
}

// fp32 weights out of the packed array into `dst` (resident_voxels() floats), on the volume's stream
static int expand_into(const tsdf_volume *v, float *dst) {
    const size_t xy = (size_t)v->g.X * v->g.Y, words = packed_words(v, v->wmode);
    const uint32_t planes = v->g.z_store_end - v->g.z_store_begin;
    const dim3 grid((unsigned)std::min<size_t>((words + 255) / 256, 4096));
    if (v->wmode == 8) hipLaunchKernelGGL(weights_expand_kernel<8>, grid, dim3(256), 0, v->stream, v->wpacked, dst, xy, planes, words);
    else hipLaunchKernelGGL(weights_expand_kernel<16>, grid, dim3(256), 0, v->stream, v->wpacked, dst, xy, planes, words);
    TSDF_HIP(hipGetLastError(), "Couldn't expand the weights");
    return TSDF_OK;
}

// get / set_weight_data in a packed mode go through a BOUNDED staging buffer, a few planes at a time (a multiple of 4: whole dwords of
// either packed mode) -- not through an fp32 copy of the whole array (4 GiB at 1024^3), which the reference's plain memcpy never needed.
constexpr size_t kWeightStageBytes = (size_t)64 << 20;
static uint32_t stage_planes(const tsdf_volume *v) {
    const size_t xy = (size_t)v->g.X * v->g.Y, planes = v->g.z_store_end - v->g.z_store_begin;
    size_t n = std::max<size_t>(4, (kWeightStageBytes / (xy * sizeof(float))) & ~(size_t)3);
    return (uint32_t)std::min(n, (planes + 3) & ~(size_t)3);
}
// planes [z0, z0 + nz) of the packed array `wp` (mode `bits`) <-> `stage` (nz * xy floats), on the volume's stream
static int stage_expand(const tsdf_volume *v, const uint32_t *wp, int bits, uint32_t z0, uint32_t nz, float *stage) {
    const size_t xy = (size_t)v->g.X * v->g.Y, per = 32 / bits, words = xy * ((nz + per - 1) / per);
    const dim3 grid((unsigned)std::min<size_t>((words + 255) / 256, 4096));
    if (bits == 8) hipLaunchKernelGGL(weights_expand_kernel<8>, grid, dim3(256), 0, v->stream, wp + (z0 / per) * xy, stage, xy, nz, words);
    else hipLaunchKernelGGL(weights_expand_kernel<16>, grid, dim3(256), 0, v->stream, wp + (z0 / per) * xy, stage, xy, nz, words);
    TSDF_HIP(hipGetLastError(), "Couldn't expand the weights");
    return TSDF_OK;
}
static int stage_pack(const tsdf_volume *v, uint32_t *wp, int bits, uint32_t z0, uint32_t nz, const float *stage) {
    const size_t xy = (size_t)v->g.X * v->g.Y, per = 32 / bits, words = xy * ((nz + per - 1) / per);
    const dim3 grid((unsigned)std::min<size_t>((words + 255) / 256, 4096));
    if (bits == 8) hipLaunchKernelGGL(weights_pack_kernel<8>, grid, dim3(256), 0, v->stream, stage, wp + (z0 / per) * xy, xy, nz, words);
    else hipLaunchKernelGGL(weights_pack_kernel<16>, grid, dim3(256), 0, v->stream, stage, wp + (z0 / per) * xy, xy, nz, words);
    TSDF_HIP(hipGetLastError(), "Couldn't pack the weights");
    return TSDF_OK;
}

// The reference's layout from here on (until clear(), unless pinned).
int weights_require_f32(tsdf_volume *v) {
    if (v->wmode == 0) return TSDF_OK;
    float *w = nullptr;
    TSDF_HIP(hipMalloc((void **)&w, v->resident_voxels() * sizeof(float)), "Couldn't allocate space for TSDF weights");
    const int rc = expand_into(v, w);
    if (rc != TSDF_OK) {
        (void)hipFree(w);
        return rc;
    }
    TSDF_HIP(hipStreamSynchronize(v->stream), "Couldn't expand the weights");
    (void)hipFree(v->wpacked);
    v->wpacked = nullptr;
    v->weight = w;
    v->wmode = 0;
    return TSDF_OK;
}

// the largest count in the packed array (one streaming read: 30 us at 512^3)
template <int BITS>
__global__ __launch_bounds__(256) void weights_max_kernel(const uint32_t *__restrict__ wp, size_t n_words, uint32_t *__restrict__ top_out) {
    uint32_t top = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (size_t)gridDim.x * 256) {
        const uint32_t w = wp[i];
        if (BITS == 8) top = max(max(top, w & 0xffu), max(max((w >> 8) & 0xffu, (w >> 16) & 0xffu), w >> 24));
        else top = max(top, max(w & 0xffffu, w >> 16));
    }
    for (int o = 32; o > 0; o >>= 1) top = max(top, (uint32_t)__shfl_down((int)top, o));
    if ((threadIdx.x & 63u) == 0 && top) atomicMax(top_out, top);
}
// `weight_bound` counts integrations, not updates of one voxel: when it reaches what the mode holds, look at the counts themselves --
// a camera that moves on leaves every voxel far below the number of frames, and the volume keeps its narrow counts.
static int refresh_bound(tsdf_volume *v) {
    uint32_t *slot = reinterpret_cast<uint32_t *>(v->counter_dev + 3);   // (scratch slot shared with verify_fast_division)
    TSDF_HIP(hipMemsetAsync(slot, 0, sizeof(uint32_t), v->stream), "weights: bound");
    const size_t words = packed_words(v, v->wmode);
    const dim3 grid((unsigned)std::min<size_t>((words + 255) / 256, 4096));
    if (v->wmode == 8) hipLaunchKernelGGL(weights_max_kernel<8>, grid, dim3(256), 0, v->stream, v->wpacked, words, slot);
    else hipLaunchKernelGGL(weights_max_kernel<16>, grid, dim3(256), 0, v->stream, v->wpacked, words, slot);
    uint32_t top = 0;
    TSDF_HIP(hipMemcpyAsync(&top, slot, sizeof(top), hipMemcpyDeviceToHost, v->stream), "weights: bound");
    TSDF_HIP(hipStreamSynchronize(v->stream), "weights: bound");
    v->weight_bound = top;
    return TSDF_OK;
}

// Before an integration in a packed mode: no count may pass what the mode holds.
// 8-bit counts -> 16-bit counts (the new array is allocated before the old one goes)
static int widen_to_16(tsdf_volume *v) {
    uint32_t *w16 = nullptr;
    int rc = alloc_packed(v, 16, &w16);
    if (rc != TSDF_OK) return rc;
    const size_t xy = (size_t)v->g.X * v->g.Y, n8 = packed_words(v, 8), n16 = packed_words(v, 16);
    hipLaunchKernelGGL(weights_widen_kernel, dim3((unsigned)std::min<size_t>((n8 + 255) / 256, 4096)), dim3(256), 0, v->stream, v->wpacked, w16, xy, n8, n16);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(v->stream);
    if (e != hipSuccess) {
        (void)hipFree(w16);
        return hip_fail(e, "Couldn't widen the weights");
    }
    (void)hipFree(v->wpacked);
    v->wpacked = w16;
    v->wmode = 16;
    return TSDF_OK;
}

// The look at the counts (a scan of the packed array and a host round trip inside integrate) must not come back every few frames: a
// region seen 250 times that then leaves the view would leave the bound at 250 and the scan due every 5th frame for the rest of the
// session.  So the mode is widened already when the largest count is in the top quarter of what it holds, and a scan that keeps the
// mode buys at least 64 (16 384) integrations without another.
int weights_make_room(tsdf_volume *v) {
    bool widen = false;
    if ((v->wmode == 8 && v->weight_bound >= 255u) || (v->wmode == 16 && v->weight_bound >= 65535u)) {
        const int rc = refresh_bound(v);
        if (rc != TSDF_OK) return rc;
        widen = v->wmode == 8 ? v->weight_bound >= 192u : v->weight_bound >= 49152u;
    }
    if (v->wmode == 8 && widen) {
        const int rc = widen_to_16(v);
        if (rc != TSDF_OK) return rc;
    }
    if (v->wmode == 16 && (v->weight_bound >= 65535u || (widen && v->weight_bound >= 49152u))) return weights_require_f32(v);
    return TSDF_OK;
}

// clear(): every weight 0 (on the volume's stream); an unpinned volume goes back to its starting mode
int weights_clear(tsdf_volume *v) {
    const int start = v->weight_pinned ? 0 : weight_pack_start();
    if (v->wmode != start) {
        TSDF_HIP(hipStreamSynchronize(v->stream), "clear");
        weights_destroy(v);
        v->wmode = start;
        if (start == 0) TSDF_HIP(hipMalloc((void **)&v->weight, v->resident_voxels() * sizeof(float)), "Couldn't allocate space for TSDF weights");
        else {
            const int rc = alloc_packed(v, start, &v->wpacked);
            if (rc != TSDF_OK) return rc;
        }
    }
    v->weight_bound = 0;
    if (v->wmode == 0) TSDF_HIP(hipMemsetAsync(v->weight, 0, v->resident_voxels() * sizeof(float), v->stream), "Couldn't clear TSDF weights");
    else TSDF_HIP(hipMemsetAsync(v->wpacked, 0, packed_words(v, v->wmode) * sizeof(uint32_t), v->stream), "Couldn't clear TSDF weights");
    return TSDF_OK;
}

// set_weight_data(): counts are packed (into the narrowest mode from the starting one that holds them), anything else is kept as fp32.
// The host array is surveyed first, staged a few planes at a time (are these counts? the largest?), and only then is anything of the
// volume's replaced: a failed allocation leaves the volume as it was.
int weights_upload(tsdf_volume *v, const float *host) {
    const size_t n = v->resident_voxels(), xy = (size_t)v->g.X * v->g.Y;
    const uint32_t planes = v->g.z_store_end - v->g.z_store_begin;
    TSDF_HIP(hipStreamSynchronize(v->stream), "Couldn't set weight data");
    const bool keep_f32 = v->weight_pinned || weight_pack_start() == 0;
    if (v->wmode == 0 && keep_f32) {
        TSDF_HIP(hipMemcpyAsync(v->weight, host, n * sizeof(float), hipMemcpyHostToDevice, v->stream), "Couldn't set weight data");
        TSDF_HIP(hipStreamSynchronize(v->stream), "Couldn't set weight data");
        return TSDF_OK;
    }
    const uint32_t chunk = stage_planes(v);
    float *stage = nullptr;
    TSDF_HIP(hipMalloc((void **)&stage, (size_t)chunk * xy * sizeof(float)), "Couldn't allocate space for TSDF weights");
    uint32_t stats[2] = {0, 0};
    uint32_t *slot = reinterpret_cast<uint32_t *>(v->counter_dev + 3);   // (scratch slot shared with verify_fast_division)
    hipError_t e = hipMemsetAsync(slot, 0, sizeof(stats), v->stream);
    for (uint32_t z0 = 0; z0 < planes && e == hipSuccess; z0 += chunk) {
        const uint32_t nz = std::min(chunk, planes - z0);
        e = hipMemcpyAsync(stage, host + (size_t)z0 * xy, (size_t)nz * xy * sizeof(float), hipMemcpyHostToDevice, v->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(weights_survey_kernel, dim3(2048), dim3(256), 0, v->stream, stage, (size_t)nz * xy, slot);
            e = hipStreamSynchronize(v->stream);   // (the staging buffer is reused)
        }
    }
    if (e == hipSuccess) e = hipMemcpyAsync(stats, slot, sizeof(stats), hipMemcpyDeviceToHost, v->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(v->stream);
    if (e == hipSuccess) e = hipMemsetAsync(slot, 0, sizeof(stats), v->stream);
    if (e != hipSuccess) {
        (void)hipFree(stage);
        return hip_fail(e, "Couldn't set weight data");
    }
    const int want = (stats[0] || keep_f32) ? 0 : (stats[1] <= 255u && weight_pack_start() == 8 ? 8 : 16);
    // the array of the mode the weights will live in: the present one, or a new one allocated BEFORE the old one goes
    float *new_f32 = nullptr;
    uint32_t *new_packed = nullptr;
    if (want == 0 && v->wmode != 0) e = hipMalloc((void **)&new_f32, n * sizeof(float));
    if (want != 0 && v->wmode != want) {
        if (alloc_packed(v, want, &new_packed) != TSDF_OK) e = hipErrorOutOfMemory;
    }
    if (e != hipSuccess) {
        (void)hipFree(stage);
        return hip_fail(e, "Couldn't allocate space for TSDF weights");
    }
    if (want == 0) {   // not counts: the reference's layout
        float *dst = new_f32 ? new_f32 : v->weight;
        e = hipMemcpyAsync(dst, host, n * sizeof(float), hipMemcpyHostToDevice, v->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(v->stream);
        (void)hipFree(stage);
        if (e != hipSuccess) {
            if (new_f32) (void)hipFree(new_f32);
            return hip_fail(e, "Couldn't set weight data");
        }
        if (new_f32) {
            (void)hipFree(v->wpacked);
            v->wpacked = nullptr;
            v->weight = new_f32;
            v->wmode = 0;
        }
        return TSDF_OK;
    }
    uint32_t *dst = new_packed ? new_packed : v->wpacked;
    int rc = TSDF_OK;
    for (uint32_t z0 = 0; z0 < planes && e == hipSuccess && rc == TSDF_OK; z0 += chunk) {
        const uint32_t nz = std::min(chunk, planes - z0);
        e = hipMemcpyAsync(stage, host + (size_t)z0 * xy, (size_t)nz * xy * sizeof(float), hipMemcpyHostToDevice, v->stream);
        if (e == hipSuccess) rc = stage_pack(v, dst, want, z0, nz, stage);
        if (e == hipSuccess && rc == TSDF_OK) e = hipStreamSynchronize(v->stream);
    }
    (void)hipFree(stage);
    if (e != hipSuccess || rc != TSDF_OK) {   // (the volume keeps what it had; its weights may be partly overwritten when the mode did not change)
        if (new_packed) (void)hipFree(new_packed);
        return e != hipSuccess ? hip_fail(e, "Couldn't set weight data") : rc;
    }
    if (new_packed) {
        if (v->wpacked) (void)hipFree(v->wpacked);
        if (v->weight) (void)hipFree(v->weight);
        v->weight = nullptr;
        v->wpacked = new_packed;
        v->wmode = want;
    }
    v->weight_bound = stats[1];
    return TSDF_OK;
}

// get_weight_data(): fp32 on the host whatever the mode
int weights_download(const tsdf_volume *v, float *host) {
    const size_t n = v->resident_voxels(), xy = (size_t)v->g.X * v->g.Y;
    if (v->wmode == 0) {
        TSDF_HIP(hipMemcpyAsync(host, v->weight, n * sizeof(float), hipMemcpyDeviceToHost, v->stream), "Couldn't read weight data");
        TSDF_HIP(hipStreamSynchronize(v->stream), "Couldn't read weight data");
        return TSDF_OK;
    }
    const uint32_t planes = v->g.z_store_end - v->g.z_store_begin, chunk = stage_planes(v);
    float *stage = nullptr;
    TSDF_HIP(hipMalloc((void **)&stage, (size_t)chunk * xy * sizeof(float)), "Couldn't read weight data");
    int rc = TSDF_OK;
    hipError_t e = hipSuccess;
    for (uint32_t z0 = 0; z0 < planes && rc == TSDF_OK && e == hipSuccess; z0 += chunk) {
        const uint32_t nz = std::min(chunk, planes - z0);
        rc = stage_expand(v, v->wpacked, v->wmode, z0, nz, stage);
        if (rc == TSDF_OK) e = hipMemcpyAsync(host + (size_t)z0 * xy, stage, (size_t)nz * xy * sizeof(float), hipMemcpyDeviceToHost, v->stream);
        if (rc == TSDF_OK && e == hipSuccess) e = hipStreamSynchronize(v->stream);   // (the staging buffer is reused)
    }
    (void)hipFree(stage);
    if (rc != TSDF_OK) return rc;
    if (e != hipSuccess) return hip_fail(e, "Couldn't read weight data");
    return TSDF_OK;
}

}  // namespace tsdf

using namespace tsdf;

extern "C" {

int tsdf_volume_set_weight_storage(tsdf_volume *v, int bits_per_weight) {
    TSDF_REQUIRE(v, "null argument");
    TSDF_REQUIRE(bits_per_weight == 8 || bits_per_weight == 16 || bits_per_weight == 32, "tsdf_volume_set_weight_storage: 8, 16 or 32 bits");
    const int now = v->wmode == 0 ? 32 : v->wmode;
    TSDF_REQUIRE(bits_per_weight >= now, "tsdf_volume_set_weight_storage: the storage can only be widened (clear() returns to the starting mode)");
    int rc = TSDF_OK;
    if (now == 8 && bits_per_weight >= 16) rc = widen_to_16(v);
    if (rc == TSDF_OK && bits_per_weight == 32) rc = weights_require_f32(v);
    return rc;
}

int tsdf_volume_weight_storage(const tsdf_volume *v, int *bits_per_weight, int *pinned) {
    TSDF_REQUIRE(v && bits_per_weight, "null argument");
    *bits_per_weight = v->wmode == 0 ? 32 : v->wmode;
    if (pinned) *pinned = v->weight_pinned;
    return TSDF_OK;
}

}  // extern "C"
