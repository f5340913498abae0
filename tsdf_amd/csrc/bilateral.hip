// Bilateral depth filter for gfx950.  Replaces BilateralFilter (src/BilateralFilter.cpp:15-130)
// of the reference, which is a single-threaded host loop.
//
// One lane per pixel, 16x16 pixel tile per workgroup (wave = 16x4 strip); the tile plus an
// r-pixel apron is staged once in LDS (uint16), so each of the (2r+1)^2 taps is an LDS read
// instead of a global gather.  The two lookup tables are built on the host with the same
// std::exp(float) calls as the reference's constructor and uploaded, so weights are
// bit-identical; the per-tap accumulation is done in double and narrowed to float each tap,
// as the reference's mixed float/double expressions do (:99-102).
//
// Border behaviour (Q12) is kept: the reference advances its kernel index only for in-image
// taps, so near a border tap (cx,cy) uses kernel[(cx - first_x) * n_valid_y + (cy - first_y)].
//
// 16-bit images: the reference indexes its 256-entry similarity table with |dI| up to 65535 and
// writes one byte per pixel into a 2-byte-per-pixel buffer -- undefined behaviour.  Defined
// semantics here: similarity(d) = exp(-d / sigma_colour^2) for every d in 0..65535
// (a 65536-entry table whose first 256 entries are the reference's), 16-bit store per pixel.
#include <cmath>
#include <cstdlib>
#include <new>
#include <type_traits>
#include <vector>

#include "common.hpp"

namespace tsdf {

constexpr int kBTile = 16;
constexpr int kMaxRadius = 24;  // LDS tile (16+2*24)^2 * 12 B = 48 KiB

constexpr int kSimLds = 4096;   // similarity entries staged in LDS (intensity differences below this; the rest stay in global memory)

// One tap of the reference's accumulation (src/BilateralFilter.cpp:99-102):
//     double conv_weight = kernel * similarity;          the float product, widened
//     sum          += conv_weight * intensity;           (float)((double)sum + conv_weight * (double)intensity)
//     total_weight += conv_weight;                       (float)((double)total_weight + conv_weight)
// The double product has at most 24 + 16 significant bits, so it is exact and mul-then-add in double equals ONE fused
// multiply-add in double (same single rounding).  total_weight + weight evaluated in double and narrowed equals the fp32 sum:
// both operands are floats, so the double sum is exact unless the smaller is below 2^-29 of the larger, and then both
// roundings return the larger operand (weights are >= 0).
__device__ inline void bilateral_tap(float weight, double intensity, float &sum, float &total_weight) {
    sum = (float)__builtin_fma((double)weight, intensity, (double)sum);
    total_weight = total_weight + weight;
}

// |a - b| of two unsigned words in one instruction
__device__ inline unsigned sad_u32(unsigned a, unsigned b) {
    unsigned d;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// R > 0: the radius is known at compile time (the tap rows unroll, LDS offsets become immediates) and the similarity table's
// head is staged in LDS; R == 0: run-time radius, plain loops.
//
// LDS: the tile as doubles (the widened operand of the reference's product, converted once per pixel instead of once per tap)
// and as 4 * intensity in 32-bit words (|4a - 4b| is the byte offset of similarity[|a - b|]); the spatial kernel; the head of
// the similarity table.
//
// The accumulation is one serial chain per pixel (every tap rounds the running sum to float, :101), and a chain that also waits
// for its own look-ups spends ~400 cycles per tap (measured in round 2: the kernel took exactly as long as ONE wave's 225
// dependent taps, whatever the instruction count -- 51 us).  A tap's weight does not depend on the sum, so the weights of
// tap column i + 1 are looked up (LDS reads only: the waits can be counted) while the chain works through column i.
template <typename PIX, int R>
__global__ __launch_bounds__(256) void bilateral_kernel(const PIX *__restrict__ in, PIX *__restrict__ out,
                                                        int width, int height, int radius_arg,
                                                        const float *__restrict__ kernel,
                                                        const float *__restrict__ similarity,
                                                        uint16_t *__restrict__ tile_max) {
    constexpr bool STAGED = R > 0;
    // tile_max != nullptr: the workgroup also leaves the largest value of its 16 x 16 output tile in tile_max[tile] -- what
    // integrate's depth_tile_max_kernel would compute from the filtered image in a launch of its own (tsdf_integrate_device_tiles)
    __shared__ unsigned wg_max;
    if (threadIdx.x == 0) wg_max = 0u;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int radius = R > 0 ? R : radius_arg;
    const int n = 2 * radius + 1;
    const int span = kBTile + 2 * radius;
    // (the double-precision copy of the tile serves the run-time-radius path only: the staged path widens the intensity from the
    // word it has read for the similarity look-up, and neither allocates nor fills the copy -- 7 KB of LDS per workgroup less)
    double *tile_d = reinterpret_cast<double *>(smem_raw);
    float *kern_lds = reinterpret_cast<float *>(tile_d + (STAGED ? 0 : span * span));
    float *sim_lds = kern_lds + n * n;
    uint32_t *tile4 = reinterpret_cast<uint32_t *>(sim_lds + (STAGED ? kSimLds : 0));
    const int tx0 = blockIdx.x * kBTile - radius;
    const int ty0 = blockIdx.y * kBTile - radius;
    // Everything the workgroup stages -- its tile with the apron, the spatial kernel, the head of the similarity table -- is requested
    // in batches and waited for once per batch.  (As plain loops these were 4 + 1 + 8 dependent memory round trips in front of the
    // first tap: a fifth of the kernel.  Pixels outside the image: the load goes to a clamped address, the value is replaced by 0.)
    {
        constexpr int kAhead = 4;
        for (int i0 = threadIdx.x; i0 < span * span; i0 += 256 * kAhead) {
            unsigned v_[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; u++) {
                const int i = min(i0 + u * 256, span * span - 1);
                const int ly = i / span, lx = i - ly * span;
                const int gx = min(max(tx0 + lx, 0), width - 1), gy = min(max(ty0 + ly, 0), height - 1);
                v_[u] = in[(size_t)gy * width + gx];
            }
#pragma unroll
            for (int u = 0; u < kAhead; u++) {
                const int i = i0 + u * 256;
                const int ly = i / span, lx = i - ly * span;
                const int gx = tx0 + lx, gy = ty0 + ly;
                const unsigned v = (gx >= 0 && gx < width && gy >= 0 && gy < height) ? v_[u] : 0u;
                if (i < span * span) {
                    tile4[i] = v * 4u;
                    if (!STAGED) tile_d[i] = (double)(int)v;
                }
            }
        }
    }
    {
        const int n_sim = !STAGED ? 0 : (sizeof(PIX) == 1 ? 256 : kSimLds);
        constexpr int kAhead = 16;
        float k_ = 0.0f;
        if ((int)threadIdx.x < n * n) k_ = kernel[threadIdx.x];                  // (n * n <= 256 for the staged radii; the loop below takes the rest)
        for (int i0 = threadIdx.x; i0 < n_sim; i0 += 256 * kAhead) {
            float s_[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; u++) s_[u] = similarity[min(i0 + u * 256, n_sim - 1)];
#pragma unroll
            for (int u = 0; u < kAhead; u++)
                if (i0 + u * 256 < n_sim) sim_lds[i0 + u * 256] = s_[u];
        }
        if ((int)threadIdx.x < n * n) kern_lds[threadIdx.x] = k_;
        for (int i = threadIdx.x + 256; i < n * n; i += 256) kern_lds[i] = kernel[i];
    }
    __syncthreads();

    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int x = blockIdx.x * kBTile + lx;
    const int y = blockIdx.y * kBTile + ly;
    const uint32_t *t0 = tile4 + ly * span + lx;      // tap (0, 0) of this lane's window
    const double *d0 = tile_d + ly * span + lx;
    const unsigned current4 = t0[radius * span + radius];
    float total_weight = 0;
    float sum = 0;
    // A wave is a 16 x 4 strip of pixels.  Waves whose every tap lies inside the image (all but the frame's rim): the
    // reference's running kernel index is the plain (column, row) of the tap, the same for every lane.
    // Rim: the reference advances its kernel index only for in-image taps (Q12), so a pixel with in-image tap columns
    // [i0, i1] and rows [j0, j1] uses kernel[(i - i0) * (j1 - j0 + 1) + (j - j0)] for tap (i, j) -- a per-lane index.  Every
    // lane walks the whole window; a tap outside the image (or of a lane outside it) gets weight 0, which leaves both
    // accumulators exactly as they are (sum + 0 * 0 and total + 0 are exact).
    const int wy0 = (blockIdx.y * kBTile + (ly & ~3)) - radius;      // first tap row of the wave's first pixel row
    const bool interior = tx0 >= 0 && tx0 + span <= width && wy0 >= 0 && wy0 + 4 + 2 * radius <= height;
    const bool inside = x < width && y < height;
    const int i0 = max(0, radius - x), i1 = inside ? min(n - 1, width - 1 - x + radius) : -1;
    const int j0 = max(0, radius - y), j1 = inside ? min(n - 1, height - 1 - y + radius) : -1;
    const int nyv = j1 - j0 + 1;

    if (STAGED) {
        constexpr int N = 2 * (R > 0 ? R : 1) + 1;
        constexpr int SPAN = kBTile + 2 * (R > 0 ? R : 1);
        constexpr unsigned kStagedBytes = (sizeof(PIX) == 1 ? 256u : (unsigned)kSimLds) * 4u;
        // A column is handled as two halves (rows 0..7 and 8..14 for N = 15): half a column of look-ups in flight is enough to
        // cover the chain's other half, and a whole column's worth of registers would cost the fifth wave per SIMD that
        // the 4 800 waves of a 640 x 480 frame need to be resident together.
        constexpr int NA = (N + 1) / 2, NB = N - NA;
        float w_a[NA], w_b[NB > 0 ? NB : 1];
        unsigned v_a[NA], v_b[NB > 0 ? NB : 1];   // 4 * intensity of the taps (the chain widens it itself: one LDS read per pixel and tap less)
        auto fetch = [&](const int i, auto first_row, auto count, float *w, unsigned *v4, const auto rim) {
            constexpr int J0 = decltype(first_row)::value, CNT = decltype(count)::value;
            unsigned delta4[CNT];
#pragma unroll
            for (int j = 0; j < CNT; j++) {
                v4[j] = t0[(J0 + j) * SPAN + i];
                delta4[j] = sad_u32(v4[j], current4);     // 4 * |conv - current|
            }
#pragma unroll
            for (int j = 0; j < CNT; j++)
                w[j] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(sim_lds) + delta4[j]);   // (no mask: see below)
            if (sizeof(PIX) != 1) {
                // differences beyond the staged head (kSimLds mm and more): one test per half column; the entries come through
                // a load the compiler cannot fold with the LDS read into a generic-address load.  (The LDS read above is not
                // masked into the head: an offset past it lands in the tile behind the table or past the workgroup's allocation,
                // where the hardware returns 0 for a read -- either way a value this test replaces.)
                unsigned big = delta4[0];
#pragma unroll
                for (int j = 1; j < CNT; j++) big = max(big, delta4[j]);
                if (__builtin_expect(big >= kStagedBytes, 0)) {
#pragma unroll
                    for (int j = 0; j < CNT; j++)
                        if (delta4[j] >= kStagedBytes) {
                            const char *p = reinterpret_cast<const char *>(similarity) + delta4[j];
                            asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(w[j]) : "v"(p) : "memory");
                        }
                }
            }
            if (!decltype(rim)::value) {
                // (the tap's spatial weight is the same for every lane: a scalar load from the table in memory, not an LDS read)
#pragma unroll
                for (int j = 0; j < CNT; j++) w[j] = kernel[i * N + J0 + j] * w[j];     // the float product the reference widens (:99)
            } else {
                const bool col_ok = i >= i0 && i <= i1;
                const int base = (i - i0) * nyv - j0;
#pragma unroll
                for (int j = 0; j < CNT; j++) {
                    const bool ok = col_ok && J0 + j >= j0 && J0 + j <= j1;
                    // (selects, not branches: the table read is unconditional at a safe index, the weight is masked)
                    const float k = kern_lds[(base + J0 + j) & (ok ? 255 : 0)];
                    w[j] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, k * w[j]) & (ok ? 0xffffffffu : 0u));
                }
            }
        };
        auto chain = [&](const int i, auto first_row, auto count, const float *w, const unsigned *v4) {
            constexpr int CNT = decltype(count)::value;
            (void)i;
#pragma unroll
            for (int j = 0; j < CNT; j++) bilateral_tap(w[j], (double)v4[j], sum, total_weight);   // (4 * intensity: `sum` runs as 4 * the reference's, see below)
        };
        using Lo = std::integral_constant<int, 0>;
        using Hi = std::integral_constant<int, NA>;
        using CntA = std::integral_constant<int, NA>;
        using CntB = std::integral_constant<int, NB>;
        auto run = [&](const auto rim) {
            fetch(0, Lo{}, CntA{}, w_a, v_a, rim);
#pragma unroll 1
            for (int i = 0; i + 1 < N; i++) {              // conv_x: outer loop of the reference
                fetch(i, Hi{}, CntB{}, w_b, v_b, rim);
                chain(i, Lo{}, CntA{}, w_a, v_a);
                fetch(i + 1, Lo{}, CntA{}, w_a, v_a, rim);
                chain(i, Hi{}, CntB{}, w_b, v_b);
            }
            fetch(N - 1, Hi{}, CntB{}, w_b, v_b, rim);     // the last column, nothing left to prefetch
            chain(N - 1, Lo{}, CntA{}, w_a, v_a);
            chain(N - 1, Hi{}, CntB{}, w_b, v_b);
        };
        if (interior) {
            run(std::false_type{});
        } else {
            __builtin_amdgcn_s_setprio(3);                 // the few rim waves do more per tap: let them run ahead of their SIMD's other waves
            run(std::true_type{});
        }
        // The chain above fed the tile's words, 4 * intensity, into the products (a shift per tap less): every partial sum was 4 times
        // the reference's, exactly -- scaling by a power of two commutes with both roundings of a tap as long as no partial sum is a
        // denormal float, and a nonzero partial sum is at least the smallest nonzero weight (terms are >= 0, intensities integers),
        // which launch_bilateral has checked to be >= 2^-120 before it picked this kernel (tsdf_bilateral::scale_exact).
        sum *= 0.25f;
    } else {
#pragma unroll 1
        for (int i = 0; i < n; i++) {                      // conv_x: outer loop of the reference
            const bool col_ok = i >= i0 && i <= i1;
            const int base = (i - i0) * nyv - j0;
#pragma unroll 1
            for (int j = 0; j < n; j++) {                  // conv_y
                const int o = j * span + i;
                const bool ok = col_ok && j >= j0 && j <= j1;
                const unsigned delta = sad_u32(t0[o], current4) >> 2;     // |conv - current|
                const float k = kern_lds[ok ? base + j : 0];
                const float weight = ok ? k * similarity[delta] : 0.0f;  // the float product the reference widens (:99)
                bilateral_tap(weight, d0[o], sum, total_weight);
            }
        }
    }
    const PIX filtered = inside ? (PIX)(int)floorf(sum / total_weight) : (PIX)0;
    if (inside) out[(size_t)y * width + x] = filtered;
    if (tile_max) {   // (uniform)
        unsigned m = (unsigned)filtered;
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_down((int)m, o));
        if ((threadIdx.x & 63) == 0 && m) atomicMax(&wg_max, m);
        __syncthreads();
        if (threadIdx.x == 0) tile_max[blockIdx.y * gridDim.x + blockIdx.x] = (uint16_t)wg_max;
    }
}

template <typename PIX>
static int launch_bilateral(const tsdf_bilateral *f, const PIX *in, PIX *out, int width, int height, hipStream_t s,
                            uint16_t *tile_max = nullptr) {
    dim3 grid((width + kBTile - 1) / kBTile, (height + kBTile - 1) / kBTile);
    const int span = kBTile + 2 * f->radius, n = 2 * f->radius + 1;
    // the depth filter of the pipeline: sigma_space 4.5 -> 15 x 15 taps (other radii, or weights so small that the staged kernel's
    // scaled sum could meet denormals: the plain loops)
    const bool staged = f->radius == 7 && f->scale_exact;
    size_t smem = (size_t)span * span * ((staged ? 0 : sizeof(double)) + sizeof(uint32_t)) + (size_t)n * n * sizeof(float) +
                  (staged ? kSimLds * sizeof(float) : 0);
    if (staged)
        hipLaunchKernelGGL((bilateral_kernel<PIX, 7>), grid, dim3(256), smem, s, in, out, width, height, f->radius, f->kernel_dev,
                           f->similarity_dev, tile_max);
    else
        hipLaunchKernelGGL((bilateral_kernel<PIX, 0>), grid, dim3(256), smem, s, in, out, width, height, f->radius, f->kernel_dev,
                           f->similarity_dev, tile_max);
    TSDF_HIP(hipGetLastError(), "bilateral filter kernel failed");
    return TSDF_OK;
}

template <typename PIX>
static int filter_host(const tsdf_bilateral *cf, PIX *host_image, int width, int height) {
    TSDF_REQUIRE(cf && host_image && width > 0 && height > 0, "tsdf_bilateral_filter: bad argument");
    tsdf_bilateral *f = const_cast<tsdf_bilateral *>(cf);
    size_t bytes = (size_t)width * height * sizeof(PIX);
    if (f->img_cap < bytes) {
        if (f->img_in) (void)hipFree(f->img_in);
        if (f->img_out) (void)hipFree(f->img_out);
        f->img_in = f->img_out = nullptr;
        f->img_cap = 0;
        TSDF_HIP(hipMalloc(&f->img_in, bytes), "bilateral image alloc");
        TSDF_HIP(hipMalloc(&f->img_out, bytes), "bilateral image alloc");
        f->img_cap = bytes;
    }
    TSDF_HIP(hipMemcpy(f->img_in, host_image, bytes, hipMemcpyHostToDevice), "bilateral H2D");
    int rc = launch_bilateral<PIX>(f, (const PIX *)f->img_in, (PIX *)f->img_out, width, height, nullptr);
    if (rc != TSDF_OK) return rc;
    // in place, like the reference's memcpy over the input (src/BilateralFilter.cpp:116)
    TSDF_HIP(hipMemcpy(host_image, f->img_out, bytes, hipMemcpyDeviceToHost), "bilateral D2H");
    return TSDF_OK;
}

}  // namespace tsdf

using namespace tsdf;

extern "C" {

int tsdf_bilateral_create(float sigma_colour, float sigma_space, tsdf_bilateral **out) {
    TSDF_REQUIRE(out, "tsdf_bilateral_create: null out pointer");
    *out = nullptr;
    TSDF_REQUIRE(sigma_colour > 0 && sigma_space > 0, "tsdf_bilateral_create: sigmas must be positive");
    // src/BilateralFilter.cpp:17-23
    int kernel_radius = (int)std::ceil(sigma_space * 1.5f);
    TSDF_REQUIRE(kernel_radius <= kMaxRadius, "tsdf_bilateral_create: kernel radius %d exceeds %d", kernel_radius, kMaxRadius);
    float inv_sigma_colour_squared = 1.0f / (sigma_colour * sigma_colour);
    float inv_sigma_space_squared = 1.0f / (sigma_space * sigma_space);
    int kernel_size = kernel_radius * 2 + 1;
    int center = (kernel_size - 1) / 2;
    std::vector<float> kernel((size_t)kernel_size * kernel_size);
    int idx = 0;
    for (int x = -center; x < kernel_size - center; x++) {
        for (int y = -center; y < kernel_size - center; y++) {
            float dist_squared = (float)(x * x + y * y);
            kernel[idx] = std::exp(-dist_squared * inv_sigma_space_squared);  // :32
            idx++;
        }
    }
    std::vector<float> similarity(65536);
    for (int i = 0; i < 65536; i++) similarity[i] = std::exp(-i * inv_sigma_colour_squared);  // :40

    tsdf_bilateral *f = new (std::nothrow) tsdf_bilateral();
    if (!f) {
        set_error("out of host memory");
        return TSDF_ERR_NOMEM;
    }
    memset(f, 0, sizeof(*f));
    f->sigma_colour = sigma_colour;
    f->sigma_space = sigma_space;
    f->radius = kernel_radius;
    {
        // smallest nonzero weight a tap can have: the float product of the smallest nonzero entries of the two tables
        float min_k = 0.0f, min_s = 0.0f;
        for (float k : kernel) if (k > 0.0f && (min_k == 0.0f || k < min_k)) min_k = k;
        for (float x : similarity) if (x > 0.0f && (min_s == 0.0f || x < min_s)) min_s = x;
        f->scale_exact = (min_k * min_s >= 0x1p-120f) ? 1 : 0;
    }
    hipError_t e = hipGetDevice(&f->device);
    if (e == hipSuccess) e = hipMalloc((void **)&f->kernel_dev, kernel.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&f->similarity_dev, similarity.size() * sizeof(float));

    if (e == hipSuccess) e = hipMemcpy(f->kernel_dev, kernel.data(), kernel.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(f->similarity_dev, similarity.data(), similarity.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        int rc = hip_fail(e, "bilateral filter tables");
        tsdf_bilateral_destroy(f);
        return rc;
    }
    *out = f;
    return TSDF_OK;
}

int tsdf_bilateral_destroy(tsdf_bilateral *f) {
    if (!f) return TSDF_OK;
    if (f->kernel_dev) (void)hipFree(f->kernel_dev);
    if (f->similarity_dev) (void)hipFree(f->similarity_dev);
    if (f->img_in) (void)hipFree(f->img_in);
    if (f->img_out) (void)hipFree(f->img_out);
    delete f;
    return TSDF_OK;
}

int tsdf_bilateral_filter_u8(const tsdf_bilateral *f, uint8_t *host_image, int width, int height) {
    return filter_host<uint8_t>(f, host_image, width, height);
}

int tsdf_bilateral_filter_u16(const tsdf_bilateral *f, uint16_t *host_image, int width, int height) {
    return filter_host<uint16_t>(f, host_image, width, height);
}

int tsdf_bilateral_filter_u8_device(const tsdf_bilateral *f, const uint8_t *in, uint8_t *out, int width,
                                    int height, void *hip_stream) {
    TSDF_REQUIRE(f && in && out && in != out && width > 0 && height > 0, "tsdf_bilateral_filter: bad argument");
    return launch_bilateral<uint8_t>(f, in, out, width, height, (hipStream_t)hip_stream);
}

int tsdf_bilateral_filter_u16_device(const tsdf_bilateral *f, const uint16_t *in, uint16_t *out, int width,
                                     int height, void *hip_stream) {
    TSDF_REQUIRE(f && in && out && in != out && width > 0 && height > 0, "tsdf_bilateral_filter: bad argument");
    return launch_bilateral<uint16_t>(f, in, out, width, height, (hipStream_t)hip_stream);
}

int tsdf_bilateral_filter_u16_device_tiles(const tsdf_bilateral *f, const uint16_t *in, uint16_t *out, int width,
                                           int height, uint16_t *tile_max, void *hip_stream) {
    TSDF_REQUIRE(f && in && out && in != out && tile_max && width > 0 && height > 0, "tsdf_bilateral_filter: bad argument");
    static_assert(kBTile == TSDF_DEPTH_TILE, "the filter's workgroup tile is the depth tile of integrate's culling");
    return launch_bilateral<uint16_t>(f, in, out, width, height, (hipStream_t)hip_stream, tile_max);
}

}  // extern "C"
