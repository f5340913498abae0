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
#include <new>
#include <vector>

#include "common.hpp"

namespace tsdf {

constexpr int kBTile = 16;
constexpr int kMaxRadius = 24;  // LDS tile (16+2*24)^2 * 10 B = 40 KiB

template <typename PIX>
__global__ __launch_bounds__(256) void bilateral_kernel(const PIX *__restrict__ in, PIX *__restrict__ out,
                                                        int width, int height, int radius,
                                                        const float *__restrict__ kernel,
                                                        const float *__restrict__ similarity) {
    // the tile twice: as doubles (the widened operand of the reference's product, converted once per pixel here instead of
    // once per tap) and as 16-bit integers (for the intensity difference)
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int span = kBTile + 2 * radius;
    double *tile_d = reinterpret_cast<double *>(smem_raw);
    uint16_t *tile = reinterpret_cast<uint16_t *>(tile_d + span * span);
    const int tx0 = blockIdx.x * kBTile - radius;
    const int ty0 = blockIdx.y * kBTile - radius;
    for (int i = threadIdx.x; i < span * span; i += 256) {
        int ly = i / span, lx = i - ly * span;
        int gx = tx0 + lx, gy = ty0 + ly;
        uint16_t v = 0;
        if (gx >= 0 && gx < width && gy >= 0 && gy < height) v = in[(size_t)gy * width + gx];
        tile[i] = v;
        tile_d[i] = (double)(int)v;
    }
    __syncthreads();

    const int x = blockIdx.x * kBTile + (threadIdx.x & 15);
    const int y = blockIdx.y * kBTile + (threadIdx.x >> 4);
    if (x >= width || y >= height) return;

    const int current = tile[(y - ty0) * span + (x - tx0)];
    // in-image tap ranges and the reference's running kernel index (Q12)
    const int fx = max(x - radius, 0), lx_ = min(x + radius, width - 1);
    const int fy = max(y - radius, 0), ly_ = min(y + radius, height - 1);
    const int ny = ly_ - fy + 1;

    float total_weight = 0;
    float sum = 0;
    for (int cx = fx; cx <= lx_; cx++) {
        const uint16_t *col = tile + (cx - tx0);
        const double *col_d = tile_d + (cx - tx0);
        const float *krow = kernel + (cx - fx) * ny;
        for (int cy = fy; cy <= ly_; cy++) {
            int conv = col[(cy - ty0) * span];
            const unsigned delta = __builtin_amdgcn_sad_u16((unsigned)conv, (unsigned)current, 0u);   // |conv - current| in one instruction (both are 0..65535: the high halves are 0)
            const float weight = krow[cy - fy] * similarity[delta];   // the float product the reference widens (:99)
            sum = (float)((double)sum + ((double)weight * col_d[(cy - ty0) * span]));
            // total_weight + weight evaluated in double and narrowed (:102) == the fp32 sum: both operands are floats, so
            // the double sum is exact unless the smaller is below 2^-29 of the larger, and then both roundings return the
            // larger operand (weights are >= 0).  One fp32 add instead of two conversions and a double add.
            total_weight = total_weight + weight;
        }
    }
    out[(size_t)y * width + x] = (PIX)(int)floorf(sum / total_weight);
}

template <typename PIX>
static int launch_bilateral(const tsdf_bilateral *f, const PIX *in, PIX *out, int width, int height, hipStream_t s) {
    dim3 grid((width + kBTile - 1) / kBTile, (height + kBTile - 1) / kBTile);
    int span = kBTile + 2 * f->radius;
    size_t smem = (size_t)span * span * (sizeof(double) + sizeof(uint16_t));
    hipLaunchKernelGGL((bilateral_kernel<PIX>), grid, dim3(256), smem, s, in, out, width, height, f->radius,
                       f->kernel_dev, f->similarity_dev);
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

}  // extern "C"
