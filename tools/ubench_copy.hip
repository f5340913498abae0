// Streaming-copy micro-benchmark (diagnostics): what the access granularity costs on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_copy tools/ubench_copy.hip && build/ubench_copy
// copy of 1 GiB (read + write = 2 GiB of traffic) with 4, 8 and 16 bytes per lane and load; contiguous, and in the shape
// integrate_kernel walks the volume: workgroups of 4 waves, each wave one 256-byte row segment, the 4 rows 2 KiB apart,
// plane after plane 1 MiB apart (a 64 x 4 x 32 brick of a 512^3 grid), read-modify-write of two arrays.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <typename T>
__global__ __launch_bounds__(256) void copy_kernel(const T *__restrict__ src, T *__restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// brick walk: X = Y = Z = 512; brick = 64 x 4 x 32; one workgroup per brick; lane <-> x.  BATCH planes in flight.
template <int BATCH, bool RMW>
__global__ __launch_bounds__(256) void brick_kernel(float *__restrict__ d, float *__restrict__ w, unsigned nbricks) {
    const unsigned b = blockIdx.x;
    if (b >= nbricks) return;
    const unsigned bx = b % 8, by = (b / 8) % 128, bz = b / (8 * 128);
    const size_t plane = 512 * 512;
    size_t idx = (size_t)bz * 32 * plane + (size_t)(by * 4 + threadIdx.y) * 512 + bx * 64 + threadIdx.x;
#pragma unroll 1
    for (int z = 0; z < 32; z += BATCH) {
        float pd[BATCH], pw[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; j++) { pd[j] = d[idx + (z + j) * plane]; pw[j] = w[idx + (z + j) * plane]; }
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
            const float nw = pw[j] + 1.0f;
            d[idx + (z + j) * plane] = RMW ? (pd[j] * pw[j] + 3.0f) / nw : pd[j] + 1.0f;
            w[idx + (z + j) * plane] = nw;
        }
    }
}

// the same bricks with 16 bytes per lane.  WIDE = false: a wave covers one plane of the brick (16 lanes x 4 voxels = 64 x, 4 rows of
// 256 B), the 4 waves of a workgroup take 8 planes each.  WIDE = true: a wave covers 256 x of one row (1 KiB contiguous), a
// workgroup 256 x 4 x 32 voxels (a quarter of the workgroups).
template <int BATCH, bool WIDE>
__global__ __launch_bounds__(256) void brick4_kernel(float4 *__restrict__ d, float4 *__restrict__ w, unsigned nbricks) {
    const unsigned b = blockIdx.x;
    if (b >= nbricks) return;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const size_t plane4 = 512 * 512 / 4;
    size_t idx;
    int z0, nz;
    if (!WIDE) {
        const unsigned bx = b % 8, by = (b / 8) % 128, bz = b / (8 * 128);
        idx = (size_t)bz * 32 * plane4 + (size_t)(by * 4 + (lane >> 4)) * 128 + bx * 16 + (lane & 15u);
        z0 = wave * 8; nz = 8;
    } else {
        const unsigned bx = b % 2, by = (b / 2) % 128, bz = b / (2 * 128);
        idx = (size_t)bz * 32 * plane4 + (size_t)(by * 4 + wave) * 128 + bx * 64 + lane;
        z0 = 0; nz = 32;
    }
#pragma unroll 1
    for (int z = z0; z < z0 + nz; z += BATCH) {
        float4 pd[BATCH], pw[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; j++) { pd[j] = d[idx + (z + j) * plane4]; pw[j] = w[idx + (z + j) * plane4]; }
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
            float4 nd = pd[j], nw = pw[j];
            nd.x += 1.0f; nd.y += 1.0f; nd.z += 1.0f; nd.w += 1.0f; nw.x += 1.0f; nw.y += 1.0f; nw.z += 1.0f; nw.w += 1.0f;
            d[idx + (z + j) * plane4] = nd;
            w[idx + (z + j) * plane4] = nw;
        }
    }
}

template <typename F>
static double timed(F launch, double bytes) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double best = 0;
    for (int r = 0; r < 6; r++) {
        (void)hipEventRecord(e0, 0);
        launch();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r > 0) best = best > bytes / (ms * 1e-3) / 1e9 ? best : bytes / (ms * 1e-3) / 1e9;
    }
    return best;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    float *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    (void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes);
    for (int wgs_per_cu : {8, 32, 0}) {
        const char *shape = wgs_per_cu ? "grid-stride" : "one element per thread";
        auto grid = [&](size_t n) { return (unsigned)(wgs_per_cu ? (size_t)256 * wgs_per_cu : (n + 255) / 256); };
        printf("%-24s %3d WG/CU: 4 B/lane %7.1f   8 B/lane %7.1f   16 B/lane %7.1f GB/s\n", shape, wgs_per_cu,
               timed([&] { hipLaunchKernelGGL(copy_kernel<float>, dim3(grid(bytes / 4)), dim3(256), 0, 0, a, b, bytes / 4); }, 2.0 * bytes),
               timed([&] { hipLaunchKernelGGL(copy_kernel<float2>, dim3(grid(bytes / 8)), dim3(256), 0, 0, (float2 *)a, (float2 *)b, bytes / 8); }, 2.0 * bytes),
               timed([&] { hipLaunchKernelGGL(copy_kernel<float4>, dim3(grid(bytes / 16)), dim3(256), 0, 0, (float4 *)a, (float4 *)b, bytes / 16); }, 2.0 * bytes));
    }
    // 512^3 x 4 B = 512 MiB per array: a = distances, b = weights (first halves of the buffers)
    const unsigned nbricks = 8 * 128 * 16;
    const double bb = 4.0 * 512.0 * 512 * 512 * 4;   // read + write of both arrays
    printf("brick walk, all bricks, in place on two arrays (read + write): batch 1 %7.1f  batch 4 %7.1f  batch 8 %7.1f GB/s;  with the blend arithmetic, batch 4: %7.1f GB/s\n",
           timed([&] { hipLaunchKernelGGL((brick_kernel<1, false>), dim3(nbricks), dim3(64, 4), 0, 0, a, b, nbricks); }, bb),
           timed([&] { hipLaunchKernelGGL((brick_kernel<4, false>), dim3(nbricks), dim3(64, 4), 0, 0, a, b, nbricks); }, bb),
           timed([&] { hipLaunchKernelGGL((brick_kernel<8, false>), dim3(nbricks), dim3(64, 4), 0, 0, a, b, nbricks); }, bb),
           timed([&] { hipLaunchKernelGGL((brick_kernel<4, true>), dim3(nbricks), dim3(64, 4), 0, 0, a, b, nbricks); }, bb));
    printf("brick walk with 16 B per lane: wave = one 64 x 4 plane of the brick, batch 1 %7.1f  batch 2 %7.1f  batch 4 %7.1f GB/s;   wave = 256 x of a row (1 KiB), batch 2 %7.1f  batch 4 %7.1f GB/s\n",
           timed([&] { hipLaunchKernelGGL((brick4_kernel<1, false>), dim3(nbricks), dim3(256), 0, 0, (float4 *)a, (float4 *)b, nbricks); }, bb),
           timed([&] { hipLaunchKernelGGL((brick4_kernel<2, false>), dim3(nbricks), dim3(256), 0, 0, (float4 *)a, (float4 *)b, nbricks); }, bb),
           timed([&] { hipLaunchKernelGGL((brick4_kernel<4, false>), dim3(nbricks), dim3(256), 0, 0, (float4 *)a, (float4 *)b, nbricks); }, bb),
           timed([&] { hipLaunchKernelGGL((brick4_kernel<2, true>), dim3(nbricks / 4), dim3(256), 0, 0, (float4 *)a, (float4 *)b, nbricks / 4); }, bb),
           timed([&] { hipLaunchKernelGGL((brick4_kernel<4, true>), dim3(nbricks / 4), dim3(256), 0, 0, (float4 *)a, (float4 *)b, nbricks / 4); }, bb));
    return 0;
}
