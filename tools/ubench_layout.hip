// What would another layout of the weights buy integrate's read-modify-write?  (diagnostics; round 4.)  Dense walks of a 512^3 grid
// in integrate_kernel's brick shape (64 x 4 x 32 voxels per workgroup, a wave per row, 4 planes per batch, bricks column by column):
//   A  distance fp32 + weight fp32, two arrays (what is built)                          16 B per voxel moved
//   B  distance fp32 + weight u8, four planes of a lane in one dword (z-packed)          10 B
//   C  distance alone                                                                     8 B
//   D  distance fp32 + weight u16, two planes per dword                                   12 B
//   E  {distance, weight} interleaved, 8 B per lane                                       16 B, half the requests
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_layout tools/ubench_layout.hip && build/ubench_layout
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float vf2 __attribute__((ext_vector_type(2)));
constexpr unsigned kX = 512, kROWS = 128 * 16;
constexpr size_t kPlane = (size_t)kX * kX, kN = kPlane * kX;

__device__ inline size_t brick_index(unsigned &bz) {
    const unsigned blk = blockIdx.x, bx = blk / kROWS, r = blk % kROWS, by = r % 128;
    bz = r / 128;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    return (size_t)(by * 4 + wave) * kX + bx * 64 + lane;   // offset inside a plane
}
__global__ __launch_bounds__(256) void k_A_two_fp32(float *d, float *w) {
    unsigned bz;
    const size_t idx = brick_index(bz) + (size_t)(bz * 32) * kPlane;
#pragma unroll 1
    for (unsigned z = 0; z < 32; z += 4) {
        float x[4], y[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { x[j] = d[idx + (z + j) * kPlane]; y[j] = w[idx + (z + j) * kPlane]; }
#pragma unroll
        for (int j = 0; j < 4; j++) { d[idx + (z + j) * kPlane] = x[j] + 1.0f; w[idx + (z + j) * kPlane] = y[j] + 1.0f; }
    }
}
__global__ __launch_bounds__(256) void k_B_zpacked_u8(float *d, uint32_t *w) {
    unsigned bz;
    const size_t in_plane = brick_index(bz), idx = in_plane + (size_t)(bz * 32) * kPlane;
#pragma unroll 1
    for (unsigned z = 0; z < 32; z += 4) {
        float x[4];
        const size_t wi = in_plane + (size_t)(bz * 8 + z / 4) * kPlane;
#pragma unroll
        for (int j = 0; j < 4; j++) x[j] = d[idx + (z + j) * kPlane];
        const uint32_t y = w[wi];
#pragma unroll
        for (int j = 0; j < 4; j++) d[idx + (z + j) * kPlane] = x[j] + (float)((y >> (8 * j)) & 255u);
        w[wi] = y + 0x01010101u;
    }
}
__global__ __launch_bounds__(256) void k_C_distance_alone(float *d) {
    unsigned bz;
    const size_t idx = brick_index(bz) + (size_t)(bz * 32) * kPlane;
#pragma unroll 1
    for (unsigned z = 0; z < 32; z += 4) {
        float x[4];
#pragma unroll
        for (int j = 0; j < 4; j++) x[j] = d[idx + (z + j) * kPlane];
#pragma unroll
        for (int j = 0; j < 4; j++) d[idx + (z + j) * kPlane] = x[j] + 1.0f;
    }
}
__global__ __launch_bounds__(256) void k_D_zpacked_u16(float *d, uint32_t *w) {
    unsigned bz;
    const size_t in_plane = brick_index(bz), idx = in_plane + (size_t)(bz * 32) * kPlane;
#pragma unroll 1
    for (unsigned z = 0; z < 32; z += 4) {
        float x[4];
        const size_t wi = in_plane + (size_t)(bz * 16 + z / 2) * kPlane;
#pragma unroll
        for (int j = 0; j < 4; j++) x[j] = d[idx + (z + j) * kPlane];
        const uint32_t y0 = w[wi], y1 = w[wi + kPlane];
        d[idx + (z + 0) * kPlane] = x[0] + (float)(y0 & 0xffffu);
        d[idx + (z + 1) * kPlane] = x[1] + (float)(y0 >> 16);
        d[idx + (z + 2) * kPlane] = x[2] + (float)(y1 & 0xffffu);
        d[idx + (z + 3) * kPlane] = x[3] + (float)(y1 >> 16);
        w[wi] = y0 + 0x00010001u;
        w[wi + kPlane] = y1 + 0x00010001u;
    }
}
__global__ __launch_bounds__(256) void k_E_interleaved(vf2 *dw) {
    unsigned bz;
    const size_t idx = brick_index(bz) + (size_t)(bz * 32) * kPlane;
#pragma unroll 1
    for (unsigned z = 0; z < 32; z += 4) {
        vf2 x[4];
#pragma unroll
        for (int j = 0; j < 4; j++) x[j] = dw[idx + (z + j) * kPlane];
#pragma unroll
        for (int j = 0; j < 4; j++) dw[idx + (z + j) * kPlane] = x[j] + 1.0f;
    }
}

template <typename F>
static void timed(const char *what, double bytes, F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    double best = 1e30;
    for (int r = 0; r < 6; r++) {
        (void)hipEventRecord(e0, 0);
        launch();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best) best = ms;
    }
    printf("%-70s %7.4f ms  %7.1f GB/s\n", what, best, bytes / (best * 1e-3) / 1e9);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

int main() {
    void *d, *w, *dw;
    if (hipMalloc(&d, kN * 4) != hipSuccess || hipMalloc(&w, kN * 4) != hipSuccess || hipMalloc(&dw, kN * 8) != hipSuccess) return 1;
    (void)hipMemset(d, 0, kN * 4); (void)hipMemset(w, 0, kN * 4); (void)hipMemset(dw, 0, kN * 8);
    const dim3 grid(8 * kROWS), blk(256);
    for (int rep = 0; rep < 2; rep++) {
        timed("A  distance fp32 + weight fp32 (16 B per voxel)", 16.0 * kN, [&] { hipLaunchKernelGGL(k_A_two_fp32, grid, blk, 0, 0, (float *)d, (float *)w); });
        timed("B  distance fp32 + weight u8 z-packed (10 B)", 10.0 * kN, [&] { hipLaunchKernelGGL(k_B_zpacked_u8, grid, blk, 0, 0, (float *)d, (uint32_t *)w); });
        timed("C  distance alone (8 B)", 8.0 * kN, [&] { hipLaunchKernelGGL(k_C_distance_alone, grid, blk, 0, 0, (float *)d); });
        timed("D  distance fp32 + weight u16 z-packed (12 B)", 12.0 * kN, [&] { hipLaunchKernelGGL(k_D_zpacked_u16, grid, blk, 0, 0, (float *)d, (uint32_t *)w); });
        timed("E  {distance, weight} interleaved, 8 B per lane (16 B)", 16.0 * kN, [&] { hipLaunchKernelGGL(k_E_interleaved, grid, blk, 0, 0, (vf2 *)dw); });
    }
    return 0;
}
