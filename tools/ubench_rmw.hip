// Why does an in-place read-modify-write of two fp32 arrays stop near 5 TB/s on gfx950 when a float4 copy reaches 6.2?
// (diagnostics; round 3)  Every kernel below moves the same bytes per array element (one read + one write, or
// one of the two) over 2 x 512 MiB; they differ in WHERE the write goes (same address / another array), in the order of the walk
// (linear / integrate's bricks) and in the width per lane.  Distinct kernel names, so that rocprofv3 --pmc separates them:
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_rmw tools/ubench_rmw.hip && build/ubench_rmw
//   rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ_DRAM_CREDIT_STALL -- build/ubench_rmw
//   rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL -- build/ubench_rmw
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float vf2 __attribute__((ext_vector_type(2)));
typedef float vf4 __attribute__((ext_vector_type(4)));
constexpr size_t kN = (size_t)512 * 512 * 512;   // floats per array

// ---- linear walks: thread t of the grid takes float4 t, t + stride, ... ; DEPTH elements in flight per thread
template <int DEPTH>
__global__ __launch_bounds__(256) void k_read_only(const vf4 *__restrict__ a, const vf4 *__restrict__ b, float *sink) {
    const size_t n = kN / 4, stride = (size_t)gridDim.x * 256;
    vf4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (DEPTH - 1) * stride < n; i += DEPTH * stride) {
#pragma unroll
        for (int j = 0; j < DEPTH; j++) acc += a[i + j * stride] + b[i + j * stride];
    }
    if (acc.x + acc.y + acc.z + acc.w == -1.0f) *sink = acc.x;
}
template <int DEPTH>
__global__ __launch_bounds__(256) void k_write_only(vf4 *__restrict__ a, vf4 *__restrict__ b) {
    const size_t n = kN / 4, stride = (size_t)gridDim.x * 256;
    const vf4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (DEPTH - 1) * stride < n; i += DEPTH * stride) {
#pragma unroll
        for (int j = 0; j < DEPTH; j++) { a[i + j * stride] = v; b[i + j * stride] = v; }
    }
}
// out of place: (a, b) -> (c, d)
template <int DEPTH>
__global__ __launch_bounds__(256) void k_copy_two(const vf4 *__restrict__ a, const vf4 *__restrict__ b, vf4 *__restrict__ c, vf4 *__restrict__ d) {
    const size_t n = kN / 4, stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (DEPTH - 1) * stride < n; i += DEPTH * stride) {
        vf4 x[DEPTH], y[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH; j++) { x[j] = a[i + j * stride]; y[j] = b[i + j * stride]; }
#pragma unroll
        for (int j = 0; j < DEPTH; j++) { c[i + j * stride] = x[j] + 1.0f; d[i + j * stride] = y[j] + 1.0f; }
    }
}
// in place: (a, b) -> (a, b)
template <int DEPTH>
__global__ __launch_bounds__(256) void k_inplace_linear(vf4 *__restrict__ a, vf4 *__restrict__ b) {
    const size_t n = kN / 4, stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (DEPTH - 1) * stride < n; i += DEPTH * stride) {
        vf4 x[DEPTH], y[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH; j++) { x[j] = a[i + j * stride]; y[j] = b[i + j * stride]; }
#pragma unroll
        for (int j = 0; j < DEPTH; j++) { a[i + j * stride] = x[j] + 1.0f; b[i + j * stride] = y[j] + 1.0f; }
    }
}
// in place, a workgroup owns one contiguous chunk of each array and walks it front to back (DRAM pages stay with one workgroup)
template <int DEPTH>
__global__ __launch_bounds__(256) void k_inplace_chunked(vf4 *__restrict__ a, vf4 *__restrict__ b) {
    const size_t n = kN / 4, per_wg = n / gridDim.x;
    const size_t base = (size_t)blockIdx.x * per_wg;
    for (size_t i = threadIdx.x; i + (DEPTH - 1) * 256 < per_wg; i += DEPTH * 256) {
        vf4 x[DEPTH], y[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH; j++) { x[j] = a[base + i + j * 256]; y[j] = b[base + i + j * 256]; }
#pragma unroll
        for (int j = 0; j < DEPTH; j++) { a[base + i + j * 256] = x[j] + 1.0f; b[base + i + j * 256] = y[j] + 1.0f; }
    }
}

// ---- integrate's walk: a workgroup owns a brick of 64 * LF (x) x 4 (y) x 32 (z) voxels, a wave per row, 4 planes in flight;
// OUT: the result goes to other arrays (c, d) instead of back in place
template <typename T> struct W { static constexpr int LF = sizeof(T) / 4; };
template <typename T, bool OUT>
__device__ inline void brick_walk(const T *a, const T *b, T *c, T *d) {
    constexpr unsigned LF = W<T>::LF, NBX = 512 / (64 * LF), ROWS = 128 * 16;
    const unsigned blk = blockIdx.x, bx = blk / ROWS, r = blk % ROWS, by = r % 128, bz = r / 128;   // column by column, as the cull kernel lists them
    (void)NBX;
    const size_t row = 512 / LF, plane = row * 512;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t idx = (size_t)(bz * 32) * plane + (size_t)(by * 4 + wave) * row + bx * 64 + lane;
#pragma unroll 1
    for (unsigned z = 0; z < 32; z += 4) {
        T x[4], y[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { x[j] = a[idx + (z + j) * plane]; y[j] = b[idx + (z + j) * plane]; }
#pragma unroll
        for (int j = 0; j < 4; j++) { (OUT ? c : const_cast<T *>(a))[idx + (z + j) * plane] = x[j] + 1.0f; (OUT ? d : const_cast<T *>(b))[idx + (z + j) * plane] = y[j] + 1.0f; }
    }
}
__global__ __launch_bounds__(256) void k_brick_inplace_4B(float *a, float *b) { brick_walk<float, false>(a, b, nullptr, nullptr); }
__global__ __launch_bounds__(256) void k_brick_outofplace_4B(const float *a, const float *b, float *c, float *d) { brick_walk<float, true>(a, b, c, d); }
__global__ __launch_bounds__(256) void k_brick_inplace_8B(vf2 *a, vf2 *b) { brick_walk<vf2, false>(a, b, nullptr, nullptr); }
__global__ __launch_bounds__(256) void k_brick_outofplace_8B(const vf2 *a, const vf2 *b, vf2 *c, vf2 *d) { brick_walk<vf2, true>(a, b, c, d); }
__global__ __launch_bounds__(256) void k_brick_inplace_16B(vf4 *a, vf4 *b) { brick_walk<vf4, false>(a, b, nullptr, nullptr); }
// the brick walk with the stores of a batch held back until the NEXT batch's loads are out (integrate_kernel's software pipeline)
__global__ __launch_bounds__(256) void k_brick_inplace_4B_pipelined(float *a, float *b) {
    const unsigned blk = blockIdx.x, ROWS = 128 * 16, bx = blk / ROWS, r = blk % ROWS, by = r % 128, bz = r / 128;
    const size_t plane = (size_t)512 * 512;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t idx = (size_t)(bz * 32) * plane + (size_t)(by * 4 + wave) * 512 + bx * 64 + lane;
    float x0[4], y0[4], x1[4], y1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { x0[j] = a[idx + j * plane]; y0[j] = b[idx + j * plane]; }
#pragma unroll
    for (unsigned z = 0; z < 32; z += 8) {
#pragma unroll
        for (int j = 0; j < 4; j++) { x1[j] = a[idx + (z + 4 + j) * plane]; y1[j] = b[idx + (z + 4 + j) * plane]; }
#pragma unroll
        for (int j = 0; j < 4; j++) { a[idx + (z + j) * plane] = x0[j] + 1.0f; b[idx + (z + j) * plane] = y0[j] + 1.0f; }
        if (z + 8 < 32) {
#pragma unroll
            for (int j = 0; j < 4; j++) { x0[j] = a[idx + (z + 8 + j) * plane]; y0[j] = b[idx + (z + 8 + j) * plane]; }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { a[idx + (z + 4 + j) * plane] = x1[j] + 1.0f; b[idx + (z + 4 + j) * plane] = y1[j] + 1.0f; }
    }
}

template <typename F>
static void timed(const char *what, double bytes, F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    double best = 0;
    for (int r = 0; r < 5; r++) {
        (void)hipEventRecord(e0, 0);
        launch();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double g = bytes / (ms * 1e-3) / 1e9;
        if (r > 0 && g > best) best = g;
    }
    printf("%-78s %7.1f GB/s\n", what, best);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

int main() {
    const size_t bytes = kN * 4;
    void *a, *b, *c, *d;
    float *sink;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&c, bytes) != hipSuccess ||
        hipMalloc(&d, bytes) != hipSuccess || hipMalloc((void **)&sink, 4) != hipSuccess)
        return 1;
    (void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes); (void)hipMemset(c, 0, bytes); (void)hipMemset(d, 0, bytes);
    const double rw = 4.0 * bytes, one = 2.0 * bytes;   // read + write of two arrays; one direction only
    const dim3 blk(256);
    for (unsigned grid : {256u * 8, 256u * 32}) {
        printf("-- linear walks, %u workgroups\n", grid);
        timed("read only (two arrays)", one, [&] { hipLaunchKernelGGL(k_read_only<4>, dim3(grid), blk, 0, 0, (const vf4 *)a, (const vf4 *)b, sink); });
        timed("write only (two arrays)", one, [&] { hipLaunchKernelGGL(k_write_only<4>, dim3(grid), blk, 0, 0, (vf4 *)a, (vf4 *)b); });
        timed("out of place: (a, b) -> (c, d), 4 float4 in flight per array", rw, [&] { hipLaunchKernelGGL(k_copy_two<4>, dim3(grid), blk, 0, 0, (const vf4 *)a, (const vf4 *)b, (vf4 *)c, (vf4 *)d); });
        timed("in place:     (a, b) -> (a, b), 4 float4 in flight per array", rw, [&] { hipLaunchKernelGGL(k_inplace_linear<4>, dim3(grid), blk, 0, 0, (vf4 *)a, (vf4 *)b); });
        timed("in place:     (a, b) -> (a, b), 1 float4 in flight per array", rw, [&] { hipLaunchKernelGGL(k_inplace_linear<1>, dim3(grid), blk, 0, 0, (vf4 *)a, (vf4 *)b); });
        timed("in place:     (a, b) -> (a, b), 8 float4 in flight per array", rw, [&] { hipLaunchKernelGGL(k_inplace_linear<8>, dim3(grid), blk, 0, 0, (vf4 *)a, (vf4 *)b); });
        timed("in place, one contiguous chunk per workgroup, 4 in flight", rw, [&] { hipLaunchKernelGGL(k_inplace_chunked<4>, dim3(grid), blk, 0, 0, (vf4 *)a, (vf4 *)b); });
    }
    printf("-- integrate's bricks (64 * LF x 4 x 32 voxels per workgroup, 4 planes in flight, column by column)\n");
    timed("bricks, 4 B per lane, in place", rw, [&] { hipLaunchKernelGGL(k_brick_inplace_4B, dim3(8 * 128 * 16), blk, 0, 0, (float *)a, (float *)b); });
    timed("bricks, 4 B per lane, out of place", rw, [&] { hipLaunchKernelGGL(k_brick_outofplace_4B, dim3(8 * 128 * 16), blk, 0, 0, (const float *)a, (const float *)b, (float *)c, (float *)d); });
    timed("bricks, 4 B per lane, in place, stores behind the next batch's loads", rw, [&] { hipLaunchKernelGGL(k_brick_inplace_4B_pipelined, dim3(8 * 128 * 16), blk, 0, 0, (float *)a, (float *)b); });
    timed("bricks, 8 B per lane, in place", rw, [&] { hipLaunchKernelGGL(k_brick_inplace_8B, dim3(4 * 128 * 16), blk, 0, 0, (vf2 *)a, (vf2 *)b); });
    timed("bricks, 8 B per lane, out of place", rw, [&] { hipLaunchKernelGGL(k_brick_outofplace_8B, dim3(4 * 128 * 16), blk, 0, 0, (const vf2 *)a, (const vf2 *)b, (vf2 *)c, (vf2 *)d); });
    timed("bricks, 16 B per lane, in place", rw, [&] { hipLaunchKernelGGL(k_brick_inplace_16B, dim3(2 * 128 * 16), blk, 0, 0, (vf4 *)a, (vf4 *)b); });
    return 0;
}
