// Does the brick shape matter to an in-place update walk when a z plane is 4 MiB (1024^3) instead of 1 MiB (512^3)?  (round 4: integrate_kernel
// at 1024^3 moves its bytes 10-20 % slower than at 512^3 on the same box, and one run on a fresh box was 20 % faster than the rest.)
// A workgroup of 4 waves walks a brick of 64 (x, one wave) x YR rows x ZP planes of two fp32 arrays, 4 (row, plane) items per wave in flight,
// bricks column by column as integrate lists them; a brick of YR x ZP = 4 x 32 touches 32 planes = 32 x 4 MiB of address space per array,
// one of 16 x 8 only 8.   hipcc --offload-arch=gfx950 -O3 -o build/ubench_tlb tools/ubench_tlb.hip && build/ubench_tlb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int YR, int ZP>
__global__ __launch_bounds__(256) void walk(float *__restrict__ d, float *__restrict__ w, unsigned N, unsigned keep_mod) {
    const unsigned NBY = N / YR, NBZ = N / ZP, ROWS = NBY * NBZ;
    const unsigned b = blockIdx.x;
    const unsigned bx = b / ROWS, r = b % ROWS, by = r % NBY, bz = r / NBY;
    if (keep_mod > 1 && (b * 2654435761u >> 16) % keep_mod != 0) return;   // a sparse list: every keep_mod-th brick, scattered
    const size_t row = N, plane = (size_t)N * N;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // items of the brick: (y, z) pairs; wave w takes items w, w + 4, ... in batches of 4; consecutive items of a wave are consecutive planes
    // when YR == 4 (integrate's shape), otherwise rows first
    constexpr int ITEMS = YR * ZP / 4;   // per wave
    const size_t base = (size_t)(bz * ZP) * plane + (size_t)(by * YR) * row + bx * 64 + lane;
#pragma unroll 1
    for (int i0 = 0; i0 < ITEMS; i0 += 4) {
        float pd[4], pw[4];
        size_t at[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int it = (i0 + j) * 4 + wave;            // 0 .. YR * ZP - 1
            const int y = it % YR, z = it / YR;
            at[j] = base + (size_t)z * plane + (size_t)y * row;
            pd[j] = d[at[j]]; pw[j] = w[at[j]];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { d[at[j]] = pd[j] + 1.0f; w[at[j]] = pw[j] + 1.0f; }
    }
}

template <int YR, int ZP>
static void run(float *a, float *b, unsigned N, unsigned keep_mod) {
    const unsigned n = (N / 64) * (N / YR) * (N / ZP);
    const double bytes = 16.0 * (double)N * N * N / keep_mod;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double best = 0;
    for (int r = 0; r < 5; r++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((walk<YR, ZP>), dim3(n), dim3(256), 0, 0, a, b, N, keep_mod);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && bytes / (ms * 1e-3) / 1e9 > best) best = bytes / (ms * 1e-3) / 1e9;
    }
    printf("N %4u  brick 64 x %2d x %2d  every %u-th brick  %7.1f GB/s\n", N, YR, ZP, keep_mod, best);
}

int main() {
    for (unsigned N : {512u, 1024u}) {
        const size_t bytes = (size_t)N * N * N * 4;
        float *a, *b;
        if (hipMalloc((void **)&a, bytes) != hipSuccess || hipMalloc((void **)&b, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
        (void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes);
        for (unsigned keep : {1u, 8u}) {
            run<4, 32>(a, b, N, keep);
            run<8, 16>(a, b, N, keep);
            run<16, 8>(a, b, N, keep);
            run<32, 4>(a, b, N, keep);
            run<4, 8>(a, b, N, keep);
            run<4, 16>(a, b, N, keep);
        }
        (void)hipFree(a); (void)hipFree(b);
    }
    return 0;
}
