// Is the rate of an in-place update walk a property of the ALLOCATION it runs on?  (round 4: tools/ubench_alias.hip showed the same
// kernel at 5.0 or 6.0 TB/s on different hipMalloc results within one process.)  Here: 8 pairs of 512 MiB arrays measured round
// robin three times (is it stable per pair?), 1 GiB windows of one 12 GiB allocation (is it a property of address ranges?), and the
// pairs again after freeing and re-allocating half of them.
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_place tools/ubench_place.hip && build/ubench_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>   // 0 in place, 2 write only
__global__ __launch_bounds__(256) void walk(float *__restrict__ d, float *__restrict__ w, unsigned N) {
    constexpr int YR = 4, ZP = 32;
    const unsigned NBY = N / YR, NBZ = N / ZP, ROWS = NBY * NBZ;
    const unsigned b = blockIdx.x;
    const unsigned bx = b / ROWS, r = b % ROWS, by = r % NBY, bz = r / NBY;
    const size_t row = N, plane = (size_t)N * N;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t base = (size_t)(bz * ZP) * plane + (size_t)(by * YR + wave) * row + bx * 64 + lane;
#pragma unroll 1
    for (int z = 0; z < ZP; z += 4) {
        float pd[4], pw[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const size_t at = base + (size_t)(z + j) * plane;
            if (MODE == 0) { pd[j] = d[at]; pw[j] = w[at]; } else { pd[j] = (float)j; pw[j] = (float)z; }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const size_t at = base + (size_t)(z + j) * plane;
            d[at] = pd[j] + 1.0f; w[at] = pw[j] + 1.0f;
        }
    }
}

template <int MODE>
static double rate(float *a, float *b) {
    const unsigned N = 512, n = (N / 64) * (N / 4) * (N / 32);
    const double bytes = (MODE == 0 ? 16.0 : 8.0) * (double)N * N * N;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double best = 0;
    for (int r = 0; r < 4; r++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((walk<MODE>), dim3(n), dim3(256), 0, 0, a, b, N);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && bytes / (ms * 1e-3) / 1e9 > best) best = bytes / (ms * 1e-3) / 1e9;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return best;
}

int main() {
    const size_t half = (size_t)512 << 20;
    const int P = 8;
    std::vector<float *> a(P), b(P);
    for (int i = 0; i < P; i++) {
        (void)hipMalloc((void **)&a[i], half); (void)hipMalloc((void **)&b[i], half);
        (void)hipMemset(a[i], 0, half); (void)hipMemset(b[i], 0, half);
    }
    for (int round = 0; round < 3; round++)
        for (int i = 0; i < P; i++)
            printf("round %d pair %d  %p %p   in place %7.1f  write only %7.1f GB/s\n", round, i, (void *)a[i], (void *)b[i], rate<0>(a[i], b[i]), rate<2>(a[i], b[i]));
    // mixed pairs: first array of pair i with the second of pair j
    for (int i = 0; i < P; i += 3)
        for (int j = 0; j < P; j += 3) printf("mixed a[%d] b[%d]   in place %7.1f GB/s\n", i, j, rate<0>(a[i], b[j]));
    for (int i = 0; i < P; i += 2) { (void)hipFree(a[i]); (void)hipFree(b[i]); }
    for (int i = 0; i < P; i += 2) {
        (void)hipMalloc((void **)&a[i], half); (void)hipMalloc((void **)&b[i], half);
        (void)hipMemset(a[i], 0, half); (void)hipMemset(b[i], 0, half);
    }
    for (int i = 0; i < P; i++)
        printf("after re-allocating the even pairs: pair %d  %p %p   in place %7.1f  write only %7.1f GB/s\n", i, (void *)a[i], (void *)b[i], rate<0>(a[i], b[i]), rate<2>(a[i], b[i]));
    for (int i = 0; i < P; i++) { (void)hipFree(a[i]); (void)hipFree(b[i]); }
    char *big;
    const size_t G = (size_t)1 << 30;
    if (hipMalloc((void **)&big, 12 * G) == hipSuccess) {
        (void)hipMemset(big, 0, 12 * G);
        for (int w = 0; w < 12; w++)
            printf("12 GiB allocation %p, window at %2d GiB (two halves of it): in place %7.1f  write only %7.1f GB/s\n", (void *)big, w,
                   rate<0>((float *)(big + w * G), (float *)(big + w * G + half)), rate<2>((float *)(big + w * G), (float *)(big + w * G + half)));
        // the second array `pad` MiB beyond "directly behind the first": which distances between the two arrays are fast?
        for (int pad : {0, 1, 8, 16, 32, 48, 64, 96, 128, 160, 192, 224, 256, 288, 320, 352, 384, 416, 448, 480, 496, 511, 512, 768, 1280})
            printf("second array %4d MiB past the end of the first: in place %7.1f  write only %7.1f GB/s\n", pad,
                   rate<0>((float *)big, (float *)(big + half + ((size_t)pad << 20))), rate<2>((float *)big, (float *)(big + half + ((size_t)pad << 20))));
        for (int pad : {0, 64, 128, 256, 384})
            printf("first array at 3 GiB + 4 MiB, second %4d MiB past its end: in place %7.1f GB/s\n", pad,
                   rate<0>((float *)(big + 3 * G + (4 << 20)), (float *)(big + 3 * G + (4 << 20) + half + ((size_t)pad << 20))));
        // the two arrays far apart inside it
        for (int w = 0; w < 6; w++)
            printf("12 GiB allocation, arrays at %d and %d GiB: in place %7.1f GB/s\n", w, w + 6, rate<0>((float *)(big + w * G), (float *)(big + (w + 6) * G)));
        (void)hipFree(big);
    }
    return 0;
}
