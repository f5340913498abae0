// Why does a dense in-place update walk of two 512^3 fp32 arrays stop at 5.0 TB/s when the same walk of two 1024^3 arrays runs at
// 6.0-6.1 (tools/ubench_tlb.hip, round 4)?  Candidates: the two arrays' lines d[i], w[i] -- requested together by every lane -- falling
// on the same channel / bank when the arrays are a power of two apart; the power-of-two row and plane pitches.  Here: the second
// array placed `pad` bytes further than "directly behind the first", read-only / write-only / in-place walks, and N = 512, 640, 768, 1024.
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_alias tools/ubench_alias.hip && build/ubench_alias
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>   // 0 in place, 1 read only, 2 write only
__global__ __launch_bounds__(256) void walk(float *__restrict__ d, float *__restrict__ w, unsigned N, float *sink) {
    constexpr int YR = 4, ZP = 32;
    const unsigned NBY = N / YR, NBZ = N / ZP, ROWS = NBY * NBZ;
    const unsigned b = blockIdx.x;
    const unsigned bx = b / ROWS, r = b % ROWS, by = r % NBY, bz = r / NBY;
    const size_t row = N, plane = (size_t)N * N;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t base = (size_t)(bz * ZP) * plane + (size_t)(by * YR + wave) * row + bx * 64 + lane;
    float acc = 0.f;
#pragma unroll 1
    for (int z = 0; z < ZP; z += 4) {
        float pd[4], pw[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const size_t at = base + (size_t)(z + j) * plane;
            if (MODE != 2) { pd[j] = d[at]; pw[j] = w[at]; } else { pd[j] = (float)j; pw[j] = (float)z; }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const size_t at = base + (size_t)(z + j) * plane;
            if (MODE != 1) { d[at] = pd[j] + 1.0f; w[at] = pw[j] + 1.0f; } else acc += pd[j] + pw[j];
        }
    }
    if (MODE == 1 && acc == 12345.678f) *sink = acc;
}

template <int MODE>
static double run(float *a, float *b, unsigned N, float *sink) {
    const unsigned n = (N / 64) * (N / 4) * (N / 32);
    const double bytes = (MODE == 0 ? 16.0 : 8.0) * (double)N * N * N;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double best = 0;
    for (int r = 0; r < 5; r++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((walk<MODE>), dim3(n), dim3(256), 0, 0, a, b, N, sink);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && bytes / (ms * 1e-3) / 1e9 > best) best = bytes / (ms * 1e-3) / 1e9;
    }
    return best;
}

int main() {
    float *sink; (void)hipMalloc((void **)&sink, 4);
    for (unsigned N : {512u, 640u, 768u, 1024u}) {
        const size_t bytes = (size_t)N * N * N * 4, slack = 64u << 20;
        char *buf;
        if (hipMalloc((void **)&buf, 2 * bytes + slack) != hipSuccess) { printf("alloc failed\n"); return 1; }
        (void)hipMemset(buf, 0, 2 * bytes + slack);
        for (size_t pad : {(size_t)0, (size_t)256, (size_t)1024, (size_t)4096, (size_t)16384, (size_t)65536, (size_t)(256 << 10), (size_t)(1 << 20) + 4096, (size_t)(17 << 20) + 8192 + 256}) {
            float *a = (float *)buf, *b = (float *)(buf + bytes + pad);
            printf("N %4u  second array %9zu B past the end of the first (base %p): in place %7.1f  read only %7.1f  write only %7.1f GB/s\n", N, pad, (void *)buf,
                   run<0>(a, b, N, sink), run<1>(a, b, N, sink), run<2>(a, b, N, sink));
        }
        (void)hipFree(buf);
        // and as two allocations, as the library makes them
        float *a, *b;
        (void)hipMalloc((void **)&a, bytes); (void)hipMalloc((void **)&b, bytes);
        (void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes);
        printf("N %4u  two hipMallocs %p %p (apart %lld B): in place %7.1f  read only %7.1f  write only %7.1f GB/s\n", N, (void *)a, (void *)b, (long long)((char *)b - (char *)a),
               run<0>(a, b, N, sink), run<1>(a, b, N, sink), run<2>(a, b, N, sink));
        (void)hipFree(a); (void)hipFree(b);
    }
    return 0;
}
