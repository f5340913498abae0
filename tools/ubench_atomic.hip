// How long do N workgroups take to each draw a slot with one returning atomicAdd, on one address or spread over 8 / 64?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/uba tools/ubench_atomic.hip && /tmp/uba
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ __launch_bounds__(256) void draw(unsigned *counters, unsigned n_addr, unsigned stride, unsigned *out, int work) {
    __shared__ unsigned base;
    float x = threadIdx.x;
    for (int i = 0; i < work; i++) x = x * 1.0001f + 0.5f;   // (something before the atomic, like the cull kernel's projection)
    if (threadIdx.x == 0) base = atomicAdd(&counters[(blockIdx.x % n_addr) * stride], 3u + (x == 7.0f));
    __syncthreads();
    if (threadIdx.x < 3) out[(blockIdx.x * 4 + threadIdx.x) & 0xffff] = base;
}
int main() {
    unsigned *c, *o;
    hipMalloc(&c, 1 << 20); hipMalloc(&o, 1 << 20); hipMemset(c, 0, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {512, 2048}) for (unsigned n_addr : {1u, 8u, 64u}) for (int work : {0, 200}) {
        float best = 1e9f;
        for (int r = 0; r < 20; r++) {
            hipExtLaunchKernelGGL(draw, dim3(wgs), dim3(256), 0, 0, e0, e1, 0, c, n_addr, 64u, o, work);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%5d workgroups, %2u addresses, %3d iterations of work before: %.2f us\n", wgs, n_addr, work, best * 1e3f);
    }
    return 0;
}
