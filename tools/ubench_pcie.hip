// Host <-> device transfer rates for the sizes the reference's blocking API moves per frame (614 KB depth up, 2 x 3.7 MB maps down):
// pageable / pinned / registered memcpy, a kernel writing straight into pinned host memory, and what hipHostRegister costs.
// hipcc --offload-arch=gfx950 -O2 tools/ubench_pcie.hip -o build/ubench_pcie
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill_kernel(float4 *dst, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_float4(v, v + 1, v + 2, v + 3);
}
int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    for (size_t bytes : {(size_t)640 * 480 * 2, (size_t)640 * 480 * 12, (size_t)2 * 640 * 480 * 12}) {
        void *dev; CK(hipMalloc(&dev, bytes));
        void *pageable = aligned_alloc(4096, (bytes + 4095) & ~(size_t)4095); memset(pageable, 1, bytes);
        void *pinned; CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
        void *reg = aligned_alloc(4096, (bytes + 4095) & ~(size_t)4095); memset(reg, 1, bytes);
        double t0 = now(); CK(hipHostRegister(reg, bytes, hipHostRegisterDefault)); double t_reg = now() - t0;
        const int R = 50;
        auto time_copy = [&](void *host, bool d2h) {
            for (int i = 0; i < 3; i++) { (void)hipMemcpyAsync(d2h ? host : dev, d2h ? dev : host, bytes, d2h ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); }
            double t = now();
            for (int i = 0; i < R; i++) { (void)hipMemcpyAsync(d2h ? host : dev, d2h ? dev : host, bytes, d2h ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); }
            return (now() - t) / R * 1e6;
        };
        printf("%8zu bytes: register %.0f us | D2H pageable %.1f pinned %.1f registered %.1f us | H2D pageable %.1f pinned %.1f registered %.1f us",
               bytes, t_reg * 1e6, time_copy(pageable, true), time_copy(pinned, true), time_copy(reg, true), time_copy(pageable, false), time_copy(pinned, false), time_copy(reg, false));
        // a kernel storing straight into pinned host memory
        void *pinned_dev; CK(hipHostGetDevicePointer(&pinned_dev, pinned, 0));
        for (int g : {256, 1024}) {
            for (int i = 0; i < 3; i++) { hipLaunchKernelGGL(fill_kernel, dim3(g), dim3(256), 0, s, (float4 *)pinned_dev, bytes / 16, 1.0f); (void)hipStreamSynchronize(s); }
            double t = now();
            for (int i = 0; i < R; i++) { hipLaunchKernelGGL(fill_kernel, dim3(g), dim3(256), 0, s, (float4 *)pinned_dev, bytes / 16, (float)i); (void)hipStreamSynchronize(s); }
            printf(" | kernel -> host (%d wg) %.1f us", g, (now() - t) / R * 1e6);
        }
        // device kernel + pinned copy + host memcpy into pageable (what a staging ring costs)
        {
            double t = now();
            for (int i = 0; i < R; i++) { (void)hipMemcpyAsync(pinned, dev, bytes, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); memcpy(pageable, pinned, bytes); }
            printf(" | D2H pinned + memcpy %.1f us", (now() - t) / R * 1e6);
        }
        printf("\n");
        (void)hipHostUnregister(reg); (void)hipHostFree(pinned); (void)hipFree(dev); free(pageable); free(reg);
    }
    return 0;
}
