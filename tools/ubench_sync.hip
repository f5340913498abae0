// What handing work to a second stream and joining it again costs the first stream, by mechanism:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_sync.hip -o build/ubench_sync
// Per iteration on the main stream: A (about 20 us) ; release the side stream ; B (about 20 us) ; join the side stream ; the side stream
// runs C (about 10 us) between release and join -- the shape of tsdf_pipeline_step (integrate ; release ; cast ; join).
//   none:    A ; B on one stream, C not run (the floor)
//   event:   hipEventRecord + hipStreamWaitEvent either way (events without the system-scope fence, as the pipeline creates them)
//   value:   hipStreamWriteValue32 + hipStreamWaitValue32 on a word of signal memory
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin_kernel(float *p, int n, int rounds) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = p[i];
    for (int r = 0; r < rounds; r++) v = v * 1.0001f + 0.5f;
    p[i] = v;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const int n = 1 << 20, N = 200;
    float *p, *q;
    CK(hipMalloc(&p, n * sizeof(float))); CK(hipMalloc(&q, n * sizeof(float)));
    CK(hipMemset(p, 0, n * sizeof(float))); CK(hipMemset(q, 0, n * sizeof(float)));
    hipStream_t m, s;
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&m, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo));
    hipEvent_t rel, join;
    CK(hipEventCreateWithFlags(&rel, hipEventDisableTiming | hipEventDisableSystemFence)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming | hipEventDisableSystemFence));
    uint32_t *sig = nullptr;
    hipError_t se = hipExtMallocWithFlags((void **)&sig, 2 * sizeof(uint32_t), hipMallocSignalMemory);
    if (se != hipSuccess) { std::printf("signal memory: %s\n", hipGetErrorString(se)); sig = nullptr; }
    else CK(hipMemset(sig, 0, 2 * sizeof(uint32_t)));
    const int ra = 3000, rc = 1500;
    auto A = [&] { hipLaunchKernelGGL(spin_kernel, dim3(n / 256), dim3(256), 0, m, p, n, ra); };
    auto C = [&] { hipLaunchKernelGGL(spin_kernel, dim3(n / 1024), dim3(256), 0, s, q, n / 4, rc); };
    uint32_t seq = 0;
    for (int rep = 0; rep < 3; rep++)
    for (int mode = 0; mode < 3; mode++) {
        if (mode == 2 && !sig) continue;
        auto iter = [&] {
            A();
            if (mode == 1) { hipEventRecord(rel, m); hipStreamWaitEvent(s, rel, 0); C(); hipEventRecord(join, s); }
            if (mode == 2) { seq++; hipStreamWriteValue32(m, sig, seq, 0); hipStreamWaitValue32(s, sig, seq, hipStreamWaitValueGte, 0xffffffffu); C(); hipStreamWriteValue32(s, sig + 1, seq, 0); }
            A();
            if (mode == 1) hipStreamWaitEvent(m, join, 0);
            if (mode == 2) hipStreamWaitValue32(m, sig + 1, seq, hipStreamWaitValueGte, 0xffffffffu);
        };
        for (int i = 0; i < 20; i++) iter();
        hipStreamSynchronize(m); hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; i++) iter();
        hipStreamSynchronize(m); hipStreamSynchronize(s);
        double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / N;
        std::printf("%-6s %.2f us per iteration\n", mode == 0 ? "none" : mode == 1 ? "event" : "value", us);
    }
    return 0;
}
