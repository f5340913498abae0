// What a kernel boundary costs on a stream, and whether a captured graph makes it cheaper:  hipcc --offload-arch=gfx950 -O3 tools/ubench_graph.hip -o build/ubench_graph
// N dependent launches of a kernel that does ~2 us of work on a few thousand waves (the size of the step's small kernels), timed
// (a) launched one by one on a stream, (b) as one captured graph launched repeatedly.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void small_kernel(float *p, int n, int rounds) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = p[i];
    for (int r = 0; r < rounds; r++) v = v * 1.0001f + 0.5f;
    p[i] = v;
}
int main() {
    const int n = 1 << 19, N = 200, reps = 20;
    float *p;
    hipMalloc(&p, n * sizeof(float));
    hipMemset(p, 0, n * sizeof(float));
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int rounds : {1, 200, 2000}) {
        auto launch_all = [&] { for (int i = 0; i < N; i++) hipLaunchKernelGGL(small_kernel, dim3(n / 256), dim3(256), 0, s, p, n, rounds); };
        launch_all(); hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) launch_all();
        hipStreamSynchronize(s);
        double stream_us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / (reps * N);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        launch_all();
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        double graph_us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / (reps * N);
        // one kernel alone, for the work's own share
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, s);
        hipLaunchKernelGGL(small_kernel, dim3(n / 256), dim3(256), 0, s, p, n, rounds);
        hipEventRecord(b, s); hipStreamSynchronize(s);
        float one = 0; hipEventElapsedTime(&one, a, b);
        printf("rounds %5d: %.2f us per dependent launch on a stream, %.2f us in a graph (one launch between two events: %.2f us)\n", rounds, stream_us, graph_us, one * 1e3);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
