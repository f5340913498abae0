// In-place update walk micro-benchmark (diagnostics): what bounds integrate's read-modify-write of the distance and weight
// arrays on gfx950, and which walk shapes / cache policies move the ceiling.
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_walk tools/ubench_walk.hip && build/ubench_walk
// 512^3 grid, two fp32 arrays, every voxel read and written once.  A wave covers LF * 64 consecutive x of one row, a workgroup 4
// rows (y), 32 planes (z), 4 planes per batch.  REMAP: consecutive bricks (x fastest) on the same XCD instead of round-robin.
// NT: nontemporal loads and stores.  ZMAJOR: a workgroup's 4 waves take 4 consecutive planes of one row instead of 4 rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int LF> struct V;
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef float vf4 __attribute__((ext_vector_type(4)));
template <> struct V<1> { typedef float T; };
template <> struct V<2> { typedef vf2 T; };
template <> struct V<4> { typedef vf4 T; };

template <typename T> __device__ inline T ld(const T *p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
template <typename T> __device__ inline void st(T *p, T v, bool nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; }
template <typename T> __device__ inline T inc(T a) { return a + 1.0f; }

// column-major launch order: workgroup i takes x-brick i / rows, row i % rows -- workgroups in flight together are rows apart
template <int LF, int YROWS, int ZPLANES, int ORDER>
__global__ __launch_bounds__(256) void walk_cm(typename V<LF>::T *__restrict__ d, typename V<LF>::T *__restrict__ w) {
    typedef typename V<LF>::T T;
    constexpr unsigned NBX = 512 / (64 * LF), NBY = 512 / YROWS, NBZ = 512 / ZPLANES, ROWS = NBY * NBZ;
    const unsigned b = blockIdx.x;
    unsigned bx = b / ROWS, r = b % ROWS;
    if (ORDER == 1) { bx = (b / 8) % NBX; r = (b % 8) + 8 * (b / (8 * NBX)); }   // 8 consecutive workgroups (one per XCD): same x-brick, 8 consecutive rows; then the next x-brick
    const unsigned by = r % NBY, bz = r / NBY;
    const size_t row = 512 / LF, plane = row * 512;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    size_t idx = (size_t)(bz * ZPLANES) * plane + (size_t)(by * YROWS + wave) * row + bx * 64 + lane;
#pragma unroll 1
    for (unsigned z = 0; z < ZPLANES; z += 4) {
        T pd[4], pw[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { pd[j] = d[idx + (z + j) * plane]; pw[j] = w[idx + (z + j) * plane]; }
#pragma unroll
        for (int j = 0; j < 4; j++) { d[idx + (z + j) * plane] = inc(pd[j]); w[idx + (z + j) * plane] = inc(pw[j]); }
    }
}
template <int LF, int YROWS, int ZPLANES, int ORDER>
static void run_cm(void *a, void *b, const char *what) {
    typedef typename V<LF>::T T;
    const unsigned n = (512 / (64 * LF)) * (512 / YROWS) * (512 / ZPLANES);
    const double bb = 4.0 * 512.0 * 512 * 512 * 4;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double best = 0;
    for (int r = 0; r < 6; r++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((walk_cm<LF, YROWS, ZPLANES, ORDER>), dim3(n), dim3(256), 0, 0, (T *)a, (T *)b);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && bb / (ms * 1e-3) / 1e9 > best) best = bb / (ms * 1e-3) / 1e9;
    }
    printf("%-58s %7.1f GB/s\n", what, best);
}

// two x-adjacent 64-wide bricks per workgroup of 8 waves (4 B per lane): waves 0-3 the left brick's rows, 4-7 the right one's
__global__ __launch_bounds__(512) void walk_pair(float *__restrict__ d, float *__restrict__ w) {
    const unsigned b = blockIdx.x;                      // 4 x-pairs, 128 rows, 16 layers; column by column
    const unsigned ROWS = 128 * 16;
    const unsigned bxp = b / ROWS, r = b % ROWS, by = r % 128, bz = r / 128;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t plane = 512 * 512;
    size_t idx = (size_t)(bz * 32) * plane + (size_t)(by * 4 + (wave & 3)) * 512 + (bxp * 2 + (wave >> 2)) * 64 + lane;
#pragma unroll 1
    for (unsigned z = 0; z < 32; z += 4) {
        float pd[4], pw[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { pd[j] = d[idx + (z + j) * plane]; pw[j] = w[idx + (z + j) * plane]; }
#pragma unroll
        for (int j = 0; j < 4; j++) { d[idx + (z + j) * plane] = pd[j] + 1.0f; w[idx + (z + j) * plane] = pw[j] + 1.0f; }
    }
}

template <int LF, bool REMAP, bool NT, int YROWS, int ZPLANES>
__global__ __launch_bounds__(256) void walk(typename V<LF>::T *__restrict__ d, typename V<LF>::T *__restrict__ w) {
    typedef typename V<LF>::T T;
    constexpr unsigned NBX = 512 / (64 * LF), NBY = 512 / YROWS, NBZ = 512 / ZPLANES;
    const unsigned n = NBX * NBY * NBZ;
    unsigned b = blockIdx.x;
    if (REMAP) b = (b % 8) * (n / 8) + b / 8;
    const unsigned bx = b % NBX, by = (b / NBX) % NBY, bz = b / (NBX * NBY);
    const size_t row = 512 / LF, plane = row * 512;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // YROWS rows of the brick over the 4 waves (YROWS = 4: one each; YROWS = 1: the waves split the planes)
    const unsigned wy = YROWS == 4 ? wave : 0, wz = YROWS == 4 ? 0 : wave * (ZPLANES / 4);
    const unsigned nz = YROWS == 4 ? ZPLANES : ZPLANES / 4;
    size_t idx = (size_t)(bz * ZPLANES + wz) * plane + (size_t)(by * YROWS + wy) * row + bx * 64 + lane;
#pragma unroll 1
    for (unsigned z = 0; z < nz; z += 4) {
        T pd[4], pw[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { pd[j] = ld(d + idx + (z + j) * plane, NT); pw[j] = ld(w + idx + (z + j) * plane, NT); }
#pragma unroll
        for (int j = 0; j < 4; j++) { st(d + idx + (z + j) * plane, inc(pd[j]), NT); st(w + idx + (z + j) * plane, inc(pw[j]), NT); }
    }
}

// general shape: a workgroup covers LF * 64 x, YR rows, ZP planes; its 4 waves take the (row, plane) slots in turn, rows fastest
// (ROWFAST) or planes fastest, 4 slots in flight per wave
template <int LF, int YR, int ZP, bool ROWFAST>
__global__ __launch_bounds__(256) void walk2(typename V<LF>::T *__restrict__ d, typename V<LF>::T *__restrict__ w) {
    typedef typename V<LF>::T T;
    constexpr unsigned NBX = 512 / (64 * LF), NBY = 512 / YR;
    const unsigned b = blockIdx.x;
    const unsigned bx = b % NBX, by = (b / NBX) % NBY, bz = b / (NBX * NBY);
    const size_t row = 512 / LF, plane = row * 512;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t base = (size_t)(bz * ZP) * plane + (size_t)(by * YR) * row + bx * 64 + lane;
#pragma unroll 1
    for (unsigned s0 = wave * 4; s0 < YR * ZP; s0 += 16) {
        T pd[4], pw[4];
        size_t idx[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned sl = s0 + j, y = ROWFAST ? sl % YR : sl / ZP, z = ROWFAST ? sl / YR : sl % ZP;
            idx[j] = base + z * plane + y * row;
            pd[j] = d[idx[j]]; pw[j] = w[idx[j]];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { d[idx[j]] = inc(pd[j]); w[idx[j]] = inc(pw[j]); }
    }
}
template <int LF, int YR, int ZP, bool ROWFAST>
static void run2(void *a, void *b, const char *what) {
    typedef typename V<LF>::T T;
    const unsigned n = (512 / (64 * LF)) * (512 / YR) * (512 / ZP);
    const double bb = 4.0 * 512.0 * 512 * 512 * 4;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double best = 0;
    for (int r = 0; r < 6; r++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((walk2<LF, YR, ZP, ROWFAST>), dim3(n), dim3(256), 0, 0, (T *)a, (T *)b);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && bb / (ms * 1e-3) / 1e9 > best) best = bb / (ms * 1e-3) / 1e9;
    }
    printf("%-58s %7.1f GB/s\n", what, best);
}

// the product's update_walk_kernel (volume.hip), for reference
__global__ __launch_bounds__(256) void update_walk_ref(float *__restrict__ d, float *__restrict__ w) {
    const unsigned b = blockIdx.x, bx = b % 8, by = (b / 8) % 128, bz = b / (8 * 128);
    const size_t plane = (size_t)512 * 512;
    const size_t idx = (size_t)bz * 32 * plane + (size_t)(by * 4 + threadIdx.y) * 512 + bx * 64 + threadIdx.x;
#pragma unroll 1
    for (int z = 0; z < 32; z += 4) {
        float pd[4], pw[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { pd[j] = d[idx + (z + j) * plane]; pw[j] = w[idx + (z + j) * plane]; }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float nw = pw[j] + 1.0f;
            d[idx + (z + j) * plane] = (pd[j] * pw[j] + 3.0f) / nw;
            w[idx + (z + j) * plane] = nw;
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

template <int LF, bool REMAP, bool NT, int YROWS, int ZPLANES>
static void run(void *a, void *b, const char *what) {
    typedef typename V<LF>::T T;
    const unsigned n = (512 / (64 * LF)) * (512 / YROWS) * (512 / ZPLANES);
    const double bb = 4.0 * 512.0 * 512 * 512 * 4;
    printf("%-58s %7.1f GB/s\n", what, timed([&] { hipLaunchKernelGGL((walk<LF, REMAP, NT, YROWS, ZPLANES>), dim3(n), dim3(256), 0, 0, (T *)a, (T *)b); }, bb));
}

int main() {
    const size_t bytes = (size_t)1 << 29;
    void *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    (void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes);
    const double bb = 4.0 * 512.0 * 512 * 512 * 4;
    printf("%-58s %7.1f GB/s\n", "update_walk_kernel of the product (64 x 4 block)", timed([&] { hipLaunchKernelGGL(update_walk_ref, dim3(16384), dim3(64, 4), 0, 0, (float *)a, (float *)b); }, bb));
    run<1, false, false, 4, 32>(a, b, "4 B/lane, 4 rows x 32 planes (integrate's walk)");
    printf("%-58s %7.1f GB/s\n", "update_walk_kernel of the product (64 x 4 block)", timed([&] { hipLaunchKernelGGL(update_walk_ref, dim3(16384), dim3(64, 4), 0, 0, (float *)a, (float *)b); }, bb));
    printf("%-58s %7.1f GB/s\n", "two x-adjacent bricks per workgroup of 8 waves", timed([&] { hipLaunchKernelGGL(walk_pair, dim3(4 * 128 * 16), dim3(512), 0, 0, (float *)a, (float *)b); }, bb));
    run_cm<1, 4, 32, 0>(a, b, "4 B/lane, 4 rows x 32 planes, column by column");
    run_cm<1, 4, 32, 1>(a, b, "4 B/lane, 4 rows x 32 planes, 8 rows then next column");
    run_cm<2, 4, 32, 0>(a, b, "8 B/lane, 4 rows x 32 planes, column by column");
    run_cm<1, 4, 16, 0>(a, b, "4 B/lane, 4 rows x 16 planes, column by column");
    run_cm<1, 4, 64, 0>(a, b, "4 B/lane, 4 rows x 64 planes, column by column");
    run<1, true, false, 4, 32>(a, b, "4 B/lane, 4 rows x 32 planes, XCD-contiguous");
    run<1, false, true, 4, 32>(a, b, "4 B/lane, 4 rows x 32 planes, nontemporal");
    run<1, true, true, 4, 32>(a, b, "4 B/lane, 4 rows x 32 planes, XCD-contiguous, nontemporal");
    run<2, false, false, 4, 32>(a, b, "8 B/lane, 4 rows x 32 planes");
    run<2, true, false, 4, 32>(a, b, "8 B/lane, 4 rows x 32 planes, XCD-contiguous");
    run<2, false, true, 4, 32>(a, b, "8 B/lane, 4 rows x 32 planes, nontemporal");
    run<2, true, true, 4, 32>(a, b, "8 B/lane, 4 rows x 32 planes, XCD-contiguous, nontemporal");
    run<4, false, false, 4, 32>(a, b, "16 B/lane, 4 rows x 32 planes");
    run<4, true, false, 4, 32>(a, b, "16 B/lane, 4 rows x 32 planes, XCD-contiguous");
    run<4, true, true, 4, 32>(a, b, "16 B/lane, 4 rows x 32 planes, XCD-contiguous, nontemporal");
    run<1, false, false, 1, 32>(a, b, "4 B/lane, 1 row x 32 planes (waves split z)");
    run<1, true, false, 1, 32>(a, b, "4 B/lane, 1 row x 32 planes, XCD-contiguous");
    run<2, true, false, 1, 32>(a, b, "8 B/lane, 1 row x 32 planes, XCD-contiguous");
    run2<1, 4, 32, true>(a, b, "walk2 4 B/lane 4 rows x 32 planes, rows fastest");
    run2<1, 4, 32, false>(a, b, "walk2 4 B/lane 4 rows x 32 planes, planes fastest");
    run2<1, 32, 4, true>(a, b, "walk2 4 B/lane 32 rows x 4 planes, rows fastest");
    run2<1, 128, 1, true>(a, b, "walk2 4 B/lane 128 rows x 1 plane");
    run2<1, 16, 8, true>(a, b, "walk2 4 B/lane 16 rows x 8 planes, rows fastest");
    run2<2, 32, 4, true>(a, b, "walk2 8 B/lane 32 rows x 4 planes, rows fastest");
    run2<2, 4, 32, true>(a, b, "walk2 8 B/lane 4 rows x 32 planes, rows fastest");
    run2<4, 32, 4, true>(a, b, "walk2 16 B/lane 32 rows x 4 planes, rows fastest");
    run<1, false, false, 4, 16>(a, b, "4 B/lane, 4 rows x 16 planes");
    run<1, true, false, 4, 16>(a, b, "4 B/lane, 4 rows x 16 planes, XCD-contiguous");
    return 0;
}
