// ICP tracking for gfx950 (wave64): the replacement of the reference's third_party/ICP_CUDA --
// pyrDown / createVMap / createNMap (Cuda/pyrdown.cu), estimateStep (Cuda/estimate.cu) and
// ICPOdometry::getIncrementalTransformation (ICPOdometry.cpp:97-136).
//
// What is different from the reference's shape, and why:
//   * The reference runs 19 iterations (10/5/4 over three pyramid levels), each = reduction kernel, second reduction
//     kernel, device synchronise, 116-byte download, 6x6 LDLT + SE3 exponential on the host, next launch.  Here the
//     pose lives on the device: the last workgroup of icp_reduce_kernel to finish adds up the partial sums, solves the
//     6x6 system in double and applies T <- exp(x) * T in place, so a frame is 19 launches queued back to back with no
//     host round trip; the host reads the pose once at the end.
//   * The reduction is deterministic: a fixed grid, per-thread fp32 partial sums (as in the reference), wave64 shuffle
//     tree, fixed-order cross-wave and cross-block sums (the last stage in double).  The reference's sums depend on the
//     threads x blocks the caller passes; those two arguments are accepted and ignored.
//   * Maps keep the reference's planar layout (component c of pixel (x,y) at [(y + c*rows)*cols + x], NaN pattern
//     0x7fffffff in component 0 of an invalid pixel); lane <-> x, so every plane access is coalesced.
// Per-pixel arithmetic follows the reference's operation order with fp contraction off.
#include <cmath>
#include <cstring>
#include <new>

#include "common.hpp"

struct tsdf_icp {
    int width, height;
    float cx, cy, fx, fy;
    float dist_thresh, angle_thresh;
    int device;
    hipStream_t stream;
    uint16_t *depth[3];                      // pyramid scratch (used by both init calls)
    uint16_t *upload;                        // the host variants' image on the device
    float *vmap_prev[3], *nmap_prev[3];      // model
    float *vmap_curr[3], *nmap_curr[3];      // current frame
    float *partial;                          // 2 x kIcpBlocks x 32 floats: per-block sums of the 29 products (two steps)
    double *state;                           // device, 2 x kIcpStateDoubles: [0..15] T (column-major), [16..17] residual,
                                             //         inliers, [18..53] A (float values), [54..59] b
    int side;                                // which copy of state / partial holds the latest step
    double *host_io;                         // pinned host memory the device reads and writes in place: [0..15] the pose an alignment starts from,
    double *host_io_dev;                     // [16..33] its result (pose, residual, inliers) -- no copy launches either side of the chain (4-5 us each)
    unsigned long long *arrivals;            // icp_persistent_kernel's grid barrier: a counter that only grows ...
    unsigned long long arrivals_base;        // ... and the value it will have when the next launch starts
    unsigned long long published_base;       // (leader variant: [1] = last published step, [2..] the published poses)
};

namespace tsdf {

constexpr int kIcpLevels = 3;
constexpr int kIcpBlocks = 256;   // one workgroup per CU (icp_finish_step adds them as 8 groups of 32)
constexpr int kIcpThreads = 256;
constexpr int kIcpStateDoubles = 64;

__device__ inline float nan_sentinel() { return __uint_as_float(0x7fffffffu); }

// pyrDownGaussKernel (Cuda/pyrdown.cu:41-78), sigma_color = 30 (:87).  (The weighted sums are exact in fp32: the
// weights are multiples of 1/256 and the values 16-bit, so the order of the additions does not matter.)
__global__ __launch_bounds__(256) void icp_pyr_down_kernel(const uint16_t *__restrict__ src, int src_rows, int src_cols,
                                                           uint16_t *__restrict__ dst) {
    const int rows = src_rows / 2, cols = src_cols / 2;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const int D = 5;
    const float sigma_color = 30;
    const int center = src[(size_t)(2 * y) * src_cols + 2 * x];
    const int x_mi = max(0, 2 * x - D / 2) - 2 * x, y_mi = max(0, 2 * y - D / 2) - 2 * y;
    const int x_ma = min(src_cols, 2 * x - D / 2 + D) - 2 * x, y_ma = min(src_rows, 2 * y - D / 2 + D) - 2 * y;
    float sum = 0, wall = 0;
    const float weights[3] = {0.375f, 0.25f, 0.0625f};
    // All 25 taps are requested at once, at clamped coordinates, and the taps outside [y_mi, y_ma) x [x_mi, x_ma) are left out of the
    // sums afterwards (the loop over the per-lane range was one dependent memory round trip per tap: 7-9 us for this kernel).
    int val_[D][D];
#pragma unroll
    for (int j = 0; j < D; j++)
#pragma unroll
        for (int i = 0; i < D; i++) {
            const int yi = min(max(j - D / 2, y_mi), y_ma - 1), xi = min(max(i - D / 2, x_mi), x_ma - 1);
            val_[j][i] = src[(size_t)(2 * y + yi) * src_cols + (2 * x + xi)];
        }
#pragma unroll
    for (int j = 0; j < D; j++)
#pragma unroll
        for (int i = 0; i < D; i++) {
            const int yi = j - D / 2, xi = i - D / 2, val = val_[j][i];
            if (yi >= y_mi && yi < y_ma && xi >= x_mi && xi < x_ma && abs(val - center) < 3 * sigma_color) {
                sum += val * weights[abs(xi)] * weights[abs(yi)];
                wall += weights[abs(xi)] * weights[abs(yi)];
            }
        }
    dst[(size_t)y * cols + x] = (uint16_t)(int)(sum / wall);
}

// computeVmapKernel (Cuda/pyrdown.cu:93-117) + computeNmapKernel (:135-172; Eigen's normalized(): n / sqrt(n.n) when n.n > 0) for all
// pyramid levels in ONE launch (blockIdx.z = level): a thread forms its own vertex and the two neighbours' its normal needs from the depth
// images with the same expressions -- the bits the reference's two kernels exchange through the vertex map -- so initICP is the pyramid (two launches) + this instead of a copy and nine launches
// (on the tracked loop's critical path for the model image: 18 us of launch chain).  Level 0 reads the caller's image and leaves
// the copy the pyramid scratch used to get from a memcpy.
struct IcpLevelMaps {
    const uint16_t *depth[3];
    float *vmap[3], *nmap[3];
    uint16_t *depth0_copy;
    int rows0, cols0;
    float fx, fy, cx, cy, depth_cutoff;
};
__device__ inline bool icp_vertex(const uint16_t *__restrict__ depth, int cols, int u, int v, float fx_inv, float fy_inv, float cx, float cy,
                                  float depth_cutoff, float &x, float &y, float &z) {
    z = depth[(size_t)v * cols + u] / 1000.f;  // mm -> metres
    if (!(z != 0 && z < depth_cutoff)) return false;
    x = z * (u - cx) * fx_inv;
    y = z * (v - cy) * fy_inv;
    return true;
}
__global__ __launch_bounds__(256) void icp_maps_kernel(const IcpLevelMaps m) {
    const int level = blockIdx.z, rows = m.rows0 >> level, cols = m.cols0 >> level, div = 1 << level;
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= cols || v >= rows) return;
    // Intr::operator()(level): every intrinsic divided by 2^level (Cuda/internal.h:63-67); 1.f / fx: createVMap (:131)
    const float fx_inv = 1.f / (m.fx / div), fy_inv = 1.f / (m.fy / div), cx = m.cx / div, cy = m.cy / div;
    const uint16_t *depth = m.depth[level];
    float *vmap = m.vmap[level], *nmap = m.nmap[level];
    if (level == 0) m.depth0_copy[(size_t)v * cols + u] = depth[(size_t)v * cols + u];
    float a0, a1, a2;
    const bool a_ok = icp_vertex(depth, cols, u, v, fx_inv, fy_inv, cx, cy, m.depth_cutoff, a0, a1, a2);
    if (a_ok) {
        vmap[(size_t)v * cols + u] = a0;
        vmap[(size_t)(v + rows) * cols + u] = a1;
        vmap[(size_t)(v + rows * 2) * cols + u] = a2;
    } else {
        vmap[(size_t)v * cols + u] = nan_sentinel();
    }
    if (u == cols - 1 || v == rows - 1) {
        nmap[(size_t)v * cols + u] = nan_sentinel();
        return;
    }
    float b0, b1, b2, c0, c1, c2;
    const bool b_ok = icp_vertex(depth, cols, u + 1, v, fx_inv, fy_inv, cx, cy, m.depth_cutoff, b0, b1, b2);
    const bool c_ok = icp_vertex(depth, cols, u, v + 1, fx_inv, fy_inv, cx, cy, m.depth_cutoff, c0, c1, c2);
    if (a_ok && b_ok && c_ok) {
        const float px = b0 - a0, py = b1 - a1, pz = b2 - a2;
        const float qx = c0 - a0, qy = c1 - a1, qz = c2 - a2;
        float rx = py * qz - pz * qy, ry = pz * qx - px * qz, rz = px * qy - py * qx;
        const float z = rx * rx + ry * ry + rz * rz;
        if (z > 0.0f) {
            const float l = sqrtf(z);
            rx = rx / l;
            ry = ry / l;
            rz = rz / l;
        }
        nmap[(size_t)v * cols + u] = rx;
        nmap[(size_t)(v + rows) * cols + u] = ry;
        nmap[(size_t)(v + 2 * rows) * cols + u] = rz;
    } else {
        nmap[(size_t)v * cols + u] = nan_sentinel();
    }
}

// __float2int_rn: to nearest even, saturating, NaN -> 0
__device__ inline int float2int_rn(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)rintf(f);
}

// Reduction::operator() (Cuda/estimate.cu:139-209): projective association of the current frame's vertices into the
// model, distance / angle gates, the 27 upper-triangular products of the row (n, v x n, n.(v_prev - v)) + inlier count.
// The pose is read from the device state (T as doubles, narrowed to float like `rotationMatrix().cast<float>()`).
template <bool COHERENT = false>
__device__ __forceinline__ void icp_finish_step(const float *partial, int n_blocks, const double *state_in, int update, double *pose_out,
                                       double *state_out);

// The sums of one Gauss-Newton step over this workgroup's share of the pixels, at the pose in `pose` (shared memory, doubles): per-thread
// fp32 partial sums as the reference's, wave64 shuffle tree, the four waves in a fixed order -> partial_out[29] of this workgroup.
__device__ __forceinline__ void icp_accumulate(const double *pose, const float *__restrict__ vmap_curr, const float *__restrict__ nmap_curr,
                                               const float *__restrict__ vmap_prev, const float *__restrict__ nmap_prev, int rows, int cols,
                                               float fx, float fy, float cx, float cy, float dist_thresh, float angle_thresh,
                                               float *__restrict__ partial_out, float (*shared)[32]) {
    float R[9], t[3];  // column-major
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) R[c * 3 + r] = (float)pose[c * 4 + r];
    for (int r = 0; r < 3; r++) t[r] = (float)pose[12 + r];

    float sum[29];
#pragma unroll
    for (int i = 0; i < 29; i++) sum[i] = 0.0f;
    const int N = rows * cols;
    for (int i = blockIdx.x * kIcpThreads + threadIdx.x; i < N; i += kIcpThreads * gridDim.x) {
        const int y = i / cols, x = i - y * cols;
        const float v0 = vmap_curr[(size_t)y * cols + x], v1 = vmap_curr[(size_t)(y + rows) * cols + x],
                    v2 = vmap_curr[(size_t)(y + 2 * rows) * cols + x];
        const float p0 = ((R[0] * v0 + R[3] * v1) + R[6] * v2) + t[0];
        const float p1 = ((R[1] * v0 + R[4] * v1) + R[7] * v2) + t[1];
        const float p2 = ((R[2] * v0 + R[5] * v1) + R[8] * v2) + t[2];
        const int px = float2int_rn(p0 * fx / p2 + cx);
        const int py = float2int_rn(p1 * fy / p2 + cy);
        if (px >= 0 && py >= 0 && px < cols && py < rows && v2 > 0 && p2 > 0) {
            const float w0 = vmap_prev[(size_t)py * cols + px], w1 = vmap_prev[(size_t)(py + rows) * cols + px],
                        w2 = vmap_prev[(size_t)(py + 2 * rows) * cols + px];
            const float n0 = nmap_curr[(size_t)y * cols + x], n1 = nmap_curr[(size_t)(y + rows) * cols + x],
                        n2 = nmap_curr[(size_t)(y + 2 * rows) * cols + x];
            const float m0 = (R[0] * n0 + R[3] * n1) + R[6] * n2;
            const float m1 = (R[1] * n0 + R[4] * n1) + R[7] * n2;
            const float m2 = (R[2] * n0 + R[5] * n1) + R[8] * n2;
            const float q0 = nmap_prev[(size_t)py * cols + px], q1 = nmap_prev[(size_t)(py + rows) * cols + px],
                        q2 = nmap_prev[(size_t)(py + 2 * rows) * cols + px];
            const float c0 = m1 * q2 - m2 * q1, c1 = m2 * q0 - m0 * q2, c2 = m0 * q1 - m1 * q0;
            const float sine = sqrtf((c0 * c0 + c1 * c1) + c2 * c2);
            const float d0 = w0 - p0, d1 = w1 - p1, d2 = w2 - p2;
            const float dist = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
            if (sine < angle_thresh && dist < dist_thresh && !(n0 != n0) && !(q0 != q0)) {
                float row[7];
                row[0] = q0;
                row[1] = q1;
                row[2] = q2;
                row[3] = p1 * q2 - p2 * q1;
                row[4] = p2 * q0 - p0 * q2;
                row[5] = p0 * q1 - p1 * q0;
                row[6] = (q0 * d0 + q1 * d1) + q2 * d2;
                int s = 0;
#pragma unroll
                for (int o = 0; o < 7; o++)
#pragma unroll
                    for (int in = o; in < 7; in++) sum[s++] += row[o] * row[in];
                sum[28] += 1.0f;
            }
        }
    }
    // wave64 shuffle tree, then the four waves of the workgroup in a fixed order
#pragma unroll
    for (int i = 0; i < 29; i++) {
        float v = sum[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        sum[i] = v;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 29; i++) shared[wave][i] = sum[i];
    }
    __syncthreads();
    if (threadIdx.x < 29) {
        partial_out[threadIdx.x] =
            ((shared[0][threadIdx.x] + shared[1][threadIdx.x]) + shared[2][threadIdx.x]) + shared[3][threadIdx.x];
    }
}

// One Gauss-Newton step's sums.  `pending` != 0: the previous launch left its per-workgroup sums in partial_prev and the
// pose they were taken at in state_in; EVERY workgroup finishes that step first (second reduction stage, solve, pose
// update: icp_finish_step, the same fixed-order arithmetic, so all arrive at the same pose) and workgroup 0 records
// it in state_out.  The kernel boundary orders the two launches -- no fence, no ticket, no workgroup waiting for the others
// across the chip's eight L2s (that hand-over cost more than the whole reduction: 24 -> 14 us per step).
__global__ __launch_bounds__(kIcpThreads) void icp_reduce_kernel(const double *__restrict__ state_in, double *__restrict__ state_out,
                                                                const float *__restrict__ partial_prev, int pending,
                                                                const float *__restrict__ vmap_curr,
                                                                const float *__restrict__ nmap_curr,
                                                                const float *__restrict__ vmap_prev,
                                                                const float *__restrict__ nmap_prev, int rows, int cols,
                                                                float fx, float fy, float cx, float cy, float dist_thresh,
                                                                float angle_thresh, float *__restrict__ partial) {
    __shared__ double pose[16];
    if (pending) {
        icp_finish_step(partial_prev, (int)gridDim.x, state_in, 1, pose, blockIdx.x == 0 ? state_out : nullptr);
    } else {
        if (threadIdx.x < 16) {
            pose[threadIdx.x] = state_in[threadIdx.x];
            if (blockIdx.x == 0) state_out[threadIdx.x] = state_in[threadIdx.x];
        }
    }
    __syncthreads();
    __shared__ float shared[4][32];
    icp_accumulate(pose, vmap_curr, nmap_curr, vmap_prev, nmap_prev, rows, cols, fx, fy, cx, cy, dist_thresh, angle_thresh,
                   partial + blockIdx.x * 32, shared);
}

// Finishes the last step of a sequence (nothing follows whose prologue would): one workgroup.
// `mirror` (pinned host memory, or null): pose, residual and inliers also go there, for the host to read after the stream's end.
__global__ __launch_bounds__(kIcpThreads) void icp_finish_kernel(const double *__restrict__ state_in, double *__restrict__ state_out,
                                                                const float *__restrict__ partial_prev, int n_blocks, int update,
                                                                double *__restrict__ mirror) {
    __shared__ double pose[16];
    icp_finish_step(partial_prev, n_blocks, state_in, update, pose, state_out);
    if (mirror && threadIdx.x == 0) {   // (thread 0 solved the step: what it has just written)
#pragma unroll
        for (int i = 0; i < 16; i++) mirror[i] = pose[i];
        mirror[16] = state_out[16];
        mirror[17] = state_out[17];
    }
}

// x = A^-1 b, 6x6 symmetric positive (semi-)definite, LDL^T with diagonal pivoting in double -- the job of
// `A_icp.cast<double>().ldlt().solve(b_icp.cast<double>())` (ICPOdometry.cpp:131).  Zero pivots give zero components.
// One lane runs this, so latency is everything: all loops are fully unrolled and the pivot exchanges are predicated
// swaps at compile-time indices, which keeps the matrices in registers instead of scratch memory.
__device__ inline void swap_if(bool c, double &a, double &b) {
    const double t = a;
    a = c ? b : a;
    b = c ? t : b;
}
// The same factorisation without pivoting: for the positive definite, reasonably conditioned normal matrix of a
// healthy ICP step it needs no row exchanges; returns false (result unused) when a pivot is not safely positive, and the
// pivoted version above/below takes over.  ~250 dependent flops instead of ~4000 predicated moves.
__device__ inline bool ldlt_solve6_unpivoted(const float *A_in, const float *b_in, double *x) {
    double A[6][6], L[6][6], D[6], Dinv[6], y[6];
    double dmax = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        y[i] = b_in[i];
#pragma unroll
        for (int j = 0; j < 6; j++) A[i][j] = A_in[i * 6 + j];
        dmax = fmax(dmax, fabs(A[i][i]));
    }
    bool ok = dmax > 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        D[k] = A[k][k];
        ok = ok && D[k] > 1.0e-9 * dmax;
        const double inv = 1.0 / D[k];
        Dinv[k] = inv;
#pragma unroll
        for (int i = k + 1; i < 6; i++) L[i][k] = A[i][k] * inv;
#pragma unroll
        for (int i = k + 1; i < 6; i++)
#pragma unroll
            for (int j = k + 1; j <= i; j++) {
                A[i][j] -= L[i][k] * D[k] * L[j][k];
                A[j][i] = A[i][j];
            }
    }
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < i; j++) y[i] -= L[i][j] * y[j];
#pragma unroll
    for (int i = 0; i < 6; i++) y[i] = y[i] * Dinv[i];   // (the pivots' reciprocals again: six divisions fewer on the critical path)
#pragma unroll
    for (int i = 5; i >= 0; i--) {
#pragma unroll
        for (int j = i + 1; j < 6; j++) y[i] -= L[j][i] * y[j];
        x[i] = y[i];
    }
    return ok;
}

__device__ inline void ldlt_solve6(const float *A_in, const float *b_in, double *x) {
    double A[6][6], L[6][6], D[6], y[6], z[6];
    int perm[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        perm[i] = i;
        y[i] = b_in[i];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            A[i][j] = A_in[i * 6 + j];
            L[i][j] = 0.0;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
        // pivot: the largest remaining diagonal entry (first one on ties)
        int p = k;
        double best = fabs(A[k][k]);
#pragma unroll
        for (int i = k + 1; i < 6; i++) {
            const double v = fabs(A[i][i]);
            if (v > best) {
                best = v;
                p = i;
            }
        }
#pragma unroll
        for (int i = k + 1; i < 6; i++) {
            const bool sw = (p == i);  // exchange rows / columns k and i of A, rows of L, the permutation, the rhs
#pragma unroll
            for (int j = 0; j < 6; j++) swap_if(sw, A[k][j], A[i][j]);
#pragma unroll
            for (int r = 0; r < 6; r++) swap_if(sw, A[r][k], A[r][i]);
#pragma unroll
            for (int j = 0; j < k; j++) swap_if(sw, L[k][j], L[i][j]);
            const int tp = perm[k];
            perm[k] = sw ? perm[i] : perm[k];
            perm[i] = sw ? tp : perm[i];
            swap_if(sw, y[k], y[i]);
        }
        D[k] = A[k][k];
        L[k][k] = 1.0;
        const bool nz = D[k] != 0.0;
#pragma unroll
        for (int i = k + 1; i < 6; i++) L[i][k] = nz ? A[i][k] / D[k] : 0.0;
#pragma unroll
        for (int i = k + 1; i < 6; i++)
#pragma unroll
            for (int j = k + 1; j < 6; j++) A[i][j] -= L[i][k] * D[k] * L[j][k];
    }
    // (y was permuted along with the rows: y = P b)
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < i; j++) y[i] -= L[i][j] * y[j];
#pragma unroll
    for (int i = 0; i < 6; i++) z[i] = (D[i] != 0.0) ? y[i] / D[i] : 0.0;
#pragma unroll
    for (int i = 5; i >= 0; i--)
#pragma unroll
        for (int j = i + 1; j < 6; j++) z[i] -= L[j][i] * z[j];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++)
            if (perm[i] == j) x[j] = z[i];
}

// Sophus::SE3d::exp(a) for a = (upsilon, omega): R = exp(hat(omega)), translation = V * upsilon; E column-major 4x4.
__device__ inline void se3_exp(const double *a, double *E) {
    const double wx = a[3], wy = a[4], wz = a[5];
    const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
    const double W[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
    double W2[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            W2[i][j] = 0;
            for (int k = 0; k < 3; k++) W2[i][j] += W[i][k] * W[k][j];
        }
    double A, B, C;  // sin th / th, (1 - cos th) / th^2, (th - sin th) / th^3
    if (th2 < 0.0625) {
        // |th| < 1/4 (every step of a converging ICP): the three entire functions by their power series in th^2, summed
        // from the smallest term (Horner, constant reciprocals: a double division costs this lane ~25 dependent instructions);
        // the first omitted terms are < 1e-19 relative.  One lane evaluates this on the
        // critical path of every iteration, and the library's double sin / cos cost several microseconds there.
        const double t = th2;
        A = 1.0 - t * (1.0 / 6.0) * (1.0 - t * (1.0 / 20.0) * (1.0 - t * (1.0 / 42.0) * (1.0 - t * (1.0 / 72.0) * (1.0 - t * (1.0 / 110.0) * (1.0 - t * (1.0 / 156.0) * (1.0 - t * (1.0 / 210.0)))))));
        B = 0.5 * (1.0 - t * (1.0 / 12.0) * (1.0 - t * (1.0 / 30.0) * (1.0 - t * (1.0 / 56.0) * (1.0 - t * (1.0 / 90.0) * (1.0 - t * (1.0 / 132.0) * (1.0 - t * (1.0 / 182.0) * (1.0 - t * (1.0 / 240.0))))))));
        C = 1.0 / 6.0 * (1.0 - t * (1.0 / 20.0) * (1.0 - t * (1.0 / 42.0) * (1.0 - t * (1.0 / 72.0) * (1.0 - t * (1.0 / 110.0) * (1.0 - t * (1.0 / 156.0) * (1.0 - t * (1.0 / 210.0) * (1.0 - t * (1.0 / 272.0))))))));
    } else {
        A = sin(th) / th;
        B = (1.0 - cos(th)) / th2;
        C = (th - sin(th)) / (th2 * th);
    }
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) E[c * 4 + r] = (r == c) ? 1.0 : 0.0;
    for (int i = 0; i < 3; i++) {
        double ti = 0;
        for (int j = 0; j < 3; j++) {
            const double I = (i == j) ? 1.0 : 0.0;
            E[j * 4 + i] = I + A * W[i][j] + B * W2[i][j];
            ti += (I + B * W[i][j] + C * W2[i][j]) * a[j];
        }
        E[12 + i] = ti;
    }
}

// Second stage of the reduction (reduceSum<29>, Cuda/estimate.cu:70-85) + the host part of estimateStep /
// getIncrementalTransformation: A, b, residual, inliers; when `update` != 0 also x = A^-1 b and T <- exp(x) * T.
// Run by all 256 threads of a workgroup; the pose after the step goes to pose_out (shared memory, for the caller's
// __syncthreads), the whole state to state_out when that is not null.  The sums were written by the previous launch -- or, COHERENT,
// by the other workgroups of THIS launch before a grid barrier (icp_persistent_kernel): then they are read past the caches that are
// not coherent across the chip's eight L2s.
template <bool COHERENT>
__device__ __forceinline__ void icp_finish_step(const float *partial, int n_blocks, const double *state_in, int update, double *pose_out,
                                       double *state_out) {
    // 29 entries x 8 groups of blocks: thread (entry, group) adds its 32 blocks in order (loads issued together), then
    // one thread per entry adds the 8 group sums in order -- a fixed tree, in double
    __shared__ double group_sum[8][32];
    __shared__ float total[32];
    const int entry = threadIdx.x & 31, group = threadIdx.x >> 5;
    // (the pose the sums were taken at: requested now, with the sums, not after them -- one memory round trip less on the
    // path every iteration waits for)
    double T[16];
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 16; i++) T[i] = state_in[i];
    }
    if (entry < 29) {
        // (The loads are unconditional -- a block past n_blocks re-reads block 0 and its value is replaced by 0 afterwards: with the
        // test in front of each load the compiler made 32 branches, each load waited for on its own: 32 dependent round trips, 8 of
        // the 9.5 us this step took, profiles/r04zz_icp_finish_phases.txt.)
        float v[32];
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int b = group * 32 + i, bb = b < n_blocks ? b : 0;
            if (COHERENT)
                v[i] = __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t *>(partial) + bb * 32 + entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            else
                v[i] = partial[bb * 32 + entry];
        }
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 32; i++) s += (double)(group * 32 + i < n_blocks ? v[i] : 0.0f);
        group_sum[group][entry] = s;
    }
    __syncthreads();
    if (threadIdx.x < 29) {
        double s = 0.0;
#pragma unroll
        for (int g = 0; g < 8; g++) s += group_sum[g][threadIdx.x];
        total[threadIdx.x] = (float)s;  // the reference hands fp32 sums to the host
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float A[36], b[6];
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            const float value = total[shift++];
            if (j == 6) b[i] = value;
            else A[j * 6 + i] = A[i * 6 + j] = value;
        }
    if (state_out) {
        state_out[16] = total[27];
        state_out[17] = total[28];
        for (int i = 0; i < 36; i++) state_out[18 + i] = A[i];
        for (int i = 0; i < 6; i++) state_out[54 + i] = b[i];
    }
    double out[16];
    for (int i = 0; i < 16; i++) out[i] = T[i];
    if (update) {
        double x[6], E[16];
        if (!ldlt_solve6_unpivoted(A, b, x)) ldlt_solve6(A, b, x);
        se3_exp(x, E);
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++) {
                double s = 0;
                for (int k = 0; k < 4; k++) s += E[k * 4 + r] * T[c * 4 + k];
                out[c * 4 + r] = s;
            }
    }
    for (int i = 0; i < 16; i++) {
        pose_out[i] = out[i];
        if (state_out) state_out[i] = out[i];
    }
}

// ---- the whole of getIncrementalTransformation in ONE launch (round 4) ---------------------------------------------------------
// The chain above costs ~20 us per iteration for 5-13 us of sums: a kernel boundary, then every workgroup fetching the 256 partial
// sums and the pose the previous launch left.  Here the 256 workgroups stay resident through all 19 iterations and meet in a grid
// barrier instead: sums -> partial[buffer] -> barrier -> every workgroup finishes the step itself (icp_finish_step, the same
// fixed-order arithmetic as the chain: the same pose bit for bit) -> next iteration at the new pose, into the other buffer.  One
// barrier per iteration is enough: a workgroup that writes buffer s for iteration i + 2 has passed the barrier of iteration i + 1,
// which every workgroup reaches only after it has read buffer s of iteration i.  The barrier is an arrival counter that only
// grows (its base is a kernel argument, so nothing has to be reset between launches).  256 workgroups of 256 threads are one per
// compute unit: all resident at once on an idle chip; beside other kernels the late ones are waited for, nothing waits for them.
struct IcpRun {
    const float *vmap_curr[3], *nmap_curr[3], *vmap_prev[3], *nmap_prev[3];
    int rows0, cols0;
    float fx, fy, cx, cy, dist_thresh, angle_thresh;
    int iterations[3];                 // per pyramid level (ICPOdometry.cpp:99-101: 10, 5, 4), coarsest level first
    const double *state_in;            // [0..15] the pose to start from
    double *state_out;                 // the state after the last step (pose, residual, inliers, A, b)
    float *partial;                    // 2 x gridDim.x x 32
    unsigned long long *arrivals;      // the barrier's counter
    unsigned long long arrivals_base;  // its value when the launch starts
    int leader;                        // 1: workgroup 0 finishes each step alone and publishes the pose (pose_pub, published)
    double *pose_pub;                  // 2 x 16
    unsigned long long *published;     // the last step whose pose is out (+ published_base)
    unsigned long long published_base;
};
__global__ __launch_bounds__(kIcpThreads) void icp_persistent_kernel(const IcpRun run) {
    __shared__ double pose[16];
    __shared__ float shared[4][32];
    if (threadIdx.x < 16) pose[threadIdx.x] = run.state_in[threadIdx.x];
    __syncthreads();
    unsigned long long step = 0;
    int buf = 0;
    int total = 0;
    for (int l = 0; l < kIcpLevels; l++) total += run.iterations[l];
    for (int level = kIcpLevels - 1; level >= 0; level--) {
        const int rows = run.rows0 >> level, cols = run.cols0 >> level;
        const float div = (float)(1 << level);
        const float fx = run.fx / div, fy = run.fy / div, cx = run.cx / div, cy = run.cy / div;
        for (int it = 0; it < run.iterations[level]; it++) {
            float *partial = run.partial + (size_t)buf * gridDim.x * 32;
            icp_accumulate(pose, run.vmap_curr[level], run.nmap_curr[level], run.vmap_prev[level], run.nmap_prev[level], rows, cols, fx, fy, cx, cy,
                           run.dist_thresh, run.angle_thresh, partial + blockIdx.x * 32, shared);
            // this workgroup's sums are out (release) ...
            __syncthreads();
            step++;
            const unsigned long long want = run.arrivals_base + step * gridDim.x;
            const bool last = (int)step == total;
            if (threadIdx.x == 0) __hip_atomic_fetch_add(run.arrivals, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (run.leader) {
                // ... workgroup 0 waits for everybody's, finishes the step alone and publishes the pose; the others wait for that word
                // (256 workgroups fetching the 30 KB of partial sums each, past the caches, cost more than the launch boundary they replace)
                double *pub = run.pose_pub + (size_t)buf * 16;
                if (blockIdx.x == 0) {
                    if (threadIdx.x == 0)
                        while (__hip_atomic_load(run.arrivals, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
                    __syncthreads();
                    icp_finish_step<true>(partial, (int)gridDim.x, pose, 1, pose, last ? run.state_out : nullptr);
                    __syncthreads();
                    if (threadIdx.x < 16) pub[threadIdx.x] = pose[threadIdx.x];
                    __syncthreads();
                    if (threadIdx.x == 0) __hip_atomic_store(run.published, run.published_base + step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    if (threadIdx.x == 0)
                        while (__hip_atomic_load(run.published, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < run.published_base + step) __builtin_amdgcn_s_sleep(1);
                    __syncthreads();
                    if (threadIdx.x < 16) {
                        const unsigned long long bits = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(pub) + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pose[threadIdx.x] = __longlong_as_double((long long)bits);
                    }
                    __syncthreads();
                }
            } else {
                // ... grid barrier: wait for everybody's (acquire), then every workgroup finishes the step itself
                if (threadIdx.x == 0)
                    while (__hip_atomic_load(run.arrivals, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(2);
                __syncthreads();
                icp_finish_step<true>(partial, (int)gridDim.x, pose, 1, pose, (last && blockIdx.x == 0) ? run.state_out : nullptr);
                __syncthreads();
            }
            buf ^= 1;
        }
    }
}

static void free_icp(tsdf_icp *f) {
    for (int i = 0; i < kIcpLevels; i++) {
        if (f->depth[i]) (void)hipFree(f->depth[i]);
        if (f->vmap_prev[i]) (void)hipFree(f->vmap_prev[i]);
        if (f->nmap_prev[i]) (void)hipFree(f->nmap_prev[i]);
        if (f->vmap_curr[i]) (void)hipFree(f->vmap_curr[i]);
        if (f->nmap_curr[i]) (void)hipFree(f->nmap_curr[i]);
    }
    if (f->upload) (void)hipFree(f->upload);
    if (f->partial) (void)hipFree(f->partial);
    if (f->state) (void)hipFree(f->state);
    if (f->arrivals) (void)hipFree(f->arrivals);
    if (f->host_io) (void)hipHostFree(f->host_io);
    delete f;
}

// pyramid + vertex / normal maps of one depth image (device pointer; f->depth[0] gets a copy as a by-product of the maps launch)
static int build_maps(tsdf_icp *f, const uint16_t *depth0, float **vmaps, float **nmaps, float depth_cutoff) {
    for (int i = 1; i < kIcpLevels; i++) {
        const int src_rows = f->height >> (i - 1), src_cols = f->width >> (i - 1);
        dim3 grid((src_cols / 2 + 63) / 64, (src_rows / 2 + 3) / 4);
        hipLaunchKernelGGL(icp_pyr_down_kernel, grid, dim3(256), 0, f->stream, i == 1 ? depth0 : (const uint16_t *)f->depth[i - 1], src_rows, src_cols,
                           f->depth[i]);
    }
    static_assert(kIcpLevels == 3, "icp_maps_kernel carries three levels");
    IcpLevelMaps m;
    m.depth[0] = depth0;
    for (int i = 1; i < kIcpLevels; i++) m.depth[i] = f->depth[i];
    for (int i = 0; i < kIcpLevels; i++) {
        m.vmap[i] = vmaps[i];
        m.nmap[i] = nmaps[i];
    }
    m.depth0_copy = f->depth[0];
    m.rows0 = f->height;
    m.cols0 = f->width;
    m.fx = f->fx; m.fy = f->fy; m.cx = f->cx; m.cy = f->cy;
    m.depth_cutoff = depth_cutoff;
    hipLaunchKernelGGL(icp_maps_kernel, dim3((f->width + 63) / 64, (f->height + 3) / 4, kIcpLevels), dim3(256), 0, f->stream, m);
    TSDF_HIP(hipGetLastError(), "ICP map kernels failed");
    return TSDF_OK;
}

// Launches the sums of one step at `level`, taken at the pose the pending step (if any) leads to.
// `start` (pending == 0 only): where the pose to start from lies, if not in the device state (the host's pinned copy).
static void launch_step(tsdf_icp *f, int level, int pending, const double *start = nullptr) {
    const int rows = f->height >> level, cols = f->width >> level, div = 1 << level;
    const int in = f->side, out = 1 - f->side;
    hipLaunchKernelGGL(icp_reduce_kernel, dim3(kIcpBlocks), dim3(kIcpThreads), 0, f->stream, start ? start : f->state + in * kIcpStateDoubles,
                       f->state + out * kIcpStateDoubles, f->partial + (size_t)in * kIcpBlocks * 32, pending, f->vmap_curr[level],
                       f->nmap_curr[level], f->vmap_prev[level], f->nmap_prev[level], rows, cols, f->fx / div, f->fy / div,
                       f->cx / div, f->cy / div, f->dist_thresh, f->angle_thresh, f->partial + (size_t)out * kIcpBlocks * 32);
    f->side = out;
}

// Finishes the step whose sums the last launch_step left (with or without the pose update).
static void launch_finish(tsdf_icp *f, int update, double *mirror = nullptr) {
    const int in = f->side, out = 1 - f->side;
    hipLaunchKernelGGL(icp_finish_kernel, dim3(1), dim3(kIcpThreads), 0, f->stream, f->state + in * kIcpStateDoubles,
                       f->state + out * kIcpStateDoubles, f->partial + (size_t)in * kIcpBlocks * 32, kIcpBlocks, update, mirror);
    f->side = out;
}

}  // namespace tsdf

using namespace tsdf;

extern "C" {

int tsdf_icp_create(int width, int height, float cx, float cy, float fx, float fy, float dist_thresh, float angle_thresh,
                    tsdf_icp **out) {
    TSDF_REQUIRE(out, "tsdf_icp_create: null argument");
    *out = nullptr;
    TSDF_REQUIRE(width >= 4 && height >= 4 && width <= 65535 && height <= 65535, "tsdf_icp_create: bad image size");
    tsdf_icp *f = new (std::nothrow) tsdf_icp();
    TSDF_REQUIRE(f, "out of host memory");
    std::memset(f, 0, sizeof(*f));
    f->width = width; f->height = height;
    f->cx = cx; f->cy = cy; f->fx = fx; f->fy = fy;
    f->dist_thresh = dist_thresh; f->angle_thresh = angle_thresh;
    hipError_t e = hipGetDevice(&f->device);
    for (int i = 0; i < kIcpLevels && e == hipSuccess; i++) {
        const size_t px = (size_t)(height >> i) * (width >> i);
        e = hipMalloc((void **)&f->depth[i], px * sizeof(uint16_t));
        float **maps[4] = {&f->vmap_prev[i], &f->nmap_prev[i], &f->vmap_curr[i], &f->nmap_curr[i]};
        for (int m = 0; m < 4 && e == hipSuccess; m++) {
            e = hipMalloc((void **)maps[m], px * 3 * sizeof(float));
            // the reference's maps start uninitialised and invalid pixels only ever get component 0 written: define the rest
            if (e == hipSuccess) e = hipMemset(*maps[m], 0, px * 3 * sizeof(float));
        }
    }
    if (e == hipSuccess) e = hipMalloc((void **)&f->partial, (size_t)2 * kIcpBlocks * 32 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&f->state, 2 * kIcpStateDoubles * sizeof(double));
    if (e == hipSuccess) e = hipMemset(f->state, 0, 2 * kIcpStateDoubles * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&f->arrivals, (2 + 32) * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(f->arrivals, 0, (2 + 32) * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc((void **)&f->host_io, (16 + 18) * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent);   // (fine-grained: the device reads and writes it past its caches)
    if (e == hipSuccess) std::memset(f->host_io, 0, (16 + 18) * sizeof(double));
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&f->host_io_dev, f->host_io, 0);
    if (e != hipSuccess) {
        free_icp(f);
        return hip_fail(e, "ICP alloc failed");
    }
    *out = f;
    return TSDF_OK;
}

void tsdf_icp_destroy(tsdf_icp *f) {
    if (f) free_icp(f);
}

int tsdf_icp_stream(const tsdf_icp *f, void **hip_stream) {
    TSDF_REQUIRE(f && hip_stream, "null argument");
    *hip_stream = f->stream;
    return TSDF_OK;
}

int tsdf_icp_set_stream(tsdf_icp *f, void *hip_stream) {
    TSDF_REQUIRE(f, "null ICP handle");
    f->stream = (hipStream_t)hip_stream;
    return TSDF_OK;
}

int tsdf_icp_init_device(tsdf_icp *f, int model, const uint16_t *device_depth, float depth_cutoff) {
    TSDF_REQUIRE(f && device_depth, "tsdf_icp_init: null argument");
    TSDF_REQUIRE(device_depth != f->depth[0], "tsdf_icp_init: the image must not be the pyramid's own level 0");
    return model ? build_maps(f, device_depth, f->vmap_prev, f->nmap_prev, depth_cutoff) : build_maps(f, device_depth, f->vmap_curr, f->nmap_curr, depth_cutoff);
}

int tsdf_icp_init(tsdf_icp *f, int model, const uint16_t *host_depth, float depth_cutoff) {
    TSDF_REQUIRE(f && host_depth, "tsdf_icp_init: null argument");
    if (!f->upload) TSDF_HIP(hipMalloc((void **)&f->upload, (size_t)f->width * f->height * sizeof(uint16_t)), "ICP upload buffer");
    TSDF_HIP(hipMemcpyAsync(f->upload, host_depth, (size_t)f->width * f->height * sizeof(uint16_t), hipMemcpyHostToDevice,
                            f->stream), "ICP depth upload");
    int rc = model ? build_maps(f, f->upload, f->vmap_prev, f->nmap_prev, depth_cutoff) : build_maps(f, f->upload, f->vmap_curr, f->nmap_curr, depth_cutoff);
    if (rc != TSDF_OK) return rc;
    TSDF_HIP(hipStreamSynchronize(f->stream), "ICP init");  // (the reference synchronises here too)
    return TSDF_OK;
}

int tsdf_icp_get_map(const tsdf_icp *f, int which, int level, float *host_map) {
    TSDF_REQUIRE(f && host_map && level >= 0 && level < kIcpLevels && which >= 0 && which < 4, "tsdf_icp_get_map: bad argument");
    const float *src[4] = {f->vmap_prev[level], f->nmap_prev[level], f->vmap_curr[level], f->nmap_curr[level]};
    const size_t bytes = (size_t)(f->height >> level) * (f->width >> level) * 3 * sizeof(float);
    TSDF_HIP(hipMemcpyAsync(host_map, src[which], bytes, hipMemcpyDeviceToHost, f->stream), "ICP map download");
    TSDF_HIP(hipStreamSynchronize(f->stream), "ICP map download");
    return TSDF_OK;
}

int tsdf_icp_get_depth_level(const tsdf_icp *f, int level, uint16_t *host_depth) {
    TSDF_REQUIRE(f && host_depth && level >= 0 && level < kIcpLevels, "tsdf_icp_get_depth_level: bad argument");
    const size_t bytes = (size_t)(f->height >> level) * (f->width >> level) * sizeof(uint16_t);
    TSDF_HIP(hipMemcpyAsync(host_depth, f->depth[level], bytes, hipMemcpyDeviceToHost, f->stream), "ICP depth download");
    TSDF_HIP(hipStreamSynchronize(f->stream), "ICP depth download");
    return TSDF_OK;
}

int tsdf_icp_estimate_step(tsdf_icp *f, int level, const float R[9], const float t[3], float A[36], float b[6],
                           float residual_inliers[2]) {
    TSDF_REQUIRE(f && R && t && A && b && residual_inliers && level >= 0 && level < kIcpLevels, "tsdf_icp_estimate_step: bad argument");
    double T[16] = {0};
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) T[c * 4 + r] = R[c * 3 + r];
    for (int r = 0; r < 3; r++) T[12 + r] = t[r];
    T[15] = 1.0;
    TSDF_HIP(hipMemcpyAsync(f->state + f->side * kIcpStateDoubles, T, sizeof(T), hipMemcpyHostToDevice, f->stream), "ICP pose upload");
    launch_step(f, level, 0);
    launch_finish(f, 0);
    TSDF_HIP(hipGetLastError(), "ICP estimate kernels failed");
    double out[kIcpStateDoubles];
    TSDF_HIP(hipMemcpyAsync(out, f->state + f->side * kIcpStateDoubles, sizeof(out), hipMemcpyDeviceToHost, f->stream), "ICP result download");
    TSDF_HIP(hipStreamSynchronize(f->stream), "ICP estimate");
    residual_inliers[0] = (float)out[16];
    residual_inliers[1] = (float)out[17];
    for (int i = 0; i < 36; i++) A[i] = (float)out[18 + i];
    for (int i = 0; i < 6; i++) b[i] = (float)out[54 + i];
    return TSDF_OK;
}

int tsdf_icp_get_incremental_transformation(tsdf_icp *f, double T_prev_curr[16], float *last_error, float *last_inliers) {
    TSDF_REQUIRE(f && T_prev_curr, "tsdf_icp_get_incremental_transformation: null argument");
    const int iterations[kIcpLevels] = {10, 5, 4};  // ICPOdometry.cpp:99-101
    const bool persistent = tuning().icp_persistent != 0;
    // The pose goes to the device and the result comes back through pinned host memory that the first and the last launch of the chain
    // address directly (every call ends with the stream idle, so the host may write it here): the two 128-byte copies were launches of
    // their own on the stream, 4-5 us each plus their boundaries, on the path every tracked frame waits for.
    if (persistent)
        TSDF_HIP(hipMemcpyAsync(f->state + f->side * kIcpStateDoubles, T_prev_curr, 16 * sizeof(double), hipMemcpyHostToDevice, f->stream),
                 "ICP pose upload");
    else
        std::memcpy(f->host_io, T_prev_curr, 16 * sizeof(double));
    if (persistent) {
        // one launch: the workgroups stay through all 19 iterations and meet in a grid barrier (icp_persistent_kernel)
        IcpRun run;
        for (int i = 0; i < kIcpLevels; i++) {
            run.vmap_curr[i] = f->vmap_curr[i]; run.nmap_curr[i] = f->nmap_curr[i];
            run.vmap_prev[i] = f->vmap_prev[i]; run.nmap_prev[i] = f->nmap_prev[i];
            run.iterations[i] = iterations[i];
        }
        run.rows0 = f->height; run.cols0 = f->width;
        run.fx = f->fx; run.fy = f->fy; run.cx = f->cx; run.cy = f->cy;
        run.dist_thresh = f->dist_thresh; run.angle_thresh = f->angle_thresh;
        run.state_in = f->state + f->side * kIcpStateDoubles;
        run.state_out = f->state + (1 - f->side) * kIcpStateDoubles;
        run.partial = f->partial;
        run.arrivals = f->arrivals;
        run.arrivals_base = f->arrivals_base;
        f->arrivals_base += (unsigned long long)kIcpBlocks * (iterations[0] + iterations[1] + iterations[2]);
        run.leader = tuning().icp_persistent == 2 ? 1 : 0;
        run.pose_pub = reinterpret_cast<double *>(f->arrivals + 2);
        run.published = f->arrivals + 1;
        run.published_base = f->published_base;
        f->published_base += (unsigned long long)(iterations[0] + iterations[1] + iterations[2]);
        hipLaunchKernelGGL(icp_persistent_kernel, dim3(kIcpBlocks), dim3(kIcpThreads), 0, f->stream, run);
        f->side = 1 - f->side;
    } else {
        int pending = 0;   // every launch finishes the step before it, the last step gets a launch of its own
        for (int i = kIcpLevels - 1; i >= 0; i--)
            for (int j = 0; j < iterations[i]; j++) {
                launch_step(f, i, pending, pending ? nullptr : f->host_io_dev);
                pending = 1;
            }
        launch_finish(f, 1, f->host_io_dev + 16);
    }
    TSDF_HIP(hipGetLastError(), "ICP kernels failed");
    double out[18];
    if (persistent) TSDF_HIP(hipMemcpyAsync(out, f->state + f->side * kIcpStateDoubles, sizeof(out), hipMemcpyDeviceToHost, f->stream), "ICP pose download");
    TSDF_HIP(hipStreamSynchronize(f->stream), "ICP");   // (polling a ticket in the pinned block instead: 0.5-1 %, not kept)
    if (!persistent) std::memcpy(out, f->host_io + 16, sizeof(out));
    std::memcpy(T_prev_curr, out, 16 * sizeof(double));
    // lastError = sqrt(residual) / inliers, lastInliers = inliers of the last iteration (ICPOdometry.cpp:127-128)
    if (last_error) *last_error = sqrtf((float)out[16]) / (float)out[17];
    if (last_inliers) *last_inliers = (float)out[17];
    return TSDF_OK;
}

}  // extern "C"
