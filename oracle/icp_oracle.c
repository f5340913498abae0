/*
 * icp_oracle.c -- CPU oracle for the ICP tracking row: a plain-C restatement of the reference's
 * third_party/ICP_CUDA (pyrdown.cu, estimate.cu, ICPOdometry.cpp).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE (see tsdf_oracle.h).
 *
 * PARITY UNPINNED: the reference holds no tests or golden vectors for ICP, and its sources need nvcc, Eigen and Sophus
 * (none in this image; Sophus is not vendored by the reference either: `#include <sophus/se3.hpp>`, no version pinned).
 * What is restated from source: the per-pixel arithmetic of the four kernels in the reference's operation order with
 * no fused multiply-adds (nvcc's default contraction would differ in the last bits of the cross products / matrix
 * products), and the published closed forms of Eigen's LDLT solve (any accurate solver of the 6x6 SPD system) and of
 * Sophus::SE3d::exp (Rodrigues + the V matrix).  The 29 sums of the reduction are accumulated in double here: the
 * reference's own fp32 sums depend on its launch configuration (threads x blocks chosen by the caller), so a GPU
 * result can only be compared within a tolerance (tests state it).
 *
 * Maps use the reference's planar layout: a map of a rows x cols level is 3*rows x cols floats, component c of pixel
 * (x, y) at [(y + c*rows) * cols + x]; an invalid pixel holds the NaN pattern 0x7fffffff in component 0 only.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float orc_nan_sentinel(void) {
    union { uint32_t u; float f; } v;
    v.u = 0x7fffffffu;
    return v.f;
}

/* pyrDownGaussKernel (third_party/ICP_CUDA/Cuda/pyrdown.cu:41-78), sigma_color = 30 (:87) */
void orc_icp_pyr_down(const uint16_t *src, int src_rows, int src_cols, uint16_t *dst) {
    const int rows = src_rows / 2, cols = src_cols / 2;
    const float sigma_color = 30;
    const float weights[3] = {0.375f, 0.25f, 0.0625f};
    for (int y = 0; y < rows; y++) {
        for (int x = 0; x < cols; x++) {
            const int D = 5;
            int center = src[(size_t)(2 * y) * src_cols + 2 * x];
            int x_mi = (0 > 2 * x - D / 2 ? 0 : 2 * x - D / 2) - 2 * x;
            int y_mi = (0 > 2 * y - D / 2 ? 0 : 2 * y - D / 2) - 2 * y;
            int x_ma = (src_cols < 2 * x - D / 2 + D ? src_cols : 2 * x - D / 2 + D) - 2 * x;
            int y_ma = (src_rows < 2 * y - D / 2 + D ? src_rows : 2 * y - D / 2 + D) - 2 * y;
            float sum = 0, wall = 0;
            for (int yi = y_mi; yi < y_ma; ++yi)
                for (int xi = x_mi; xi < x_ma; ++xi) {
                    int val = src[(size_t)(2 * y + yi) * src_cols + (2 * x + xi)];
                    if (abs(val - center) < 3 * sigma_color) {
                        sum += val * weights[abs(xi)] * weights[abs(yi)];
                        wall += weights[abs(xi)] * weights[abs(yi)];
                    }
                }
            dst[(size_t)y * cols + x] = (uint16_t)(int)(sum / wall);
        }
    }
}

/* computeVmapKernel (pyrdown.cu:93-117); fx_inv = 1.f / fx is formed by createVMap (:131) */
void orc_icp_vmap(const uint16_t *depth, int rows, int cols, float fx, float fy, float cx, float cy, float depth_cutoff,
                  float *vmap) {
    const float fx_inv = 1.f / fx, fy_inv = 1.f / fy;
    for (int v = 0; v < rows; v++)
        for (int u = 0; u < cols; u++) {
            float z = depth[(size_t)v * cols + u] / 1000.f;
            if (z != 0 && z < depth_cutoff) {
                float vx = z * (u - cx) * fx_inv;
                float vy = z * (v - cy) * fy_inv;
                vmap[(size_t)v * cols + u] = vx;
                vmap[(size_t)(v + rows) * cols + u] = vy;
                vmap[(size_t)(v + rows * 2) * cols + u] = z;
            } else {
                vmap[(size_t)v * cols + u] = orc_nan_sentinel();
            }
        }
}

/* computeNmapKernel (pyrdown.cu:135-172): (v01 - v00) x (v10 - v00), normalised (Eigen: n / sqrt(n.n) when n.n > 0) */
void orc_icp_nmap(const float *vmap, int rows, int cols, float *nmap) {
    for (int v = 0; v < rows; v++)
        for (int u = 0; u < cols; u++) {
            if (u == cols - 1 || v == rows - 1) {
                nmap[(size_t)v * cols + u] = orc_nan_sentinel();
                continue;
            }
            float a0 = vmap[(size_t)v * cols + u], b0 = vmap[(size_t)v * cols + u + 1], c0 = vmap[(size_t)(v + 1) * cols + u];
            if (!isnan(a0) && !isnan(b0) && !isnan(c0)) {
                float a1 = vmap[(size_t)(v + rows) * cols + u], b1 = vmap[(size_t)(v + rows) * cols + u + 1],
                      c1 = vmap[(size_t)(v + 1 + rows) * cols + u];
                float a2 = vmap[(size_t)(v + 2 * rows) * cols + u], b2 = vmap[(size_t)(v + 2 * rows) * cols + u + 1],
                      c2 = vmap[(size_t)(v + 1 + 2 * rows) * cols + u];
                float px = b0 - a0, py = b1 - a1, pz = b2 - a2;   /* v01 - v00 */
                float qx = c0 - a0, qy = c1 - a1, qz = c2 - a2;   /* v10 - v00 */
                float rx = py * qz - pz * qy;
                float ry = pz * qx - px * qz;
                float rz = px * qy - py * qx;
                float z = rx * rx + ry * ry + rz * rz;
                if (z > 0.0f) {
                    float l = sqrtf(z);
                    rx = rx / l;
                    ry = ry / l;
                    rz = rz / l;
                }
                nmap[(size_t)v * cols + u] = rx;
                nmap[(size_t)(v + rows) * cols + u] = ry;
                nmap[(size_t)(v + 2 * rows) * cols + u] = rz;
            } else {
                nmap[(size_t)v * cols + u] = orc_nan_sentinel();
            }
        }
}

/* __float2int_rn: round to nearest even, saturating, NaN -> 0 */
static int orc_float2int_rn(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)nearbyintf(f);   /* default rounding mode: to nearest even */
}

/* Reduction::operator() + the unpacking of estimateStep (estimate.cu:87-209, 262-281).
 * R: 3x3 column-major (Eigen's data()), t: 3.  A: 6x6 (symmetric, either major), b: 6, residual_inliers: 2.
 * The 29 sums are accumulated in double in pixel order (see the header); sums29 (optional) receives them. */
void orc_icp_step(const float *R, const float *t, const float *vmap_curr, const float *nmap_curr, const float *vmap_prev,
                  const float *nmap_prev, int rows, int cols, float fx, float fy, float cx, float cy, float dist_thresh,
                  float angle_thresh, float *A, float *b, float *residual_inliers, double *sums29) {
    double sum[29];
    for (int i = 0; i < 29; i++) sum[i] = 0.0;
    const int N = rows * cols;
    for (int i = 0; i < N; i++) {
        const int y = i / cols, x = i - y * cols;
        const float v0 = vmap_curr[(size_t)y * cols + x], v1 = vmap_curr[(size_t)(y + rows) * cols + x],
                    v2 = vmap_curr[(size_t)(y + 2 * rows) * cols + x];
        /* R * v + t, coefficient-wise left to right */
        const float p0 = ((R[0] * v0 + R[3] * v1) + R[6] * v2) + t[0];
        const float p1 = ((R[1] * v0 + R[4] * v1) + R[7] * v2) + t[1];
        const float p2 = ((R[2] * v0 + R[5] * v1) + R[8] * v2) + t[2];
        const int px = orc_float2int_rn(p0 * fx / p2 + cx);
        const int py = orc_float2int_rn(p1 * fy / p2 + cy);
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
            /* |n_curr_in_prev x n_prev| and |v_prev - v_curr_in_prev| */
            const float c0 = m1 * q2 - m2 * q1, c1 = m2 * q0 - m0 * q2, c2 = m0 * q1 - m1 * q0;
            const float sine = sqrtf((c0 * c0 + c1 * c1) + c2 * c2);
            const float d0 = w0 - p0, d1 = w1 - p1, d2 = w2 - p2;
            const float dist = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
            if (sine < angle_thresh && dist < dist_thresh && !isnan(n0) && !isnan(q0)) {
                float row[7];
                row[0] = q0;
                row[1] = q1;
                row[2] = q2;
                row[3] = p1 * q2 - p2 * q1;   /* v_curr_in_prev x n_prev */
                row[4] = p2 * q0 - p0 * q2;
                row[5] = p0 * q1 - p1 * q0;
                row[6] = (q0 * d0 + q1 * d1) + q2 * d2;
                int s = 0;
                for (int o = 0; o < 7; o++)
                    for (int in = o; in < 7; in++) sum[s++] += (double)(row[o] * row[in]);
                sum[28] += 1.0;
            }
        }
    }
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            float value = (float)sum[shift++];
            if (j == 6) b[i] = value;
            else A[j * 6 + i] = A[i * 6 + j] = value;
        }
    residual_inliers[0] = (float)sum[27];
    residual_inliers[1] = (float)sum[28];
    if (sums29) memcpy(sums29, sum, sizeof(sum));
}

/* x = A^-1 b for a symmetric positive (semi-)definite 6x6 system, in double: what
 * `A_icp.cast<double>().ldlt().solve(b_icp.cast<double>())` computes (ICPOdometry.cpp:131).  LDL^T with diagonal
 * pivoting, zero pivots give zero components (Eigen's solve does the same for a singular D). */
void orc_ldlt_solve6(const float *A_in, const float *b_in, double *x) {
    double A[6][6], b[6];
    int perm[6];
    for (int i = 0; i < 6; i++) {
        perm[i] = i;
        b[i] = b_in[i];
        for (int j = 0; j < 6; j++) A[i][j] = A_in[i * 6 + j];
    }
    double L[6][6], D[6];
    memset(L, 0, sizeof(L));
    for (int k = 0; k < 6; k++) {
        /* pivot: largest remaining diagonal entry */
        int p = k;
        for (int i = k + 1; i < 6; i++)
            if (fabs(A[i][i]) > fabs(A[p][p])) p = i;
        if (p != k) {
            for (int j = 0; j < 6; j++) { double s = A[k][j]; A[k][j] = A[p][j]; A[p][j] = s; }
            for (int i = 0; i < 6; i++) { double s = A[i][k]; A[i][k] = A[i][p]; A[i][p] = s; }
            for (int j = 0; j < k; j++) { double s = L[k][j]; L[k][j] = L[p][j]; L[p][j] = s; }
            int s = perm[k]; perm[k] = perm[p]; perm[p] = s;
        }
        D[k] = A[k][k];
        L[k][k] = 1.0;
        if (D[k] != 0.0) {
            for (int i = k + 1; i < 6; i++) L[i][k] = A[i][k] / D[k];
            for (int i = k + 1; i < 6; i++)
                for (int j = k + 1; j < 6; j++) A[i][j] -= L[i][k] * D[k] * L[j][k];
        }
    }
    double y[6], z[6];
    for (int i = 0; i < 6; i++) {
        y[i] = b[perm[i]];
        for (int j = 0; j < i; j++) y[i] -= L[i][j] * y[j];
    }
    for (int i = 0; i < 6; i++) z[i] = (D[i] != 0.0) ? y[i] / D[i] : 0.0;
    for (int i = 5; i >= 0; i--)
        for (int j = i + 1; j < 6; j++) z[i] -= L[j][i] * z[j];
    for (int i = 0; i < 6; i++) x[perm[i]] = z[i];
}

/* Sophus::SE3d::exp(a), a = (upsilon, omega): rotation = exp(hat(omega)) (Rodrigues), translation = V * upsilon with
 * V = I + (1 - cos th)/th^2 W + (th - sin th)/th^3 W^2, W = hat(omega), V = R for th -> 0.  T: 4x4 column-major. */
void orc_se3_exp(const double *a, double *T) {
    const double wx = a[3], wy = a[4], wz = a[5];
    const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
    double W[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}}, W2[3][3], R[3][3], V[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            W2[i][j] = 0;
            for (int k = 0; k < 3; k++) W2[i][j] += W[i][k] * W[k][j];
        }
    double A, B, C;   /* sin th / th, (1 - cos th) / th^2, (th - sin th) / th^3 */
    if (th < 1e-10) {
        A = 1.0 - th2 / 6.0;
        B = 0.5 - th2 / 24.0;
        C = 1.0 / 6.0 - th2 / 120.0;
    } else {
        A = sin(th) / th;
        B = (1.0 - cos(th)) / th2;
        C = (th - sin(th)) / (th2 * th);
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            const double I = (i == j) ? 1.0 : 0.0;
            R[i][j] = I + A * W[i][j] + B * W2[i][j];
            V[i][j] = I + B * W[i][j] + C * W2[i][j];
        }
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) T[c * 4 + r] = (r == c) ? 1.0 : 0.0;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[j * 4 + i] = R[i][j];
        T[12 + i] = V[i][0] * a[0] + V[i][1] * a[1] + V[i][2] * a[2];
    }
}

/* 4x4 column-major product C = A * B (double) */
void orc_mat4d_mul(const double *A, const double *B, double *C) {
    double out[16];
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += A[k * 4 + r] * B[c * 4 + k];
            out[c * 4 + r] = s;
        }
    memcpy(C, out, sizeof(out));
}

/* ICPOdometry::initICP / initICPModel + getIncrementalTransformation (ICPOdometry.cpp:69-136): three pyramid levels,
 * 10/5/4 iterations from the coarsest level down, T <- exp(update) * T.  depth_curr feeds initICP, depth_model
 * initICPModel.  T: 4x4 column-major double, in/out.  last_error / last_inliers as the class members. */
void orc_icp_incremental_transformation(const uint16_t *depth_curr, const uint16_t *depth_model, int width, int height,
                                        float cx, float cy, float fx, float fy, float dist_thresh, float angle_thresh,
                                        float depth_cutoff, double *T, float *last_error, float *last_inliers) {
    enum { NUM_PYRS = 3 };
    const int iterations[NUM_PYRS] = {10, 5, 4};
    uint16_t *dc[NUM_PYRS], *dm[NUM_PYRS];
    float *vc[NUM_PYRS], *nc[NUM_PYRS], *vp[NUM_PYRS], *np_[NUM_PYRS];
    for (int i = 0; i < NUM_PYRS; i++) {
        const int rows = height >> i, cols = width >> i;
        dc[i] = (uint16_t *)malloc((size_t)rows * cols * 2);
        dm[i] = (uint16_t *)malloc((size_t)rows * cols * 2);
        /* (the reference's maps start uninitialised; zero here so that the never-written components are defined) */
        vc[i] = (float *)calloc((size_t)rows * cols * 3, 4);
        nc[i] = (float *)calloc((size_t)rows * cols * 3, 4);
        vp[i] = (float *)calloc((size_t)rows * cols * 3, 4);
        np_[i] = (float *)calloc((size_t)rows * cols * 3, 4);
    }
    memcpy(dc[0], depth_curr, (size_t)width * height * 2);
    memcpy(dm[0], depth_model, (size_t)width * height * 2);
    for (int i = 1; i < NUM_PYRS; i++) {
        orc_icp_pyr_down(dc[i - 1], height >> (i - 1), width >> (i - 1), dc[i]);
        orc_icp_pyr_down(dm[i - 1], height >> (i - 1), width >> (i - 1), dm[i]);
    }
    for (int i = 0; i < NUM_PYRS; i++) {
        const int rows = height >> i, cols = width >> i, div = 1 << i;
        /* Intr::operator()(level): every intrinsic divided by 2^level (internal.h:63-67) */
        orc_icp_vmap(dc[i], rows, cols, fx / div, fy / div, cx / div, cy / div, depth_cutoff, vc[i]);
        orc_icp_nmap(vc[i], rows, cols, nc[i]);
        orc_icp_vmap(dm[i], rows, cols, fx / div, fy / div, cx / div, cy / div, depth_cutoff, vp[i]);
        orc_icp_nmap(vp[i], rows, cols, np_[i]);
    }
    for (int i = NUM_PYRS - 1; i >= 0; i--) {
        const int rows = height >> i, cols = width >> i, div = 1 << i;
        for (int j = 0; j < iterations[i]; j++) {
            float R[9], t[3], A[36], b[6], ri[2];
            for (int c = 0; c < 3; c++)
                for (int r = 0; r < 3; r++) R[c * 3 + r] = (float)T[c * 4 + r];
            for (int r = 0; r < 3; r++) t[r] = (float)T[12 + r];
            orc_icp_step(R, t, vc[i], nc[i], vp[i], np_[i], rows, cols, fx / div, fy / div, cx / div, cy / div, dist_thresh,
                         angle_thresh, A, b, ri, NULL);
            if (last_error) *last_error = sqrtf(ri[0]) / ri[1];
            if (last_inliers) *last_inliers = ri[1];
            double update[6], E[16];
            orc_ldlt_solve6(A, b, update);
            orc_se3_exp(update, E);
            orc_mat4d_mul(E, T, T);
        }
    }
    for (int i = 0; i < NUM_PYRS; i++) {
        free(dc[i]); free(dm[i]); free(vc[i]); free(nc[i]); free(vp[i]); free(np_[i]);
    }
}

/* ---- mesh deformation: deformation_kernel + get_trilinear_elements + rotate
 * (src/TSDF/TSDFVolume.cu:101-263).  nodes: X*Y*Z x {translation xyz, rotation xyz} or NULL for the regular grid that
 * clear() leaves (voxel centre + offset_at_clear).  The two cases the reference leaves undefined follow the product's
 * definition: points outside the volume are unchanged, neighbour indices past the array are clamped to the last node.
 * PARITY UNPINNED (no reference tests; cosf / sinf are the host libm's here and in the product). */
void orc_deform_points(const uint32_t dims[3], const float vs[3], const float offset[3], const float offset_at_clear[3],
                       const float *nodes, const float global_rotation[3], const float global_translation[3], int num_points,
                       float *points) {
    const float eps = 0.001f;
    const float c1 = cosf(global_rotation[0]), c2 = cosf(global_rotation[1]), c3 = cosf(global_rotation[2]);
    const float s1 = sinf(global_rotation[0]), s2 = sinf(global_rotation[1]), s3 = sinf(global_rotation[2]);
    const long long X = dims[0], Y = dims[1], Z = dims[2], n_nodes = X * Y * Z;
    for (int p = 0; p < num_points; p++) {
        float pt[3], adj[3];
        int vox[3], lower[3];
        float uvw[3];
        int inside = 1;
        for (int a = 0; a < 3; a++) {
            pt[a] = points[p * 3 + a] - offset[a];
            const float mx = dims[a] * vs[a];
            adj[a] = pt[a];
            if ((pt[a] > mx) && (pt[a] - mx < eps)) adj[a] = mx - eps;
            if (pt[a] < -eps) adj[a] = 0.0f;
            const float f = floorf(adj[a] / vs[a]);
            vox[a] = (f != f) ? 0 : (f >= 2147483648.0f ? 2147483647 : (f <= -2147483648.0f ? (-2147483647 - 1) : (int)f));
            if (!(vox[a] >= 0 && (uint32_t)vox[a] < dims[a])) inside = 0;
        }
        if (!inside) continue;
        for (int a = 0; a < 3; a++) {
            const float centre = (vox[a] + 0.5f) * vs[a] + 0.0f;
            lower[a] = (adj[a] < centre) ? vox[a] - 1 : vox[a];
            if (lower[a] < 0) lower[a] = 0;
            const float lc = (lower[a] + 0.5f) * vs[a] + 0.0f;
            uvw[a] = (adj[a] - lc) / vs[a];
        }
        const float u = uvw[0], v = uvw[1], w = uvw[2];
        const long long dx = 1, dy = X, dz = X * Y;
        long long ind[8];
        ind[0] = lower[0] + (lower[1] * dy) + (lower[2] * dz);
        ind[1] = ind[0] + dx; ind[2] = ind[1] + dz; ind[3] = ind[0] + dz;
        ind[4] = ind[0] + dy; ind[5] = ind[1] + dy; ind[6] = ind[2] + dy; ind[7] = ind[3] + dy;
        float co[8];
        co[0] = (1 - u) * (1 - v) * (1 - w);
        co[1] = u * (1 - v) * (1 - w);
        co[2] = u * (1 - v) * w;
        co[3] = (1 - u) * (1 - v) * w;
        co[4] = (1 - u) * v * (1 - w);
        co[5] = u * v * (1 - w);
        co[6] = (1 - u) * v * w;
        co[7] = u * v * w;
        float d[3] = {0.0f, 0.0f, 0.0f};
        for (int k = 0; k < 8; k++) {
            long long i = ind[k] < n_nodes - 1 ? ind[k] : n_nodes - 1;
            float t[3];
            if (nodes) {
                t[0] = nodes[i * 6 + 0]; t[1] = nodes[i * 6 + 1]; t[2] = nodes[i * 6 + 2];
            } else {
                const int nz = (int)(i / dz), ny = (int)((i - nz * dz) / dy), nx = (int)(i - nz * dz - ny * dy);
                t[0] = ((nx + 0.5f) * vs[0]) + offset_at_clear[0];
                t[1] = ((ny + 0.5f) * vs[1]) + offset_at_clear[1];
                t[2] = ((nz + 0.5f) * vs[2]) + offset_at_clear[2];
            }
            for (int a = 0; a < 3; a++) d[a] = (t[a] * co[k]) + d[a];
        }
        const float rx = (c2 * c3) * d[0] - (c2 * s3) * d[1] + s2 * d[2];
        const float ry = (c1 * s3 + s1 * s2 * c3) * d[0] + (c1 * c3 - s1 * s2 * s3) * d[1] - (s1 * c2) * d[2];
        const float rz = (s1 * s3 - c1 * s2 * c3) * d[0] + (s1 * c3 + c1 * s2 * s3) * d[1] + (c1 * c2) * d[2];
        points[p * 3 + 0] = global_translation[0] + rx;
        points[p * 3 + 1] = global_translation[1] + ry;
        points[p * 3 + 2] = global_translation[2] + rz;
    }
}
