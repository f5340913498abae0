// Minimal stand-in for <sophus/se3.hpp>: the part of Sophus::SE3 that the reference's ICP interface and its
// src/Tools/tsdf_icp.cpp use (default construction = identity, rotationMatrix(), translation(), matrix(), cast<>(),
// exp(), operator*, inverse()).  Sophus is not vendored by the reference and not installed in this image; with a real
// Sophus on the include path this directory is simply not used (Makefile: SOPHUS_INC).
#ifndef TSDF_AMD_SOPHUS_COMPAT_SE3
#define TSDF_AMD_SOPHUS_COMPAT_SE3

#include <cmath>

#include <Eigen/Core>

namespace Sophus {

template <typename Scalar>
class SE3 {
public:
    typedef Eigen::Matrix<Scalar, 3, 3> Rotation;
    typedef Eigen::Matrix<Scalar, 3, 1> Point;
    typedef Eigen::Matrix<Scalar, 4, 4> Transformation;
    typedef Eigen::Matrix<Scalar, 6, 1> Tangent;

    SE3() : r_(Rotation::Identity()), t_(Point::Zero()) {}
    SE3(const Rotation &r, const Point &t) : r_(r), t_(t) {}
    explicit SE3(const Transformation &m) {
        for (int c = 0; c < 3; c++)
            for (int r = 0; r < 3; r++) r_(r, c) = m(r, c);
        for (int r = 0; r < 3; r++) t_[r] = m(r, 3);
    }

    Rotation rotationMatrix() const { return r_; }
    Point &translation() { return t_; }
    const Point &translation() const { return t_; }
    Transformation matrix() const {
        Transformation m = Transformation::Identity();
        for (int c = 0; c < 3; c++)
            for (int r = 0; r < 3; r++) m(r, c) = r_(r, c);
        for (int r = 0; r < 3; r++) m(r, 3) = t_[r];
        return m;
    }
    template <typename NewScalar>
    SE3<NewScalar> cast() const {
        return SE3<NewScalar>(r_.template cast<NewScalar>(), t_.template cast<NewScalar>());
    }
    SE3 operator*(const SE3 &o) const { return SE3(r_ * o.r_, r_ * o.t_ + t_); }
    SE3 &operator*=(const SE3 &o) { return *this = *this * o; }
    Point operator*(const Point &p) const { return r_ * p + t_; }
    SE3 inverse() const {
        Rotation rt = r_.transpose();
        return SE3(rt, -(rt * t_));
    }

    // a = (upsilon, omega): rotation exp(hat(omega)) by Rodrigues' formula, translation V * upsilon with
    // V = I + (1 - cos th)/th^2 W + (th - sin th)/th^3 W^2
    static SE3 exp(const Tangent &a) {
        const Scalar wx = a[3], wy = a[4], wz = a[5];
        const Scalar th2 = wx * wx + wy * wy + wz * wz, th = std::sqrt(th2);
        Rotation W, W2;
        W << 0, -wz, wy, wz, 0, -wx, -wy, wx, 0;
        W2 = W * W;
        Scalar A, B, C;
        if (th < Scalar(1e-10)) {
            A = Scalar(1) - th2 / Scalar(6);
            B = Scalar(0.5) - th2 / Scalar(24);
            C = Scalar(1) / Scalar(6) - th2 / Scalar(120);
        } else {
            A = std::sin(th) / th;
            B = (Scalar(1) - std::cos(th)) / th2;
            C = (th - std::sin(th)) / (th2 * th);
        }
        const Rotation I = Rotation::Identity();
        const Rotation R = I + W * A + W2 * B, V = I + W * B + W2 * C;
        return SE3(R, V * Point(a[0], a[1], a[2]));
    }

private:
    Rotation r_;
    Point t_;
};

typedef SE3<double> SE3d;
typedef SE3<float> SE3f;

}  // namespace Sophus
#endif
