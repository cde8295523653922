// rmd::SE3<T>: rigid transform stored as a 3x4 row-major [R|t] (reference: include/rmd/se3.cuh:27-168,
// matrix.cuh).  Host side only; the device kernels receive the 12 floats through the C ABI.
// Arithmetic follows the reference's operation order so poses composed here are bit-identical.
#ifndef RMD_SE3_CUH_
#define RMD_SE3_CUH_

#include <iomanip>
#include <ostream>

#include <rmd/host_types.h>

namespace rmd {

template <typename Type, unsigned R, unsigned C>
struct Matrix {
  Type operator()(int row, int col) const { return data[row * C + col]; }
  Type& operator()(int row, int col) { return data[row * C + col]; }
  Type operator[](int ind) const { return data[ind]; }
  Type& operator[](int ind) { return data[ind]; }
  friend std::ostream& operator<<(std::ostream& out, const Matrix<Type, R, C>& m) {
    for (unsigned row = 0; row < R; ++row) {
      for (unsigned col = 0; col < C; ++col) out << std::setprecision(9) << m(row, col) << " ";
      out << std::endl;
    }
    return out;
  }
  Type data[R * C];
};

template <typename Type>
struct SE3 {
  SE3() {
    for (int i = 0; i < 12; ++i) data[i] = Type(0);
    data[0] = data[5] = data[10] = Type(1);
  }

  // unit quaternion (qw, qx, qy, qz) and translation
  SE3(Type qw, Type qx, Type qy, Type qz, Type tx, Type ty, Type tz) {
    const Type x = 2 * qx, y = 2 * qy, z = 2 * qz;
    const Type wx = x * qw, wy = y * qw, wz = z * qw;
    const Type xx = x * qx, xy = y * qx, xz = z * qx;
    const Type yy = y * qy, yz = z * qy, zz = z * qz;
    const Type rot[9] = {1 - (yy + zz), xy - wz, xz + wy, xy + wz, 1 - (xx + zz), yz - wx, xz - wy, yz + wx, 1 - (xx + yy)};
    const Type trans[3] = {tx, ty, tz};
    set(rot, trans);
  }

  // r: 3x3 rotation, row major; t: translation
  SE3(Type* r, Type* t) { set(r, t); }

  SE3<Type> inv() const {
    SE3<Type> out;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) out.data(i, j) = data(j, i);
      out.data(i, 3) = -data(0, i) * data(0, 3) - data(1, i) * data(1, 3) - data(2, i) * data(2, 3);
    }
    return out;
  }

  Type operator()(int r, int c) const { return data(r, c); }
  Type& operator()(int r, int c) { return data(r, c); }

  float3 rotate(const float3& p) const {
    return make_float3(data(0, 0) * p.x + data(0, 1) * p.y + data(0, 2) * p.z, data(1, 0) * p.x + data(1, 1) * p.y + data(1, 2) * p.z,
                       data(2, 0) * p.x + data(2, 1) * p.y + data(2, 2) * p.z);
  }
  float3 translate(const float3& p) const { return make_float3(p.x + data(0, 3), p.y + data(1, 3), p.z + data(2, 3)); }
  float3 getTranslation() const { return make_float3(data(0, 3), data(1, 3), data(2, 3)); }

  friend std::ostream& operator<<(std::ostream& out, const SE3& m) { return out << m.data; }

  Matrix<Type, 3, 4> data;

 private:
  void set(const Type* r, const Type* t) {
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) data(i, j) = r[3 * i + j];
      data(i, 3) = t[i];
    }
  }
};

template <typename Type>
inline SE3<Type> operator*(const SE3<Type>& lhs, const SE3<Type>& rhs) {
  SE3<Type> out;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) out.data(i, j) = lhs.data(i, 0) * rhs.data(0, j) + lhs.data(i, 1) * rhs.data(1, j) + lhs.data(i, 2) * rhs.data(2, j);
    out.data(i, 3) = lhs.data(i, 3) + lhs.data(i, 0) * rhs.data(0, 3) + lhs.data(i, 1) * rhs.data(1, 3) + lhs.data(i, 2) * rhs.data(2, 3);
  }
  return out;
}

inline float3 operator*(const SE3<float>& se3, const float3& p) { return se3.translate(se3.rotate(p)); }

}  // namespace rmd

#endif  // RMD_SE3_CUH_
