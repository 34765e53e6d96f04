// drt_common.h -- small vector types shared by the gfx950 kernels.
//
// Everything here is plain C++ so the per-item bodies of the kernels can also be
// compiled by g++ for tests/hostsim (a CPU unit-test harness for the device math;
// it is never part of the product path).  Arithmetic that defines results
// (ray/triangle test, float64 shading) is written one operation per rounding and
// the library is built with -ffp-contract=off; explicit fma is used only where a
// result is allowed to be conservative (box slabs).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DRT_HD __host__ __device__ __forceinline__
#define DRT_D __device__ __forceinline__
#else
#define DRT_HD inline
#define DRT_D inline
#endif
#if defined(__clang__)
#define DRT_UNROLL _Pragma("unroll")
#else
#define DRT_UNROLL
#endif

namespace drt {

template <typename T>
struct V3 {
    T x, y, z;
};
using f3 = V3<float>;
using d3 = V3<double>;

template <typename T> DRT_HD V3<T> mk(T x, T y, T z) { return V3<T>{x, y, z}; }
template <typename T> DRT_HD V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> DRT_HD V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> DRT_HD V3<T> operator-(V3<T> a) { return {-a.x, -a.y, -a.z}; }
template <typename T> DRT_HD V3<T> operator*(T s, V3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename T> DRT_HD V3<T> operator*(V3<T> a, T s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename T> DRT_HD V3<T> operator/(V3<T> a, T s) { return {a.x / s, a.y / s, a.z / s}; }
template <typename T> DRT_HD V3<T>& operator+=(V3<T>& a, V3<T> b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
template <typename T> DRT_HD V3<T>& operator-=(V3<T>& a, V3<T> b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }

// (a0*b0 + a1*b1) + a2*b2 -- the evaluation order of the reference's `dot` (DiffRender.py:26).
template <typename T> DRT_HD T dot(V3<T> a, V3<T> b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
template <typename T> DRT_HD V3<T> cross(V3<T> a, V3<T> b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

DRT_HD f3 to_f32(d3 a) { return {(float)a.x, (float)a.y, (float)a.z}; }
DRT_HD d3 to_f64(f3 a) { return {(double)a.x, (double)a.y, (double)a.z}; }

DRT_HD d3 load_d3(const double* p, int64_t i) { return {p[3 * i + 0], p[3 * i + 1], p[3 * i + 2]}; }
DRT_HD void store_d3(double* p, int64_t i, d3 v) { p[3 * i + 0] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }

}  // namespace drt
