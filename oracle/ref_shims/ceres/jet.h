// Stand-in for ceres/jet.h (ceres-solver is absent from /root/reference; pinned at 58c5edae...,
// bazel/repositories.bzl:134-144): dual numbers a + v . eps with the arithmetic of Ceres'
// published jet.h -- product f.a*g.v + f.v*g.a, quotient through g_a_inverse = 1/g.a and
// f_a_by_g_a, sqrt through two_a_inverse, sin / cos / atan2 / abs, comparisons on the scalar part.
// Ours, written from the published formulas; TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REF_SHIMS_CERES_JET_H_
#define ORACLE_REF_SHIMS_CERES_JET_H_
#include <cmath>
namespace ceres {

template <typename T, int N>
struct Jet {
  enum { DIMENSION = N };
  typedef T Scalar;
  T a;
  T v[N];
  Jet() : a() { for (int i = 0; i < N; ++i) v[i] = T(); }
  explicit Jet(const T& value) : a(value) { for (int i = 0; i < N; ++i) v[i] = T(); }
  Jet(const T& value, int k) : a(value) {
    for (int i = 0; i < N; ++i) v[i] = T();
    v[k] = T(1.0);
  }
  Jet& operator+=(const Jet& y) { *this = *this + y; return *this; }
  Jet& operator-=(const Jet& y) { *this = *this - y; return *this; }
  Jet& operator*=(const Jet& y) { *this = *this * y; return *this; }
  Jet& operator/=(const Jet& y) { *this = *this / y; return *this; }
  Jet& operator+=(const T& s) { a += s; return *this; }
  Jet& operator-=(const T& s) { a -= s; return *this; }
  Jet& operator*=(const T& s) { *this = *this * s; return *this; }
  Jet& operator/=(const T& s) { *this = *this / s; return *this; }
};

template <typename T, int N> Jet<T, N> operator+(const Jet<T, N>& f) { return f; }
template <typename T, int N> Jet<T, N> operator-(const Jet<T, N>& f) {
  Jet<T, N> h; h.a = -f.a;
  for (int i = 0; i < N; ++i) h.v[i] = -f.v[i];
  return h;
}
template <typename T, int N> Jet<T, N> operator+(const Jet<T, N>& f, const Jet<T, N>& g) {
  Jet<T, N> h; h.a = f.a + g.a;
  for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i];
  return h;
}
template <typename T, int N> Jet<T, N> operator+(const Jet<T, N>& f, T s) {
  Jet<T, N> h = f; h.a = f.a + s; return h;
}
template <typename T, int N> Jet<T, N> operator+(T s, const Jet<T, N>& f) {
  Jet<T, N> h = f; h.a = f.a + s; return h;
}
template <typename T, int N> Jet<T, N> operator-(const Jet<T, N>& f, const Jet<T, N>& g) {
  Jet<T, N> h; h.a = f.a - g.a;
  for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i];
  return h;
}
template <typename T, int N> Jet<T, N> operator-(const Jet<T, N>& f, T s) {
  Jet<T, N> h = f; h.a = f.a - s; return h;
}
template <typename T, int N> Jet<T, N> operator-(T s, const Jet<T, N>& f) {
  Jet<T, N> h; h.a = s - f.a;
  for (int i = 0; i < N; ++i) h.v[i] = -f.v[i];
  return h;
}
template <typename T, int N> Jet<T, N> operator*(const Jet<T, N>& f, const Jet<T, N>& g) {
  Jet<T, N> h; h.a = f.a * g.a;
  for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a;
  return h;
}
template <typename T, int N> Jet<T, N> operator*(const Jet<T, N>& f, T s) {
  Jet<T, N> h; h.a = f.a * s;
  for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s;
  return h;
}
template <typename T, int N> Jet<T, N> operator*(T s, const Jet<T, N>& f) {
  Jet<T, N> h; h.a = f.a * s;
  for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s;
  return h;
}
template <typename T, int N> Jet<T, N> operator/(const Jet<T, N>& f, const Jet<T, N>& g) {
  const T g_a_inverse = T(1.0) / g.a;
  const T f_a_by_g_a = f.a * g_a_inverse;
  Jet<T, N> h; h.a = f_a_by_g_a;
  for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - f_a_by_g_a * g.v[i]) * g_a_inverse;
  return h;
}
template <typename T, int N> Jet<T, N> operator/(T s, const Jet<T, N>& g) {
  const T minus_s_g_a_inverse2 = -s / (g.a * g.a);
  Jet<T, N> h; h.a = s / g.a;
  for (int i = 0; i < N; ++i) h.v[i] = g.v[i] * minus_s_g_a_inverse2;
  return h;
}
template <typename T, int N> Jet<T, N> operator/(const Jet<T, N>& f, T s) {
  const T s_inverse = T(1.0) / s;
  Jet<T, N> h; h.a = f.a * s_inverse;
  for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s_inverse;
  return h;
}

#define ORACLE_JET_COMPARISON(op)                                                          \
  template <typename T, int N> bool operator op(const Jet<T, N>& f, const Jet<T, N>& g) {  \
    return f.a op g.a;                                                                     \
  }                                                                                        \
  template <typename T, int N> bool operator op(const T& s, const Jet<T, N>& g) {          \
    return s op g.a;                                                                       \
  }                                                                                        \
  template <typename T, int N> bool operator op(const Jet<T, N>& f, const T& s) {          \
    return f.a op s;                                                                       \
  }
ORACLE_JET_COMPARISON(<)
ORACLE_JET_COMPARISON(<=)
ORACLE_JET_COMPARISON(>)
ORACLE_JET_COMPARISON(>=)
ORACLE_JET_COMPARISON(==)
ORACLE_JET_COMPARISON(!=)
#undef ORACLE_JET_COMPARISON

// Scalar versions live in ceres:: too (jet.h pulls std:: in), so that templated code can write
// ceres::sqrt(x) for doubles and Jets alike.
using std::abs;
using std::atan2;
using std::cos;
using std::sin;
using std::sqrt;

template <typename T, int N> Jet<T, N> abs(const Jet<T, N>& f) { return f.a < T(0.0) ? -f : f; }
template <typename T, int N> Jet<T, N> sqrt(const Jet<T, N>& f) {
  const T tmp = std::sqrt(f.a);
  const T two_a_inverse = T(1.0) / (T(2.0) * tmp);
  Jet<T, N> h; h.a = tmp;
  for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * two_a_inverse;
  return h;
}
template <typename T, int N> Jet<T, N> cos(const Jet<T, N>& f) {
  const T minus_sin = -std::sin(f.a);
  Jet<T, N> h; h.a = std::cos(f.a);
  for (int i = 0; i < N; ++i) h.v[i] = minus_sin * f.v[i];
  return h;
}
template <typename T, int N> Jet<T, N> sin(const Jet<T, N>& f) {
  const T c = std::cos(f.a);
  Jet<T, N> h; h.a = std::sin(f.a);
  for (int i = 0; i < N; ++i) h.v[i] = c * f.v[i];
  return h;
}
template <typename T, int N> Jet<T, N> atan2(const Jet<T, N>& g, const Jet<T, N>& f) {
  const T tmp = T(1.0) / (f.a * f.a + g.a * g.a);
  Jet<T, N> h; h.a = std::atan2(g.a, f.a);
  for (int i = 0; i < N; ++i) h.v[i] = tmp * (-g.a * f.v[i] + f.a * g.v[i]);
  return h;
}

}  // namespace ceres
#endif  // ORACLE_REF_SHIMS_CERES_JET_H_
