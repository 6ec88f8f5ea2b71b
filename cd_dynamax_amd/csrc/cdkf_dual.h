// cdkf_dual.h -- forward-mode dual numbers for user-supplied drifts (launch_custom.hip).
//
// The reference differentiates whatever callable it is given -- jacfwd for the Jacobian of the predict step
// (inference_ekf.py:95), a second jacfwd for grad(div f) (inference_ekf.py:108-116), jax.value_and_grad for the parameter
// gradient of fit_sgd (ssm_temissions.py:550-568).  A drift that arrives here as C source (cdkf_custom_drift_register) gets the same
// service from operator overloading: its statements are compiled a second time with the scalar type T = Dual<...> instead of the
// compute type, and the derivatives fall out of the arithmetic -- Dual<R, D> seeded with the unit directions gives the Jacobian,
// Dual<Dual<R, D>, D> the second derivatives grad(div f) contracts, Dual<Dual<R, 1>, D> the directional derivatives of f and of its
// Jacobian along (m', e_p) that the forward-sensitivity sweep of cdkf_grad_kernels.h needs for parameter p.
// Device code only; no standard headers (it is compiled by hipRTC as well as by hipcc).
#pragma once

namespace cdkf {

// the math functions of the compute type stay visible beside the overloads below
using ::sin;
using ::cos;
using ::tan;
using ::tanh;
using ::sinh;
using ::cosh;
using ::exp;
using ::log;
using ::sqrt;
using ::pow;
using ::fabs;
using ::atan;

template <typename S, int N>
struct Dual {
  S v;
  S g[N];
  __device__ Dual() {}
  __device__ Dual(double c) : v(c) {
#pragma unroll
    for (int k = 0; k < N; ++k) g[k] = S(0.0);
  }
  __device__ Dual(float c) : Dual((double)c) {}
  __device__ Dual(int c) : Dual((double)c) {}
  __device__ Dual& operator+=(const Dual& b) {
    v = v + b.v;
#pragma unroll
    for (int k = 0; k < N; ++k) g[k] = g[k] + b.g[k];
    return *this;
  }
  __device__ Dual& operator-=(const Dual& b) {
    v = v - b.v;
#pragma unroll
    for (int k = 0; k < N; ++k) g[k] = g[k] - b.g[k];
    return *this;
  }
  __device__ Dual& operator*=(const Dual& b) {
#pragma unroll
    for (int k = 0; k < N; ++k) g[k] = g[k] * b.v + v * b.g[k];
    v = v * b.v;
    return *this;
  }
  __device__ Dual& operator/=(const Dual& b) {
    const S q = v / b.v;
#pragma unroll
    for (int k = 0; k < N; ++k) g[k] = (g[k] - q * b.g[k]) / b.v;
    v = q;
    return *this;
  }
};

template <typename S, int N>
__device__ inline Dual<S, N> operator+(Dual<S, N> a, const Dual<S, N>& b) { return a += b; }
template <typename S, int N>
__device__ inline Dual<S, N> operator-(Dual<S, N> a, const Dual<S, N>& b) { return a -= b; }
template <typename S, int N>
__device__ inline Dual<S, N> operator*(Dual<S, N> a, const Dual<S, N>& b) { return a *= b; }
template <typename S, int N>
__device__ inline Dual<S, N> operator/(Dual<S, N> a, const Dual<S, N>& b) { return a /= b; }
template <typename S, int N>
__device__ inline Dual<S, N> operator-(const Dual<S, N>& a) {
  Dual<S, N> r;
  r.v = -a.v;
#pragma unroll
  for (int k = 0; k < N; ++k) r.g[k] = -a.g[k];
  return r;
}
template <typename S, int N>
__device__ inline Dual<S, N> operator+(const Dual<S, N>& a) { return a; }

// constants of the compute type (and integer literals) on either side
#define CDKF_DUAL_SCALAR_OPS(C)                                                                                              \
  template <typename S, int N> __device__ inline Dual<S, N> operator+(const Dual<S, N>& a, C c) { return a + Dual<S, N>(c); } \
  template <typename S, int N> __device__ inline Dual<S, N> operator+(C c, const Dual<S, N>& a) { return Dual<S, N>(c) + a; } \
  template <typename S, int N> __device__ inline Dual<S, N> operator-(const Dual<S, N>& a, C c) { return a - Dual<S, N>(c); } \
  template <typename S, int N> __device__ inline Dual<S, N> operator-(C c, const Dual<S, N>& a) { return Dual<S, N>(c) - a; } \
  template <typename S, int N> __device__ inline Dual<S, N> operator*(const Dual<S, N>& a, C c) { return a * Dual<S, N>(c); } \
  template <typename S, int N> __device__ inline Dual<S, N> operator*(C c, const Dual<S, N>& a) { return Dual<S, N>(c) * a; } \
  template <typename S, int N> __device__ inline Dual<S, N> operator/(const Dual<S, N>& a, C c) { return a / Dual<S, N>(c); } \
  template <typename S, int N> __device__ inline Dual<S, N> operator/(C c, const Dual<S, N>& a) { return Dual<S, N>(c) / a; } \
  template <typename S, int N> __device__ inline bool operator<(const Dual<S, N>& a, C c) { return a.v < S(c); }              \
  template <typename S, int N> __device__ inline bool operator>(const Dual<S, N>& a, C c) { return a.v > S(c); }              \
  template <typename S, int N> __device__ inline bool operator<=(const Dual<S, N>& a, C c) { return a.v <= S(c); }            \
  template <typename S, int N> __device__ inline bool operator>=(const Dual<S, N>& a, C c) { return a.v >= S(c); }            \
  template <typename S, int N> __device__ inline bool operator<(C c, const Dual<S, N>& a) { return S(c) < a.v; }              \
  template <typename S, int N> __device__ inline bool operator>(C c, const Dual<S, N>& a) { return S(c) > a.v; }
CDKF_DUAL_SCALAR_OPS(double)
CDKF_DUAL_SCALAR_OPS(float)
CDKF_DUAL_SCALAR_OPS(int)
#undef CDKF_DUAL_SCALAR_OPS

// comparisons look at the values (a branch of the drift takes the side its primal takes)
template <typename S, int N>
__device__ inline bool operator<(const Dual<S, N>& a, const Dual<S, N>& b) { return a.v < b.v; }
template <typename S, int N>
__device__ inline bool operator>(const Dual<S, N>& a, const Dual<S, N>& b) { return a.v > b.v; }
template <typename S, int N>
__device__ inline bool operator<=(const Dual<S, N>& a, const Dual<S, N>& b) { return a.v <= b.v; }
template <typename S, int N>
__device__ inline bool operator>=(const Dual<S, N>& a, const Dual<S, N>& b) { return a.v >= b.v; }

// y = h(a): value h(a.v), derivative h'(a.v) a.g  (hv, dh: of the inner scalar type, which may itself be a Dual)
template <typename S, int N>
__device__ inline Dual<S, N> dual_chain(const Dual<S, N>& a, const S& hv, const S& dh) {
  Dual<S, N> r;
  r.v = hv;
#pragma unroll
  for (int k = 0; k < N; ++k) r.g[k] = dh * a.g[k];
  return r;
}
template <typename S, int N>
__device__ inline Dual<S, N> sin(const Dual<S, N>& a) { return dual_chain(a, sin(a.v), cos(a.v)); }
template <typename S, int N>
__device__ inline Dual<S, N> cos(const Dual<S, N>& a) { return dual_chain(a, cos(a.v), -sin(a.v)); }
template <typename S, int N>
__device__ inline Dual<S, N> tan(const Dual<S, N>& a) {
  const S t = tan(a.v);
  return dual_chain(a, t, S(1.0) + t * t);
}
template <typename S, int N>
__device__ inline Dual<S, N> tanh(const Dual<S, N>& a) {
  const S t = tanh(a.v);
  return dual_chain(a, t, S(1.0) - t * t);
}
template <typename S, int N>
__device__ inline Dual<S, N> sinh(const Dual<S, N>& a) { return dual_chain(a, sinh(a.v), cosh(a.v)); }
template <typename S, int N>
__device__ inline Dual<S, N> cosh(const Dual<S, N>& a) { return dual_chain(a, cosh(a.v), sinh(a.v)); }
template <typename S, int N>
__device__ inline Dual<S, N> exp(const Dual<S, N>& a) {
  const S e = exp(a.v);
  return dual_chain(a, e, e);
}
template <typename S, int N>
__device__ inline Dual<S, N> log(const Dual<S, N>& a) { return dual_chain(a, log(a.v), S(1.0) / a.v); }
template <typename S, int N>
__device__ inline Dual<S, N> sqrt(const Dual<S, N>& a) {
  const S s = sqrt(a.v);
  return dual_chain(a, s, S(0.5) / s);
}
template <typename S, int N>
__device__ inline Dual<S, N> atan(const Dual<S, N>& a) { return dual_chain(a, atan(a.v), S(1.0) / (S(1.0) + a.v * a.v)); }
template <typename S, int N>
__device__ inline Dual<S, N> fabs(const Dual<S, N>& a) { return (a.v < S(0.0)) ? -a : a; }
// a^c for a constant exponent; a^b = exp(b log a) otherwise
template <typename S, int N>
__device__ inline Dual<S, N> pow(const Dual<S, N>& a, int c);
template <typename S, int N>
__device__ inline Dual<S, N> pow(const Dual<S, N>& a, double c) {
  if (c >= 0.0 && c <= 16.0 && c == (double)(int)c) return pow(a, (int)c);  // (a whole exponent: by multiplications, below)
  const S pv = S(pow(a.v, c)), dv = S(c) * S(pow(a.v, c - 1.0));  // (pow(float, double) is a double: back to the inner type)
  return dual_chain(a, pv, dv);
}
template <typename S, int N>
__device__ inline Dual<S, N> pow(const Dual<S, N>& a, float c) { return pow(a, (double)c); }
// a^c for a small whole exponent: by multiplications (exact derivatives of every order the nesting asks for, and no call into the
// maths library: the library's pow is not inlined into the larger run-time compiled kernels)
template <typename S, int N>
__device__ inline Dual<S, N> pow(const Dual<S, N>& a, int c) {
  if (c >= 0 && c <= 16) {
    Dual<S, N> r(1.0), b = a;
    for (int e = c; e; e >>= 1) {
      if (e & 1) r *= b;
      if (e >> 1) b *= b;
    }
    return r;
  }
  return pow(a, (double)c);
}
template <typename S, int N>
__device__ inline Dual<S, N> pow(const Dual<S, N>& a, const Dual<S, N>& b) { return exp(b * log(a)); }

}  // namespace cdkf
