// empc_dev_math.hpp -- scalar / dual FP64 helpers shared by the HIP kernels (host+device inlines).
//
// Everything here is per-lane code: no LDS, no cross-lane traffic.  The kernels in empc_kernels.hip combine these
// pieces with LDS staging.  The same header compiles with g++ (EMPC_HD empty) so that tests/ can run the per-lane
// phases through a CPU lane emulator when no GPU is present; that emulator is test infrastructure, not a fallback.
//
// Lie-group conventions: Pinocchio free-flyer as used by crocoddyl::StateMultibody:
// spatial vectors [linear; angular], quaternion xyzw, right-perturbation Jacobians.
#pragma once
#include <math.h>
#include "empc_variants.hpp"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define EMPC_HD __host__ __device__ __forceinline__
#else
#define EMPC_HD inline
#endif

// Read-only problem data (model, cost tables, solver constants) is addressed through the AMDGPU constant address
// space on the device: wave-uniform loads then become scalar (s_load) instructions and the values live in SGPRs
// instead of occupying vector registers and the vector-memory pipeline.
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define EMPC_K __attribute__((address_space(4)))
#define EMPC_KREF(T, ptr) (*(const EMPC_K T*)(unsigned long long)(ptr))
#define EMPC_KPTR(T, ptr) ((const EMPC_K T*)(unsigned long long)(ptr))
#else
#define EMPC_K
#define EMPC_KREF(T, ptr) (*(ptr))
#define EMPC_KPTR(T, ptr) (ptr)
#endif

namespace empc {

// ---- one-direction dual number -------------------------------------------------------------------
struct D1 {
  double v, d;
  EMPC_HD D1() : v(0.0), d(0.0) {}
  EMPC_HD D1(double x) : v(x), d(0.0) {}
  EMPC_HD D1(double x, double y) : v(x), d(y) {}
};
EMPC_HD D1 operator+(const D1& a, const D1& b) { return D1(a.v + b.v, a.d + b.d); }
EMPC_HD D1 operator-(const D1& a, const D1& b) { return D1(a.v - b.v, a.d - b.d); }
EMPC_HD D1 operator-(const D1& a) { return D1(-a.v, -a.d); }
EMPC_HD D1 operator*(const D1& a, const D1& b) { return D1(a.v * b.v, a.d * b.v + a.v * b.d); }
EMPC_HD D1 operator*(double a, const D1& b) { return D1(a * b.v, a * b.d); }
EMPC_HD D1 operator*(const D1& b, double a) { return D1(a * b.v, a * b.d); }
EMPC_HD D1 operator+(const D1& a, double b) { return D1(a.v + b, a.d); }
EMPC_HD D1 operator+(double b, const D1& a) { return D1(a.v + b, a.d); }
EMPC_HD D1 operator-(const D1& a, double b) { return D1(a.v - b, a.d); }
EMPC_HD D1 operator-(double b, const D1& a) { return D1(b - a.v, -a.d); }
EMPC_HD D1& operator+=(D1& a, const D1& b) {
  a.v += b.v;
  a.d += b.d;
  return a;
}
EMPC_HD D1& operator-=(D1& a, const D1& b) {
  a.v -= b.v;
  a.d -= b.d;
  return a;
}
EMPC_HD double val(double a) { return a; }
EMPC_HD double val(const D1& a) { return a.v; }

// ---- scalar primitives of the hot loops --------------------------------------------------------------------
// The per-knot chain of the rollout / linearize kernels is made of scalar SE(3) exp/log work, so the cost of one
// division, square root or sine matters.  These forms are a few ulp from the correctly rounded results (the parity
// tolerance of the solver is 1e-4 on xs/us; kernel-vs-oracle tests hold 1e-11) and cost a fraction of the IEEE
// expansions: rcp/rsq hardware seeds + Newton steps on the device, fdlibm polynomial kernels without the
// Payne-Hanek path for sin/cos (arguments here are joint angles and rotation half-angles, |x| << 2^20 pi/2).
EMPC_HD double frcp(double x) {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
#else
  return 1.0 / x;
#endif
}
EMPC_HD double fdiv(double a, double b) {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
  const double r = frcp(b);
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
#else
  return a / b;
#endif
}
// 1 / sqrt(x), x > 0
EMPC_HD double frsqrt(double x) {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  r = r * fma(-h * r, r, 1.5);
  const double e = fma(-x * r, r, 1.0);  // 1 - x r^2
  return fma(0.5 * r, e, r);
#else
  return 1.0 / sqrt(x);
#endif
}
// sqrt(x), x >= 0
EMPC_HD double fsqrt(double x) {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#if EMPC_FSQRT_BITS
  // the failure value from the bit pattern: under -fno-honor-nans (the baked units) `x == x` folds to true and a product with
  // a NaN constant is poison, so neither may decide or make the result
  if (!(x > 1e-300)) {
    unsigned long long u;
    __builtin_memcpy(&u, &x, sizeof(u));
    const bool nan_or_negative = (u & 0x7fffffffffffffffull) > 0x7ff0000000000000ull || ((u >> 63) != 0 && (u << 1) != 0);
    u = nan_or_negative ? 0x7ff8000000000000ull : 0ull;
    double r_;
    __builtin_memcpy(&r_, &u, sizeof(r_));
    return r_;
  }
#else
  if (!(x > 1e-300)) return (x == x && x >= 0.0) ? 0.0 : x * __builtin_nan("");
#endif
  const double r = frsqrt(x);
  const double s = x * r;
  return fma(fma(-s, s, x), 0.5 * r, s);
#else
  return sqrt(x);
#endif
}
// sin and cos of x: Cody-Waite reduction by pi/2 (33-bit pieces taken off with fma: the products need not be representable,
// only the differences, which they are far beyond |k| = 2^20 -- measured within 2e-16 of sinl / cosl for |x| up to 1.6e12,
// round 5) and the fdlibm kernels on [-pi/4, pi/4]
EMPC_HD void fsincos(double x, double* sn, double* cs) {
  const double kf = rint(x * 6.36619772367581382433e-01);
  double r = fma(-kf, 1.57079632673412561417e+00, x);
  r = fma(-kf, 6.07710050630396597660e-11, r);
  r = fma(-kf, 2.02226624871116645580e-21, r);
  r = fma(-kf, 8.47842766036889956997e-32, r);
  const double z = r * r;
  const double ps = -1.66666666666666324348e-01 +
                    z * (8.33333333332248946124e-03 +
                         z * (-1.98412698298579493134e-04 +
                              z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
  const double s = fma(r * z, ps, r);
  const double pc = 4.16666666666666019037e-02 +
                    z * (-1.38888888888741095749e-03 +
                         z * (2.48015872894767294178e-05 +
                              z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double c = w + (((1.0 - w) - hz) + z * z * pc);
  const int q = (int)((long long)kf & 3);
  const double s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
  *sn = (q & 2) ? -s1 : s1;
  *cs = ((q + 1) & 2) ? -c1 : c1;
}
// atan2(y, x) for y >= 0, x >= 0 (first quadrant; the only use is the half-angle of a unit quaternion):
// fdlibm atan kernel on the ratio min/max, with the interval transforms folded into one division
EMPC_HD double fatan2_pos(double y, double x) {
  const bool sw = y > x;
  const double mn = sw ? x : y, mx = sw ? y : x;
  double num, den, hi, lo;
  if (16.0 * mn < 7.0 * mx) {  // z < 7/16
    num = mn;
    den = mx;
    hi = 0.0;
    lo = 0.0;
  } else if (16.0 * mn < 11.0 * mx) {  // atan(z) = atan(1/2) + atan((2z - 1) / (2 + z))
    num = 2.0 * mn - mx;
    den = 2.0 * mx + mn;
    hi = 4.63647609000806093515e-01;
    lo = 2.26987774529616870924e-17;
  } else {  // atan(z) = pi/4 + atan((z - 1) / (z + 1))
    num = mn - mx;
    den = mx + mn;
    hi = 7.85398163397448278999e-01;
    lo = 3.06161699786838301793e-17;
  }
  const double z = (den > 0.0) ? fdiv(num, den) : 0.0;
  const double z2 = z * z, w = z2 * z2;
  const double s1 = z2 * (3.33333333333329318027e-01 +
                          w * (1.42857142725034663711e-01 +
                               w * (9.09088713343650656196e-02 +
                                    w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
  const double s2 = w * (-1.99999999998764832476e-01 +
                         w * (-1.11111104054623557880e-01 +
                              w * (-7.69187620504482999495e-02 + w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
  const double a = hi - ((z * (s1 + s2) - lo) - z);
  return sw ? (1.57079632679489655800e+00 - a) + 6.12323399573676603587e-17 : a;
}

// ---- 3-vector / 3x3 (row-major) helpers, generic in the scalar ----------------------------------------
template <class S, class A, class B>
EMPC_HD void cross3(const A* a, const B* b, S* r) {
  S r0 = a[1] * b[2] - a[2] * b[1];
  S r1 = a[2] * b[0] - a[0] * b[2];
  S r2 = a[0] * b[1] - a[1] * b[0];
  r[0] = r0;
  r[1] = r1;
  r[2] = r2;
}
template <class S, class A, class B>
EMPC_HD S dot3(const A* a, const B* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
template <class S, class A, class B>
EMPC_HD void matvec3(const A* M, const B* v, S* r) {
  S r0 = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
  S r1 = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
  S r2 = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  r[0] = r0;
  r[1] = r1;
  r[2] = r2;
}
template <class S, class A, class B>
EMPC_HD void matTvec3(const A* M, const B* v, S* r) {
  S r0 = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
  S r1 = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
  S r2 = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
  r[0] = r0;
  r[1] = r1;
  r[2] = r2;
}
template <class S, class A, class B>
EMPC_HD void matmul3(const A* a, const B* b, S* r) {
  S t[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
#pragma unroll
  for (int i = 0; i < 9; ++i) r[i] = t[i];
}
template <class S, class A, class B>
EMPC_HD void matTmul3(const A* a, const B* b, S* r) {
  S t[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) t[3 * i + j] = a[i] * b[j] + a[3 + i] * b[3 + j] + a[6 + i] * b[6 + j];
#pragma unroll
  for (int i = 0; i < 9; ++i) r[i] = t[i];
}
EMPC_HD void skew3(const double* w, double* M) {
  M[0] = 0;
  M[1] = -w[2];
  M[2] = w[1];
  M[3] = w[2];
  M[4] = 0;
  M[5] = -w[0];
  M[6] = -w[1];
  M[7] = w[0];
  M[8] = 0;
}

// ---- quaternions (x,y,z,w) -----------------------------------------------------------------------------
EMPC_HD void quat_to_R(const double* q, double* R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z);
  R[1] = 2 * (x * y - z * w);
  R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);
  R[4] = 1 - 2 * (x * x + z * z);
  R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);
  R[7] = 2 * (y * z + x * w);
  R[8] = 1 - 2 * (x * x + y * y);
}
EMPC_HD void quat_mul(const double* a, const double* b, double* r) {
  const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  const double y = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  const double z = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  r[0] = x;
  r[1] = y;
  r[2] = z;
  r[3] = w;
}
EMPC_HD void quat_normalize(double* q) {
  const double n = frsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] *= n;
  q[1] *= n;
  q[2] *= n;
  q[3] *= n;
}
EMPC_HD void quat_exp3(const double* w, double* q) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double t = fsqrt(t2);
  double k, sh, ch;
  fsincos(0.5 * t, &sh, &ch);
  if (t < 1e-4)
    k = 0.5 - t2 * (1.0 / 48.0) + t2 * t2 * (1.0 / 3840.0);
  else
    k = fdiv(sh, t);
  q[0] = k * w[0];
  q[1] = k * w[1];
  q[2] = k * w[2];
  q[3] = ch;
}
// log of a unit quaternion; optionally returns the half-angle sine / cosine (n, |q.w|) and the angle
EMPC_HD void quat_log3(const double* q, double* w, double* half = nullptr) {
  const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  const double n = fsqrt(n2);
  const double sgn = q[3] >= 0 ? 1.0 : -1.0;
  const double aw = fabs(q[3]);
  double k, t;
  if (n < 1e-6) {
    k = fdiv(2.0, aw) * (1.0 - fdiv(n2, 3.0 * aw * aw));
    t = k * n;
  } else {
    t = 2.0 * fatan2_pos(n, aw);
    k = fdiv(t, n);
  }
  k *= sgn;
  w[0] = k * q[0];
  w[1] = k * q[1];
  w[2] = k * q[2];
  if (half) {
    half[0] = n;
    half[1] = aw;
    half[2] = t;
  }
}
EMPC_HD void R_to_quat(const double* R, double* q) {
  const double tr = R[0] + R[4] + R[8];
  // s = 2 sqrt(d): the dominant component is s / 4 = d * (0.5 / sqrt(d)), the others are differences * (0.5 / sqrt(d))
  if (tr > 0) {
    const double d = tr + 1.0, h = 0.5 * frsqrt(d);
    q[3] = d * h;
    q[0] = (R[7] - R[5]) * h;
    q[1] = (R[2] - R[6]) * h;
    q[2] = (R[3] - R[1]) * h;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const double d = 1.0 + R[0] - R[4] - R[8], h = 0.5 * frsqrt(d);
    q[3] = (R[7] - R[5]) * h;
    q[0] = d * h;
    q[1] = (R[1] + R[3]) * h;
    q[2] = (R[2] + R[6]) * h;
  } else if (R[4] > R[8]) {
    const double d = 1.0 + R[4] - R[0] - R[8], h = 0.5 * frsqrt(d);
    q[3] = (R[2] - R[6]) * h;
    q[0] = (R[1] + R[3]) * h;
    q[1] = d * h;
    q[2] = (R[5] + R[7]) * h;
  } else {
    const double d = 1.0 + R[8] - R[0] - R[4], h = 0.5 * frsqrt(d);
    q[3] = (R[3] - R[1]) * h;
    q[0] = (R[2] + R[6]) * h;
    q[1] = (R[5] + R[7]) * h;
    q[2] = d * h;
  }
  quat_normalize(q);
}

// coefficient set shared by exp6 / log6 / Jexp6 / Jlog6 (all functions of theta only)
struct SO3Coef {
  double a;     // sin t / t
  double b;     // (1 - cos t) / t^2
  double c;     // (t - sin t) / t^3
  double e;     // 1/t^2 - (1+cos t)/(2 t sin t)  == beta of log6
  double al;    // t sin t / (2 (1 - cos t))       alpha of log6
  double bdot;  // d(beta)/dt / t
};
// from the angle t (t2 = t^2) and the sine / cosine of HALF the angle: sin t = 2 sh ch, 1 - cos t = 2 sh^2
EMPC_HD void so3_coef_half(double t2, double t, double sh, double ch, SO3Coef& k) {
  if (t < 1e-2) {
    const double t4 = t2 * t2;
    const double t6 = t4 * t2;  // series coefficients as reciprocal constants: no divisions on this branch
    k.a = 1.0 - t2 * (1.0 / 6.0) + t4 * (1.0 / 120.0) - t6 * (1.0 / 5040.0);
    k.b = 0.5 - t2 * (1.0 / 24.0) + t4 * (1.0 / 720.0) - t6 * (1.0 / 40320.0);
    k.c = 1.0 / 6.0 - t2 * (1.0 / 120.0) + t4 * (1.0 / 5040.0) - t6 * (1.0 / 362880.0);
    k.e = 1.0 / 12.0 + t2 * (1.0 / 720.0) + t4 * (1.0 / 30240.0) + t6 * (1.0 / 1209600.0);
    k.al = 1.0 - t2 * (1.0 / 12.0) - t4 * (1.0 / 720.0) - t6 * (1.0 / 30240.0);
    k.bdot = 1.0 / 360.0 + t2 * (1.0 / 7560.0) + t4 * (1.0 / 201600.0);
  } else {
    const double st = 2.0 * sh * ch;
    const double it = frcp(t), it2 = it * it;
    const double cot = fdiv(ch, sh);      // cot(t/2) = sin t / (1 - cos t)
    k.a = st * it;
    k.b = 2.0 * sh * sh * it2;
    k.c = (t - st) * it2 * it;
    k.e = it2 - 0.5 * it * cot;           // 1/t^2 - sin t / (2 t (1 - cos t))
    k.al = 0.5 * t * cot;                 // t sin t / (2 (1 - cos t))
    k.bdot = -2.0 * it2 * it2 + (1.0 + k.a) * it2 * fdiv(0.25, sh * sh);  // (1 + sin t / t) / (2 t^2 (1 - cos t))
  }
}
EMPC_HD void so3_coef(double t2, SO3Coef& k) {
  const double t = fsqrt(t2);
  double sh, ch;
  fsincos(0.5 * t, &sh, &ch);
  so3_coef_half(t2, t, sh, ch, k);
}

// exp6 of [v; w] -> rotation as quaternion and translation
EMPC_HD void exp6_quat(const double* xi, double* q, double* p) {
  const double* v = xi;
  const double* w = xi + 3;
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  SO3Coef k;
  so3_coef(t2, k);
  double wxv[3];
  cross3<double>(w, v, wxv);
  const double wv = dot3<double>(w, v);
#pragma unroll
  for (int i = 0; i < 3; ++i) p[i] = k.a * v[i] + k.c * wv * w[i] + k.b * wxv[i];
  quat_exp3(w, q);
}
// log6 from a unit quaternion and a translation
EMPC_HD void log6_quat(const double* q, const double* p, double* xi) {
  double w[3], half[3];
  quat_log3(q, w, half);
  const double t = half[2], t2 = t * t;
  SO3Coef k;
  so3_coef_half(t2, t, half[0], half[1], k);  // unit quaternion: (|q.xyz|, |q.w|) = (sin, cos) of t / 2
  double wxp[3];
  cross3<double>(w, p, wxp);
  const double wp = dot3<double>(w, p);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    xi[i] = k.al * p[i] - 0.5 * wxp[i] + k.e * wp * w[i];
    xi[3 + i] = w[i];
  }
}
// Jr(w) = I - b [w]x + c [w]x^2 ; Jlog3(w) = I + 1/2 [w]x + e [w]x^2
EMPC_HD void Jexp3(const double* w, const SO3Coef& k, double* J) {
  double K[9], K2[9];
  skew3(w, K);
  matmul3<double>(K, K, K2);
#pragma unroll
  for (int i = 0; i < 9; ++i) J[i] = -k.b * K[i] + k.c * K2[i];
  J[0] += 1;
  J[4] += 1;
  J[8] += 1;
}
EMPC_HD void Jlog3(const double* w, const SO3Coef& k, double* J) {
  double K[9], K2[9];
  skew3(w, K);
  matmul3<double>(K, K, K2);
#pragma unroll
  for (int i = 0; i < 9; ++i) J[i] = 0.5 * K[i] + k.e * K2[i];
  J[0] += 1;
  J[4] += 1;
  J[8] += 1;
}
// The 3x3 coupling block shared by Jexp6 / Jlog6 (Pinocchio explog form):
//   C(p, w) = (bdot w.p) w w^T - (t^2 bdot + 2 beta) p w^T + (w.p) beta I + beta w p^T + 1/2 [p]x
// with p the translation of exp6([v; w]).  Jlog6 = [[Jl, C Jl],[0, Jl]],  Jexp6 = [[Jr, -Jr C],[0, Jr]].
EMPC_HD void se3_C(const double* p, const double* w, double t2, const SO3Coef& k, double* C) {
  const double wp = dot3<double>(w, p);
  const double c1 = k.bdot * wp, c2 = t2 * k.bdot + 2.0 * k.e;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = c1 * w[i] * w[j] - c2 * p[i] * w[j] + k.e * w[i] * p[j];
  C[0] += wp * k.e;
  C[4] += wp * k.e;
  C[8] += wp * k.e;
  C[1] -= 0.5 * p[2];
  C[2] += 0.5 * p[1];
  C[3] += 0.5 * p[2];
  C[5] -= 0.5 * p[0];
  C[6] -= 0.5 * p[1];
  C[7] += 0.5 * p[0];
}
// Jlog6 with xi = log6(M), p = translation of M.  Output 6x6 row-major.
EMPC_HD void Jlog6(const double* xi, const double* p, double* J) {
  const double* w = xi + 3;
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  SO3Coef k;
  so3_coef(t2, k);
  double Jl[9], C[9], CJ[9];
  Jlog3(w, k, Jl);
  se3_C(p, w, t2, k, C);
  matmul3<double>(C, Jl, CJ);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      J[6 * i + j] = Jl[3 * i + j];
      J[6 * i + 3 + j] = CJ[3 * i + j];
      J[6 * (3 + i) + j] = 0;
      J[6 * (3 + i) + 3 + j] = Jl[3 * i + j];
    }
}
// Jexp6 at xi with p = translation of exp6(xi).
EMPC_HD void Jexp6(const double* xi, const double* p, double* J) {
  const double* w = xi + 3;
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  SO3Coef k;
  so3_coef(t2, k);
  double Jr[9], C[9], JC[9];
  Jexp3(w, k, Jr);
  se3_C(p, w, t2, k, C);
  matmul3<double>(Jr, C, JC);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      J[6 * i + j] = Jr[3 * i + j];
      J[6 * i + 3 + j] = -JC[3 * i + j];
      J[6 * (3 + i) + j] = 0;
      J[6 * (3 + i) + 3 + j] = Jr[3 * i + j];
    }
}

// NaN / overflow guards on the bit pattern: the same answers as `v != v` and `!(fabs(v) < 1e30)`, whatever the compiler is
// told about NaNs (the baked-model instantiations are built with -fno-honor-nans -fno-signed-zeros so that products with
// the structural zeros of a robot fold away; a floating-point self-comparison would fold away with them)
EMPC_HD unsigned long long double_bits(double v) {
  unsigned long long u;
  __builtin_memcpy(&u, &v, sizeof(u));
  return u;
}
EMPC_HD bool is_nan(double v) { return (double_bits(v) & 0x7fffffffffffffffull) > 0x7ff0000000000000ull; }
// NaN, inf or >= 1e30 (crocoddyl::raiseIfNaN); 0x46293e5939a08cea = 1e30
EMPC_HD bool bad_number(double v) { return (double_bits(v) & 0x7fffffffffffffffull) >= 0x46293e5939a08ceaull; }

}  // namespace empc
