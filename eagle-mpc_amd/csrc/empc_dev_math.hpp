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
  const double n = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] *= n;
  q[1] *= n;
  q[2] *= n;
  q[3] *= n;
}
EMPC_HD void quat_exp3(const double* w, double* q) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double t = sqrt(t2);
  double k;
  if (t < 1e-4)
    k = 0.5 - t2 / 48.0 + t2 * t2 / 3840.0;
  else
    k = sin(0.5 * t) / t;
  q[0] = k * w[0];
  q[1] = k * w[1];
  q[2] = k * w[2];
  q[3] = cos(0.5 * t);
}
EMPC_HD void quat_log3(const double* q, double* w) {
  const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  const double n = sqrt(n2);
  const double sgn = q[3] >= 0 ? 1.0 : -1.0;
  const double aw = fabs(q[3]);
  double k;
  if (n < 1e-6)
    k = 2.0 / aw * (1.0 - n2 / (3.0 * aw * aw));
  else
    k = 2.0 * atan2(n, aw) / n;
  k *= sgn;
  w[0] = k * q[0];
  w[1] = k * q[1];
  w[2] = k * q[2];
}
EMPC_HD void R_to_quat(const double* R, double* q) {
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    const double s = sqrt(tr + 1.0) * 2;
    q[3] = 0.25 * s;
    q[0] = (R[7] - R[5]) / s;
    q[1] = (R[2] - R[6]) / s;
    q[2] = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
    q[3] = (R[7] - R[5]) / s;
    q[0] = 0.25 * s;
    q[1] = (R[1] + R[3]) / s;
    q[2] = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    const double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
    q[3] = (R[2] - R[6]) / s;
    q[0] = (R[1] + R[3]) / s;
    q[1] = 0.25 * s;
    q[2] = (R[5] + R[7]) / s;
  } else {
    const double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
    q[3] = (R[3] - R[1]) / s;
    q[0] = (R[2] + R[6]) / s;
    q[1] = (R[5] + R[7]) / s;
    q[2] = 0.25 * s;
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
EMPC_HD void so3_coef(double t2, SO3Coef& k) {
  const double t = sqrt(t2);
  if (t < 1e-2) {
    const double t4 = t2 * t2;
    k.a = 1.0 - t2 / 6.0 + t4 / 120.0 - t4 * t2 / 5040.0;
    k.b = 0.5 - t2 / 24.0 + t4 / 720.0 - t4 * t2 / 40320.0;
    k.c = 1.0 / 6.0 - t2 / 120.0 + t4 / 5040.0 - t4 * t2 / 362880.0;
    k.e = 1.0 / 12.0 + t2 / 720.0 + t4 / 30240.0 + t4 * t2 / 1209600.0;
    k.al = 1.0 - t2 / 12.0 - t4 / 720.0 - t4 * t2 / 30240.0;
    k.bdot = 1.0 / 360.0 + t2 / 7560.0 + t4 / 201600.0;
  } else {
    const double st = sin(t), ct = cos(t);
    k.a = st / t;
    k.b = (1.0 - ct) / t2;
    k.c = (t - st) / (t2 * t);
    k.e = 1.0 / t2 - st / (2.0 * t * (1.0 - ct));
    k.al = t * st / (2.0 * (1.0 - ct));
    k.bdot = -2.0 / (t2 * t2) + (1.0 + st / t) / (2.0 * t2 * (1.0 - ct));
  }
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
  double w[3];
  quat_log3(q, w);
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  SO3Coef k;
  so3_coef(t2, k);
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

EMPC_HD bool bad_number(double v) { return !(fabs(v) < 1e30); }  // NaN, inf or >= 1e30 (crocoddyl::raiseIfNaN)

}  // namespace empc
