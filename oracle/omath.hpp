// ORACLE (test infrastructure only -- never linked into or called from the product path).
//
// omath.hpp: scalar/dual arithmetic and SO(3)/SE(3) maps used by the CPU restatement.
// The Lie-group conventions restate Pinocchio 2.x as used by crocoddyl::StateMultibody
// (reference call sites src/trajectory.cpp:47, src/sbfddp.cpp:430):
// spatial order [linear; angular], right-perturbation Jacobians, quaternion xyzw.
// Parity status: UNPINNED (the reference ships no tests or golden vectors; Pinocchio/Crocoddyl are
// not vendored).  Every closed form here is checked by finite differences in tests/test_oracle_math.py.
#pragma once
#include <cmath>
#include <cstring>

namespace oracle {

// ----------------------------------------------------------------------------------------------
// Forward-mode dual number with ND tangent directions (vector mode).
// ----------------------------------------------------------------------------------------------
constexpr int ND = 24;  // >= 2*nv for every supported robot (tilt5: 22)

struct Dual {
  double v;
  double d[ND];
  Dual() : v(0) { std::memset(d, 0, sizeof(d)); }
  Dual(double x) : v(x) { std::memset(d, 0, sizeof(d)); }
};

inline Dual operator+(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v + b.v;
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
inline Dual operator-(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v - b.v;
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
inline Dual operator-(const Dual& a) {
  Dual r;
  r.v = -a.v;
  for (int i = 0; i < ND; ++i) r.d[i] = -a.d[i];
  return r;
}
inline Dual operator*(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v * b.v;
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
inline Dual operator*(double a, const Dual& b) {
  Dual r;
  r.v = a * b.v;
  for (int i = 0; i < ND; ++i) r.d[i] = a * b.d[i];
  return r;
}
inline Dual operator*(const Dual& b, double a) { return a * b; }
inline Dual operator+(const Dual& a, double b) {
  Dual r = a;
  r.v += b;
  return r;
}
inline Dual operator+(double b, const Dual& a) { return a + b; }
inline Dual operator-(const Dual& a, double b) {
  Dual r = a;
  r.v -= b;
  return r;
}
inline Dual operator-(double b, const Dual& a) { return (-a) + b; }
inline Dual& operator+=(Dual& a, const Dual& b) {
  a.v += b.v;
  for (int i = 0; i < ND; ++i) a.d[i] += b.d[i];
  return a;
}
inline Dual& operator-=(Dual& a, const Dual& b) {
  a.v -= b.v;
  for (int i = 0; i < ND; ++i) a.d[i] -= b.d[i];
  return a;
}
inline Dual dsin(const Dual& a) {
  Dual r;
  r.v = std::sin(a.v);
  const double c = std::cos(a.v);
  for (int i = 0; i < ND; ++i) r.d[i] = c * a.d[i];
  return r;
}
inline Dual dcos(const Dual& a) {
  Dual r;
  r.v = std::cos(a.v);
  const double s = -std::sin(a.v);
  for (int i = 0; i < ND; ++i) r.d[i] = s * a.d[i];
  return r;
}
inline double dsin(double a) { return std::sin(a); }
inline double dcos(double a) { return std::cos(a); }
inline double val(double a) { return a; }
inline double val(const Dual& a) { return a.v; }

// ----------------------------------------------------------------------------------------------
// 3-vectors / 3x3 matrices (row-major), generic in the scalar.
// ----------------------------------------------------------------------------------------------
template <class S, class A, class B>
inline void cross3(const A* a, const B* b, S* r) {
  S r0 = a[1] * b[2] - a[2] * b[1];
  S r1 = a[2] * b[0] - a[0] * b[2];
  S r2 = a[0] * b[1] - a[1] * b[0];
  r[0] = r0;
  r[1] = r1;
  r[2] = r2;
}
template <class S, class A, class B>
inline S dot3(const A* a, const B* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
// r = M v
template <class S, class A, class B>
inline void matvec3(const A* M, const B* v, S* r) {
  S r0 = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
  S r1 = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
  S r2 = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  r[0] = r0;
  r[1] = r1;
  r[2] = r2;
}
// r = M^T v
template <class S, class A, class B>
inline void matTvec3(const A* M, const B* v, S* r) {
  S r0 = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
  S r1 = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
  S r2 = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
  r[0] = r0;
  r[1] = r1;
  r[2] = r2;
}
// R = A B
template <class S, class A, class B>
inline void matmul3(const A* a, const B* b, S* r) {
  S t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  for (int i = 0; i < 9; ++i) r[i] = t[i];
}
// R = A^T B
template <class S, class A, class B>
inline void matTmul3(const A* a, const B* b, S* r) {
  S t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) t[3 * i + j] = a[i] * b[j] + a[3 + i] * b[3 + j] + a[6 + i] * b[6 + j];
  for (int i = 0; i < 9; ++i) r[i] = t[i];
}
inline void skew3(const double* w, double* M) {
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

// quaternion (x,y,z,w) -> rotation matrix
inline void quat_to_R(const double* q, double* R) {
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
// Hamilton product r = a (x) b, xyzw
inline void quat_mul(const double* a, const double* b, double* r) {
  const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  const double y = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  const double z = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  r[0] = x;
  r[1] = y;
  r[2] = z;
  r[3] = w;
}
inline void quat_conj(const double* a, double* r) {
  r[0] = -a[0];
  r[1] = -a[1];
  r[2] = -a[2];
  r[3] = a[3];
}
inline void quat_normalize(double* q) {
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] /= n;
}
// rotate a vector by a unit quaternion: R(q) v
inline void quat_rotate(const double* q, const double* v, double* r) {
  double R[9];
  quat_to_R(q, R);
  matvec3<double>(R, v, r);
}

// quaternion exponential of a rotation vector: (sin(t/2)/t w, cos(t/2))
inline void quat_exp3(const double* w, double* q) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double t = std::sqrt(t2);
  double k;  // sin(t/2)/t
  if (t < 1e-4)
    k = 0.5 - t2 / 48.0 + t2 * t2 / 3840.0;
  else
    k = std::sin(0.5 * t) / t;
  q[0] = k * w[0];
  q[1] = k * w[1];
  q[2] = k * w[2];
  q[3] = std::cos(0.5 * t);
}
// rotation vector of a unit quaternion (angle in [0, pi])
inline void quat_log3(const double* q, double* w) {
  const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  const double n = std::sqrt(n2);
  const double sgn = q[3] >= 0 ? 1.0 : -1.0;
  const double aw = std::fabs(q[3]);
  double k;  // theta / |vec|
  if (n < 1e-6) {
    // theta = 2 atan2(n, w) = 2 n / w (1 - n^2/(3 w^2) + ...)
    k = 2.0 / aw * (1.0 - n2 / (3.0 * aw * aw));
  } else {
    k = 2.0 * std::atan2(n, aw) / n;
  }
  k *= sgn;
  w[0] = k * q[0];
  w[1] = k * q[1];
  w[2] = k * q[2];
}

// exp3 as a rotation matrix (Rodrigues)
inline void exp3(const double* w, double* R) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double t = std::sqrt(t2);
  double a, b;  // sin t / t, (1 - cos t)/t^2
  if (t < 1e-4) {
    a = 1.0 - t2 / 6.0 + t2 * t2 / 120.0;
    b = 0.5 - t2 / 24.0 + t2 * t2 / 720.0;
  } else {
    a = std::sin(t) / t;
    b = (1.0 - std::cos(t)) / t2;
  }
  double K[9];
  skew3(w, K);
  double K2[9];
  matmul3<double>(K, K, K2);
  for (int i = 0; i < 9; ++i) R[i] = a * K[i] + b * K2[i];
  R[0] += 1;
  R[4] += 1;
  R[8] += 1;
}

// log3 of a rotation matrix (through a quaternion; robust near pi is not needed by the OCPs here,
// the extraction below picks the largest pivot so it stays accurate for any angle)
inline void R_to_quat(const double* R, double* q) {
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    double s = std::sqrt(tr + 1.0) * 2;
    q[3] = 0.25 * s;
    q[0] = (R[7] - R[5]) / s;
    q[1] = (R[2] - R[6]) / s;
    q[2] = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    double s = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
    q[3] = (R[7] - R[5]) / s;
    q[0] = 0.25 * s;
    q[1] = (R[1] + R[3]) / s;
    q[2] = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    double s = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
    q[3] = (R[2] - R[6]) / s;
    q[0] = (R[1] + R[3]) / s;
    q[1] = 0.25 * s;
    q[2] = (R[5] + R[7]) / s;
  } else {
    double s = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
    q[3] = (R[3] - R[1]) / s;
    q[0] = (R[2] + R[6]) / s;
    q[1] = (R[5] + R[7]) / s;
    q[2] = 0.25 * s;
  }
  quat_normalize(q);
}
inline void log3(const double* R, double* w) {
  double q[4];
  R_to_quat(R, q);
  quat_log3(q, w);
}

// Right Jacobian of SO(3): exp(w + d) ~ exp(w) exp(Jr d)
inline void Jexp3(const double* w, double* J) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double t = std::sqrt(t2);
  double b, c;  // (1-cos t)/t^2, (t - sin t)/t^3
  if (t < 1e-4) {
    b = 0.5 - t2 / 24.0 + t2 * t2 / 720.0;
    c = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0;
  } else {
    b = (1.0 - std::cos(t)) / t2;
    c = (t - std::sin(t)) / (t2 * t);
  }
  double K[9], K2[9];
  skew3(w, K);
  matmul3<double>(K, K, K2);
  for (int i = 0; i < 9; ++i) J[i] = -b * K[i] + c * K2[i];
  J[0] += 1;
  J[4] += 1;
  J[8] += 1;
}
// Inverse right Jacobian (Pinocchio's Jlog3): log(R exp(d)) ~ log(R) + Jlog3 d, with w = log(R)
inline void Jlog3(const double* w, double* J) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double t = std::sqrt(t2);
  double e;  // 1/t^2 - (1+cos t)/(2 t sin t)
  if (t < 1e-3) {
    e = 1.0 / 12.0 + t2 / 720.0 + t2 * t2 / 30240.0;
  } else {
    e = 1.0 / t2 - (1.0 + std::cos(t)) / (2.0 * t * std::sin(t));
  }
  double K[9], K2[9];
  skew3(w, K);
  matmul3<double>(K, K, K2);
  for (int i = 0; i < 9; ++i) J[i] = 0.5 * K[i] + e * K2[i];
  J[0] += 1;
  J[4] += 1;
  J[8] += 1;
}

// SE(3): exp6 of a twist [v; w] -> (R, p)
inline void exp6(const double* xi, double* R, double* p) {
  const double* v = xi;
  const double* w = xi + 3;
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double t = std::sqrt(t2);
  double a, b, c;  // sin t/t, (1-cos t)/t^2, (t - sin t)/t^3
  if (t < 1e-4) {
    a = 1.0 - t2 / 6.0 + t2 * t2 / 120.0;
    b = 0.5 - t2 / 24.0 + t2 * t2 / 720.0;
    c = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0;
  } else {
    a = std::sin(t) / t;
    b = (1.0 - std::cos(t)) / t2;
    c = (t - std::sin(t)) / (t2 * t);
  }
  exp3(w, R);
  double wxv[3];
  cross3<double>(w, v, wxv);
  const double wv = dot3<double>(w, v);
  // p = V v with V = I + b [w]x + c [w]x^2 = a v + c (w.v) w + b (w x v)
  for (int i = 0; i < 3; ++i) p[i] = a * v[i] + c * wv * w[i] + b * wxv[i];
}

// log6 of (R, p) -> [v; w]
inline void log6(const double* R, const double* p, double* xi) {
  double w[3];
  log3(R, w);
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double t = std::sqrt(t2);
  double al, be;
  if (t < 1e-3) {
    al = 1.0 - t2 / 12.0 - t2 * t2 / 720.0;
    be = 1.0 / 12.0 + t2 / 720.0 + t2 * t2 / 30240.0;
  } else {
    const double st = std::sin(t), ct = std::cos(t);
    al = t * st / (2.0 * (1.0 - ct));
    be = 1.0 / t2 - st / (2.0 * t * (1.0 - ct));
  }
  double wxp[3];
  cross3<double>(w, p, wxp);
  const double wp = dot3<double>(w, p);
  for (int i = 0; i < 3; ++i) {
    xi[i] = al * p[i] - 0.5 * wxp[i] + be * wp * w[i];
    xi[3 + i] = w[i];
  }
}

// 6x6 adjoint action matrix of (R,p) on motions, order [lin; ang]:  Ad = [[R, [p]x R],[0, R]]
inline void adjoint6(const double* R, const double* p, double* A) {
  double P[9], PR[9];
  skew3(p, P);
  matmul3<double>(P, R, PR);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      A[6 * i + j] = R[3 * i + j];
      A[6 * i + 3 + j] = PR[3 * i + j];
      A[6 * (3 + i) + j] = 0;
      A[6 * (3 + i) + 3 + j] = R[3 * i + j];
    }
}
// adjoint of the inverse placement: [[R^T, -R^T [p]x],[0, R^T]]
inline void adjoint6_inv(const double* R, const double* p, double* A) {
  double P[9], RtP[9];
  skew3(p, P);
  matTmul3<double>(R, P, RtP);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      A[6 * i + j] = R[3 * j + i];
      A[6 * i + 3 + j] = -RtP[3 * i + j];
      A[6 * (3 + i) + j] = 0;
      A[6 * (3 + i) + 3 + j] = R[3 * j + i];
    }
}

// Right Jacobian of SE(3) (Pinocchio Jexp6): exp6(xi + d) ~ exp6(xi) exp6(J d); J = [[Jr, Q],[0, Jr]].
// Q is Barfoot's closed form evaluated at -xi (right Jacobian = left Jacobian at -xi).
inline void Jexp6(const double* xi, double* J) {
  double Jr[9];
  Jexp3(xi + 3, Jr);
  double rho[3] = {-xi[0], -xi[1], -xi[2]};
  double phi[3] = {-xi[3], -xi[4], -xi[5]};
  const double t2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const double t = std::sqrt(t2);
  // Barfoot, "State Estimation for Robotics", eq. (7.86):
  //   c1 = (t - sin t)/t^3, c2 = (t^2 + 2 cos t - 2)/(2 t^4), c3 = (2t - 3 sin t + t cos t)/(2 t^5)
  double c1, c2, c3;
  if (t < 0.1) {
    const double t4 = t2 * t2, t6 = t4 * t2;
    c1 = 1.0 / 6.0 - t2 / 120.0 + t4 / 5040.0 - t6 / 362880.0;
    c2 = 1.0 / 24.0 - t2 / 720.0 + t4 / 40320.0 - t6 / 3628800.0;
    c3 = 1.0 / 120.0 - t2 / 2520.0 + t4 / 120960.0;
  } else {
    const double st = std::sin(t), ct = std::cos(t);
    c1 = (t - st) / (t2 * t);
    c2 = (0.5 * t2 + ct - 1.0) / (t2 * t2);
    c3 = (2.0 * t - 3.0 * st + t * ct) / (2.0 * t2 * t2 * t);
  }
  double P[9], W[9];
  skew3(rho, P);
  skew3(phi, W);
  double WP[9], PW[9], WPW[9], WWP[9], PWW[9], WPWW[9], WWPW[9];
  matmul3<double>(W, P, WP);
  matmul3<double>(P, W, PW);
  matmul3<double>(WP, W, WPW);
  matmul3<double>(W, WP, WWP);
  matmul3<double>(PW, W, PWW);
  matmul3<double>(WPW, W, WPWW);
  matmul3<double>(W, WPW, WWPW);
  double Q[9];
  for (int i = 0; i < 9; ++i)
    Q[i] = 0.5 * P[i] + c1 * (WP[i] + PW[i] + WPW[i]) + c2 * (WWP[i] + PWW[i] - 3.0 * WPW[i]) +
           c3 * (WPWW[i] + WWPW[i]);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      J[6 * i + j] = Jr[3 * i + j];
      J[6 * i + 3 + j] = Q[3 * i + j];
      J[6 * (3 + i) + j] = 0;
      J[6 * (3 + i) + 3 + j] = Jr[3 * i + j];
    }
}

// Jlog6 with xi = log6(M): inverse of Jexp6(xi):  [[Jl, -Jl Q Jl],[0, Jl]],  Jl = Jlog3(w)
inline void Jlog6(const double* xi, double* J) {
  double Je[36];
  Jexp6(xi, Je);
  double Jl[9];
  Jlog3(xi + 3, Jl);
  double Q[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Q[3 * i + j] = Je[6 * i + 3 + j];
  double JQ[9], JQJ[9];
  matmul3<double>(Jl, Q, JQ);
  matmul3<double>(JQ, Jl, JQJ);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      J[6 * i + j] = Jl[3 * i + j];
      J[6 * i + 3 + j] = -JQJ[3 * i + j];
      J[6 * (3 + i) + j] = 0;
      J[6 * (3 + i) + 3 + j] = Jl[3 * i + j];
    }
}

// dense helpers (row-major)
inline void matmul(const double* A, const double* B, double* C, int n, int k, int m) {  // C[n x m] = A[n x k] B[k x m]
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      double s = 0;
      for (int l = 0; l < k; ++l) s += A[i * k + l] * B[l * m + j];
      C[i * m + j] = s;
    }
}

// Cholesky LLT in place on the lower triangle of an n x n row-major matrix; returns false if not PD.
inline bool cholesky(double* A, int n) {
  for (int j = 0; j < n; ++j) {
    double s = A[j * n + j];
    for (int k = 0; k < j; ++k) s -= A[j * n + k] * A[j * n + k];
    if (!(s > 0.0) || !std::isfinite(s)) return false;
    const double d = std::sqrt(s);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double t = A[i * n + j];
      for (int k = 0; k < j; ++k) t -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = t / d;
    }
  }
  return true;
}
// The same factorisation for the constraint matrix Jc M^-1 Jc^T of a stage with SEVERAL contacts, which can be rank deficient
// (two point contacts on a stretched chain: the rows along the chain coincide -- and crocoddyl's default initial guess, every
// knot at the zero state, is that configuration).  Eigen's LLT (pinocchio::forwardDynamics) stops at the first non-positive
// pivot and leaves the rest of the matrix untouched, as cholesky() above does; the solves that follow give the redundant row a
// multiplier of ~0.  But on an exactly singular matrix that pivot is rounding noise whose SIGN depends on the summation order:
// with a tiny positive one the factorisation runs on and the redundant force component is split arbitrarily between the
// rows (5.6 + 8.0 instead of 13.6 + 0 on the eagle_catch pair; the accelerations are the same, the problem does not determine
// the split).  Stated deviation: a pivot below 1e-13 of its diagonal entry counts as non-positive, so that the oracle, its FMA
// build and the device code (chol_packed_stop, empc_dev_model.hpp) all take the branch the reference takes half of the time.
inline bool cholesky_rank_deficient(double* A, int n) {
  for (int j = 0; j < n; ++j) {
    const double raw = A[j * n + j];
    double s = raw;
    for (int k = 0; k < j; ++k) s -= A[j * n + k] * A[j * n + k];
    if (!(s > 1e-13 * raw) || !std::isfinite(s)) return false;
    const double d = std::sqrt(s);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double t = A[i * n + j];
      for (int k = 0; k < j; ++k) t -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = t / d;
    }
  }
  return true;
}
// solve L L^T x = b in place
inline void cholesky_solve(const double* L, int n, double* b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * n + k] * b[k];
    b[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * b[k];
    b[i] = s / L[i * n + i];
  }
}

}  // namespace oracle
