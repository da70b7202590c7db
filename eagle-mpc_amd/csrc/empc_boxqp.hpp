// empc_boxqp.hpp -- crocoddyl::BoxQP::solve (core/solvers/box-qp.cpp, ~1.8) for one lane: the box-constrained QP behind
// SolverBoxDDP::computeGains / SolverBoxFDDP::computeGains, i.e. the `SolverBoxFDDP` / `SolverBoxDDP` back ends MpcAbstract
// accepts (include/eagle_mpc/mpc-base.hpp:36-47) and the reference's examples fall back to with useSquash = False.
//   min 1/2 x'Hx + q'x  s.t. lb <= x <= ub : projected Newton, active set from the sign of the gradient at the bounds,
//   Armijo backtracking over alpha = 2^-n.  Oracle: oracle/solver.hpp box_qp (same structure, same constants).
// The free block of H is factored as an M x M problem with the clamped dimensions decoupled (unit pivots), so that nothing
// is indexed through a run-time list.
#pragma once
#include "empc_dev_model.hpp"

namespace empc {

// EMPC_BOX_LDS: g, xnew, dx and the held free set in caller-provided memory `ws` (3 M doubles + M ints; LDS in the backward pass)
// ... and the QP is a real function (s_swappc) instead of ~8k instructions inlined into the knot loop: its register demand no
// longer adds to everything that is live across it (Qxx accumulators, prefetched record, offset tables), which is what put the
// box instantiation at 469 spilled registers / 1.5 KB of scratch per lane; the trajectories that take the plain gains (gaps
// still open) run the plain pass's code.  Measured statically (9-DoF): 0 spilled VGPRs, 20 B scratch (the call frame).
#if EMPC_BOX_LDS && defined(__HIPCC__)
#define EMPC_BOXQP_NOINLINE __attribute__((noinline))
#else
#define EMPC_BOXQP_NOINLINE
#endif
#if EMPC_BOX_LDS
#define EMPC_BOXQP_WORK(ws)   \
  double* const g = (ws);     \
  double* const xnew = (ws) + M; \
  double* const dx = (ws) + 2 * M; \
  int* const prev_mask = reinterpret_cast<int*>((ws) + 3 * M);
#define EMPC_BOXQP_WS_PARAM , double* ws
#else
#define EMPC_BOXQP_WORK(ws) \
  double g[M], xnew[M], dx[M]; \
  int prev_mask[M];
#define EMPC_BOXQP_WS_PARAM
#endif

// H: full M x M (row-major), symmetric positive definite on the free block.  x: in = warm start, out = solution.
// free_mask[i] = 1 for free components; Hinv: inverse of the free block, zero rows / columns for clamped components.
// Returns false when a factorisation fails (crocoddyl throws "backward_error").
#if EMPC_BOXQP_ONE_EXIT
// (commit b94f9cd: one exit, unrolled loops -- fewer spilled registers in the box backward pass; never run on hardware)
template <int M>
EMPC_BOXQP_NOINLINE EMPC_HD bool box_qp_lane(const double* H, const double* q, const double* lb, const double* ub, double* x, int* free_mask, double* Hinv,
                         int maxiter, double th_acceptstep, double th_grad, double reg EMPC_BOXQP_WS_PARAM) {
  // crocoddyl::BoxQP builds its own ten step lengths 2^-n whatever the outer solver's line search uses
  constexpr int n_alphas = 10;
  EMPC_BOXQP_WORK(ws)
  bool have_inv = false, ok = true;
#pragma unroll
  for (int i = 0; i < M; ++i) {
    x[i] = fmax(fmin(x[i], ub[i]), lb[i]);
    prev_mask[i] = -1;
  }
  // LLT of the free block (kept: the Newton step is one pair of triangular solves, crocoddyl's Hff_inv_llt_.solveInPlace);
  // the inverse itself (solution_.Hff_inv, which SolverBox*::computeGains multiplies with Qux) is formed at the exits only
  double L[M * (M + 1) / 2];
  auto factor_free = [&](const int* mask) {
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j)
        L[i * (i + 1) / 2 + j] = (mask[i] && mask[j]) ? H[i * M + j] + ((i == j) ? reg : 0.0) : ((i == j) ? 1.0 : 0.0);
    if (!chol_packed<M>(L)) return false;
#pragma unroll
    for (int i = 0; i < M; ++i) prev_mask[i] = mask[i];
    have_inv = true;
    return true;
  };
  auto invert_free = [&]() {
#pragma unroll
    for (int c = 0; c < M; ++c) {
      double col[M];
#pragma unroll
      for (int i = 0; i < M; ++i) col[i] = (i == c) ? 1.0 : 0.0;
      chol_solve_packed<M>(L, col);
#pragma unroll
      for (int i = 0; i < M; ++i) Hinv[i * M + c] = (prev_mask[i] && prev_mask[c]) ? col[i] : 0.0;
    }
  };
  auto fval = [&](const double* z) {
    double f = 0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      double a = 0;
#pragma unroll
      for (int j = 0; j < M; ++j) a += H[i * M + j] * z[j];
      f += 0.5 * z[i] * a + q[i] * z[i];
    }
    return f;
  };
  // One exit: every way out of the iteration ends in the same `invert_free` (one copy of its code; with the loops above
  // unrolled every array is indexed by compile-time constants and lives in registers -- rounds 2-3 kept them in scratch memory)
  bool want_inv = false;
  for (int k = 0; k < maxiter; ++k) {
    double gmax = 0;
    int nf = 0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      double a = q[i];
#pragma unroll
      for (int j = 0; j < M; ++j) a += H[i * M + j] * x[j];
      g[i] = a;
      gmax = fmax(gmax, fabs(a));
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const bool clamped = (x[j] == lb[j] && g[j] > 0.0) || (x[j] == ub[j] && g[j] < 0.0);
      free_mask[j] = clamped ? 0 : 1;
      nf += free_mask[j];
    }
    // the factorisation of an unchanged free set is the one already held
    bool same = have_inv;
#pragma unroll
    for (int i = 0; i < M; ++i) same = same && prev_mask[i] == free_mask[i];
    if (!same) ok = factor_free(free_mask);
    if (gmax <= th_grad || nf == 0) {
      want_inv = ok;
      break;
    }
    if (!ok) return false;
    // Newton step on the free space: dxf = -Hff^-1 (qf + Hfc xc) - xf (the clamped rows of the factor are unit pivots)
    {
      double r[M];
#pragma unroll
      for (int j = 0; j < M; ++j) {
        double a = -q[j];
#pragma unroll
        for (int c = 0; c < M; ++c)
          if (!free_mask[c]) a -= H[j * M + c] * x[c];
        r[j] = free_mask[j] ? a : 0.0;
      }
      chol_solve_packed<M>(L, r);
#pragma unroll
      for (int i = 0; i < M; ++i) dx[i] = free_mask[i] ? r[i] - x[i] : 0.0;
    }
    const double fold = fval(x);
    bool moved = false;
    for (int ia = 0; ia < n_alphas; ++ia) {
      const double alpha = ldexp(1.0, -ia);
#pragma unroll
      for (int i = 0; i < M; ++i) xnew[i] = fmax(fmin(x[i] + alpha * dx[i], ub[i]), lb[i]);
      const double fnew = fval(xnew);
      double gd = 0;
#pragma unroll
      for (int i = 0; i < M; ++i) gd += g[i] * (x[i] - xnew[i]);
      if (fold - fnew > th_acceptstep * gd) {
#pragma unroll
        for (int i = 0; i < M; ++i) {
          moved = moved || x[i] != xnew[i];
          x[i] = xnew[i];
        }
        break;
      }
    }
    // an iteration that left x where it was repeats itself (same gradient, same free set, same step) until maxiter: the
    // result is the one at hand (a control clamped with a non-zero multiplier keeps the gradient norm above th_grad for ever)
    want_inv = have_inv;  // (the exit by maxiter: whatever factor is held)
    if (!moved) break;
  }
  if (!ok) return false;
  if (want_inv) invert_free();
  return true;
}

#else
template <int M>
EMPC_BOXQP_NOINLINE EMPC_HD bool box_qp_lane(const double* H, const double* q, const double* lb, const double* ub, double* x, int* free_mask, double* Hinv,
                         int maxiter, double th_acceptstep, double th_grad, double reg EMPC_BOXQP_WS_PARAM) {
  // crocoddyl::BoxQP builds its own ten step lengths 2^-n whatever the outer solver's line search uses
  constexpr int n_alphas = 10;
  EMPC_BOXQP_WORK(ws)
  bool have_inv = false, ok = true;
#pragma unroll
  for (int i = 0; i < M; ++i) {
    x[i] = fmax(fmin(x[i], ub[i]), lb[i]);
    prev_mask[i] = -1;
  }
  // LLT of the free block (kept: the Newton step is one pair of triangular solves, crocoddyl's Hff_inv_llt_.solveInPlace);
  // the inverse itself (solution_.Hff_inv, which SolverBox*::computeGains multiplies with Qux) is formed at the exits only
  double L[M * (M + 1) / 2];
  auto factor_free = [&](const int* mask) {
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j)
        L[i * (i + 1) / 2 + j] = (mask[i] && mask[j]) ? H[i * M + j] + ((i == j) ? reg : 0.0) : ((i == j) ? 1.0 : 0.0);
    if (!chol_packed<M>(L)) return false;
#pragma unroll
    for (int i = 0; i < M; ++i) prev_mask[i] = mask[i];
    have_inv = true;
    return true;
  };
  auto invert_free = [&]() {
    for (int c = 0; c < M; ++c) {
      double col[M];
#pragma unroll
      for (int i = 0; i < M; ++i) col[i] = (i == c) ? 1.0 : 0.0;
      chol_solve_packed<M>(L, col);
#pragma unroll
      for (int i = 0; i < M; ++i) Hinv[i * M + c] = (prev_mask[i] && prev_mask[c]) ? col[i] : 0.0;
    }
  };
  auto fval = [&](const double* z) {
    double f = 0;
    for (int i = 0; i < M; ++i) {
      double a = 0;
#pragma unroll
      for (int j = 0; j < M; ++j) a += H[i * M + j] * z[j];
      f += 0.5 * z[i] * a + q[i] * z[i];
    }
    return f;
  };
  for (int k = 0; k < maxiter; ++k) {
    double gmax = 0;
    int nf = 0;
    for (int i = 0; i < M; ++i) {
      double a = q[i];
#pragma unroll
      for (int j = 0; j < M; ++j) a += H[i * M + j] * x[j];
      g[i] = a;
      gmax = fmax(gmax, fabs(a));
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const bool clamped = (x[j] == lb[j] && g[j] > 0.0) || (x[j] == ub[j] && g[j] < 0.0);
      free_mask[j] = clamped ? 0 : 1;
      nf += free_mask[j];
    }
    if (gmax <= th_grad || nf == 0) {
      bool same = have_inv;
#pragma unroll
      for (int i = 0; i < M; ++i) same = same && prev_mask[i] == free_mask[i];
      if (!same) ok = factor_free(free_mask);
      if (ok) invert_free();
      return ok;
    }
    {
      // the factorisation of an unchanged free set is the one already held
      bool same = have_inv;
#pragma unroll
      for (int i = 0; i < M; ++i) same = same && prev_mask[i] == free_mask[i];
      if (!same && !factor_free(free_mask)) return false;
    }
    // Newton step on the free space: dxf = -Hff^-1 (qf + Hfc xc) - xf (the clamped rows of the factor are unit pivots)
    {
      double r[M];
#pragma unroll
      for (int j = 0; j < M; ++j) {
        double a = -q[j];
#pragma unroll
        for (int c = 0; c < M; ++c)
          if (!free_mask[c]) a -= H[j * M + c] * x[c];
        r[j] = free_mask[j] ? a : 0.0;
      }
      chol_solve_packed<M>(L, r);
#pragma unroll
      for (int i = 0; i < M; ++i) dx[i] = free_mask[i] ? r[i] - x[i] : 0.0;
    }
    const double fold = fval(x);
    bool moved = false;
    for (int ia = 0; ia < n_alphas; ++ia) {
      const double alpha = ldexp(1.0, -ia);
#pragma unroll
      for (int i = 0; i < M; ++i) xnew[i] = fmax(fmin(x[i] + alpha * dx[i], ub[i]), lb[i]);
      const double fnew = fval(xnew);
      double gd = 0;
#pragma unroll
      for (int i = 0; i < M; ++i) gd += g[i] * (x[i] - xnew[i]);
      if (fold - fnew > th_acceptstep * gd) {
#pragma unroll
        for (int i = 0; i < M; ++i) {
          moved = moved || x[i] != xnew[i];
          x[i] = xnew[i];
        }
        break;
      }
    }
    // an iteration that left x where it was repeats itself (same gradient, same free set, same step) until maxiter: the
    // result is the one at hand (a control clamped with a non-zero multiplier keeps the gradient norm above th_grad for ever)
    if (!moved) {
      invert_free();
      return true;
    }
  }
  if (have_inv) invert_free();
  return true;
}

#endif

}  // namespace empc
