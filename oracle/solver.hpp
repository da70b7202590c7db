// ORACLE (test infrastructure only -- never linked into or called from the product path).
//
// solver.hpp: scalar, single-trajectory restatement of SolverSbFDDP.
//   SolverSbFDDP::solve / solveFDDP / solveDDP / forwardPassDDP / expectedImprovementDDP /
//   barrierInit / barrierUpdate / squashingUpdate / fillSquashedOutputs  -> reference src/sbfddp.cpp (whole file)
//   crocoddyl::SolverDDP::{calcDiff, backwardPass, computeGains, increase/decreaseRegularization, setCandidate},
//   crocoddyl::SolverFDDP::{updateExpectedImprovement, expectedImprovement, forwardPass}
//                                                   -> SURVEY.md Appendix A.1/A.2 (un-vendored Crocoddyl fork)
// Parity status: UNPINNED.  The fork-only stopping functions (sbfddp.cpp:27-31,301,309,379,387) are
// selectable through EmpcSolverParams (assumption register SURVEY A.8 U1..U5).
#pragma once
#include <cmath>
#include <vector>

#include "action.hpp"

namespace oracle {

struct IterRecord {
  double phase, iter, cost, stop, xreg, steplength, feasible, dV, dVexp, gapnorm, d0, d1;
};

inline bool bad_number(double v) { return std::isnan(v) || std::isinf(v) || std::fabs(v) >= 1e30; }

// Candidate and solver scalars at the TOP of one iteration of solveFDDP / solveDDP (before computeDirection), with the
// outcome of that iteration filled in when it is over.  Recorded on request (Solver::record_iterates) for the
// teacher-forced parity tests: the GPU solver is put at exactly this iterate and must reproduce the iteration.
struct Iterate {
  std::vector<double> xs, us;  // (T+1) x nx, T x nu
  std::vector<double> k;       // T x nu: feed-forward terms of the previous iteration (warm start of the box QPs)
  int phase, iter, is_feasible, was_feasible, recalc;
  double xreg, smooth, th_stop, cost, cost_prev;
  int trace_index;     // index of this iteration's IterRecord in Solver::trace, -1 when it ended without one
  int accepted_alpha;  // index n of the accepted step length 2^-n, -1 = none
  int ended;           // the pass ended with this iteration
  int returned;        // ... because the stopping test passed (solveFDDP / solveDDP returned true)
};

struct Solver {
  Problem P;
  int T, nx, ndx, nu;
  std::vector<double> x0;
  std::vector<std::vector<double>> xs, us, xs_try, us_try, fs, Vxx, Vx, Qxx, Qxu, Quu, Qx, Qu, K, k, Quuk, dx;
  std::vector<std::vector<double>> us_lastcalc;  // control of the last IAM.calc at every running node
  std::vector<NodeData> datas;
  std::vector<double> alphas;
  double cost, cost_prev, cost_try, xreg, ureg, steplength, dV, dVexp, d[2], dg, dq, dv, stop, th_stop;
  bool is_feasible, was_feasible;
  int iter, total_iters, status;
  std::vector<IterRecord> trace;
  int phase;
  bool record_iterates = false;
  std::vector<Iterate> iterates;

  void open_iterate(bool recalc) {
    if (!record_iterates) return;
    Iterate it;
    it.xs.reserve((size_t)(T + 1) * nx);
    for (int t = 0; t <= T; ++t) it.xs.insert(it.xs.end(), xs[t].begin(), xs[t].end());
    it.us.reserve((size_t)T * nu);
    for (int t = 0; t < T; ++t) it.us.insert(it.us.end(), us[t].begin(), us[t].end());
    for (int t = 0; t < T; ++t) it.k.insert(it.k.end(), k[t].begin(), k[t].end());
    it.phase = phase;
    it.iter = iter;
    it.is_feasible = is_feasible ? 1 : 0;
    it.was_feasible = was_feasible ? 1 : 0;
    it.recalc = recalc ? 1 : 0;
    it.xreg = xreg;
    it.smooth = P.smooth;
    it.th_stop = th_stop;
    it.cost = cost;
    it.cost_prev = cost_prev;
    it.trace_index = -1;
    it.accepted_alpha = -1;
    it.ended = 0;
    it.returned = 0;
    iterates.push_back(std::move(it));
  }
  void close_iterate(int accepted_alpha, bool recorded, bool ended, bool returned) {
    if (!record_iterates || iterates.empty()) return;
    Iterate& it = iterates.back();
    it.accepted_alpha = accepted_alpha;
    it.trace_index = recorded ? (int)trace.size() - 1 : -1;
    it.ended = ended ? 1 : 0;
    it.returned = returned ? 1 : 0;
  }

  // SolverSbFDDP ctor (src/sbfddp.cpp:5-38) + barrierInit (:169-190)
  void init(const EmpcProblemDesc& desc, const EmpcSolverParams& prm) {
    const EmpcProblemDesc& d_ = desc;
    P.d = d_;
    P.prm = prm;
    P.sets.assign(d_.sets, d_.sets + d_.n_sets);
    P.knot_set.assign(d_.knot_set, d_.knot_set + d_.T + 1);
    P.d.sets = nullptr;
    P.d.knot_set = nullptr;
    P.smooth = prm.smooth_init;
    T = d_.T;
    nx = d_.nx;
    ndx = d_.ndx;
    nu = d_.nu;
    x0.assign(d_.x0, d_.x0 + nx);
    if (prm.solver_type == EMPC_SOLVER_SBFDDP) barrier_init();  // SolverSbFDDP's own cost; the crocoddyl solvers add nothing
    auto mk = [&](std::vector<std::vector<double>>& v, int n, int sz) { v.assign(n, std::vector<double>(sz, 0.0)); };
    mk(xs, T + 1, nx);
    mk(xs_try, T + 1, nx);
    mk(us, T, nu);
    mk(us_try, T, nu);
    mk(us_lastcalc, T, nu);
    mk(fs, T + 1, ndx);
    mk(dx, T + 1, ndx);
    mk(Vxx, T + 1, ndx * ndx);
    mk(Vx, T + 1, ndx);
    mk(Qxx, T, ndx * ndx);
    mk(Qxu, T, ndx * nu);
    mk(Quu, T, nu * nu);
    mk(Qx, T, ndx);
    mk(Qu, T, nu);
    mk(K, T, nu * ndx);
    mk(k, T, nu);
    mk(Quuk, T, nu);
    datas.resize(T + 1);
    alphas.resize(prm.n_alphas);
    for (int n = 0; n < prm.n_alphas; ++n) alphas[n] = 1.0 / std::pow(2.0, (double)n);
    cost = cost_prev = cost_try = 0;
    stop = 0;
    xreg = ureg = prm.reg_init;
    is_feasible = was_feasible = false;
    steplength = 1;
    dV = dVexp = 0;
    d[0] = d[1] = 0;
    dg = dq = dv = 0;
    iter = total_iters = status = 0;
    th_stop = prm.convergence_init;
    phase = 0;
  }

  // Add the "barrier" cost to the CostModelSum of every distinct running model (sbfddp.cpp:181-186).
  void barrier_init() {
    std::vector<char> done(P.sets.size(), 0);
    for (int t = 0; t < T; ++t) {
      const int si = P.knot_set[t];
      if (done[si]) continue;
      done[si] = 1;
      EmpcCostSet& s = P.sets[si];
      bool found = false;
      for (int i = 0; i < s.ncosts; ++i)
        if (std::strcmp(s.costs[i].name, "barrier") == 0) found = true;
      if (found) continue;
      EmpcCost c;
      std::memset(&c, 0, sizeof(c));
      std::strcpy(c.name, "barrier");
      c.type = EMPC_COST_CONTROL;
      c.activation = EMPC_ACT_WEIGHTED_QUADRATIC_BARRIER;
      c.active = 1;
      c.frame = -1;
      c.nr = nu;
      c.is_barrier = 1;
      c.weight = P.prm.barrier_weight;
      for (int i = 0; i < nu; ++i) {
        c.ref[i] = 0;
        c.lb[i] = P.d.u_lb[i];  // s_lb = u_lb, s_ub = u_ub (SquashingModelSmoothSat)
        c.ub[i] = P.d.u_ub[i];
        c.act_w[i] = 1.0;
      }
      // keep the table sorted by name (std::map order)
      int pos = 0;
      while (pos < s.ncosts && std::strcmp(s.costs[pos].name, "barrier") < 0) ++pos;
      for (int i = s.ncosts; i > pos; --i) s.costs[i] = s.costs[i - 1];
      s.costs[pos] = c;
      s.ncosts++;
    }
    barrier_update(P.prm.smooth_init);
  }
  // sbfddp.cpp:464-477
  void barrier_update(double smooth) {
    for (auto& s : P.sets)
      for (int i = 0; i < s.ncosts; ++i)
        if (s.costs[i].is_barrier) {
          for (int j = 0; j < nu; ++j) {
            const double aux = smooth * (P.d.u_ub[j] - P.d.u_lb[j]);
            s.costs[i].act_w[j] = 1.0 / (aux * aux);
          }
          s.costs[i].weight = P.prm.barrier_weight;
        }
  }

  void calc_node(int t, const double* x, const double* u, bool diff, NodeData& D) {
    node_calc(P, t, x, u, diff, D);
    if (u && t < T) us_lastcalc[t].assign(u, u + nu);
  }

  void set_candidate(const double* xs_in, const double* us_in, bool feasible) {
    for (int t = 0; t <= T; ++t) {
      if (xs_in)
        xs[t].assign(xs_in + t * nx, xs_in + (t + 1) * nx);
      else
        state_zero(P, xs[t].data());
    }
    for (int t = 0; t < T; ++t) {
      if (us_in)
        us[t].assign(us_in + t * nu, us_in + (t + 1) * nu);
      else
        us[t].assign(nu, 0.0);
    }
    is_feasible = feasible;
  }

  void increase_reg() {
    xreg *= P.prm.reg_incfactor;
    if (xreg > P.prm.reg_max) xreg = P.prm.reg_max;
    ureg = xreg;
  }
  void decrease_reg() {
    xreg /= P.prm.reg_decfactor;
    if (xreg < P.prm.reg_min) xreg = P.prm.reg_min;
    ureg = xreg;
  }

  // SolverDDP::calcDiff (A.2).  iter==0 runs problem.calc first; re-evaluating calc inside calcDiff gives the same values (U5).
  void calc_diff() {
    cost = 0;
    for (int t = 0; t < T; ++t) {
      calc_node(t, xs[t].data(), us[t].data(), true, datas[t]);
      cost += datas[t].cost;
    }
    calc_node(T, xs[T].data(), nullptr, true, datas[T]);
    cost += datas[T].cost;
    if (!is_feasible) {
      state_diff(P, xs[0].data(), x0.data(), fs[0].data());
      bool could = true;
      auto inf_norm = [&](const std::vector<double>& v) {
        double m = 0;
        for (double e : v) m = std::max(m, std::fabs(e));
        return m;
      };
      if (inf_norm(fs[0]) >= P.prm.th_gaptol) could = false;
      for (int t = 0; t < T; ++t) {
        state_diff(P, xs[t + 1].data(), datas[t].xnext, fs[t + 1].data());
        if (could && inf_norm(fs[t + 1]) >= P.prm.th_gaptol) could = false;
      }
      is_feasible = could;
    } else if (!was_feasible) {
      for (auto& f : fs) std::fill(f.begin(), f.end(), 0.0);
    }
  }

  // crocoddyl::BoxQP::solve (core/solvers/box-qp.cpp, ~1.8): projected Newton on min 1/2 x'Hx + q'x, lb <= x <= ub, warm
  // started at xinit.  Outputs x, the free set (mask) and the inverse of the free block of H (embedded in an m x m matrix
  // with zero clamped rows / columns).  Returns false when a factorisation fails.
  bool box_qp(const double* H, const double* q, const double* lb, const double* ub, const double* xinit, int m, double* x,
              int* free_mask, double* Hinv) {
    double g[NU], xnew[NU], dx[NU];
    int prev_mask[NU];
    bool have_inv = false;
    for (int i = 0; i < m; ++i) {
      x[i] = std::max(std::min(xinit[i], ub[i]), lb[i]);
      prev_mask[i] = -1;
    }
    // LLT of the free block, as an m x m problem with the clamped dimensions decoupled (unit pivots): same arithmetic on the
    // free entries as factoring Hff alone.  The factor is kept: the Newton step is a pair of triangular solves (crocoddyl's
    // Hff_inv_llt_.solveInPlace); the inverse (solution_.Hff_inv) is formed at the exits.
    double Hm[NU * NU];
    auto factor_free = [&](const int* mask) {
      for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j)
          Hm[i * m + j] = (mask[i] && mask[j]) ? H[i * m + j] + ((i == j) ? P.prm.boxqp_reg : 0.0) : ((i == j) ? 1.0 : 0.0);
      if (!cholesky(Hm, m)) return false;
      for (int i = 0; i < m; ++i) prev_mask[i] = mask[i];
      have_inv = true;
      return true;
    };
    auto invert_free = [&]() {
      for (int c = 0; c < m; ++c) {
        double col[NU];
        for (int i = 0; i < m; ++i) col[i] = (i == c) ? 1.0 : 0.0;
        cholesky_solve(Hm, m, col);
        for (int i = 0; i < m; ++i) Hinv[i * m + c] = (prev_mask[i] && prev_mask[c]) ? col[i] : 0.0;
      }
    };
    for (int k = 0; k < P.prm.boxqp_maxiter; ++k) {
      double gmax = 0;
      int nf = 0;
      for (int i = 0; i < m; ++i) {
        double a = q[i];
        for (int j = 0; j < m; ++j) a += H[i * m + j] * x[j];
        g[i] = a;
        gmax = std::max(gmax, std::fabs(a));
      }
      for (int j = 0; j < m; ++j) {
        const bool clamped = (x[j] == lb[j] && g[j] > 0.0) || (x[j] == ub[j] && g[j] < 0.0);
        free_mask[j] = clamped ? 0 : 1;
        nf += free_mask[j];
      }
      if (gmax <= P.prm.boxqp_th_grad || nf == 0) {
        bool same = have_inv;
        for (int i = 0; i < m; ++i) same = same && prev_mask[i] == free_mask[i];
        if (!same && !factor_free(free_mask)) return false;  // (the reference factors at k == 0; later it keeps the last one)
        invert_free();
        return true;
      }
      {
        bool same = have_inv;
        for (int i = 0; i < m; ++i) same = same && prev_mask[i] == free_mask[i];
        if (!same && !factor_free(free_mask)) return false;  // an unchanged free set keeps its factorisation
      }
      // Newton step on the free space: dxf = -Hff^-1 (qf + Hfc xc) - xf (the clamped rows of the factor are unit pivots)
      {
        double r[NU];
        for (int j = 0; j < m; ++j) {
          double a = -q[j];
          for (int c = 0; c < m; ++c)
            if (!free_mask[c]) a -= H[j * m + c] * x[c];
          r[j] = free_mask[j] ? a : 0.0;
        }
        cholesky_solve(Hm, m, r);
        for (int i = 0; i < m; ++i) dx[i] = free_mask[i] ? r[i] - x[i] : 0.0;
      }
      auto fval = [&](const double* z) {
        double f = 0;
        for (int i = 0; i < m; ++i) {
          double a = 0;
          for (int j = 0; j < m; ++j) a += H[i * m + j] * z[j];
          f += 0.5 * z[i] * a + q[i] * z[i];
        }
        return f;
      };
      const double fold = fval(x);
      for (int ia = 0; ia < 10; ++ia) {  // BoxQP's own step lengths (n_alphas_ = 10 in its constructor), not the solver's
        const double alpha = std::ldexp(1.0, -ia);
        for (int i = 0; i < m; ++i) xnew[i] = std::max(std::min(x[i] + alpha * dx[i], ub[i]), lb[i]);
        const double fnew = fval(xnew);
        double gd = 0;
        for (int i = 0; i < m; ++i) gd += g[i] * (x[i] - xnew[i]);
        if (fold - fnew > P.prm.boxqp_th_acceptstep * gd) {
          for (int i = 0; i < m; ++i) x[i] = xnew[i];
          break;
        }
      }
    }
    if (have_inv) invert_free();
    return true;
  }

  // SolverDDP::backwardPass + computeGains (A.2). Returns false on "backward_error".
  bool backward_pass() {
    const int n = ndx, m = nu;
    Vxx[T].assign(datas[T].Lxx, datas[T].Lxx + n * n);
    Vx[T].assign(datas[T].Lx, datas[T].Lx + n);
    if (!std::isnan(xreg))
      for (int i = 0; i < n; ++i) Vxx[T][i * n + i] += xreg;
    if (!is_feasible)
      for (int i = 0; i < n; ++i) {
        double acc = 0;
        for (int j = 0; j < n; ++j) acc += Vxx[T][i * n + j] * fs[T][j];
        Vx[T][i] += acc;
      }
    std::vector<double> FxTV(n * n), FuTV(m * n), L(m * m), Kt(m * n);
    for (int t = T - 1; t >= 0; --t) {
      const NodeData& D = datas[t];
      const std::vector<double>& Vp = Vxx[t + 1];
      const std::vector<double>& vp = Vx[t + 1];
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
          double acc = 0;
          for (int l = 0; l < n; ++l) acc += D.Fx[l * n + i] * Vp[l * n + j];
          FxTV[i * n + j] = acc;
        }
      for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
          double acc = 0;
          for (int l = 0; l < n; ++l) acc += D.Fu[l * m + i] * Vp[l * n + j];
          FuTV[i * n + j] = acc;
        }
      for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
          double acc = D.Lxx[i * n + j];
          for (int l = 0; l < n; ++l) acc += FxTV[i * n + l] * D.Fx[l * n + j];
          Qxx[t][i * n + j] = acc;
        }
        for (int j = 0; j < m; ++j) {
          double acc = D.Lxu[i * m + j];
          for (int l = 0; l < n; ++l) acc += FxTV[i * n + l] * D.Fu[l * m + j];
          Qxu[t][i * m + j] = acc;
        }
        double acc = D.Lx[i];
        for (int l = 0; l < n; ++l) acc += D.Fx[l * n + i] * vp[l];
        Qx[t][i] = acc;
      }
      for (int i = 0; i < m; ++i) {
        for (int j = 0; j < m; ++j) {
          double acc = D.Luu[i * m + j];
          for (int l = 0; l < n; ++l) acc += FuTV[i * n + l] * D.Fu[l * m + j];
          Quu[t][i * m + j] = acc;
        }
        double acc = D.Lu[i];
        for (int l = 0; l < n; ++l) acc += D.Fu[l * m + i] * vp[l];
        Qu[t][i] = acc;
      }
      if (!std::isnan(ureg))
        for (int i = 0; i < m; ++i) Quu[t][i * m + i] += ureg;
      // computeGains.  SolverBoxDDP::computeGains and SolverBoxFDDP::computeGains (crocoddyl ~1.8 box-ddp.cpp / box-fddp.cpp)
      // start with `if (!has_control_limits || !is_feasible_) { SolverDDP::computeGains(t); return; }`: the box QP runs once
      // the trajectory is feasible.  k from the box QP over [u_lb - us, u_ub - us], K on the free controls only, Qu zeroed on
      // the clamped ones
      const bool box_gains = is_feasible && (P.prm.solver_type == EMPC_SOLVER_BOXDDP || P.prm.solver_type == EMPC_SOLVER_BOXFDDP);
      if (box_gains) {
        double lb[NU], ub[NU], xq[NU], Hinv[NU * NU];
        int fm[NU];
        for (int i = 0; i < m; ++i) {
          lb[i] = P.d.u_lb[i] - us[t][i];
          ub[i] = P.d.u_ub[i] - us[t][i];
        }
        if (!box_qp(Quu[t].data(), Qu[t].data(), lb, ub, k[t].data(), m, xq, fm, Hinv)) return false;
        for (int i = 0; i < m; ++i)
          for (int j = 0; j < n; ++j) {
            double acc = 0;
            for (int l = 0; l < m; ++l) acc += Hinv[i * m + l] * Qxu[t][j * m + l];
            K[t][i * n + j] = acc;
          }
        for (int i = 0; i < m; ++i) {
          k[t][i] = -xq[i];
          if (!fm[i]) Qu[t][i] = 0.0;
        }
      } else {
        L = Quu[t];
        if (!cholesky(L.data(), m)) return false;
        for (int j = 0; j < n; ++j) {
          double col[NU];
          for (int i = 0; i < m; ++i) col[i] = Qxu[t][j * m + i];
          cholesky_solve(L.data(), m, col);
          for (int i = 0; i < m; ++i) K[t][i * n + j] = col[i];
        }
        k[t] = Qu[t];
        cholesky_solve(L.data(), m, k[t].data());
      }
      // value function
      for (int i = 0; i < m; ++i) {
        double acc = 0;
        for (int j = 0; j < m; ++j) acc += Quu[t][i * m + j] * k[t][j];
        Quuk[t][i] = acc;
      }
      for (int i = 0; i < n; ++i) {
        double acc = Qx[t][i];
        for (int l = 0; l < m; ++l) acc += K[t][l * n + i] * Quuk[t][l];
        for (int l = 0; l < m; ++l) acc -= 2.0 * K[t][l * n + i] * Qu[t][l];
        Vx[t][i] = acc;
      }
      std::vector<double>& V = Vxx[t];
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
          double acc = Qxx[t][i * n + j];
          for (int l = 0; l < m; ++l) acc -= Qxu[t][i * m + l] * K[t][l * n + j];
          V[i * n + j] = acc;
        }
      for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
          const double s = 0.5 * (V[i * n + j] + V[j * n + i]);
          V[i * n + j] = s;
          V[j * n + i] = s;
        }
      if (!std::isnan(xreg))
        for (int i = 0; i < n; ++i) V[i * n + i] += xreg;
      if (!is_feasible)
        for (int i = 0; i < n; ++i) {
          double acc = 0;
          for (int j = 0; j < n; ++j) acc += V[i * n + j] * fs[t][j];
          Vx[t][i] += acc;
        }
      double mx = 0;
      for (double e : Vx[t]) mx = std::max(mx, std::fabs(e));
      if (bad_number(mx)) return false;
      mx = 0;
      bool nanf = false;
      for (double e : V) {
        if (std::isnan(e)) nanf = true;
        mx = std::max(mx, std::fabs(e));
      }
      if (nanf || bad_number(mx)) return false;
    }
    return true;
  }

  // computeDirection with the retry loop of sbfddp.cpp:242-255 / :330-343. Returns false when reg_max is hit.
  bool compute_direction(bool& recalc) {
    while (true) {
      if (recalc) calc_diff();
      if (backward_pass()) return true;
      recalc = false;
      increase_reg();
      if (xreg == P.prm.reg_max) return false;
    }
  }

  // SolverFDDP::updateExpectedImprovement (A.2)
  void update_expected_improvement() {
    dg = 0;
    dq = 0;
    const int n = ndx;
    auto gap_terms = [&](int t) {
      double a = 0, b = 0;
      for (int i = 0; i < n; ++i) {
        a += Vx[t][i] * fs[t][i];
        double acc = 0;
        for (int j = 0; j < n; ++j) acc += Vxx[t][i * n + j] * fs[t][j];
        b += fs[t][i] * acc;
      }
      dg -= a;
      dq += b;
    };
    if (!is_feasible) gap_terms(T);
    for (int t = 0; t < T; ++t) {
      for (int i = 0; i < nu; ++i) {
        dg += Qu[t][i] * k[t][i];
        dq -= k[t][i] * Quuk[t][i];
      }
      if (!is_feasible) gap_terms(t);
    }
  }
  // SolverFDDP::expectedImprovement (A.2)
  void expected_improvement() {
    dv = 0;
    const int n = ndx;
    if (!is_feasible) {
      for (int t = T; t >= 0; --t) {  // reference order: terminal first, then t = 0..T-1 (sum is order-insensitive up to rounding)
        state_diff(P, xs_try[t].data(), xs[t].data(), dx[t].data());
      }
      auto term = [&](int t) {
        double b = 0;
        for (int i = 0; i < n; ++i) {
          double acc = 0;
          for (int j = 0; j < n; ++j) acc += Vxx[t][i * n + j] * dx[t][j];
          b += fs[t][i] * acc;
        }
        dv -= b;
      };
      term(T);
      for (int t = 0; t < T; ++t) term(t);
    }
    d[0] = dg + dv;
    d[1] = dq - 2 * dv;
  }
  // sbfddp.cpp:395-408
  void expected_improvement_ddp() {
    d[0] = d[1] = 0;
    for (int t = 0; t < T; ++t)
      for (int i = 0; i < nu; ++i) {
        d[0] += Qu[t][i] * k[t][i];
        d[1] -= k[t][i] * Quuk[t][i];
      }
  }

  // SolverFDDP::forwardPass (gap-aware, A.2) / SolverSbFDDP::forwardPassDDP (sbfddp.cpp:416-460, gaps ignored).
  // Returns false on "forward_error".
  bool forward_pass(double alpha, bool ddp) {
    cost_try = 0;
    std::vector<double> xnext(x0), tmp(ndx);
    NodeData D;
    for (int t = 0; t < T; ++t) {
      if (ddp) {
        if (t == 0) xs_try[0] = x0;  // set once in solve(): xs_try_[0] = problem_->get_x0()
      } else if (is_feasible || alpha == 1.0) {
        xs_try[t] = xnext;
      } else {
        for (int i = 0; i < ndx; ++i) tmp[i] = fs[t][i] * (alpha - 1.0);
        state_integrate(P, xnext.data(), tmp.data(), xs_try[t].data());
      }
      state_diff(P, xs[t].data(), xs_try[t].data(), dx[t].data());
      for (int i = 0; i < nu; ++i) {
        double acc = us[t][i] - k[t][i] * alpha;
        for (int j = 0; j < ndx; ++j) acc -= K[t][i * ndx + j] * dx[t][j];
        us_try[t][i] = acc;
      }
      if (P.prm.solver_type != EMPC_SOLVER_SBFDDP)  // SolverBox{DDP,FDDP}::forwardPass: clamp to the control limits
        for (int i = 0; i < nu; ++i) us_try[t][i] = std::min(std::max(us_try[t][i], P.d.u_lb[i]), P.d.u_ub[i]);
      calc_node(t, xs_try[t].data(), us_try[t].data(), false, D);
      xnext.assign(D.xnext, D.xnext + nx);
      if (ddp) xs_try[t + 1] = xnext;
      cost_try += D.cost;
      if (bad_number(cost_try)) return false;
      double mx = 0;
      for (double e : xnext) {
        if (std::isnan(e)) return false;
        mx = std::max(mx, std::fabs(e));
      }
      if (bad_number(mx)) return false;
    }
    if (!ddp) {
      if (is_feasible || alpha == 1.0) {
        xs_try[T] = xnext;
      } else {
        for (int i = 0; i < ndx; ++i) tmp[i] = fs[T][i] * (alpha - 1.0);
        state_integrate(P, xnext.data(), tmp.data(), xs_try[T].data());
      }
    }
    calc_node(T, xs_try[T].data(), nullptr, false, D);
    cost_try += D.cost;
    if (bad_number(cost_try)) return false;
    return true;
  }

  double gap_norm() const {
    double g = 0;
    for (int t = 0; t <= T; ++t) {
      double l1 = 0, li = 0;
      for (double e : fs[t]) {
        l1 += std::fabs(e);
        li = std::max(li, std::fabs(e));
      }
      if (P.prm.gap_norm == EMPC_GAP_L1)
        g += l1;
      else
        g = std::max(g, li);
    }
    return g;
  }
  // fork-only stoppingCriteria() (U1)
  void stopping_criteria() {
    // the crocoddyl solvers use SolverDDP::stoppingCriteria (sum |Qu|^2), whatever the fork-only option says
    switch (P.prm.solver_type != EMPC_SOLVER_SBFDDP ? (int)EMPC_STOP_QU_NORM : P.prm.stop_criteria) {
      case EMPC_STOP_COST_REDUCTION:
        stop = std::fabs(cost_prev - cost);
        break;
      case EMPC_STOP_EXPECTED_REDUCTION:
        stop = std::fabs(d[0] + 0.5 * d[1]);
        break;
      default: {
        stop = 0;
        for (int t = 0; t < T; ++t)
          for (int i = 0; i < nu; ++i) stop += Qu[t][i] * Qu[t][i];
      }
    }
  }
  bool stopping_test_gaps() { return stop < th_stop && gap_norm() < P.prm.th_stop_gaps; }
  bool stopping_test_feasible() { return was_feasible && stop < th_stop; }

  void accept(bool feasible_next) {
    was_feasible = is_feasible;
    xs = xs_try;
    us = us_try;
    is_feasible = feasible_next;
    cost_prev = cost;
    cost = cost_try;
  }
  void record() {
    IterRecord r{(double)phase, (double)iter, cost,  stop,       xreg, steplength,
                 is_feasible ? 1.0 : 0.0, dV, dVexp, gap_norm(), d[0], d[1]};
    trace.push_back(r);
  }

  // One pass through the loop body of solveFDDP (sbfddp.cpp:241-311).  Returns 0 = go on, 1 = solveFDDP returns true,
  // -1 = solveFDDP returns false.  upstream = crocoddyl::SolverFDDP::solve's own stopping test (was_feasible_ && stop_ <
  // th_stop_), used by SolverBoxFDDP.  acc_idx: index of the accepted step length (-1 none); recorded: the iteration reached
  // stoppingCriteria() and left an IterRecord.
  int fddp_iteration(bool& recalc, bool upstream, int& acc_idx, bool& recorded) {
    acc_idx = -1;
    recorded = false;
    if (!compute_direction(recalc)) {
      status |= EMPC_STATUS_REG_MAX;
      return -1;
    }
    update_expected_improvement();
    recalc = false;
    int ai = 0;
    for (double alpha : alphas) {
      steplength = alpha;
      const int this_ai = ai++;
      if (!forward_pass(alpha, false)) continue;
      dV = cost - cost_try;
      expected_improvement();
      dVexp = steplength * (d[0] + 0.5 * steplength * d[1]);
      if (dVexp >= 0) {
        if (d[0] < P.prm.th_grad || dV > P.prm.th_acceptstep * dVexp) {
          accept(is_feasible || steplength == 1.0);
          recalc = true;
          acc_idx = this_ai;
          break;
        }
      } else {
        if (dV > P.prm.th_acceptnegstep * dVexp) {
          accept(is_feasible || steplength == 1.0);
          recalc = true;
          acc_idx = this_ai;
          break;
        }
      }
    }
    if (steplength > P.prm.th_stepdec) decrease_reg();
    if (steplength <= P.prm.th_stepinc) {
      increase_reg();
      if (xreg == P.prm.reg_max) {
        status |= EMPC_STATUS_REG_MAX;
        return -1;
      }
    }
    stopping_criteria();
    record();
    recorded = true;
    if (upstream ? stopping_test_feasible() : stopping_test_gaps()) return 1;
    return 0;
  }
  // sbfddp.cpp:228-315
  bool solve_fddp(int maxiter, bool feasible, double reginit, bool upstream = false) {
    is_feasible = feasible;
    xreg = ureg = std::isnan(reginit) ? P.prm.reg_min : reginit;
    was_feasible = false;
    bool recalc = true;
    for (iter = 0; iter < maxiter; ++iter) {
      open_iterate(recalc);
      int acc_idx;
      bool recorded;
      const int r = fddp_iteration(recalc, upstream, acc_idx, recorded);
      if (r != 0) {
        close_iterate(acc_idx, recorded, true, r > 0);
        return r > 0;
      }
      close_iterate(acc_idx, recorded, iter + 1 >= maxiter, false);
    }
    iter = iter >= maxiter ? maxiter - 1 : iter;
    status |= EMPC_STATUS_MAXITER;
    return false;
  }

  // one pass through the loop body of solveDDP (sbfddp.cpp:329-389); return value as fddp_iteration
  int ddp_iteration(bool& recalc, int& acc_idx, bool& recorded) {
    acc_idx = -1;
    recorded = false;
    if (!compute_direction(recalc)) {
      status |= EMPC_STATUS_REG_MAX;
      return -1;
    }
    expected_improvement_ddp();
    recalc = false;
    int ai = 0;
    for (double alpha : alphas) {
      steplength = alpha;
      const int this_ai = ai++;
      if (!forward_pass(alpha, true)) continue;
      dV = cost - cost_try;
      dVexp = steplength * (d[0] + 0.5 * steplength * d[1]);
      if (dVexp >= 0) {
        if (d[0] < P.prm.th_grad || !is_feasible || dV > P.prm.th_acceptstep * dVexp) {
          accept(true);
          recalc = true;
          acc_idx = this_ai;
          break;
        }
      }
    }
    if (steplength > P.prm.th_stepdec) decrease_reg();
    if (steplength <= P.prm.th_stepinc) {
      increase_reg();
      if (xreg == P.prm.reg_max) {
        status |= EMPC_STATUS_REG_MAX;
        return -1;
      }
    }
    stopping_criteria();
    record();
    recorded = true;
    if (stopping_test_feasible()) return 1;
    return 0;
  }
  // sbfddp.cpp:317-393
  bool solve_ddp(int maxiter, double reginit) {
    xreg = ureg = std::isnan(reginit) ? P.prm.reg_min : reginit;
    was_feasible = false;
    bool recalc = true;
    for (iter = 0; iter < maxiter; ++iter) {
      open_iterate(recalc);
      int acc_idx;
      bool recorded;
      const int r = ddp_iteration(recalc, acc_idx, recorded);
      if (r != 0) {
        close_iterate(acc_idx, recorded, true, r > 0);
        return r > 0;
      }
      close_iterate(acc_idx, recorded, iter + 1 >= maxiter, false);
    }
    iter = iter >= maxiter ? maxiter - 1 : iter;
    status |= EMPC_STATUS_MAXITER;
    return false;
  }

  // sbfddp.cpp:192-226
  bool solve(const double* init_xs, const double* init_us, int maxiter, bool feasible_arg) {
    xs_try[0] = x0;
    set_candidate(init_xs, init_us, feasible_arg);
    double smooth = P.prm.smooth_init;
    double convergence = P.prm.convergence_init;
    total_iters = 0;
    status = 0;
    trace.clear();
    iterates.clear();
    phase = 0;
    bool last = false;
    if (P.prm.solver_type != EMPC_SOLVER_SBFDDP) {
      // crocoddyl::SolverBoxFDDP / SolverBoxDDP::solve(init_xs, init_us, maxiter, is_feasible): one loop, th_stop_ = 5e-5.
      // The BoxQP of knot t is warm-started at k_[t]: zeros for a fresh solver.  Every solve starts from zeros here (a
      // solver object serves many rollouts of a batch; the warm start moves the QP's path, not its optimum).
      for (auto& kt : k) std::fill(kt.begin(), kt.end(), 0.0);
      th_stop = P.prm.box_th_stop;
      if (P.prm.solver_type == EMPC_SOLVER_BOXFDDP) {
        last = solve_fddp(maxiter, feasible_arg, P.prm.reg_init, true);
      } else {
        phase = 100;
        last = solve_ddp(maxiter, P.prm.reg_init);
      }
      total_iters = iter + 1;
      iter = total_iters - 1;
      if (last) status |= EMPC_STATUS_CONVERGED;
      return true;
    }
    while (convergence >= P.prm.convergence_stop) {
      P.smooth = smooth;       // squashingUpdate
      barrier_update(smooth);  // barrierUpdate
      th_stop = convergence;
      status &= ~(EMPC_STATUS_MAXITER | EMPC_STATUS_REG_MAX);
      last = solve_fddp(maxiter, false, P.prm.reg_init);
      smooth *= P.prm.smooth_mult;
      convergence *= P.prm.convergence_mult;
      total_iters += iter + 1;
      ++phase;
    }
    if (!is_feasible) {
      phase = 100;
      status &= ~(EMPC_STATUS_MAXITER | EMPC_STATUS_REG_MAX);
      status |= EMPC_STATUS_DDP_CLEANUP;
      last = solve_ddp(maxiter, P.prm.reg_init);
      total_iters += iter + 1;
    }
    iter = total_iters - 1;
    if (last) status |= EMPC_STATUS_CONVERGED;
    return true;
  }

  // fillSquashedOutputs (sbfddp.cpp:479-486): squashing data of the LAST calc at every running node
  void squashed_outputs(double* out) {
    for (int t = 0; t < T; ++t) {
      double u[NU];
      if (P.d.use_squash)
        squash(P, us_lastcalc[t].data(), u, nullptr);
      else
        for (int i = 0; i < nu; ++i) u[i] = us[t][i];  // no squashing data: the accepted controls
      for (int i = 0; i < nu; ++i) out[t * nu + i] = u[i];
    }
  }
};

}  // namespace oracle
