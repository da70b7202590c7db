// ORACLE (test infrastructure only -- never linked into or called from the product path).
//
// C entry points of the CPU restatement, loaded with ctypes by tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg only.  Parity status: UNPINNED (see oracle/README.md and DESIGN.md).
#include <chrono>
#include <cstdio>
#include <cstring>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "solver.hpp"

using namespace oracle;

extern "C" {

void oracle_solver_params_default(EmpcSolverParams* p) {
  std::memset(p, 0, sizeof(*p));
  p->smooth_init = 0.1;        // src/sbfddp.cpp:9
  p->smooth_mult = 0.5;        // :10
  p->barrier_weight = 1e-3;    // :11
  p->convergence_init = 1e-2;  // :12
  p->convergence_stop = 1e-3;  // :13
  p->convergence_mult = 1e-1;  // :14
  p->reg_init = 1e-9;          // :16
  p->th_acceptnegstep = 2;     // :17
  p->th_stop_gaps = 1.0;       // :27
  p->th_grad = 1e-12;          // crocoddyl SolverDDP defaults
  p->th_acceptstep = 0.1;
  p->th_stepdec = 0.5;
  p->th_stepinc = 0.01;
  p->reg_incfactor = 10;
  p->reg_decfactor = 10;
  p->reg_min = 1e-9;
  p->reg_max = 1e9;
  p->th_gaptol = 1e-16;
  p->n_alphas = 10;
  p->stop_criteria = EMPC_STOP_COST_REDUCTION;
  p->gap_norm = EMPC_GAP_L1;
  p->terminal_dt_scaling = 1;
  p->smoothsat_power = 2;
  p->solver_type = EMPC_SOLVER_SBFDDP;
  p->box_th_stop = 5e-5;  // crocoddyl SolverBox{DDP,FDDP} constructors (~1.8)
  p->boxqp_th_acceptstep = 0.1;
  p->boxqp_th_grad = 1e-5;
  p->boxqp_reg = 0.0;
  p->boxqp_maxiter = 100;
}

void* oracle_solver_create(const EmpcProblemDesc* d, const EmpcSolverParams* p) {
  Solver* s = new Solver();
  s->init(*d, *p);
  return s;
}
void oracle_solver_destroy(void* h) { delete static_cast<Solver*>(h); }

void oracle_solver_set_x0(void* h, const double* x0) {
  Solver* s = static_cast<Solver*>(h);
  s->x0.assign(x0, x0 + s->nx);
}
void oracle_solver_set_convergence_init(void* h, double c) { static_cast<Solver*>(h)->P.prm.convergence_init = c; }

// MPC updates: overwrite reference / weight / active flag of a cost of one cost set
int oracle_solver_set_cost_ref(void* h, int set, const char* name, const double* ref, int nref, int active,
                               double weight) {
  Solver* s = static_cast<Solver*>(h);
  if (set < 0 || set >= (int)s->P.sets.size()) return -1;
  EmpcCostSet& cs = s->P.sets[set];
  for (int i = 0; i < cs.ncosts; ++i)
    if (std::strcmp(cs.costs[i].name, name) == 0) {
      if (ref)
        for (int j = 0; j < nref; ++j) cs.costs[i].ref[j] = ref[j];
      if (active >= 0) cs.costs[i].active = active;
      if (weight >= 0) cs.costs[i].weight = weight;
      return 0;
    }
  return -2;
}

int oracle_solver_solve(void* h, const double* xs, const double* us, int maxiter, int is_feasible) {
  Solver* s = static_cast<Solver*>(h);
  s->solve(xs, us, maxiter, is_feasible != 0);
  return s->status;
}

void oracle_solver_get(void* h, double* xs, double* us, double* us_squash, double* cost, int* iters, int* status,
                       double* stop) {
  Solver* s = static_cast<Solver*>(h);
  if (xs)
    for (int t = 0; t <= s->T; ++t) std::memcpy(xs + t * s->nx, s->xs[t].data(), sizeof(double) * s->nx);
  if (us)
    for (int t = 0; t < s->T; ++t) std::memcpy(us + t * s->nu, s->us[t].data(), sizeof(double) * s->nu);
  if (us_squash) s->squashed_outputs(us_squash);
  if (cost) *cost = s->cost;
  if (iters) *iters = s->iter;
  if (status) *status = s->status;
  if (stop) *stop = s->stop;
}

int oracle_solver_trace(void* h, double* out, int max_records) {
  Solver* s = static_cast<Solver*>(h);
  const int n = std::min<int>(max_records, (int)s->trace.size());
  if (out) std::memcpy(out, s->trace.data(), sizeof(IterRecord) * n);
  return (int)s->trace.size();
}
int oracle_trace_record_len(void) { return (int)(sizeof(IterRecord) / sizeof(double)); }

// ---- iterate recording + one-iteration probe (teacher-forced parity tests) ----------------------------------------------
void oracle_solver_record_iterates(void* h, int on) { static_cast<Solver*>(h)->record_iterates = on != 0; }
int oracle_solver_n_iterates(void* h) { return (int)static_cast<Solver*>(h)->iterates.size(); }
// iterate i of the last solve: candidate (xs, us), integers {phase, iter, is_feasible, was_feasible, recalc, trace_index,
// accepted_alpha, ended, returned}, reals {xreg, smooth, th_stop, cost, cost_prev}
int oracle_solver_get_iterate(void* h, int i, double* xs, double* us, int* ints, double* reals, double* k) {
  Solver* s = static_cast<Solver*>(h);
  if (i < 0 || i >= (int)s->iterates.size()) return -1;
  const Iterate& it = s->iterates[i];
  if (xs) std::memcpy(xs, it.xs.data(), sizeof(double) * it.xs.size());
  if (us) std::memcpy(us, it.us.data(), sizeof(double) * it.us.size());
  if (k) std::memcpy(k, it.k.data(), sizeof(double) * it.k.size());
  if (ints) {
    const int v[9] = {it.phase, it.iter, it.is_feasible, it.was_feasible, it.recalc, it.trace_index, it.accepted_alpha, it.ended, it.returned};
    std::memcpy(ints, v, sizeof(v));
  }
  if (reals) {
    const double v[5] = {it.xreg, it.smooth, it.th_stop, it.cost, it.cost_prev};
    std::memcpy(reals, v, sizeof(v));
  }
  return 0;
}
// One iteration's worth of work from a given iterate, WITHOUT the line search's early exit: calcDiff, computeDirection
// (regularisation retries included), the expected-improvement sums, then forwardPass for EVERY step length.
//   scal_in  {is_feasible, was_feasible, ddp (solveDDP iteration), xreg, smooth}
//   scal_out {cost, is_feasible after calcDiff, gap norm, xreg after computeDirection, dg, dq, direction ok}
//   per step length n: ok[n], cost_try[n], d0[n], d1[n] (expectedImprovement after that trial; the DDP values for ddp)
// The solver object afterwards holds the tape (oracle_phase_tape) and the gains of that iterate.
int oracle_iter_probe(void* h, const double* xs, const double* us, const double* k_in, const double* scal_in, double* scal_out,
                      int* ok, double* cost_try, double* d0, double* d1) {
  Solver* s = static_cast<Solver*>(h);
  if (k_in)  // box solvers: the QP of knot t starts at k_[t] of the previous iteration
    for (int t = 0; t < s->T; ++t) s->k[t].assign(k_in + (size_t)t * s->nu, k_in + (size_t)(t + 1) * s->nu);
  const bool ddp = scal_in[2] != 0.0;
  s->P.smooth = scal_in[4];
  if (s->P.prm.solver_type == EMPC_SOLVER_SBFDDP) s->barrier_update(scal_in[4]);
  s->set_candidate(xs, us, scal_in[0] != 0.0);
  s->was_feasible = scal_in[1] != 0.0;
  s->xreg = s->ureg = scal_in[3];
  s->xs_try[0] = s->x0;
  bool recalc = true;
  const bool dir_ok = s->compute_direction(recalc);
  scal_out[0] = s->cost;
  scal_out[1] = s->is_feasible ? 1.0 : 0.0;
  scal_out[2] = s->gap_norm();
  scal_out[3] = s->xreg;
  scal_out[6] = dir_ok ? 1.0 : 0.0;
  if (!dir_ok) return 0;
  if (ddp) {
    s->expected_improvement_ddp();
    scal_out[4] = s->d[0];
    scal_out[5] = s->d[1];
  } else {
    s->update_expected_improvement();
    scal_out[4] = s->dg;
    scal_out[5] = s->dq;
  }
  const double d0_ddp = s->d[0], d1_ddp = s->d[1];
  for (size_t n = 0; n < s->alphas.size(); ++n) {
    const bool fok = s->forward_pass(s->alphas[n], ddp);
    ok[n] = fok ? 1 : 0;
    cost_try[n] = s->cost_try;
    if (fok && !ddp) s->expected_improvement();
    d0[n] = ddp ? d0_ddp : s->d[0];
    d1[n] = ddp ? d1_ddp : s->d[1];
  }
  return 1;
}

// Exactly one pass through the loop body of solveFDDP / solveDDP from a given iterate (the function the solve loops call).
//   scal_in  {is_feasible, was_feasible, ddp, xreg, smooth, th_stop, cost, cost_prev, iter, upstream test (box FDDP)}
//   ints_out {result (0 go on, 1 returned true, -1 returned false), accepted alpha index, is_feasible, was_feasible, recorded}
//   scal_out {steplength, xreg, cost, cost_prev, stop, dV, dVexp, d0, d1, gap norm}
// The new candidate is read with oracle_solver_get.
void oracle_iter_step(void* h, const double* xs, const double* us, const double* k_in, const double* scal_in, int* ints_out,
                      double* scal_out) {
  Solver* s = static_cast<Solver*>(h);
  const bool ddp = scal_in[2] != 0.0;
  if (k_in)
    for (int t = 0; t < s->T; ++t) s->k[t].assign(k_in + (size_t)t * s->nu, k_in + (size_t)(t + 1) * s->nu);
  s->P.smooth = scal_in[4];
  if (s->P.prm.solver_type == EMPC_SOLVER_SBFDDP) s->barrier_update(scal_in[4]);
  s->set_candidate(xs, us, scal_in[0] != 0.0);
  s->was_feasible = scal_in[1] != 0.0;
  s->xreg = s->ureg = scal_in[3];
  s->th_stop = scal_in[5];
  s->cost = scal_in[6];
  s->cost_prev = scal_in[7];
  s->iter = (int)scal_in[8];
  s->xs_try[0] = s->x0;
  s->phase = ddp ? 100 : 0;
  bool recalc = true;
  int acc_idx = -1;
  bool recorded = false;
  const int r = ddp ? s->ddp_iteration(recalc, acc_idx, recorded) : s->fddp_iteration(recalc, scal_in[9] != 0.0, acc_idx, recorded);
  ints_out[0] = r;
  ints_out[1] = acc_idx;
  ints_out[2] = s->is_feasible ? 1 : 0;
  ints_out[3] = s->was_feasible ? 1 : 0;
  ints_out[4] = recorded ? 1 : 0;
  const double o[10] = {s->steplength, s->xreg, s->cost, s->cost_prev, s->stop, s->dV, s->dVexp, s->d[0], s->d[1], s->gap_norm()};
  std::memcpy(scal_out, o, sizeof(o));
}

// gains and value-function gradient as the last backward pass left them (no recomputation: the box solvers' QPs depend on
// their warm start, so a second backward pass is not the same computation)
void oracle_get_gains(void* h, double* K, double* k, double* Vx) {
  Solver* s = static_cast<Solver*>(h);
  const int n = s->ndx, m = s->nu;
  if (K)
    for (int t = 0; t < s->T; ++t) std::memcpy(K + (size_t)t * m * n, s->K[t].data(), sizeof(double) * m * n);
  if (k)
    for (int t = 0; t < s->T; ++t) std::memcpy(k + (size_t)t * m, s->k[t].data(), sizeof(double) * m);
  if (Vx)
    for (int t = 0; t <= s->T; ++t) std::memcpy(Vx + (size_t)t * n, s->Vx[t].data(), sizeof(double) * n);
}

// squashingUpdate + barrierUpdate (src/sbfddp.cpp:462-477)
void oracle_solver_set_smooth(void* h, double smooth) {
  Solver* s = static_cast<Solver*>(h);
  s->P.smooth = smooth;
  s->barrier_update(smooth);
}

// One node: IAM.calc / calcDiff at (x,u); u == NULL -> terminal call.  Any output pointer may be NULL.
void oracle_node_calc(void* h, int t, const double* x, const double* u, int diff, double* xnext, double* cost,
                      double* Fx, double* Fu, double* Lx, double* Lu, double* Lxx, double* Lxu, double* Luu,
                      double* acc, double* u_squash, double* lambda) {
  Solver* s = static_cast<Solver*>(h);
  static thread_local NodeData D;
  node_calc(s->P, t, x, u, diff != 0, D);
  const int n = s->ndx, m = s->nu;
  if (xnext) std::memcpy(xnext, D.xnext, sizeof(double) * s->nx);
  if (cost) *cost = D.cost;
  if (acc) std::memcpy(acc, D.xout, sizeof(double) * s->P.nv());
  if (u_squash) std::memcpy(u_squash, D.u_squash, sizeof(double) * m);
  if (lambda) std::memcpy(lambda, D.lambda, sizeof(double) * 6);
  if (diff) {
    if (Fx) std::memcpy(Fx, D.Fx, sizeof(double) * n * n);
    if (Fu) std::memcpy(Fu, D.Fu, sizeof(double) * n * m);
    if (Lx) std::memcpy(Lx, D.Lx, sizeof(double) * n);
    if (Lu) std::memcpy(Lu, D.Lu, sizeof(double) * m);
    if (Lxx) std::memcpy(Lxx, D.Lxx, sizeof(double) * n * n);
    if (Lxu) std::memcpy(Lxu, D.Lxu, sizeof(double) * n * m);
    if (Luu) std::memcpy(Luu, D.Luu, sizeof(double) * m * m);
  }
}

// ---- phase-level entry points (mirror the HIP kernels: linearize / backward / rollout) --------------
// calcDiff at (xs, us): fills the solver's tapes; returns the total cost, writes gaps fs[(T+1) x ndx] and the
// feasibility flag the reference would derive.
double oracle_phase_calcdiff(void* h, const double* xs, const double* us, int is_feasible, int was_feasible,
                             double* fs, int* feasible_out) {
  Solver* s = static_cast<Solver*>(h);
  s->set_candidate(xs, us, is_feasible != 0);
  s->was_feasible = was_feasible != 0;
  s->calc_diff();
  if (fs)
    for (int t = 0; t <= s->T; ++t) std::memcpy(fs + t * s->ndx, s->fs[t].data(), sizeof(double) * s->ndx);
  if (feasible_out) *feasible_out = s->is_feasible ? 1 : 0;
  return s->cost;
}
// backwardPass with the given regularisation; returns 1 on success, 0 on "backward_error".
int oracle_phase_backward(void* h, double xreg, double* K, double* k, double* Vx, double* Vxx, double* dgdq) {
  Solver* s = static_cast<Solver*>(h);
  s->xreg = s->ureg = xreg;
  const bool ok = s->backward_pass();
  const int n = s->ndx, m = s->nu;
  if (ok) {
    if (K)
      for (int t = 0; t < s->T; ++t) std::memcpy(K + t * m * n, s->K[t].data(), sizeof(double) * m * n);
    if (k)
      for (int t = 0; t < s->T; ++t) std::memcpy(k + t * m, s->k[t].data(), sizeof(double) * m);
    if (Vx)
      for (int t = 0; t <= s->T; ++t) std::memcpy(Vx + t * n, s->Vx[t].data(), sizeof(double) * n);
    if (Vxx)
      for (int t = 0; t <= s->T; ++t) std::memcpy(Vxx + t * n * n, s->Vxx[t].data(), sizeof(double) * n * n);
    s->update_expected_improvement();
    if (dgdq) {
      dgdq[0] = s->dg;
      dgdq[1] = s->dq;
    }
  }
  return ok ? 1 : 0;
}
// forwardPass(alpha) (FDDP when ddp == 0, forwardPassDDP otherwise); returns 1 on success.
int oracle_phase_forward(void* h, double alpha, int ddp, double* xs_try, double* us_try, double* cost_try,
                         double* d01) {
  Solver* s = static_cast<Solver*>(h);
  if (ddp) s->xs_try[0] = s->x0;
  const bool ok = s->forward_pass(alpha, ddp != 0);
  if (xs_try)
    for (int t = 0; t <= s->T; ++t) std::memcpy(xs_try + t * s->nx, s->xs_try[t].data(), sizeof(double) * s->nx);
  if (us_try)
    for (int t = 0; t < s->T; ++t) std::memcpy(us_try + t * s->nu, s->us_try[t].data(), sizeof(double) * s->nu);
  if (cost_try) *cost_try = s->cost_try;
  if (ok && d01) {
    if (ddp)
      s->expected_improvement_ddp();
    else
      s->expected_improvement();
    d01[0] = s->d[0];
    d01[1] = s->d[1];
  }
  return ok ? 1 : 0;
}
// crocoddyl::BoxQP::solve restated (oracle/solver.hpp box_qp), for the known-answer test: returns 1 on success
int oracle_box_qp(void* h, int m, const double* H, const double* q, const double* lb, const double* ub, const double* xinit,
                  double* x, int* free_mask, double* Hinv) {
  return static_cast<Solver*>(h)->box_qp(H, q, lb, ub, xinit, m, x, free_mask, Hinv) ? 1 : 0;
}
// expectedImprovementDDP (src/sbfddp.cpp:395-408) after oracle_phase_backward: d0 = sum Qu.k, d1 = -sum k.Quu k
void oracle_phase_expected_ddp(void* h, double* d01) {
  Solver* s = static_cast<Solver*>(h);
  s->expected_improvement_ddp();
  d01[0] = s->d[0];
  d01[1] = s->d[1];
}
// read one node's tape after oracle_phase_calcdiff
void oracle_phase_tape(void* h, int t, double* Fx, double* Fu, double* Lx, double* Lu, double* Lxx, double* Lxu,
                       double* Luu, double* xnext, double* cost) {
  Solver* s = static_cast<Solver*>(h);
  const NodeData& D = s->datas[t];
  const int n = s->ndx, m = s->nu;
  if (Fx) std::memcpy(Fx, D.Fx, sizeof(double) * n * n);
  if (Fu) std::memcpy(Fu, D.Fu, sizeof(double) * n * m);
  if (Lx) std::memcpy(Lx, D.Lx, sizeof(double) * n);
  if (Lu) std::memcpy(Lu, D.Lu, sizeof(double) * m);
  if (Lxx) std::memcpy(Lxx, D.Lxx, sizeof(double) * n * n);
  if (Lxu) std::memcpy(Lxu, D.Lxu, sizeof(double) * n * m);
  if (Luu) std::memcpy(Luu, D.Luu, sizeof(double) * m * m);
  if (xnext) std::memcpy(xnext, D.xnext, sizeof(double) * s->nx);
  if (cost) *cost = D.cost;
}

// ---- batch driver: B independent solves of the same problem from different x0 (the CPU baseline) -----
// x0s: B x nx. Outputs may be NULL. Returns wall seconds.  nthreads <= 1: serial.
double oracle_solve_batch(const EmpcProblemDesc* d, const EmpcSolverParams* p, int B, const double* x0s,
                          int maxiter, int nthreads, double* xs, double* us, double* us_squash, double* cost,
                          int* iters, int* status) {
  const int nx = d->nx, nu = d->nu, T = d->T;
  auto t0 = std::chrono::steady_clock::now();
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
  {
    Solver* s = new Solver();
    s->init(*d, *p);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int b = 0; b < B; ++b) {
      s->x0.assign(x0s + (size_t)b * nx, x0s + (size_t)(b + 1) * nx);
      s->solve(nullptr, nullptr, maxiter, false);
      oracle_solver_get(s, xs ? xs + (size_t)b * (T + 1) * nx : nullptr, us ? us + (size_t)b * T * nu : nullptr,
                        us_squash ? us_squash + (size_t)b * T * nu : nullptr, cost ? cost + b : nullptr,
                        iters ? iters + b : nullptr, status ? status + b : nullptr, nullptr);
    }
    delete s;
  }
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}

// ---- math entry points for the finite-difference tests ------------------------------------------------
void oracle_exp6(const double* xi, double* R, double* p) { exp6(xi, R, p); }
void oracle_log6(const double* R, const double* p, double* xi) { log6(R, p, xi); }
void oracle_Jexp6(const double* xi, double* J) { Jexp6(xi, J); }
void oracle_Jlog6(const double* xi, double* J) { Jlog6(xi, J); }
void oracle_exp3(const double* w, double* R) { exp3(w, R); }
void oracle_log3(const double* R, double* w) { log3(R, w); }
void oracle_Jexp3(const double* w, double* J) { Jexp3(w, J); }
void oracle_Jlog3(const double* w, double* J) { Jlog3(w, J); }
void oracle_state_integrate(void* h, const double* x, const double* dx, double* xout) {
  state_integrate(static_cast<Solver*>(h)->P, x, dx, xout);
}
void oracle_state_diff(void* h, const double* x0, const double* x1, double* dx) {
  state_diff(static_cast<Solver*>(h)->P, x0, x1, dx);
}
// RNEA / CRBA on the bare model
void oracle_rnea(const EmpcModelDesc* m, const double* q, const double* v, const double* a, double* tau) {
  double R0[9], cs[NB], sn[NB];
  plain_state(*m, q, R0, cs, sn);
  Kin<double> kin;
  rnea<double>(*m, R0, q, cs, sn, v, a, nullptr, tau, kin);
}
void oracle_crba(const EmpcModelDesc* m, const double* q, double* M) {
  double R0[9], cs[NB], sn[NB], z[NV] = {0};
  plain_state(*m, q, R0, cs, sn);
  Kin<double> kin;
  forward_kin<double>(*m, R0, q, cs, sn, z, z, false, kin);
  crba(*m, kin, M);
}
// total mechanical energy (kinetic + potential) -- used by the energy-conservation identity test
double oracle_energy(const EmpcModelDesc* m, const double* q, const double* v) {
  double R0[9], cs[NB], sn[NB], z[NV] = {0};
  plain_state(*m, q, R0, cs, sn);
  Kin<double> kin;
  forward_kin<double>(*m, R0, q, cs, sn, v, z, false, kin);
  double E = 0;
  for (int b = 0; b < m->nbodies; ++b) {
    double Iv[6];
    inertia_apply<double>(*m, b, kin.v[b], Iv);
    double ke = 0;
    for (int i = 0; i < 6; ++i) ke += 0.5 * kin.v[b][i] * Iv[i];
    double cw[3];
    matvec3<double>(kin.R[b], m->com[b], cw);
    double pe = 0;
    for (int i = 0; i < 3; ++i) pe -= m->mass[b] * m->gravity[i] * (kin.p[b][i] + cw[i]);
    E += ke + pe;
  }
  return E;
}
// Plant of the closed-loop MPC runs (reference: bindings/python/eagle_mpc/utils/simulator.py:8-29):
// FreeFwdDynamics with the unsquashed multicopter actuation, crocoddyl::IntegratedActionModelRK4:
//   k_i = [v_i; a(y_i, u)], y_i = x (+) c_i dt k_{i-1}, c = {0, 1/2, 1/2, 1}, xnext = x (+) dt/6 (k0 + 2 k1 + 2 k2 + k3)
static void plant_acc(const Problem& P, const double* x, const double* u, double* a) {
  const EmpcModelDesc& m = P.d.model;
  const int nv = m.nv, nq = m.nq, nrot = P.d.n_rotors;
  double tau[NV], h[NV], z[NV] = {0}, M[NV * NV];
  for (int r = 0; r < 6; ++r) {
    double s = 0;
    for (int c = 0; c < nrot; ++c) s += P.d.tau_f[r * nrot + c] * u[c];
    tau[r] = s;
  }
  for (int i = 6; i < nv; ++i) tau[i] = u[nrot + i - 6];
  oracle_rnea(&m, x, x + nq, z, h);
  oracle_crba(&m, x, M);
  // dense Cholesky solve M a = tau - h
  double L[NV * NV] = {0}, y[NV];
  for (int i = 0; i < nv; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = M[i * nv + j];
      for (int k = 0; k < j; ++k) s -= L[i * nv + k] * L[j * nv + k];
      L[i * nv + j] = (i == j) ? std::sqrt(s) : s / L[j * nv + j];
    }
  for (int i = 0; i < nv; ++i) {
    double s = tau[i] - h[i];
    for (int k = 0; k < i; ++k) s -= L[i * nv + k] * y[k];
    y[i] = s / L[i * nv + i];
  }
  for (int i = nv - 1; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < nv; ++k) s -= L[k * nv + i] * a[k];
    a[i] = s / L[i * nv + i];
  }
}
void oracle_plant_rk4(const EmpcProblemDesc* d, const double* x, const double* u, double dt, double* xnext) {
  Problem P;
  P.d = *d;
  const int nv = d->model.nv, nq = d->model.nq, ndx = 2 * nv, nx = nq + nv;
  const double c[4] = {0.0, 0.5, 0.5, 1.0};
  double k[4][2 * NV], y[EMPC_MAX_NX], dx[2 * NV];
  for (int i = 0; i < nx; ++i) y[i] = x[i];
  for (int st = 0; st < 4; ++st) {
    if (st > 0) {
      for (int i = 0; i < ndx; ++i) dx[i] = c[st] * k[st - 1][i] * dt;
      state_integrate(P, x, dx, y);
    }
    for (int i = 0; i < nv; ++i) k[st][i] = y[nq + i];
    plant_acc(P, y, u, k[st] + nv);
  }
  for (int i = 0; i < ndx; ++i) dx[i] = (k[0][i] + 2.0 * k[1][i] + 2.0 * k[2][i] + k[3][i]) * dt / 6.0;
  state_integrate(P, x, dx, xnext);
}
}
