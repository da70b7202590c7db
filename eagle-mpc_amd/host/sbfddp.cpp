// SolverSbFDDP: C++ mirror of include/eagle_mpc/sbfddp.hpp:34-126 on top of the C ABI (include/empc.h).
// All arithmetic runs in the HIP kernels; this class only moves buffers and keeps the reference's getters.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "eagle_mpc.hpp"

namespace eagle_mpc {

SolverSbFDDP::SolverSbFDDP(const std::shared_ptr<ShootingProblem>& problem, std::size_t batch_size, int device, int solver_type)
    : problem_(problem), batch_(batch_size) {
  if (!problem) throw std::invalid_argument("SolverSbFDDP: problem is null");
  EmpcSolverParams prm;
  empc_solver_params_default(&prm);
  prm.solver_type = solver_type;
  handle_ = empc_solver_create(&problem_->desc(), &prm, (int)batch_size, device);
  if (!handle_) throw std::runtime_error(std::string("SolverSbFDDP: ") + empc_last_error());
  const std::size_t T = problem_->get_T();
  us_squash_.assign(T, VectorXd(problem_->get_nu(), 0.0));  // src/sbfddp.cpp:33-37
}

SolverSbFDDP::SolverSbFDDP(const std::shared_ptr<ShootingProblem>& problem, const std::shared_ptr<SquashingModelSmoothSat>& squashing_model,
                           std::size_t batch_size, int device)
    : SolverSbFDDP(problem, batch_size, device, EMPC_SOLVER_SBFDDP) {
  if (!squashing_model) throw std::invalid_argument("SolverSbFDDP: squashing model is null");
  const EmpcProblemDesc& d = problem_->desc();
  bool same = squashing_model->get_ns() == (std::size_t)d.nu;
  for (int i = 0; same && i < d.nu; ++i)
    same = squashing_model->get_u_lb()[i] == d.u_lb[i] && squashing_model->get_u_ub()[i] == d.u_ub[i];
  if (!same) throw std::invalid_argument("SolverSbFDDP: squashing model does not belong to this problem (control limits differ)");
  squashing_model_ = squashing_model;
}

SolverSbFDDP::~SolverSbFDDP() { empc_solver_destroy(handle_); }

void SolverSbFDDP::setCallbacks(const std::vector<std::shared_ptr<CallbackAbstract>>& callbacks) {
  callbacks_ = callbacks;
  if (!callbacks_.empty() && empc_solver_enable_trace(handle_, 512) != EMPC_OK) throw std::runtime_error(empc_last_error());
}

void CallbackVerbose::operator()(const IterationRecord& r) {
  if (r.iter % 10 == 0) std::printf("iter \t cost \t      stop \t    grad \t  xreg \t      ureg \t step \t feas\n");
  std::printf("%4d  %0.5e  %0.5e  %0.5e  %10.5e  %10.5e   %0.4f     %d\n", r.iter, r.cost, r.stop, -r.d1, r.x_reg, r.u_reg, r.steplength,
              r.is_feasible ? 1 : 0);
}

void SolverSbFDDP::set_convergence_init(double convergence_init) {
  convergence_init_ = convergence_init;
  if (empc_solver_set_convergence_init(handle_, convergence_init) != EMPC_OK) throw std::runtime_error(empc_last_error());
}

void SolverSbFDDP::syncProblem() {
  if (empc_solver_update_problem(handle_, &problem_->desc()) != EMPC_OK) throw std::runtime_error(empc_last_error());
}

bool SolverSbFDDP::solve(const std::vector<VectorXd>& init_xs, const std::vector<VectorXd>& init_us, std::size_t maxiter,
                         bool is_feasible, double /*regInit: ignored by the reference too, src/sbfddp.cpp:210*/) {
  const std::size_t T = problem_->get_T(), nx = problem_->get_nx(), nu = problem_->get_nu();
  std::vector<double> x0s, xs, us;
  for (std::size_t b = 0; b < batch_; ++b) x0s.insert(x0s.end(), problem_->get_x0().begin(), problem_->get_x0().end());
  if (!init_xs.empty()) {
    if (init_xs.size() != T + 1) throw std::invalid_argument("Warm start state has wrong dimension, got " + std::to_string(init_xs.size()) + " expecting " + std::to_string(T + 1));
    for (std::size_t b = 0; b < batch_; ++b)
      for (const auto& x : init_xs) {
        if (x.size() != nx) throw std::invalid_argument("Invalid argument: xs[i] has wrong dimension");
        xs.insert(xs.end(), x.begin(), x.end());
      }
  }
  if (!init_us.empty()) {
    if (init_us.size() != T) throw std::invalid_argument("Warm start control has wrong dimension, got " + std::to_string(init_us.size()) + " expecting " + std::to_string(T));
    for (std::size_t b = 0; b < batch_; ++b)
      for (const auto& u : init_us) {
        if (u.size() != nu) throw std::invalid_argument("Invalid argument: us[i] has wrong dimension");
        us.insert(us.end(), u.begin(), u.end());
      }
  }
  return solveBatch(x0s, xs, us, maxiter, is_feasible);
}

bool SolverSbFDDP::solveBatch(const std::vector<double>& x0s, const std::vector<double>& init_xs,
                              const std::vector<double>& init_us, std::size_t maxiter, bool is_feasible) {
  const std::size_t T = problem_->get_T(), nx = problem_->get_nx(), nu = problem_->get_nu();
  if (x0s.size() != batch_ * nx) throw std::invalid_argument("x0s must be batch x nx");
  if (!init_xs.empty() && init_xs.size() != batch_ * (T + 1) * nx) throw std::invalid_argument("init_xs must be batch x (T+1) x nx");
  if (!init_us.empty() && init_us.size() != batch_ * T * nu) throw std::invalid_argument("init_us must be batch x T x nu");
  if (empc_solver_set_x0(handle_, x0s.data()) != EMPC_OK) throw std::runtime_error(empc_last_error());
  if (empc_solver_set_warmstart(handle_, init_xs.empty() ? nullptr : init_xs.data(),
                                init_us.empty() ? nullptr : init_us.data()) != EMPC_OK)
    throw std::runtime_error(empc_last_error());
  if (empc_solver_solve(handle_, (int)maxiter, is_feasible ? 1 : 0) != EMPC_OK) throw std::runtime_error(empc_last_error());
  fetch();
  return true;  // the reference returns true unconditionally (src/sbfddp.cpp:225)
}

void SolverSbFDDP::fetch() {
  const std::size_t T = problem_->get_T(), nx = problem_->get_nx(), nu = problem_->get_nu();
  xs_b_.resize(batch_ * (T + 1) * nx);
  us_b_.resize(batch_ * T * nu);
  us_squash_b_.resize(batch_ * T * nu);
  cost_b_.resize(batch_);
  iter_b_.resize(batch_);
  status_b_.resize(batch_);
  std::vector<double> stop(batch_);
  if (empc_solver_get_xs(handle_, xs_b_.data()) != EMPC_OK || empc_solver_get_us(handle_, us_b_.data()) != EMPC_OK ||
      empc_solver_get_us_squash(handle_, us_squash_b_.data()) != EMPC_OK ||
      empc_solver_get_cost(handle_, cost_b_.data()) != EMPC_OK || empc_solver_get_iters(handle_, iter_b_.data()) != EMPC_OK ||
      empc_solver_get_status(handle_, status_b_.data()) != EMPC_OK || empc_solver_get_stop(handle_, stop.data()) != EMPC_OK)
    throw std::runtime_error(empc_last_error());
  xs_.assign(T + 1, VectorXd(nx));
  us_.assign(T, VectorXd(nu));
  us_squash_.assign(T, VectorXd(nu));
  for (std::size_t t = 0; t <= T; ++t) std::memcpy(xs_[t].data(), &xs_b_[t * nx], sizeof(double) * nx);
  for (std::size_t t = 0; t < T; ++t) {
    std::memcpy(us_[t].data(), &us_b_[t * nu], sizeof(double) * nu);
    std::memcpy(us_squash_[t].data(), &us_squash_b_[t * nu], sizeof(double) * nu);
  }
  iter_ = (std::size_t)iter_b_[0];
  cost_ = cost_b_[0];
  stop_ = stop[0];
  if (!callbacks_.empty()) {
    int n = 0;
    if (empc_solver_get_trace(handle_, 0, nullptr, 0, &n) != EMPC_OK) throw std::runtime_error(empc_last_error());
    const int k = std::min(n, 512);
    std::vector<double> tr((std::size_t)k * EMPC_TRACE_WORDS);
    if (k > 0 && empc_solver_get_trace(handle_, 0, tr.data(), k, &n) != EMPC_OK) throw std::runtime_error(empc_last_error());
    for (int i = 0; i < k; ++i) {
      const double* r = &tr[(std::size_t)i * EMPC_TRACE_WORDS];
      const IterationRecord rec{(int)r[0], (int)r[1], r[2], r[3], r[4], r[4], r[5], r[7], r[8], r[9], r[10], r[11], r[6] != 0.0};
      for (auto& cb : callbacks_) (*cb)(rec);
    }
  }
}

}  // namespace eagle_mpc
