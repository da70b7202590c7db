// MpcAbstract / CarrotMpc / RailMpc / WeightedMpc: host-side mirror of src/mpc-base.cpp and
// src/mpc-controllers/{carrot,rail,weighted}-mpc.cpp.
// The controller only edits cost tables (references, active flags); the arithmetic of every solve runs in the HIP
// kernels behind the C ABI.  Where the reference mutates shared Crocoddyl cost models in place, this class edits the
// private EmpcCostSet of each knot and SolverSbFDDP::syncProblem() uploads the tables.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <stdexcept>

#include "eagle_mpc.hpp"

namespace eagle_mpc {

// -----------------------------------------------------------------------------------------------------
// MpcAbstract (src/mpc-base.cpp)
// -----------------------------------------------------------------------------------------------------
MpcAbstract::MpcAbstract(const std::string& yaml_path) {  // :5-17
  ParserYaml parser(yaml_path);
  params_server_ = std::make_shared<ParamsServer>(parser.get_params());
  initializeRobotObjects();
  loadParams();
}

void MpcAbstract::initializeRobotObjects() {  // :19-38
  const std::string prefix_robot = "robot/";
  robot_model_path_ = getUrdfPath(params_server_->getParam<std::string>(prefix_robot + "urdf"));
  robot_model_ = std::make_shared<RobotModel>(RobotModel::fromUrdf(robot_model_path_));
  platform_params_ = std::make_shared<MultiCopterBaseParams>();
  platform_params_->autoSetup(prefix_robot + "platform/", params_server_, robot_model_);
}

void MpcAbstract::loadParams() {  // :40-60
  const std::string prefix = "mpc_controller/";
  const std::string integration_method = params_server_->getParam<std::string>(prefix + "integration_method");
  if (integration_method == "IntegratedActionModelEuler")
    params_.integrator_type = IntegratedActionModelTypes::IntegratedActionModelEuler;
  else if (integration_method == "IntegratedActionModelRK4")
    params_.integrator_type = IntegratedActionModelTypes::IntegratedActionModelRK4;
  else
    throw std::out_of_range("map::at");  // IntegratedActionModelTypes_map.at (:44)
  params_.knots = (std::size_t)params_server_->getParam<int>(prefix + "knots");
  params_.iters = (std::size_t)params_server_->getParam<int>(prefix + "iters");
  params_.dt = (std::size_t)params_server_->getParam<int>(prefix + "dt");
  const std::string solver = params_server_->getParam<std::string>(prefix + "solver");
  if (solver == "SolverSbFDDP")
    params_.solver_type = SolverTypes::SolverSbFDDP;
  else if (solver == "SolverBoxFDDP")
    params_.solver_type = SolverTypes::SolverBoxFDDP;
  else if (solver == "SolverBoxDDP")
    params_.solver_type = SolverTypes::SolverBoxDDP;
  else
    throw std::out_of_range("map::at");  // SolverTypes_map.at (:52)
  try {
    params_.callback = params_server_->getParam<bool>(prefix + "callback");
  } catch (const std::exception&) {
    params_.callback = false;
  }
}

VectorXd MpcAbstract::zero_state() const {
  VectorXd x(get_nx(), 0.0);
  x[6] = 1.0;
  return x;
}

const std::shared_ptr<SolverSbFDDP>& MpcAbstract::get_solver(std::size_t batch_size, int device) {
  if (!solver_) {
    // src/mpc-controllers/carrot-mpc.cpp:232-242: eagle_mpc::SolverSbFDDP, crocoddyl::SolverBoxFDDP or crocoddyl::SolverBoxDDP
    const int st = params_.solver_type == SolverTypes::SolverSbFDDP    ? EMPC_SOLVER_SBFDDP
                   : params_.solver_type == SolverTypes::SolverBoxFDDP ? EMPC_SOLVER_BOXFDDP
                                                                       : EMPC_SOLVER_BOXDDP;
    solver_ = std::make_shared<SolverSbFDDP>(problem_, batch_size, device, st);
  } else if (solver_->get_batch_size() != batch_size) {
    throw std::invalid_argument("MpcAbstract: the solver already exists with a different batch size");
  }
  return solver_;
}

// -----------------------------------------------------------------------------------------------------
// CarrotMpc (src/mpc-controllers/carrot-mpc.cpp)
// -----------------------------------------------------------------------------------------------------
CarrotMpc::CarrotMpc(const std::shared_ptr<Trajectory>& trajectory, const std::vector<VectorXd>& state_ref,
                     std::size_t dt_ref, const std::string& yaml_path)
    : MpcAbstract(yaml_path), trajectory_(trajectory) {  // :15-50
  if (!trajectory) throw std::invalid_argument("CarrotMpc: trajectory is null");
  state_ref_ = state_ref;
  for (const auto& x : state_ref_)
    if (x.size() != get_nx()) throw std::invalid_argument("CarrotMpc: state_ref[i] has wrong dimension");
  for (std::size_t i = 0; i < state_ref_.size(); ++i) t_ref_.push_back(dt_ref * i);

  loadCostParams();

  const auto& stages = trajectory_->get_stages();
  if (stages.empty()) throw std::invalid_argument("CarrotMpc: trajectory has no stages");
  t_stages_.reserve(stages.size() + 1);
  t_stages_.push_back(0);
  for (std::size_t i = 1; i < stages.size(); ++i) {
    const std::size_t duration = stages[i - 1]->get_duration() <= params_.dt ? params_.dt : stages[i - 1]->get_duration();
    t_stages_.push_back(t_stages_.back() + duration);
  }
  const std::size_t duration = stages.back()->get_duration() <= params_.dt ? params_.dt : stages.back()->get_duration();
  t_stages_.push_back(t_stages_.back() + duration);

  createProblem();
  update_vars_.state_ref = zero_state();
}

void CarrotMpc::loadCostParams() {  // :53-176
  const std::string p = "mpc_controller/";
  auto scalar = [&](const char* key, double fallback) {
    try {
      return params_server_->getParam<double>(p + key);
    } catch (const std::exception&) {
      return fallback;
    }
  };
  auto vec = [&](const char* key, std::size_t n) {
    try {
      return converter<VectorXd>::convert(params_server_->getParam<std::string>(p + key));
    } catch (const std::exception&) {
      return VectorXd(n, 1.0);
    }
  };
  carrot_weight_ = scalar("carrot_weight", 10.0);
  carrot_tail_weight_ = scalar("carrot_tail_weight", 5.0);
  carrot_tail_act_weights_ = vec("carrot_tail_act_weights", get_ndx());
  control_reg_weight_ = scalar("carrot_control_reg_weight", 1e-2);
  control_reg_act_weights_ = vec("carrot_control_reg_act_weights", get_nu());
  state_reg_weight_ = scalar("carrot_state_reg_weight", 1e-3);
  state_ref_act_weights_ = vec("carrot_state_ref_act_weights", get_ndx());
  state_limits_weight_ = scalar("carrot_state_limits_weight", 100);
  state_limits_act_weights_ = vec("carrot_state_limits_act_weights", get_ndx());
  // the two bound vectors have no default: a missing key propagates the ParamsServer exception (:162-175)
  state_limits_l_bound_ = converter<VectorXd>::convert(params_server_->getParam<std::string>(p + "carrot_state_limits_l_bound"));
  state_limits_u_bound_ = converter<VectorXd>::convert(params_server_->getParam<std::string>(p + "carrot_state_limits_u_bound"));
  // The reference builds its std::runtime_error objects for wrong vector sizes without throwing them (:83,:104,...);
  // the mismatch surfaces when the Crocoddyl cost is constructed.  Same here: createCosts() throws.
}

static EmpcCost blank_cost(const char* name, int type, int activation, int nr, double weight, bool active) {
  EmpcCost c;
  std::memset(&c, 0, sizeof(c));
  std::strncpy(c.name, name, EMPC_NAME_LEN - 1);
  c.type = type;
  c.activation = activation;
  c.active = active ? 1 : 0;
  c.frame = -1;
  c.nr = nr;
  c.ref_share = -1;
  c.weight = weight;
  const double inf = std::numeric_limits<double>::infinity();
  for (int i = 0; i < EMPC_MAX_NR; ++i) {
    c.act_w[i] = 1.0;
    c.lb[i] = -inf;
    c.ub[i] = inf;
  }
  return c;
}

static void set_weights(EmpcCost& c, const VectorXd& w, std::size_t nr, const char* what) {
  if (w.size() != nr)
    throw std::invalid_argument(std::string("CarrotMpc: ") + what + " has dimension " + std::to_string(w.size()) +
                                ", should be " + std::to_string(nr));
  for (std::size_t i = 0; i < nr; ++i) c.act_w[i] = w[i];
}

EmpcCostSet CarrotMpc::createCosts() const {  // :250-296
  const std::size_t nx = get_nx(), ndx = get_ndx(), nu = get_nu();
  const VectorXd zero = zero_state();
  CostModelSum costs;

  EmpcCost state_reg = blank_cost("state_reg", EMPC_COST_STATE, EMPC_ACT_WEIGHTED_QUAD, (int)ndx, state_reg_weight_, true);
  set_weights(state_reg, state_ref_act_weights_, ndx, "state regularization activation weights vector");
  for (std::size_t i = 0; i < nx; ++i) state_reg.ref[i] = zero[i];
  costs.addCost("state_reg", state_reg, state_reg_weight_, true);

  EmpcCost control_reg = blank_cost("control_reg", EMPC_COST_CONTROL, EMPC_ACT_WEIGHTED_QUAD, (int)nu, control_reg_weight_, true);
  set_weights(control_reg, control_reg_act_weights_, nu, "control activation weights vector");
  costs.addCost("control_reg", control_reg, control_reg_weight_, true);

  EmpcCost limits = blank_cost("state_limits", EMPC_COST_STATE, EMPC_ACT_WEIGHTED_QUADRATIC_BARRIER, (int)ndx,
                               state_limits_weight_, true);
  set_weights(limits, state_limits_act_weights_, ndx, "state limits activation weights vector");
  if (state_limits_l_bound_.size() != ndx || state_limits_u_bound_.size() != ndx)
    throw std::invalid_argument("CarrotMpc: the dimension for the state limits vectors should be " + std::to_string(ndx));
  for (std::size_t i = 0; i < ndx; ++i) {  // ActivationBounds(lb, ub, beta = 1)
    limits.lb[i] = state_limits_l_bound_[i];
    limits.ub[i] = state_limits_u_bound_[i];
  }
  for (std::size_t i = 0; i < nx; ++i) limits.ref[i] = zero[i];
  costs.addCost("state_limits", limits, state_limits_weight_, true);

  EmpcCost carrot = blank_cost("carrot_state", EMPC_COST_STATE, EMPC_ACT_QUAD, (int)ndx, carrot_weight_, false);
  for (std::size_t i = 0; i < nx; ++i) carrot.ref[i] = zero[i];
  costs.addCost("carrot_state", carrot, carrot_weight_, false);

  EmpcCost tail = blank_cost("carrot_tail", EMPC_COST_STATE, EMPC_ACT_WEIGHTED_QUAD, (int)ndx, carrot_tail_weight_, false);
  set_weights(tail, carrot_tail_act_weights_, ndx, "tail activation weights vector");
  for (std::size_t i = 0; i < nx; ++i) tail.ref[i] = zero[i];
  costs.addCost("carrot_tail", tail, carrot_tail_weight_, false);

  return makeCostSet(costs, ContactModelMultiple());
}

void CarrotMpc::createProblem() {  // :178-248
  if (trajectory_->get_has_contact()) throw std::runtime_error("Carrot with contact has not been implemented");  // :204
  if (params_.knots < 2) throw std::invalid_argument("CarrotMpc: knots must be >= 2");
  std::vector<EmpcCostSet> sets;
  std::vector<int> knot_set;
  for (std::size_t i = 0; i < params_.knots; ++i) {  // one private action model per knot
    sets.push_back(createCosts());
    knot_set.push_back((int)i);
  }
  const int integrator = params_.integrator_type == IntegratedActionModelTypes::IntegratedActionModelEuler
                             ? EMPC_INTEGRATOR_EULER
                             : EMPC_INTEGRATOR_RK4;
  const bool squash = params_.solver_type == SolverTypes::SolverSbFDDP;  // actuation_squash_ vs actuation_ (:188-193)
  // running models = the first knots-1, terminal = the last (:226-229); x0 = state->zero()
  problem_ = std::make_shared<ShootingProblem>(zero_state(), robot_model_->descWithFrames(std::vector<int>()), sets, knot_set,
                                               platform_params_->tau_f_, platform_params_->u_lb, platform_params_->u_ub,
                                               double(params_.dt) / 1000.0, false, squash, integrator);
}

void CarrotMpc::updateProblem(const std::size_t& current_time) {  // :298-313
  computeActiveStage(current_time);
  update_vars_.idx_last_stage = update_vars_.idx_stage;
  for (std::size_t i = 0; i < params_.knots; ++i) {
    update_vars_.node_time = current_time + i * params_.dt;
    computeActiveStage(update_vars_.node_time);
    if (trajectory_->get_has_contact())
      updateContactCosts(i);
    else
      updateFreeCosts(i, current_time);
    update_vars_.idx_last_stage = update_vars_.idx_stage;
  }
}

void CarrotMpc::computeActiveStage(const std::size_t& current_time) {  // :315-319
  update_vars_.idx_stage = std::size_t(std::upper_bound(t_stages_.begin(), t_stages_.end(), current_time) - t_stages_.begin()) - 1;
}

void CarrotMpc::updateContactCosts(const std::size_t&) {}  // :329

static EmpcCost& find_cost(EmpcCostSet& set, const char* name) {
  for (int i = 0; i < set.ncosts; ++i)
    if (std::strcmp(set.costs[i].name, name) == 0) return set.costs[i];
  throw std::out_of_range("map::at");
}

void CarrotMpc::updateFreeCosts(const std::size_t& idx, const std::size_t&) {  // :331-362
  EmpcCostSet& set = problem_->get_sets().at(idx);
  const auto& stages = trajectory_->get_stages();
  EmpcCost& carrot = find_cost(set, "carrot_state");
  if (update_vars_.idx_stage < stages.size()) {
    if (!stages[update_vars_.idx_stage]->get_is_transition() || idx == params_.knots - 1) {
      carrot.active = 1;
      computeStateReference(update_vars_.node_time);
      for (std::size_t i = 0; i < get_nx(); ++i) carrot.ref[i] = update_vars_.state_ref[i];
    } else {
      carrot.active = 0;
    }
  } else {
    EmpcCost& tail = find_cost(set, "carrot_tail");
    carrot.active = 0;
    tail.active = 1;
    computeStateReference(update_vars_.node_time);
    for (std::size_t i = 0; i < get_nx(); ++i) tail.ref[i] = update_vars_.state_ref[i];
  }
}

const VectorXd& CarrotMpc::computeStateReference(const std::size_t& time) {  // :384-403
  const std::size_t nq = (std::size_t)robot_model_->nq(), nv = (std::size_t)robot_model_->nv();
  if (state_ref_.empty()) throw std::runtime_error("CarrotMpc: empty state reference");
  update_vars_.idx_state = std::size_t(std::upper_bound(t_ref_.begin(), t_ref_.end(), time) - t_ref_.begin());
  if (update_vars_.idx_state >= state_ref_.size()) {
    update_vars_.state_ref = zero_state();
    for (std::size_t i = 0; i < nq; ++i) update_vars_.state_ref[i] = state_ref_.back()[i];
  } else {
    // The reference divides two std::size_t values (:390-391): the quotient is 0 for every time inside
    // [t_ref[idx-1], t_ref[idx]), so pinocchio::interpolate(q0, q1, 0) returns q0 and the velocity part is v0.
    const std::size_t i1 = update_vars_.idx_state, i0 = i1 - 1;
    const std::size_t quotient = (time - t_ref_[i0]) / (t_ref_[i1] - t_ref_[i0]);
    update_vars_.alpha = (double)quotient;
    if (quotient != 0) throw std::logic_error("CarrotMpc: non-zero integer interpolation factor");
    for (std::size_t i = 0; i < nq; ++i) update_vars_.state_ref[i] = state_ref_[i0][i];
    for (std::size_t i = 0; i < nv; ++i) update_vars_.state_ref[nq + i] = state_ref_[i0][nq + i];
  }
  return update_vars_.state_ref;
}

// -----------------------------------------------------------------------------------------------------
// RailMpc (src/mpc-controllers/rail-mpc.cpp)
// -----------------------------------------------------------------------------------------------------
RailMpc::RailMpc(const std::vector<VectorXd>& state_ref, std::size_t dt_ref, const std::string& yaml_path)
    : MpcAbstract(yaml_path) {  // :14-60
  state_ref_ = state_ref;
  for (const auto& x : state_ref_)
    if (x.size() != get_nx()) throw std::invalid_argument("RailMpc: state_ref[i] has wrong dimension");
  for (std::size_t i = 0; i < state_ref_.size(); ++i) t_ref_.push_back(dt_ref * i);

  const std::string p = "mpc_controller/";
  try {
    state_weight_ = params_server_->getParam<double>(p + "rail_weight");
  } catch (const std::exception&) {
    state_weight_ = 10;
  }
  try {
    state_activation_weights_ = converter<VectorXd>::convert(params_server_->getParam<std::string>(p + "rail_activation_weights"));
  } catch (const std::exception&) {
    state_activation_weights_ = VectorXd(get_ndx(), 1.0);
  }
  try {
    control_weight_ = params_server_->getParam<double>(p + "rail_control_weight");
  } catch (const std::exception&) {
    control_weight_ = 1e-1;
  }
  createProblem();
  update_vars_.state_ref = zero_state();
}

EmpcCostSet RailMpc::createCosts() const {  // :128-149
  const std::size_t nx = get_nx(), ndx = get_ndx(), nu = get_nu();
  const VectorXd zero = zero_state();
  CostModelSum costs;
  EmpcCost rail = blank_cost("rail_state", EMPC_COST_STATE, EMPC_ACT_WEIGHTED_QUAD, (int)ndx, state_weight_, true);
  // the reference builds this error without throwing it (:42-46); the mismatch then fails inside CostModelResidual
  if (state_activation_weights_.size() != ndx)
    throw std::invalid_argument("RailMPC: the dimension for the state activation weights vector is " +
                                std::to_string(state_activation_weights_.size()) + ", should be " + std::to_string(ndx));
  for (std::size_t i = 0; i < ndx; ++i) rail.act_w[i] = state_activation_weights_[i];
  for (std::size_t i = 0; i < nx; ++i) rail.ref[i] = zero[i];
  costs.addCost("rail_state", rail, state_weight_, true);
  EmpcCost control = blank_cost("control", EMPC_COST_CONTROL, EMPC_ACT_QUAD, (int)nu, control_weight_, true);
  costs.addCost("control", control, control_weight_, true);
  return makeCostSet(costs, ContactModelMultiple());
}

// the receding-horizon problem shared by the controllers: knots private action models, x0 = state->zero()
static std::shared_ptr<ShootingProblem> horizon_problem(const MpcAbstract& mpc, const MpcParams& params,
                                                        const std::vector<EmpcCostSet>& sets, const std::vector<int>& frames) {
  if (params.knots < 2) throw std::invalid_argument("MPC: knots must be >= 2");
  std::vector<int> knot_set;
  for (std::size_t i = 0; i < params.knots; ++i) knot_set.push_back((int)i);
  const int integrator = params.integrator_type == IntegratedActionModelTypes::IntegratedActionModelEuler ? EMPC_INTEGRATOR_EULER
                                                                                                           : EMPC_INTEGRATOR_RK4;
  const bool squash = params.solver_type == SolverTypes::SolverSbFDDP;  // actuation_squash_ vs actuation_
  const auto& platform = mpc.get_platform_params();
  return std::make_shared<ShootingProblem>(mpc.zero_state(), mpc.get_robot_model()->descWithFrames(frames), sets, knot_set,
                                           platform->tau_f_, platform->u_lb, platform->u_ub, double(params.dt) / 1000.0, false,
                                           squash, integrator);
}

void RailMpc::createProblem() {  // :64-126 (always the free-flight dynamics, :66-67)
  std::vector<EmpcCostSet> sets;
  for (std::size_t i = 0; i < params_.knots; ++i) sets.push_back(createCosts());
  problem_ = horizon_problem(*this, params_, sets, std::vector<int>());
}

void RailMpc::updateProblem(const std::size_t& current_time) {  // :151-161
  for (std::size_t i = 0; i < params_.knots; ++i) {
    update_vars_.node_time = current_time + i * params_.dt;
    updateFreeCosts(i);
  }
}

void RailMpc::updateFreeCosts(const std::size_t& idx) {  // :165-174
  EmpcCost& rail = find_cost(problem_->get_sets().at(idx), "rail_state");
  computeStateReference(update_vars_.node_time);
  for (std::size_t i = 0; i < get_nx(); ++i) rail.ref[i] = update_vars_.state_ref[i];
}

const VectorXd& RailMpc::computeStateReference(const std::size_t& time) {  // :176-200
  const std::size_t nq = (std::size_t)robot_model_->nq(), nv = (std::size_t)robot_model_->nv();
  if (state_ref_.empty()) throw std::runtime_error("RailMpc: empty state reference");
  update_vars_.idx_state = std::size_t(std::upper_bound(t_ref_.begin(), t_ref_.end(), time) - t_ref_.begin());
  if (update_vars_.idx_state >= state_ref_.size()) {
    // hover at the last planned configuration, yaw kept: q = last q with (qz, qw) replaced by the normalised
    // (0, 0, qz, qw) quaternion; qx, qy stay as copied (the reference overwrites entries 5 and 6 only, :181-185)
    update_vars_.state_ref = zero_state();
    const VectorXd& last = state_ref_.back();
    for (std::size_t i = 0; i < nq; ++i) update_vars_.state_ref[i] = last[i];
    double w = last[6], z = last[5];
    const double n2 = w * w + z * z;
    if (n2 > 0) {  // Eigen's normalize() leaves a zero quaternion untouched
      const double n = std::sqrt(n2);
      w /= n;
      z /= n;
    }
    update_vars_.state_ref[5] = z;
    update_vars_.state_ref[6] = w;
  } else {
    // same std::size_t quotient as CarrotMpc (:187-188): always 0, so the reference is the sample at or before `time`
    const std::size_t i1 = update_vars_.idx_state, i0 = i1 - 1;
    const std::size_t quotient = (time - t_ref_[i0]) / (t_ref_[i1] - t_ref_[i0]);
    update_vars_.alpha = (double)quotient;
    if (quotient != 0) throw std::logic_error("RailMpc: non-zero integer interpolation factor");
    for (std::size_t i = 0; i < nq + nv; ++i) update_vars_.state_ref[i] = state_ref_[i0][i];
  }
  return update_vars_.state_ref;
}

// -----------------------------------------------------------------------------------------------------
// WeightedMpc (src/mpc-controllers/weighted-mpc.cpp)
// -----------------------------------------------------------------------------------------------------
WeightedMpc::WeightedMpc(const std::shared_ptr<Trajectory>& trajectory, std::size_t, const std::string& yaml_path)
    : MpcAbstract(yaml_path), trajectory_(trajectory) {  // :16-72 (dt_ref is unused by the reference too)
  if (!trajectory) throw std::invalid_argument("WeightedMpc: trajectory is null");
  auto scalar = [&](const char* key, double fallback) {
    try {
      return params_server_->getParam<double>(std::string("mpc_controller/") + key);
    } catch (const std::exception&) {
      return fallback;
    }
  };
  alpha_ = scalar("weighted_alpha", 20.0);
  beta_ = scalar("weighted_beta", 1.0);
  state_reg_ = scalar("weighted_state_reg", 1e-1);      // loaded and never used, as in the reference
  control_reg_ = scalar("weighted_control_reg", 1e-1);  // idem

  // every transition stage is folded into its successor, which then starts where the transition started (:57-69)
  for (std::size_t i = 0; i < trajectory_->get_stages().size(); ++i) {
    const auto& stages = trajectory_->get_stages();
    if (stages[i]->get_is_transition()) {
      if (i + 1 >= stages.size())  // the reference indexes past the end here
        throw std::runtime_error("WeightedMpc: the last stage of the trajectory is a transition");
      stages[i + 1]->set_duration(stages[i]->get_duration() + stages[i + 1]->get_duration());
      stages[i + 1]->set_t_ini(stages[i]->get_t_ini());
      trajectory_->removeStage(i);
    }
    t_stages_.push_back(trajectory_->get_stages()[i]->get_t_ini());
  }
  createProblem();
}

CostModelSum WeightedMpc::createCosts() const {  // :145-168
  CostModelSum costs;
  for (const auto& stage : trajectory_->get_stages()) {
    if (stage->get_is_transition()) continue;
    // The reference re-creates each cost from the trajectory's parameter server with the same path and type
    // (:156-158); the stage already holds exactly that object, so it is copied.  Added inactive, with the stage's weight.
    for (const auto& ctype : stage->get_cost_types()) {
      const EmpcCost& cost = stage->get_costs()->get_costs().at(ctype.first);
      const std::string name = stage->get_name() + "/" + ctype.first;
      if (name.size() >= EMPC_NAME_LEN) throw std::runtime_error("WeightedMpc: cost name too long: " + name);
      costs.addCost(name, cost, cost.weight, false);
    }
  }
  return costs;
}

void WeightedMpc::createProblem() {  // :76-143
  if (trajectory_->get_has_contact()) throw std::runtime_error("Weighted with contact has not been implemented");  // :100
  const CostModelSum costs = createCosts();
  cost_names_.clear();
  for (const auto& kv : costs.get_costs()) cost_names_.push_back(kv.first);
  const EmpcCostSet set = makeCostSet(costs, ContactModelMultiple());
  std::vector<EmpcCostSet> sets(params_.knots, set);
  problem_ = horizon_problem(*this, params_, sets, trajectory_->frame_table().ids());
}

void WeightedMpc::updateProblem(const std::size_t& current_time) {  // :170-185
  computeActiveStage(current_time);
  update_vars_.idx_last_stage = update_vars_.idx_stage;
  for (std::size_t i = 0; i < params_.knots; ++i) {
    update_vars_.node_time = current_time + i * params_.dt;
    computeActiveStage(update_vars_.node_time, update_vars_.idx_last_stage);
    update_vars_.name_stage = trajectory_->get_stages().at(update_vars_.idx_stage)->get_name();
    updateFreeCosts(i);
    update_vars_.idx_last_stage = update_vars_.idx_stage;
  }
}

void WeightedMpc::computeActiveStage(const std::size_t& current_time) {  // :187-191
  update_vars_.idx_stage = std::size_t(std::upper_bound(t_stages_.begin(), t_stages_.end(), current_time) - t_stages_.begin()) - 1;
}

void WeightedMpc::computeActiveStage(const std::size_t& current_time, const std::size_t& last_stage) {  // :193-199
  computeActiveStage(current_time);
  if (update_vars_.idx_stage == last_stage + 2) update_vars_.idx_stage -= 1;  // never skip a stage between two knots
}

void WeightedMpc::updateFreeCosts(const std::size_t& idx) {  // :203-228
  EmpcCostSet& set = problem_->get_sets().at(idx);
  const std::string& stage_name = update_vars_.name_stage;
  const auto& stage = trajectory_->get_stages().at(update_vars_.idx_stage);
  for (std::size_t c = 0; c < cost_names_.size(); ++c) {
    const std::string& name = cost_names_[c];
    EmpcCost& cost = set.costs[c];
    if (name.compare(0, stage_name.size(), stage_name) == 0) {  // prefix match, as in the reference
      cost.active = 1;
      if (name.compare(stage_name.size(), 4, "/reg") != 0 && name.compare(stage_name.size(), 7, "/limits") != 0) {
        computeWeight(update_vars_.node_time);
        // .at() throws std::out_of_range when the prefix matched a longer stage name, like the reference's map::at
        cost.weight = stage->get_costs()->get_costs().at(name.substr(stage_name.size() + 1)).weight * update_vars_.weight * beta_;
      }
    } else {
      cost.active = 0;  // ("barrier" is exempt in the reference, :223; here the barrier lives inside the solver)
    }
  }
}

void WeightedMpc::computeWeight(const std::size_t& time) {  // :230-243
  if (time > trajectory_->get_duration()) {  // saturate once the knot is beyond the end of the trajectory
    update_vars_.weight_time = 0.0;
  } else {
    const auto& stage = trajectory_->get_stages().at(update_vars_.idx_stage);
    update_vars_.weight_time = ((int)time - ((int)stage->get_t_ini() + (int)stage->get_duration())) / 1000.0;
  }
  update_vars_.weight = std::exp(alpha_ * update_vars_.weight_time);
}

}  // namespace eagle_mpc
