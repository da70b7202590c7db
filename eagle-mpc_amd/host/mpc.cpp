// MpcAbstract / CarrotMpc: host-side mirror of src/mpc-base.cpp and src/mpc-controllers/carrot-mpc.cpp.
// The controller only edits cost tables (references, active flags); the arithmetic of every solve runs in the HIP
// kernels behind the C ABI.  Where the reference mutates shared Crocoddyl cost models in place, this class edits the
// private EmpcCostSet of each knot and SolverSbFDDP::syncProblem() uploads the tables.
#include <algorithm>
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
    // only the squash-box solver is built here (SolverBoxFDDP / SolverBoxDDP are out of scope)
    if (params_.solver_type != SolverTypes::SolverSbFDDP)
      throw std::runtime_error("MpcAbstract: only solver 'SolverSbFDDP' is available in this build");
    solver_ = std::make_shared<SolverSbFDDP>(problem_, batch_size, device);
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

}  // namespace eagle_mpc
