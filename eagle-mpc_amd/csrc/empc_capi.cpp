// C ABI: error plumbing + YAML problem factory entry points (host only).  Solver entry points live in
// empc_solver.hip.  See include/empc.h for the reference interfaces each function replaces.
#include <cstring>
#include <exception>
#include <string>

#include "../host/eagle_mpc.hpp"
#include "empc_internal.hpp"

using namespace eagle_mpc;

namespace empc {
static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
}  // namespace empc

struct EmpcTrajectory {
  std::shared_ptr<Trajectory> t;
};
struct EmpcProblem {
  std::shared_ptr<ShootingProblem> p;
};

#define EMPC_TRY try {
#define EMPC_CATCH(ret)                       \
  }                                           \
  catch (const std::exception& e) {           \
    empc::set_last_error(e.what());           \
    return ret;                               \
  }                                           \
  catch (...) {                               \
    empc::set_last_error("unknown exception"); \
    return ret;                               \
  }

extern "C" {

const char* empc_last_error(void) { return empc::g_last_error.c_str(); }
const char* empc_version(void) { return "eagle-mpc_amd 0.1 (gfx950)"; }

int empc_set_data_dirs(const char* yaml_dir, const char* robot_data_dir) {
  EMPC_TRY
  if (yaml_dir) set_yaml_dir(yaml_dir);
  if (robot_data_dir) set_robot_data_dir(robot_data_dir);
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}

EmpcTrajectory* empc_trajectory_create(const char* yaml_path) {
  EMPC_TRY
  if (!yaml_path) throw std::invalid_argument("yaml_path is NULL");
  auto t = Trajectory::create();
  t->autoSetup(yaml_path);
  return new EmpcTrajectory{t};
  EMPC_CATCH(nullptr)
}
void empc_trajectory_destroy(EmpcTrajectory* t) { delete t; }

int empc_trajectory_dims(const EmpcTrajectory* t, int* nx, int* ndx, int* nu, int* n_stages, int* has_contact,
                         int* duration_ms) {
  EMPC_TRY
  if (!t) throw std::invalid_argument("trajectory is NULL");
  if (nx) *nx = (int)t->t->get_nx();
  if (ndx) *ndx = (int)t->t->get_ndx();
  if (nu) *nu = (int)t->t->get_nu();
  if (n_stages) *n_stages = (int)t->t->get_stages().size();
  if (has_contact) *has_contact = t->t->get_has_contact() ? 1 : 0;
  if (duration_ms) *duration_ms = (int)t->t->get_duration();
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}

/* Trajectory::removeStage (src/trajectory.cpp:145-150): erases the stage; like the reference it leaves duration_ and the
 * t_ini of the other stages alone (WeightedMpc merges the durations itself before it removes a transition stage) */
int empc_trajectory_remove_stage(EmpcTrajectory* t, int stage) {
  EMPC_TRY
  if (!t) throw std::invalid_argument("trajectory is NULL");
  if (stage < 0 || (size_t)stage >= t->t->get_stages().size()) throw std::out_of_range("removeStage: no such stage");
  t->t->removeStage((size_t)stage);
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}
/* Trajectory::get_robot_model_path (src/trajectory.cpp:160): the URDF file the robot was built from; returns its length */
int empc_trajectory_robot_model_path(const EmpcTrajectory* t, char* path, int path_len) {
  EMPC_TRY
  if (!t) throw std::invalid_argument("trajectory is NULL");
  const std::string& p = t->t->get_robot_model_path();
  if (path && path_len > 0) {
    std::strncpy(path, p.c_str(), (size_t)path_len - 1);
    path[path_len - 1] = 0;
  }
  return (int)p.size();
  EMPC_CATCH(EMPC_ERR_INVALID)
}

int empc_trajectory_stage_info(const EmpcTrajectory* t, int stage, char* name, int name_len, int* duration_ms,
                               int* is_transition, int* n_costs, int* n_contacts) {
  EMPC_TRY
  if (!t) throw std::invalid_argument("trajectory is NULL");
  const auto& st = t->t->get_stages().at((size_t)stage);
  if (name && name_len > 0) {
    std::strncpy(name, st->get_name().c_str(), (size_t)name_len - 1);
    name[name_len - 1] = 0;
  }
  if (duration_ms) *duration_ms = (int)st->get_duration();
  if (is_transition) *is_transition = st->get_is_transition() ? 1 : 0;
  if (n_costs) *n_costs = (int)st->get_costs()->get_costs().size();
  if (n_contacts) *n_contacts = (int)st->get_contacts()->get_contacts().size();
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}

long long empc_trajectory_stage_t_ini(const EmpcTrajectory* t, int stage) {
  EMPC_TRY
  if (!t) throw std::invalid_argument("trajectory is NULL");
  return (long long)t->t->get_stages().at((size_t)stage)->get_t_ini();
  EMPC_CATCH(-1)
}
int empc_trajectory_stage_cost(const EmpcTrajectory* t, int stage, int cost, char* name, int name_len, double* weight,
                               int* active) {
  EMPC_TRY
  if (!t) throw std::invalid_argument("trajectory is NULL");
  const auto& costs = t->t->get_stages().at((size_t)stage)->get_costs()->get_costs();
  if (cost < 0 || (size_t)cost >= costs.size()) throw std::out_of_range("cost index out of range");
  auto it = costs.begin();
  std::advance(it, cost);
  if (name && name_len > 0) {
    std::strncpy(name, it->first.c_str(), (size_t)name_len - 1);
    name[name_len - 1] = 0;
  }
  if (weight) *weight = it->second.weight;
  if (active) *active = it->second.active;
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}

static void copy_out(const std::string& v, char* dst, int len) {
  if (dst && len > 0) {
    std::strncpy(dst, v.c_str(), (size_t)len - 1);
    dst[len - 1] = 0;
  }
}
int empc_trajectory_stage_cost_type(const EmpcTrajectory* t, int stage, const char* cost_name, char* type, int type_len) {
  EMPC_TRY
  if (!t || !cost_name) throw std::invalid_argument("NULL argument");
  static const char* names[] = {"CostModelState", "CostModelControl", "CostModelFramePlacement", "CostModelFrameRotation",
                                "CostModelFrameVelocity", "CostModelFrameTranslation", "CostModelContactFrictionCone"};
  const auto& types = t->t->get_stages().at((size_t)stage)->get_cost_types();
  const auto it = types.find(cost_name);
  if (it == types.end()) throw std::out_of_range(std::string("no cost named '") + cost_name + "' in this stage");
  copy_out(names[(int)it->second], type, type_len);
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}
int empc_trajectory_stage_contact(const EmpcTrajectory* t, int stage, int contact, char* name, int name_len, char* type, int type_len) {
  EMPC_TRY
  if (!t) throw std::invalid_argument("trajectory is NULL");
  static const char* names[] = {"ContactModel2D", "ContactModel3D", "ContactModel6D"};
  const auto& st = t->t->get_stages().at((size_t)stage);
  const auto& contacts = st->get_contacts()->get_contacts();
  if (contact < 0 || (size_t)contact >= contacts.size()) throw std::out_of_range("contact index out of range");
  auto it = contacts.begin();
  std::advance(it, contact);
  copy_out(it->first, name, name_len);
  copy_out(names[(int)st->get_contact_types().at(it->first)], type, type_len);
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}
int empc_trajectory_get_platform_params(const EmpcTrajectory* t, double* scalars, char* base_link_name, int name_len) {
  EMPC_TRY
  if (!t) throw std::invalid_argument("trajectory is NULL");
  const auto& pp = t->t->get_platform_params();
  if (scalars) {
    const double v[6] = {pp->cf_, pp->cm_, pp->max_thrust_, pp->min_thrust_, pp->max_prop_speed_, pp->min_prop_speed_};
    std::memcpy(scalars, v, sizeof(v));
  }
  copy_out(pp->base_link_name_, base_link_name, name_len);
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}
int empc_trajectory_get_rotor_pose(const EmpcTrajectory* t, int rotor, double* R, double* p, int* spin_direction) {
  EMPC_TRY
  if (!t) throw std::invalid_argument("trajectory is NULL");
  const auto& pp = t->t->get_platform_params();
  if (rotor < 0 || (size_t)rotor >= pp->rotors_pose_.size()) throw std::out_of_range("rotor index out of range");
  if (R) std::memcpy(R, pp->rotors_pose_[(size_t)rotor].R, sizeof(double) * 9);
  if (p) std::memcpy(p, pp->rotors_pose_[(size_t)rotor].p, sizeof(double) * 3);
  if (spin_direction) *spin_direction = pp->rotors_spin_dir_[(size_t)rotor];
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}

int empc_trajectory_get_initial_state(const EmpcTrajectory* t, double* x0) {
  EMPC_TRY
  if (!t || !x0) throw std::invalid_argument("NULL argument");
  const auto& x = t->t->get_initial_state();
  std::memcpy(x0, x.data(), sizeof(double) * x.size());
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}
int empc_trajectory_set_initial_state(EmpcTrajectory* t, const double* x0) {
  EMPC_TRY
  if (!t || !x0) throw std::invalid_argument("NULL argument");
  t->t->set_initial_state(VectorXd(x0, x0 + t->t->get_nx()));
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}
int empc_trajectory_get_platform(const EmpcTrajectory* t, double* tau_f, double* u_lb, double* u_ub, int* n_rotors) {
  EMPC_TRY
  if (!t) throw std::invalid_argument("trajectory is NULL");
  const auto& pp = t->t->get_platform_params();
  if (n_rotors) *n_rotors = (int)pp->n_rotors_;
  if (tau_f) std::memcpy(tau_f, pp->tau_f_.data.data(), sizeof(double) * pp->tau_f_.data.size());
  if (u_lb) std::memcpy(u_lb, pp->u_lb.data(), sizeof(double) * pp->u_lb.size());
  if (u_ub) std::memcpy(u_ub, pp->u_ub.data(), sizeof(double) * pp->u_ub.size());
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}

EmpcProblem* empc_trajectory_create_problem(const EmpcTrajectory* t, int dt_ms, int squash, const char* integration_method) {
  EMPC_TRY
  if (!t) throw std::invalid_argument("trajectory is NULL");
  std::shared_ptr<ShootingProblem> p;
  if (dt_ms <= 0)
    p = t->t->createProblem();
  else
    p = t->t->createProblem((std::size_t)dt_ms, squash != 0,
                            integration_method ? integration_method : "IntegratedActionModelEuler");
  return new EmpcProblem{p};
  EMPC_CATCH(nullptr)
}
void empc_problem_destroy(EmpcProblem* p) { delete p; }
const EmpcProblemDesc* empc_problem_desc(EmpcProblem* p) {
  EMPC_TRY
  if (!p) throw std::invalid_argument("problem is NULL");
  return &p->p->desc();
  EMPC_CATCH(nullptr)
}
int empc_problem_set_x0(EmpcProblem* p, const double* x0) {
  EMPC_TRY
  if (!p || !x0) throw std::invalid_argument("NULL argument");
  p->p->set_x0(VectorXd(x0, x0 + p->p->get_nx()));
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}
int empc_trajectory_get_param(const EmpcTrajectory* t, const char* key, char* value, int value_len) {
  EMPC_TRY
  if (!t || !key) throw std::invalid_argument("NULL argument");
  const std::string v = t->t->get_params_server()->getParam<std::string>(key);
  if (value && value_len > 0) {
    std::strncpy(value, v.c_str(), (size_t)value_len - 1);
    value[value_len - 1] = 0;
  }
  return (int)v.size();
  EMPC_CATCH(EMPC_ERR_INVALID)
}

// ---- Carrot MPC ------------------------------------------------------------------------------------------------
struct EmpcCarrotMpc {
  std::shared_ptr<CarrotMpc> m;
};

EmpcCarrotMpc* empc_carrot_mpc_create(const EmpcTrajectory* t, const double* state_ref, int n_ref, int dt_ref_ms,
                                      const char* mpc_yaml_path) {
  EMPC_TRY
  if (!t || !mpc_yaml_path) throw std::invalid_argument("NULL argument");
  if (n_ref < 0 || dt_ref_ms < 0 || (n_ref > 0 && !state_ref)) throw std::invalid_argument("bad state reference");
  const std::size_t nx = t->t->get_nx();
  std::vector<VectorXd> ref((std::size_t)n_ref);
  for (int i = 0; i < n_ref; ++i) ref[(std::size_t)i].assign(state_ref + (std::size_t)i * nx, state_ref + (std::size_t)(i + 1) * nx);
  return new EmpcCarrotMpc{std::make_shared<CarrotMpc>(t->t, ref, (std::size_t)dt_ref_ms, mpc_yaml_path)};
  EMPC_CATCH(nullptr)
}
void empc_carrot_mpc_destroy(EmpcCarrotMpc* m) { delete m; }
int empc_carrot_mpc_params(const EmpcCarrotMpc* m, int* knots, int* iters, int* dt_ms, int* nx, int* ndx, int* nu,
                           int* n_t_stages) {
  EMPC_TRY
  if (!m) throw std::invalid_argument("controller is NULL");
  if (knots) *knots = (int)m->m->get_knots();
  if (iters) *iters = (int)m->m->get_iters();
  if (dt_ms) *dt_ms = (int)m->m->get_dt();
  if (nx) *nx = (int)m->m->get_nx();
  if (ndx) *ndx = (int)m->m->get_ndx();
  if (nu) *nu = (int)m->m->get_nu();
  if (n_t_stages) *n_t_stages = (int)m->m->get_t_stages().size();
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}
int empc_carrot_mpc_t_stages(const EmpcCarrotMpc* m, long long* t_stages) {
  EMPC_TRY
  if (!m || !t_stages) throw std::invalid_argument("NULL argument");
  const auto& ts = m->m->get_t_stages();
  for (std::size_t i = 0; i < ts.size(); ++i) t_stages[i] = (long long)ts[i];
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}
int empc_carrot_mpc_update_problem(EmpcCarrotMpc* m, long long current_time_ms) {
  EMPC_TRY
  if (!m) throw std::invalid_argument("controller is NULL");
  if (current_time_ms < 0) throw std::invalid_argument("current_time must be >= 0");
  m->m->updateProblem((std::size_t)current_time_ms);
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}
int empc_carrot_mpc_state_reference(EmpcCarrotMpc* m, long long time_ms, double* xref) {
  EMPC_TRY
  if (!m || !xref) throw std::invalid_argument("NULL argument");
  if (time_ms < 0) throw std::invalid_argument("time must be >= 0");
  const VectorXd& r = m->m->computeStateReference((std::size_t)time_ms);
  std::memcpy(xref, r.data(), sizeof(double) * r.size());
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}
int empc_carrot_mpc_set_x0(EmpcCarrotMpc* m, const double* x0) {
  EMPC_TRY
  if (!m || !x0) throw std::invalid_argument("NULL argument");
  m->m->get_problem()->set_x0(VectorXd(x0, x0 + m->m->get_nx()));
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}
const EmpcProblemDesc* empc_carrot_mpc_problem_desc(EmpcCarrotMpc* m) {
  EMPC_TRY
  if (!m) throw std::invalid_argument("controller is NULL");
  return &m->m->get_problem()->desc();
  EMPC_CATCH(nullptr)
}

// ---- Rail / Weighted MPC -----------------------------------------------------------------------------------------
struct EmpcMpc {
  std::shared_ptr<MpcAbstract> m;
  std::shared_ptr<RailMpc> rail;
  std::shared_ptr<WeightedMpc> weighted;
};

EmpcMpc* empc_rail_mpc_create(const double* state_ref, int n_ref, int nx, int dt_ref_ms, const char* mpc_yaml_path) {
  EMPC_TRY
  if (!mpc_yaml_path) throw std::invalid_argument("NULL argument");
  if (n_ref < 0 || nx <= 0 || dt_ref_ms < 0 || (n_ref > 0 && !state_ref)) throw std::invalid_argument("bad state reference");
  std::vector<VectorXd> ref((std::size_t)n_ref);
  for (int i = 0; i < n_ref; ++i)
    ref[(std::size_t)i].assign(state_ref + (std::size_t)i * (std::size_t)nx, state_ref + (std::size_t)(i + 1) * (std::size_t)nx);
  auto r = std::make_shared<RailMpc>(ref, (std::size_t)dt_ref_ms, mpc_yaml_path);
  return new EmpcMpc{r, r, nullptr};
  EMPC_CATCH(nullptr)
}
EmpcMpc* empc_weighted_mpc_create(EmpcTrajectory* t, int dt_ref_ms, const char* mpc_yaml_path) {
  EMPC_TRY
  if (!t || !mpc_yaml_path) throw std::invalid_argument("NULL argument");
  if (dt_ref_ms < 0) throw std::invalid_argument("dt_ref must be >= 0");
  auto w = std::make_shared<WeightedMpc>(t->t, (std::size_t)dt_ref_ms, mpc_yaml_path);
  return new EmpcMpc{w, nullptr, w};
  EMPC_CATCH(nullptr)
}
void empc_mpc_destroy(EmpcMpc* m) { delete m; }
int empc_mpc_params(const EmpcMpc* m, int* knots, int* iters, int* dt_ms, int* nx, int* ndx, int* nu) {
  EMPC_TRY
  if (!m) throw std::invalid_argument("controller is NULL");
  if (knots) *knots = (int)m->m->get_knots();
  if (iters) *iters = (int)m->m->get_iters();
  if (dt_ms) *dt_ms = (int)m->m->get_dt();
  if (nx) *nx = (int)m->m->get_nx();
  if (ndx) *ndx = (int)m->m->get_ndx();
  if (nu) *nu = (int)m->m->get_nu();
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}
/* EmpcSolverType named by the controller's YAML (`solver:`), for either controller handle (NULL for the other) */
int empc_mpc_solver_type(const EmpcCarrotMpc* carrot, const EmpcMpc* other) {
  const eagle_mpc::MpcAbstract* a = carrot ? static_cast<const eagle_mpc::MpcAbstract*>(carrot->m.get())
                                           : (other ? static_cast<const eagle_mpc::MpcAbstract*>(other->m.get()) : nullptr);
  if (!a) return -1;
  switch (a->get_solver_type()) {
    case eagle_mpc::SolverTypes::SolverSbFDDP:
      return EMPC_SOLVER_SBFDDP;
    case eagle_mpc::SolverTypes::SolverBoxFDDP:
      return EMPC_SOLVER_BOXFDDP;
    default:
      return EMPC_SOLVER_BOXDDP;
  }
}
int empc_mpc_update_problem(EmpcMpc* m, long long current_time_ms) {
  EMPC_TRY
  if (!m) throw std::invalid_argument("controller is NULL");
  if (current_time_ms < 0) throw std::invalid_argument("current_time must be >= 0");
  m->m->updateProblem((std::size_t)current_time_ms);
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}
int empc_mpc_set_x0(EmpcMpc* m, const double* x0) {
  EMPC_TRY
  if (!m || !x0) throw std::invalid_argument("NULL argument");
  m->m->get_problem()->set_x0(VectorXd(x0, x0 + m->m->get_nx()));
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}
const EmpcProblemDesc* empc_mpc_problem_desc(EmpcMpc* m) {
  EMPC_TRY
  if (!m) throw std::invalid_argument("controller is NULL");
  return &m->m->get_problem()->desc();
  EMPC_CATCH(nullptr)
}
int empc_rail_mpc_state_reference(EmpcMpc* m, long long time_ms, double* xref) {
  EMPC_TRY
  if (!m || !xref) throw std::invalid_argument("NULL argument");
  if (!m->rail) throw std::invalid_argument("not a RailMpc handle");
  if (time_ms < 0) throw std::invalid_argument("time must be >= 0");
  const VectorXd& r = m->rail->computeStateReference((std::size_t)time_ms);
  std::memcpy(xref, r.data(), sizeof(double) * r.size());
  return EMPC_OK;
  EMPC_CATCH(EMPC_ERR_INVALID)
}
int empc_weighted_mpc_t_stages(const EmpcMpc* m, long long* t_stages, int capacity) {
  EMPC_TRY
  if (!m) throw std::invalid_argument("controller is NULL");
  if (!m->weighted) throw std::invalid_argument("not a WeightedMpc handle");
  const auto& ts = m->weighted->get_t_stages();
  if (t_stages)
    for (std::size_t i = 0; i < ts.size() && (int)i < capacity; ++i) t_stages[i] = (long long)ts[i];
  return (int)ts.size();
  EMPC_CATCH(EMPC_ERR_INVALID)
}

void empc_solver_params_default(EmpcSolverParams* p) {
  if (!p) return;
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
  p->th_grad = 1e-12;          // crocoddyl::SolverDDP defaults
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

}  // extern "C"
