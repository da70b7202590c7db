// eagle_mpc.hpp: host-side C++ mirror of the eagle-mpc classes that sit on the north-star hot path.
//
// Same class / method names and argument meaning as the reference, with
// std::shared_ptr and plain std::vector<double> in place of boost::shared_ptr / Eigen / Crocoddyl types:
//   MultiCopterBaseParams  include/eagle_mpc/multicopter-base-params.hpp:25-58
//   Stage                  include/eagle_mpc/stage.hpp:33-83
//   Trajectory             include/eagle_mpc/trajectory.hpp:43-103
//   factories              include/eagle_mpc/factory/{cost,activation,contacts,diff-action,int-action}.hpp
//   ShootingProblem        crocoddyl::ShootingProblem as used at src/trajectory.cpp:139-140 (x0, running models,
//                          terminal model) -- here: the flat EmpcProblemDesc the HIP solver consumes
//   SolverSbFDDP           include/eagle_mpc/sbfddp.hpp:34-126 (+ a batched overload; B = 1 is the reference call)
//   MpcAbstract/CarrotMpc  include/eagle_mpc/mpc-base.hpp:62-119, mpc-controllers/carrot-mpc.hpp:23-88
// The arithmetic of the hot path runs in HIP kernels behind the C ABI of include/empc.h; nothing in this
// directory computes dynamics or Riccati recursions on the CPU.
#pragma once
#include <cstddef>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/empc.h"
#include "params.hpp"
#include "robot_model.hpp"

namespace eagle_mpc {

struct MatrixXd {
  int rows = 0, cols = 0;
  std::vector<double> data;  // row-major
  MatrixXd() {}
  MatrixXd(int r, int c) : rows(r), cols(c), data((size_t)r * c, 0.0) {}
  double& operator()(int r, int c) { return data[(size_t)r * cols + c]; }
  double operator()(int r, int c) const { return data[(size_t)r * cols + c]; }
};

struct SE3 {
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double p[3] = {0, 0, 0};
};
// Eigen::Quaterniond(Vector4d(x,y,z,w)).normalized().toRotationMatrix()
void quaternionToRotation(const VectorXd& xyzw, double* R);

enum class CostModelTypes {
  CostModelState,
  CostModelControl,
  CostModelFramePlacement,
  CostModelFrameRotation,
  CostModelFrameVelocity,
  CostModelFrameTranslation,
  CostModelContactFrictionCone,
  NbCostModelTypes
};
enum class ActivationModelTypes {
  ActivationModelQuad,
  ActivationModelQuadFlatExp,
  ActivationModelQuadFlatLog,
  ActivationModelSmooth1Norm,
  ActivationModelSmooth2Norm,
  ActivationModelWeightedQuad,
  ActivationModelQuadraticBarrier,
  ActivationModelWeightedQuadraticBarrier,
  NbActivationModelTypes
};
enum class ContactModelTypes { ContactModel2D, ContactModel3D, ContactModel6D, NbContactModelTypes };
enum class DifferentialActionModelTypes {
  DifferentialActionModelFreeFwdDynamics,
  DifferentialActionModelContactFwdDynamics,
  NbDifferentialActionModelTypes
};
enum class IntegratedActionModelTypes { IntegratedActionModelEuler, IntegratedActionModelRK4, NbIntegratedActionModelTypes };

class MultiCopterBaseParams {
 public:
  MultiCopterBaseParams() {}
  MultiCopterBaseParams(double cf, double cm, const MatrixXd& tau_f, double max_th, double min_th, const std::string& base_link);
  void autoSetup(const std::string& path_to_platform, const std::shared_ptr<ParamsServer>& server);
  void autoSetup(const std::string& path_to_platform, const std::shared_ptr<ParamsServer>& server,
                 const std::shared_ptr<RobotModel>& robot_model);
  void setControlLimits(const std::shared_ptr<RobotModel>& robot_model);

  double cf_ = 0, cm_ = 0;
  std::size_t n_rotors_ = 0;
  MatrixXd tau_f_;  // 6 x n_rotors
  double max_thrust_ = 0, min_thrust_ = 0, max_prop_speed_ = 0, min_prop_speed_ = 0;
  std::string base_link_name_;
  std::vector<SE3> rotors_pose_;
  std::vector<int> rotors_spin_dir_;
  VectorXd u_lb, u_ub;
};

class Trajectory;

// crocoddyl::SquashingModelSmoothSat(u_lb, u_ub, ns) as the reference builds it (src/trajectory.cpp:49-50): plain data -- the
// limits of the smooth saturation and its smoothing.  The kernels take the limits from the problem (the same numbers);
// SolverSbFDDP(problem, squashing_model) checks that they agree.
struct SquashingModelSmoothSat {
  SquashingModelSmoothSat(const VectorXd& u_lb, const VectorXd& u_ub, std::size_t ns) : u_lb_(u_lb), u_ub_(u_ub), ns_(ns) {}
  const VectorXd& get_u_lb() const { return u_lb_; }
  const VectorXd& get_u_ub() const { return u_ub_; }
  const VectorXd& get_s_lb() const { return u_lb_; }  // s_lb = u_lb, s_ub = u_ub
  const VectorXd& get_s_ub() const { return u_ub_; }
  std::size_t get_ns() const { return ns_; }
  double get_smooth() const { return smooth_; }
  void set_smooth(double s) { smooth_ = s; }

 private:
  VectorXd u_lb_, u_ub_;
  std::size_t ns_;
  double smooth_ = 0.1;
};

// Registry of the operational frames a problem references (index = position in EmpcModelDesc's frame table).
class FrameTable {
 public:
  int use(int model_frame_id);
  const std::vector<int>& ids() const { return ids_; }

 private:
  std::vector<int> ids_;
};

class ActivationModelFactory {
 public:
  // fills activation / act_w / lb / ub of `cost` (reference: src/factory/activation.cpp:17-102)
  void create(const std::string& path_to_cost, const std::shared_ptr<ParamsServer>& server, std::size_t nr, EmpcCost& cost) const;
};
class CostModelFactory {
 public:
  // reference: src/factory/cost.cpp:17-171
  EmpcCost create(const std::string& path_to_cost, const std::shared_ptr<ParamsServer>& server,
                  const std::shared_ptr<RobotModel>& robot_model, FrameTable& frames, std::size_t nu,
                  CostModelTypes& cost_type) const;

 private:
  ActivationModelFactory activation_factory_;
};
class ContactModelFactory {
 public:
  // reference: src/factory/contacts.cpp:17-82
  EmpcContact create(const std::string& path_to_contact, const std::shared_ptr<ParamsServer>& server,
                     const std::shared_ptr<RobotModel>& robot_model, FrameTable& frames, std::size_t nu,
                     ContactModelTypes& contact_type) const;
};

// CostModelSum analogue: name-ordered table (std::map iteration order)
class CostModelSum {
 public:
  void addCost(const std::string& name, const EmpcCost& cost, double weight, bool active = true);
  void removeCost(const std::string& name);
  const std::map<std::string, EmpcCost>& get_costs() const { return costs_; }
  std::map<std::string, EmpcCost>& get_costs() { return costs_; }

 private:
  std::map<std::string, EmpcCost> costs_;
};
class ContactModelMultiple {
 public:
  void addContact(const std::string& name, const EmpcContact& contact);
  const std::map<std::string, EmpcContact>& get_contacts() const { return contacts_; }

 private:
  std::map<std::string, EmpcContact> contacts_;
};
EmpcCostSet makeCostSet(const CostModelSum& costs, const ContactModelMultiple& contacts);

class Stage : public std::enable_shared_from_this<Stage> {
 public:
  static std::shared_ptr<Stage> create(const std::shared_ptr<Trajectory>& trajectory);
  void autoSetup(const std::string& path_to_stages, const std::map<std::string, std::string>& stage,
                 const std::shared_ptr<ParamsServer>& server, std::size_t t_ini);
  void set_t_ini(std::size_t t_ini) { t_ini_ = t_ini; }
  void set_duration(std::size_t duration) { duration_ = duration; }
  // the owning trajectory (empty once it is gone).  The back-reference is weak: the reference keeps a strong
  // boost::shared_ptr here (include/eagle_mpc/stage.hpp), a Trajectory <-> Stage cycle that never frees either.
  std::shared_ptr<Trajectory> get_trajectory() const { return trajectory_.lock(); }
  const std::shared_ptr<CostModelSum>& get_costs() const { return costs_; }
  const std::shared_ptr<ContactModelMultiple>& get_contacts() const { return contacts_; }
  const std::map<std::string, CostModelTypes>& get_cost_types() const { return cost_types_; }
  const std::map<std::string, ContactModelTypes>& get_contact_types() const { return contact_types_; }
  std::size_t get_duration() const { return duration_; }
  std::size_t get_t_ini() const { return t_ini_; }
  const std::string& get_name() const { return name_; }
  bool get_is_terminal() const { return is_terminal_; }
  bool get_is_transition() const { return is_transition_; }

 private:
  explicit Stage(const std::shared_ptr<Trajectory>& trajectory);
  std::weak_ptr<Trajectory> trajectory_;
  std::shared_ptr<CostModelSum> costs_;
  std::shared_ptr<ContactModelMultiple> contacts_;
  std::map<std::string, CostModelTypes> cost_types_;
  std::map<std::string, ContactModelTypes> contact_types_;
  std::string name_;
  std::size_t duration_ = 0, t_ini_ = 0;
  bool is_transition_ = false, is_terminal_ = false;
};

// What crocoddyl::ShootingProblem is to the reference: initial state + one action model per node.  Action models
// are rows of a cost-set table (all knots of a Stage share one row, exactly like the shared iam pointer at
// src/trajectory.cpp:133-136); desc() flattens it for the C ABI.
class ShootingProblem {
 public:
  ShootingProblem(const VectorXd& x0, const EmpcModelDesc& model, const std::vector<EmpcCostSet>& sets,
                  const std::vector<int>& knot_set, const MatrixXd& tau_f, const VectorXd& u_lb, const VectorXd& u_ub,
                  double dt, bool has_contact, bool use_squash, int integrator);
  std::size_t get_T() const { return (std::size_t)(knot_set_.size() - 1); }
  const VectorXd& get_x0() const { return x0_; }
  void set_x0(const VectorXd& x0);
  std::size_t get_nx() const { return (std::size_t)desc_.nx; }
  std::size_t get_ndx() const { return (std::size_t)desc_.ndx; }
  std::size_t get_nu() const { return (std::size_t)desc_.nu; }
  double get_dt() const { return desc_.dt; }
  std::vector<EmpcCostSet>& get_sets() { return sets_; }
  const std::vector<int>& get_knot_set() const { return knot_set_; }
  const EmpcProblemDesc& desc();  // pointers valid until the problem is modified or destroyed

 private:
  VectorXd x0_;
  std::vector<EmpcCostSet> sets_;
  std::vector<int> knot_set_;
  EmpcProblemDesc desc_;
};

struct ProblemParams {
  bool use_squash = false;
  std::size_t dt = 0;
  std::string integrator;
};

class Trajectory : public std::enable_shared_from_this<Trajectory> {
 public:
  static std::shared_ptr<Trajectory> create();
  void autoSetup(const std::string& yaml_path);
  std::shared_ptr<ShootingProblem> createProblem() const;
  std::shared_ptr<ShootingProblem> createProblem(std::size_t dt, bool squash, const std::string& integration_method) const;
  void removeStage(std::size_t idx_stage);
  void set_initial_state(const VectorXd& initial_state);

  const std::vector<std::shared_ptr<Stage>>& get_stages() const { return stages_; }
  const std::shared_ptr<RobotModel>& get_robot_model() const { return robot_model_; }
  const std::string& get_robot_model_path() const { return robot_model_path_; }
  const std::shared_ptr<MultiCopterBaseParams>& get_platform_params() const { return platform_params_; }
  const VectorXd& get_initial_state() const { return initial_state_; }
  const std::shared_ptr<ParamsServer>& get_params_server() const { return params_server_; }
  bool get_has_contact() const { return has_contact_; }
  std::size_t get_duration() const { return duration_; }
  std::size_t get_nx() const { return (std::size_t)(robot_model_->nq() + robot_model_->nv()); }
  std::size_t get_ndx() const { return (std::size_t)(2 * robot_model_->nv()); }
  std::size_t get_nu() const;  // actuation->get_nu() = n_rotors + (nv - 6)
  VectorXd zero_state() const;
  FrameTable& frame_table() { return frames_; }
  const FrameTable& frame_table() const { return frames_; }
  const ProblemParams& get_problem_params() const { return problem_params_; }
  // get_squash() (include/eagle_mpc/trajectory.hpp:68): the platform's SquashingModelSmoothSat
  std::shared_ptr<SquashingModelSmoothSat> get_squash() const {
    return std::make_shared<SquashingModelSmoothSat>(platform_params_->u_lb, platform_params_->u_ub, get_nu());
  }

 private:
  Trajectory();
  std::vector<std::shared_ptr<Stage>> stages_;
  std::shared_ptr<RobotModel> robot_model_;
  std::string robot_model_path_;
  std::shared_ptr<MultiCopterBaseParams> platform_params_;
  VectorXd initial_state_;
  std::shared_ptr<ParamsServer> params_server_;
  ProblemParams problem_params_;
  bool has_contact_ = false;
  std::size_t duration_ = 0;
  FrameTable frames_;
};

// Batched Squash-box FDDP on the GPU.  The scalar interface below is the reference's; the *Batch methods are the
// data-parallel extension (B independent rollouts of the same problem from different initial states).
// What a crocoddyl callback reads from the solver after one iteration (crocoddyl::CallbackAbstract::operator()(SolverAbstract&)).
struct IterationRecord {
  int phase, iter;
  double cost, stop, x_reg, u_reg, steplength, dV, dVexp, gap_norm, d0, d1;
  bool is_feasible;
};
class CallbackAbstract {
 public:
  virtual ~CallbackAbstract() {}
  virtual void operator()(const IterationRecord& rec) = 0;
};
// crocoddyl::CallbackVerbose: one line per iteration
class CallbackVerbose : public CallbackAbstract {
 public:
  void operator()(const IterationRecord& rec) override;
};

class SolverSbFDDP {
 public:
  // the reference's constructor (include/eagle_mpc/sbfddp.hpp:39-40): SolverSbFDDP(problem, trajectory->get_squash())
  SolverSbFDDP(const std::shared_ptr<ShootingProblem>& problem, const std::shared_ptr<SquashingModelSmoothSat>& squashing_model,
               std::size_t batch_size = 1, int device = 0);
  // SolverAbstract::setCallbacks: invoked once per DDP iteration of trajectory 0, replayed from the device iteration trace
  // right after solve() returns (the solve itself never leaves the device)
  void setCallbacks(const std::vector<std::shared_ptr<CallbackAbstract>>& callbacks);
  const std::vector<std::shared_ptr<CallbackAbstract>>& getCallbacks() const { return callbacks_; }
  // squashing model == the problem's (u_lb, u_ub) smooth saturation; batch_size trajectories on HIP device `device`
  // solver_type: EMPC_SOLVER_SBFDDP (the fork's solver, default) or the crocoddyl back ends MpcAbstract also accepts --
  // EMPC_SOLVER_BOXFDDP / EMPC_SOLVER_BOXDDP (include/eagle_mpc/mpc-base.hpp:36-47); same handle type, same getters
  explicit SolverSbFDDP(const std::shared_ptr<ShootingProblem>& problem, std::size_t batch_size = 1, int device = 0,
                        int solver_type = EMPC_SOLVER_SBFDDP);
  ~SolverSbFDDP();
  SolverSbFDDP(const SolverSbFDDP&) = delete;
  SolverSbFDDP& operator=(const SolverSbFDDP&) = delete;

  // reference signature (include/eagle_mpc/sbfddp.hpp:42-46); always returns true (src/sbfddp.cpp:225)
  bool solve(const std::vector<VectorXd>& init_xs = std::vector<VectorXd>(),
             const std::vector<VectorXd>& init_us = std::vector<VectorXd>(), std::size_t maxiter = 100,
             bool is_feasible = false, double regInit = 1e-9);
  // batched: x0s is B x nx; init_xs B x (T+1) x nx and init_us B x T x nu (empty = zero state / zero control)
  bool solveBatch(const std::vector<double>& x0s, const std::vector<double>& init_xs = std::vector<double>(),
                  const std::vector<double>& init_us = std::vector<double>(), std::size_t maxiter = 100,
                  bool is_feasible = false);

  const std::vector<VectorXd>& getSquashControls() const { return us_squash_; }
  const std::vector<VectorXd>& get_xs() const { return xs_; }
  const std::vector<VectorXd>& get_us() const { return us_; }
  std::size_t get_iter() const { return iter_; }
  double get_cost() const { return cost_; }
  double get_stop() const { return stop_; }
  double get_convergence_init() const { return convergence_init_; }
  void set_convergence_init(double convergence_init);
  const std::shared_ptr<ShootingProblem>& get_problem() const { return problem_; }
  std::size_t get_batch_size() const { return batch_; }
  // re-upload the problem's cost tables / x0 after an MPC updateProblem()
  void syncProblem();

  // batched results, row-major B x ...
  const std::vector<double>& get_xs_batch() const { return xs_b_; }
  const std::vector<double>& get_us_batch() const { return us_b_; }
  const std::vector<double>& get_us_squash_batch() const { return us_squash_b_; }
  const std::vector<double>& get_cost_batch() const { return cost_b_; }
  const std::vector<int>& get_iter_batch() const { return iter_b_; }
  const std::vector<int>& get_status_batch() const { return status_b_; }
  EmpcSolver* handle() const { return handle_; }

 private:
  void fetch();
  std::shared_ptr<ShootingProblem> problem_;
  std::size_t batch_;
  EmpcSolver* handle_ = nullptr;
  double convergence_init_ = 1e-2;
  std::vector<VectorXd> xs_, us_, us_squash_;
  std::size_t iter_ = 0;
  double cost_ = 0, stop_ = 0;
  std::vector<double> xs_b_, us_b_, us_squash_b_, cost_b_;
  std::vector<int> iter_b_, status_b_;
  std::vector<std::shared_ptr<CallbackAbstract>> callbacks_;
  std::shared_ptr<SquashingModelSmoothSat> squashing_model_;
};

// ------------------------------------------------------------------------------------------------------------
// MPC controllers (include/eagle_mpc/mpc-base.hpp:34-119, include/eagle_mpc/mpc-controllers/{carrot,rail,weighted}-mpc.hpp)
// ------------------------------------------------------------------------------------------------------------
enum class SolverTypes { SolverSbFDDP, SolverBoxFDDP, SolverBoxDDP, NbSolverTypes };  // mpc-base.hpp:34-39

struct MpcParams {  // mpc-base.hpp:49-57
  IntegratedActionModelTypes integrator_type = IntegratedActionModelTypes::IntegratedActionModelEuler;
  std::size_t knots = 0, iters = 0, dt = 0;
  SolverTypes solver_type = SolverTypes::SolverSbFDDP;
  bool callback = false;
};

// MpcAbstract: reads the `mpc_controller:` YAML, builds robot / platform objects, owns problem and solver.  The
// reference's per-knot dif/int action models are the rows of the problem's cost-set table here (one private
// EmpcCostSet per knot); get_problem()->get_sets()[i] is what dif_models_[i]->get_costs() is to the reference.
class MpcAbstract {
 public:
  explicit MpcAbstract(const std::string& yaml_path);
  virtual ~MpcAbstract() {}
  virtual void createProblem() = 0;
  virtual void updateProblem(const std::size_t& current_time) = 0;

  const std::shared_ptr<RobotModel>& get_robot_model() const { return robot_model_; }
  const std::string& get_robot_model_path() const { return robot_model_path_; }
  const std::shared_ptr<MultiCopterBaseParams>& get_platform_params() const { return platform_params_; }
  const std::shared_ptr<ParamsServer>& get_params_server() const { return params_server_; }
  const std::shared_ptr<ShootingProblem>& get_problem() const { return problem_; }
  // created lazily on the first call (needs a HIP device); batch_size rollouts share the problem
  const std::shared_ptr<SolverSbFDDP>& get_solver(std::size_t batch_size = 1, int device = 0);
  const std::size_t& get_dt() const { return params_.dt; }
  const std::size_t& get_knots() const { return params_.knots; }
  const std::size_t& get_iters() const { return params_.iters; }
  const SolverTypes& get_solver_type() const { return params_.solver_type; }
  std::size_t get_nx() const { return (std::size_t)(robot_model_->nq() + robot_model_->nv()); }
  std::size_t get_ndx() const { return (std::size_t)(2 * robot_model_->nv()); }
  std::size_t get_nu() const { return platform_params_->n_rotors_ + (std::size_t)(robot_model_->nv() - 6); }
  VectorXd zero_state() const;

 protected:
  void initializeRobotObjects();  // src/mpc-base.cpp:19-38
  void loadParams();              // src/mpc-base.cpp:40-60
  std::shared_ptr<ParamsServer> params_server_;
  std::shared_ptr<RobotModel> robot_model_;
  std::string robot_model_path_;
  std::shared_ptr<MultiCopterBaseParams> platform_params_;
  MpcParams params_;
  std::shared_ptr<ShootingProblem> problem_;
  std::shared_ptr<SolverSbFDDP> solver_;
};

class CarrotMpc : public MpcAbstract {
 public:
  // reference ctor: src/mpc-controllers/carrot-mpc.cpp:15-50
  CarrotMpc(const std::shared_ptr<Trajectory>& trajectory, const std::vector<VectorXd>& state_ref, std::size_t dt_ref,
            const std::string& yaml_path);
  void createProblem() override;                                 // :178-248
  void updateProblem(const std::size_t& current_time) override;  // :298-313

  const std::shared_ptr<Trajectory>& get_trajectory() const { return trajectory_; }
  const std::vector<std::size_t>& get_t_stages() const { return t_stages_; }
  const std::vector<std::size_t>& get_t_ref() const { return t_ref_; }
  const std::vector<VectorXd>& get_state_ref() const { return state_ref_; }
  const VectorXd& computeStateReference(const std::size_t& time);  // :384-403

 private:
  void loadCostParams();                                                      // :53-176
  EmpcCostSet createCosts() const;                                            // :250-296
  void computeActiveStage(const std::size_t& current_time);                   // :315-319
  void updateContactCosts(const std::size_t& idx);                            // :329 (empty in the reference)
  void updateFreeCosts(const std::size_t& idx, const std::size_t& current_time);  // :331-362

  std::shared_ptr<Trajectory> trajectory_;
  std::vector<VectorXd> state_ref_;
  std::vector<std::size_t> t_ref_, t_stages_;
  double carrot_weight_ = 10, carrot_tail_weight_ = 5, control_reg_weight_ = 1e-2, state_reg_weight_ = 1e-3,
         state_limits_weight_ = 100;
  VectorXd carrot_tail_act_weights_, control_reg_act_weights_, state_ref_act_weights_, state_limits_act_weights_,
      state_limits_l_bound_, state_limits_u_bound_;
  struct UpdateVars {
    std::size_t idx_stage = 0, idx_last_stage = 0, node_time = 0, idx_state = 0;
    double alpha = 0;
    VectorXd state_ref;
  } update_vars_;
};

// RailMpc (include/eagle_mpc/mpc-controllers/rail-mpc.hpp:24-60): every knot tracks the planned state at its own
// time ("rail_state", weighted quadratic) plus a plain control regularisation ("control").
class RailMpc : public MpcAbstract {
 public:
  // reference ctor: src/mpc-controllers/rail-mpc.cpp:14-60
  RailMpc(const std::vector<VectorXd>& state_ref, std::size_t dt_ref, const std::string& yaml_path);
  void createProblem() override;                                 // :64-126
  void updateProblem(const std::size_t& current_time) override;  // :151-161

  const std::vector<VectorXd>& get_state_ref() const { return state_ref_; }
  const std::vector<std::size_t>& get_t_ref() const { return t_ref_; }
  const VectorXd& computeStateReference(const std::size_t& time);  // :176-200

 private:
  EmpcCostSet createCosts() const;                // :128-149
  void updateContactCosts(const std::size_t&) {}  // :163 (empty in the reference)
  void updateFreeCosts(const std::size_t& idx);   // :165-174

  std::vector<VectorXd> state_ref_;
  std::vector<std::size_t> t_ref_;
  VectorXd state_activation_weights_;
  double state_weight_ = 10, control_weight_ = 1e-1;
  struct UpdateVars {
    std::size_t node_time = 0, idx_state = 0;
    double alpha = 0;
    VectorXd state_ref;
  } update_vars_;
};

// WeightedMpc (include/eagle_mpc/mpc-controllers/weighted-mpc.hpp:24-70): every knot carries the task costs of all
// (non-transition) stages of the trajectory; the costs of the stage active at the knot's time are switched on, the
// task ones weighted by beta * exp(alpha * (time - end of the stage)).
class WeightedMpc : public MpcAbstract {
 public:
  // reference ctor: src/mpc-controllers/weighted-mpc.cpp:16-72.  NOTE: like the reference, the ctor edits the trajectory
  // it is given: every transition stage is merged into the stage that follows it.
  WeightedMpc(const std::shared_ptr<Trajectory>& trajectory, std::size_t dt_ref, const std::string& yaml_path);
  void createProblem() override;                                 // :76-143
  void updateProblem(const std::size_t& current_time) override;  // :170-185

  const std::shared_ptr<Trajectory>& get_trajectory() const { return trajectory_; }
  const std::vector<std::size_t>& get_t_stages() const { return t_stages_; }
  double get_alpha() const { return alpha_; }
  double get_beta() const { return beta_; }

 private:
  CostModelSum createCosts() const;                                                          // :145-168
  void computeActiveStage(const std::size_t& current_time);                                  // :187-191
  void computeActiveStage(const std::size_t& current_time, const std::size_t& last_stage);   // :193-199
  void computeWeight(const std::size_t& time);                                               // :230-243
  void updateContactCosts(const std::size_t&) {}                                             // :201 (empty)
  void updateFreeCosts(const std::size_t& idx);                                              // :203-228

  std::shared_ptr<Trajectory> trajectory_;
  std::vector<std::size_t> t_stages_;
  std::vector<std::string> cost_names_;  // full names "<stage>/<cost>" in the order of every knot's cost set
  double alpha_ = 20, beta_ = 1, state_reg_ = 1e-1, control_reg_ = 1e-1;
  struct UpdateVars {
    std::size_t idx_stage = 0, idx_last_stage = 0, node_time = 0;
    std::string name_stage;
    double weight = 1, weight_time = 0;
  } update_vars_;
};

}  // namespace eagle_mpc
