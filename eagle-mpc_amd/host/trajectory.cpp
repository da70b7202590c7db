// Host-side problem factory: MultiCopterBaseParams, cost/activation/contact factories, Stage, Trajectory and
// ShootingProblem.  Restates (not copies) src/multicopter-base-params.cpp, src/factory/*.cpp, src/stage.cpp and
// src/trajectory.cpp of the reference; each function cites the lines it follows.
#include <cmath>
#include <cstring>
#include <iostream>
#include <limits>

#include "eagle_mpc.hpp"

namespace eagle_mpc {

void quaternionToRotation(const VectorXd& q, double* R) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  const double n = std::sqrt(x * x + y * y + z * z + w * w);
  x /= n;
  y /= n;
  z /= n;
  w /= n;
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

// -----------------------------------------------------------------------------------------------------
// MultiCopterBaseParams  (src/multicopter-base-params.cpp)
// -----------------------------------------------------------------------------------------------------
MultiCopterBaseParams::MultiCopterBaseParams(double cf, double cm, const MatrixXd& tau_f, double max_th, double min_th,
                                             const std::string& base_link)
    : cf_(cf), cm_(cm), n_rotors_(tau_f.cols), tau_f_(tau_f), max_thrust_(max_th), min_thrust_(min_th), base_link_name_(base_link) {}

void MultiCopterBaseParams::autoSetup(const std::string& path, const std::shared_ptr<ParamsServer>& server) {  // :27-79
  try {
    cf_ = server->getParam<double>(path + "cf");
    cm_ = server->getParam<double>(path + "cm");
    max_thrust_ = server->getParam<double>(path + "max_thrust");
    min_thrust_ = server->getParam<double>(path + "min_thrust");
    max_prop_speed_ = std::sqrt(max_thrust_ / cf_);
    min_prop_speed_ = std::sqrt(min_thrust_ / cf_);
    base_link_name_ = server->getParam<std::string>(path + "base_link_name");
    n_rotors_ = (std::size_t)server->getParam<int>(path + "n_rotors");
    std::vector<std::string> rotors = server->getParam<std::vector<std::string>>(path + "rotors");
    if (n_rotors_ != rotors.size())
      throw std::runtime_error("'n_rotors' field and the number of rotor poses specified must be the same.");
    for (std::size_t i = 0; i < n_rotors_; ++i) {
      std::map<std::string, VectorXd> rotor = converter<std::map<std::string, VectorXd>>::convert(rotors[i]);
      SE3 pose;
      const VectorXd& tr = rotor["translation"];
      for (int k = 0; k < 3; ++k) pose.p[k] = tr.at(k);
      quaternionToRotation(rotor["orientation"], pose.R);
      rotors_pose_.push_back(pose);
      rotors_spin_dir_.push_back(int(rotor["spin_direction"].at(0)));
    }
  } catch (const std::exception& e) {
    std::cerr << e.what() << '\n';  // the reference swallows the error here too (:63-65)
  }
  // tau_f: thrust direction rows, then p x thrust + spin * cm/cf * thrust (:67-78)
  tau_f_ = MatrixXd(6, (int)n_rotors_);
  for (std::size_t i = 0; i < rotors_pose_.size(); ++i) {
    const SE3& M = rotors_pose_[i];
    const double th[3] = {M.R[2], M.R[5], M.R[8]};  // R e3
    const double k = rotors_spin_dir_[i] * cm_ / cf_;
    const double cr[3] = {M.p[1] * th[2] - M.p[2] * th[1], M.p[2] * th[0] - M.p[0] * th[2], M.p[0] * th[1] - M.p[1] * th[0]};
    for (int r = 0; r < 3; ++r) {
      tau_f_(r, (int)i) = th[r];
      tau_f_(3 + r, (int)i) = cr[r] + k * th[r];
    }
  }
}
void MultiCopterBaseParams::autoSetup(const std::string& path, const std::shared_ptr<ParamsServer>& server,
                                      const std::shared_ptr<RobotModel>& robot_model) {  // :81-87
  autoSetup(path, server);
  setControlLimits(robot_model);
}
void MultiCopterBaseParams::setControlLimits(const std::shared_ptr<RobotModel>& robot_model) {  // :89-101
  const std::size_t n_arm = (std::size_t)robot_model->nq() - 7;
  u_lb.assign(n_arm + n_rotors_, 0.0);
  u_ub = u_lb;
  for (std::size_t i = 0; i < n_rotors_; ++i) {
    u_lb[i] = min_thrust_;
    u_ub[i] = max_thrust_;
  }
  const std::vector<double>& eff = robot_model->effortLimit();
  for (std::size_t i = 0; i < n_arm; ++i) {
    const double e = eff[eff.size() - n_arm + i];
    u_lb[n_rotors_ + i] = -e;
    u_ub[n_rotors_ + i] = e;
  }
}

// -----------------------------------------------------------------------------------------------------
// factories
// -----------------------------------------------------------------------------------------------------
int FrameTable::use(int model_frame_id) {
  for (std::size_t i = 0; i < ids_.size(); ++i)
    if (ids_[i] == model_frame_id) return (int)i;
  if (ids_.size() >= EMPC_MAX_FRAMES) throw std::runtime_error("too many distinct operational frames in one problem");
  ids_.push_back(model_frame_id);
  return (int)ids_.size() - 1;
}

static const std::map<std::string, ActivationModelTypes>& activationTypes() {  // include/eagle_mpc/factory/activation.hpp:40-53
  static const std::map<std::string, ActivationModelTypes> m = {
      {"ActivationModelQuad", ActivationModelTypes::ActivationModelQuad},
      {"ActivationModelQuadFlatExp", ActivationModelTypes::ActivationModelQuadFlatExp},
      {"ActivationModelQuadFlatLog", ActivationModelTypes::ActivationModelQuadFlatLog},
      {"ActivationModelSmooth1Norm", ActivationModelTypes::ActivationModelSmooth1Norm},
      {"ActivationModelSmooth2Norm", ActivationModelTypes::ActivationModelSmooth2Norm},
      {"ActivationModelWeightedQuad", ActivationModelTypes::ActivationModelWeightedQuad},
      {"ActivationModelQuadraticBarrier", ActivationModelTypes::ActivationModelQuadraticBarrier},
      {"ActivationModelWeightedQuadraticBarrier", ActivationModelTypes::ActivationModelWeightedQuadraticBarrier}};
  return m;
}
static const std::map<std::string, CostModelTypes>& costTypes() {  // include/eagle_mpc/factory/cost.hpp:49-63
  static const std::map<std::string, CostModelTypes> m = {
      {"CostModelState", CostModelTypes::CostModelState},
      {"CostModelControl", CostModelTypes::CostModelControl},
      {"CostModelFramePlacement", CostModelTypes::CostModelFramePlacement},
      {"CostModelFrameRotation", CostModelTypes::CostModelFrameRotation},
      {"CostModelFrameVelocity", CostModelTypes::CostModelFrameVelocity},
      {"CostModelFrameTranslation", CostModelTypes::CostModelFrameTranslation},
      {"CostModelContactFrictionCone", CostModelTypes::CostModelContactFrictionCone}};
  return m;
}

static void check_size(const VectorXd& v, std::size_t n, const std::string& what, const std::string& path) {
  if (v.size() != n)
    throw std::runtime_error(what + " vector @" + path + " has dimension " + std::to_string(v.size()) + ". Should be " +
                             std::to_string(n));
}

void ActivationModelFactory::create(const std::string& path_to_cost, const std::shared_ptr<ParamsServer>& server,
                                    std::size_t nr, EmpcCost& cost) const {  // src/factory/activation.cpp:17-102
  std::string name;
  try {
    name = server->getParam<std::string>(path_to_cost + "activation");
  } catch (const std::exception&) {
    name = "ActivationModelQuad";
  }
  const double inf = std::numeric_limits<double>::infinity();
  for (int i = 0; i < EMPC_MAX_NR; ++i) {
    cost.act_w[i] = 1.0;
    cost.lb[i] = -inf;
    cost.ub[i] = inf;
  }
  cost.nr = (int)nr;
  auto weights_or_ones = [&]() {
    VectorXd w;
    try {
      w = converter<VectorXd>::convert(server->getParam<std::string>(path_to_cost + "weights"));
    } catch (const std::exception&) {
      w.assign(nr, 1.0);
    }
    check_size(w, nr, "Weights", path_to_cost + "weights");
    for (std::size_t i = 0; i < nr; ++i) cost.act_w[i] = w[i];
  };
  auto bounds = [&]() {
    VectorXd lb = converter<VectorXd>::convert(server->getParam<std::string>(path_to_cost + "l_bound"));
    VectorXd ub = converter<VectorXd>::convert(server->getParam<std::string>(path_to_cost + "u_bound"));
    check_size(lb, nr, "l_bound", path_to_cost + "l_bound");
    check_size(ub, nr, "u_bound", path_to_cost + "u_bound");
    for (std::size_t i = 0; i < nr; ++i) {
      cost.lb[i] = lb[i];
      cost.ub[i] = ub[i];
    }
  };
  auto it = activationTypes().find(name);
  if (it == activationTypes().end()) throw std::out_of_range("map::at");  // ActivationModelTypes_map.at(name)
  switch (it->second) {
    case ActivationModelTypes::ActivationModelQuad:
      cost.activation = EMPC_ACT_QUAD;
      break;
    case ActivationModelTypes::ActivationModelWeightedQuad:
      weights_or_ones();
      cost.activation = EMPC_ACT_WEIGHTED_QUAD;
      break;
    case ActivationModelTypes::ActivationModelQuadraticBarrier:
      bounds();
      cost.activation = EMPC_ACT_QUADRATIC_BARRIER;
      break;
    case ActivationModelTypes::ActivationModelWeightedQuadraticBarrier:
      bounds();
      weights_or_ones();
      cost.activation = EMPC_ACT_WEIGHTED_QUADRATIC_BARRIER;
      break;
    default:
      throw std::runtime_error("Activation '" + name + "' @" + path_to_cost + "activation not found");
  }
}

static int frame_or_throw(const std::shared_ptr<RobotModel>& model, FrameTable& frames, const std::string& link_name) {
  const std::size_t id = model->getFrameId(link_name);
  if (id == model->frames().size()) throw std::runtime_error("Link " + link_name + "does no exists");
  return frames.use((int)id);
}

EmpcCost CostModelFactory::create(const std::string& path_to_cost, const std::shared_ptr<ParamsServer>& server,
                                  const std::shared_ptr<RobotModel>& model, FrameTable& frames, std::size_t nu,
                                  CostModelTypes& cost_type) const {  // src/factory/cost.cpp:17-171
  EmpcCost cost;
  std::memset(&cost, 0, sizeof(cost));
  cost.frame = -1;
  cost.active = 1;
  const std::size_t nx = (std::size_t)(model->nq() + model->nv());
  const std::size_t ndx = (std::size_t)(2 * model->nv());
  {
    const std::string type = server->getParam<std::string>(path_to_cost + "type");
    auto it = costTypes().find(type);
    if (it == costTypes().end())
      throw std::runtime_error("Cost " + type + " not found. Please make sure the specified cost exists.");
    cost_type = it->second;
  }
  auto vec = [&](const std::string& key) { return converter<VectorXd>::convert(server->getParam<std::string>(path_to_cost + key)); };
  switch (cost_type) {
    case CostModelTypes::CostModelState: {
      activation_factory_.create(path_to_cost, server, ndx, cost);
      VectorXd reference;
      try {
        reference = vec("reference");
      } catch (const std::exception&) {
        reference.assign(nx, 0.0);
        reference[6] = 1.0;  // state->zero()
      }
      if (reference.size() != nx)
        throw std::runtime_error("State reference vector @" + path_to_cost + "reference has dimension " +
                                 std::to_string(reference.size()) + ". Should be " + std::to_string(nx));
      for (std::size_t i = 0; i < nx; ++i) cost.ref[i] = reference[i];
      cost.type = EMPC_COST_STATE;
    } break;
    case CostModelTypes::CostModelControl: {
      activation_factory_.create(path_to_cost, server, nu, cost);
      VectorXd reference;
      try {
        reference = vec("reference");
      } catch (const std::exception&) {
        reference.assign(nu, 0.0);
      }
      if (reference.size() != nu)
        throw std::runtime_error("Control reference vector @" + path_to_cost + "reference has dimension " +
                                 std::to_string(reference.size()) + ". Should be " + std::to_string(nu));
      for (std::size_t i = 0; i < nu; ++i) cost.ref[i] = reference[i];
      cost.type = EMPC_COST_CONTROL;
    } break;
    case CostModelTypes::CostModelFramePlacement: {
      activation_factory_.create(path_to_cost, server, 6, cost);
      VectorXd position = vec("position");
      VectorXd orientation = vec("orientation");
      cost.frame = frame_or_throw(model, frames, server->getParam<std::string>(path_to_cost + "link_name"));
      for (int i = 0; i < 3; ++i) cost.ref[i] = position.at(i);
      if (orientation.size() != 4) throw std::runtime_error("orientation @" + path_to_cost + " must have 4 entries");
      quaternionToRotation(orientation, cost.ref + 3);
      cost.type = EMPC_COST_FRAME_PLACEMENT;
    } break;
    case CostModelTypes::CostModelFrameRotation: {
      activation_factory_.create(path_to_cost, server, 3, cost);
      VectorXd orientation = vec("orientation");
      cost.frame = frame_or_throw(model, frames, server->getParam<std::string>(path_to_cost + "link_name"));
      if (orientation.size() != 4) throw std::runtime_error("orientation @" + path_to_cost + " must have 4 entries");
      quaternionToRotation(orientation, cost.ref);
      cost.type = EMPC_COST_FRAME_ROTATION;
    } break;
    case CostModelTypes::CostModelFrameVelocity: {
      activation_factory_.create(path_to_cost, server, 6, cost);
      VectorXd linear = vec("linear");
      VectorXd angular = vec("angular");
      cost.frame = frame_or_throw(model, frames, server->getParam<std::string>(path_to_cost + "link_name"));
      for (int i = 0; i < 3; ++i) {
        cost.ref[i] = linear.at(i);
        cost.ref[3 + i] = angular.at(i);
      }
      cost.type = EMPC_COST_FRAME_VELOCITY;
    } break;
    case CostModelTypes::CostModelFrameTranslation: {
      activation_factory_.create(path_to_cost, server, 3, cost);
      VectorXd position = vec("position");
      cost.frame = frame_or_throw(model, frames, server->getParam<std::string>(path_to_cost + "link_name"));
      for (int i = 0; i < 3; ++i) cost.ref[i] = position.at(i);
      cost.type = EMPC_COST_FRAME_TRANSLATION;
    } break;
    case CostModelTypes::CostModelContactFrictionCone: {
      VectorXd n_surf = vec("n_surf");
      const double mu = server->getParam<double>(path_to_cost + "mu");
      cost.frame = frame_or_throw(model, frames, server->getParam<std::string>(path_to_cost + "link_name"));
      // FrictionCone(n_surf, mu, 4, false) with ActivationModelQuadraticBarrier on its bounds (:154-163):
      // four facet rows bounded above by 0, the normal-force row bounded below by 0
      const double inf = std::numeric_limits<double>::infinity();
      for (int i = 0; i < EMPC_MAX_NR; ++i) {
        cost.act_w[i] = 1.0;
        cost.lb[i] = -inf;
        cost.ub[i] = inf;
      }
      for (int i = 0; i < 4; ++i) cost.ub[i] = 0.0;
      cost.lb[4] = 0.0;
      cost.nr = 5;
      cost.activation = EMPC_ACT_QUADRATIC_BARRIER;
      for (int i = 0; i < 3; ++i) cost.ref[i] = n_surf.at(i);
      cost.ref[3] = mu;
      cost.type = EMPC_COST_CONTACT_FRICTION_CONE;
    } break;
    default:
      break;
  }
  return cost;
}

EmpcContact ContactModelFactory::create(const std::string& path, const std::shared_ptr<ParamsServer>& server,
                                        const std::shared_ptr<RobotModel>& model, FrameTable& frames, std::size_t /*nu*/,
                                        ContactModelTypes& contact_type) const {  // src/factory/contacts.cpp:17-82
  EmpcContact c;
  std::memset(&c, 0, sizeof(c));
  c.ref_R[0] = c.ref_R[4] = c.ref_R[8] = 1.0;
  const std::string type = server->getParam<std::string>(path + "type");
  if (type == "ContactModel2D")
    contact_type = ContactModelTypes::ContactModel2D;
  else if (type == "ContactModel3D")
    contact_type = ContactModelTypes::ContactModel3D;
  else if (type == "ContactModel6D")
    contact_type = ContactModelTypes::ContactModel6D;
  else
    throw std::runtime_error("Contact " + type + "not found. Please make sure the specified contact exists.");
  auto vec = [&](const std::string& key) { return converter<VectorXd>::convert(server->getParam<std::string>(path + key)); };
  auto gains = [&]() {
    VectorXd g;
    try {
      g = vec("gains");
    } catch (const std::exception&) {
      g.assign(2, 0.0);
    }
    c.gains[0] = g.at(0);
    c.gains[1] = g.at(1);
  };
  switch (contact_type) {
    case ContactModelTypes::ContactModel3D: {
      VectorXd position = vec("position");
      c.frame = frame_or_throw(model, frames, server->getParam<std::string>(path + "link_name"));
      for (int i = 0; i < 3; ++i) c.ref_p[i] = position.at(i);
      gains();
      c.type = EMPC_CONTACT_3D;
    } break;
    case ContactModelTypes::ContactModel6D: {
      VectorXd position = vec("position");
      VectorXd orientation = vec("orientation");
      c.frame = frame_or_throw(model, frames, server->getParam<std::string>(path + "link_name"));
      for (int i = 0; i < 3; ++i) c.ref_p[i] = position.at(i);
      quaternionToRotation(orientation, c.ref_R);
      gains();
      c.type = EMPC_CONTACT_6D;
    } break;
    default:
      // the reference's switch has no ContactModel2D case and returns a null contact (:45-80)
      throw std::runtime_error("Contact " + type + " is enumerated but not implemented by the factory.");
  }
  return c;
}

void CostModelSum::addCost(const std::string& name, const EmpcCost& cost, double weight, bool active) {
  EmpcCost c = cost;
  std::memset(c.name, 0, sizeof(c.name));
  std::strncpy(c.name, name.c_str(), EMPC_NAME_LEN - 1);
  c.weight = weight;
  c.active = active ? 1 : 0;
  costs_.insert({name, c});  // like crocoddyl: an existing name is left untouched
}
void CostModelSum::removeCost(const std::string& name) { costs_.erase(name); }
void ContactModelMultiple::addContact(const std::string& name, const EmpcContact& contact) {
  EmpcContact c = contact;
  std::memset(c.name, 0, sizeof(c.name));
  std::strncpy(c.name, name.c_str(), EMPC_NAME_LEN - 1);
  contacts_.insert({name, c});
}
EmpcCostSet makeCostSet(const CostModelSum& costs, const ContactModelMultiple& contacts) {
  EmpcCostSet s;
  std::memset(&s, 0, sizeof(s));
  if (costs.get_costs().size() > EMPC_MAX_COSTS - 1)  // one slot is reserved for the solver's barrier cost
    throw std::runtime_error("too many costs in one stage (max " + std::to_string(EMPC_MAX_COSTS - 1) + ")");
  if (contacts.get_contacts().size() > EMPC_MAX_CONTACTS) throw std::runtime_error("too many contacts in one stage");
  for (const auto& kv : costs.get_costs()) s.costs[s.ncosts++] = kv.second;
  for (const auto& kv : contacts.get_contacts()) s.contacts[s.ncontacts++] = kv.second;
  return s;
}

// -----------------------------------------------------------------------------------------------------
// Stage  (src/stage.cpp)
// -----------------------------------------------------------------------------------------------------
Stage::Stage(const std::shared_ptr<Trajectory>& trajectory) : trajectory_(trajectory) {
  costs_ = std::make_shared<CostModelSum>();
  contacts_ = std::make_shared<ContactModelMultiple>();
}
std::shared_ptr<Stage> Stage::create(const std::shared_ptr<Trajectory>& trajectory) {
  return std::shared_ptr<Stage>(new Stage(trajectory));
}
void Stage::autoSetup(const std::string& path_to_stages, const std::map<std::string, std::string>& stage,
                      const std::shared_ptr<ParamsServer>& server, std::size_t t_ini) {  // :26-71
  const std::string path_to_stage = path_to_stages + stage.at("name") + "/";
  name_ = stage.at("name");
  duration_ = std::size_t(converter<int>::convert(stage.at("duration")));
  t_ini_ = t_ini;
  is_transition_ = converter<bool>::convert(stage.at("transition"));
  ContactModelFactory contact_factory;
  CostModelFactory cost_factory;
  const std::shared_ptr<Trajectory> trajectory = trajectory_.lock();
  if (!trajectory) throw std::runtime_error("Stage: the owning trajectory no longer exists");
  const std::size_t nu = trajectory->get_nu();
  try {
    std::vector<std::string> contact_names = converter<std::vector<std::string>>::convert(stage.at("contacts"));
    for (const auto& contact_name : contact_names) {
      ContactModelTypes contact_type;
      EmpcContact contact = contact_factory.create(path_to_stage + "contacts/" + contact_name + "/", server,
                                                   trajectory->get_robot_model(), trajectory->frame_table(), nu, contact_type);
      contacts_->addContact(contact_name, contact);
      contact_types_.insert({contact_name, contact_type});
    }
  } catch (const std::exception&) {
    // stage without contacts (the reference swallows every exception raised in this block, :38-50)
  }
  std::vector<std::string> cost_names = converter<std::vector<std::string>>::convert(stage.at("costs"));
  for (const auto& cost_name : cost_names) {
    const double weight = server->getParam<double>(path_to_stage + "costs/" + cost_name + "/weight");
    // quirk kept from :55-61: a cost is INACTIVE only when an 'active' key exists and parses as a number
    bool active = false;
    try {
      server->getParam<double>(path_to_stage + "costs/" + cost_name + "/active");
    } catch (const std::exception&) {
      active = true;
    }
    CostModelTypes cost_type;
    EmpcCost cost = cost_factory.create(path_to_stage + "costs/" + cost_name + "/", server, trajectory->get_robot_model(),
                                        trajectory->frame_table(), nu, cost_type);
    costs_->addCost(cost_name, cost, weight, active);
    cost_types_.insert({cost_name, cost_type});
  }
}

// -----------------------------------------------------------------------------------------------------
// ShootingProblem
// -----------------------------------------------------------------------------------------------------
ShootingProblem::ShootingProblem(const VectorXd& x0, const EmpcModelDesc& model, const std::vector<EmpcCostSet>& sets,
                                 const std::vector<int>& knot_set, const MatrixXd& tau_f, const VectorXd& u_lb,
                                 const VectorXd& u_ub, double dt, bool has_contact, bool use_squash, int integrator)
    : x0_(x0), sets_(sets), knot_set_(knot_set) {
  std::memset(&desc_, 0, sizeof(desc_));
  desc_.model = model;
  desc_.nx = model.nq + model.nv;
  desc_.ndx = 2 * model.nv;
  desc_.n_rotors = tau_f.cols;
  desc_.nu = tau_f.cols + model.nv - 6;
  if (desc_.n_rotors > EMPC_MAX_ROTORS || desc_.nu > EMPC_MAX_NU) throw std::runtime_error("too many rotors / controls");
  if ((int)x0.size() != desc_.nx) throw std::runtime_error("x0 has wrong dimension");
  desc_.T = (int)knot_set.size() - 1;
  desc_.n_sets = (int)sets.size();
  desc_.has_contact = has_contact ? 1 : 0;
  desc_.use_squash = use_squash ? 1 : 0;
  desc_.integrator = integrator;
  desc_.dt = dt;
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < tau_f.cols; ++c) desc_.tau_f[r * tau_f.cols + c] = tau_f(r, c);
  for (int i = 0; i < desc_.nu; ++i) {
    desc_.u_lb[i] = u_lb.at(i);
    desc_.u_ub[i] = u_ub.at(i);
  }
}
void ShootingProblem::set_x0(const VectorXd& x0) {
  if (x0.size() != x0_.size()) throw std::invalid_argument("x0 has wrong dimension");
  x0_ = x0;
}
const EmpcProblemDesc& ShootingProblem::desc() {
  for (int i = 0; i < desc_.nx; ++i) desc_.x0[i] = x0_[i];
  desc_.n_sets = (int)sets_.size();
  desc_.sets = sets_.data();
  desc_.knot_set = knot_set_.data();
  return desc_;
}

// -----------------------------------------------------------------------------------------------------
// Trajectory  (src/trajectory.cpp)
// -----------------------------------------------------------------------------------------------------
Trajectory::Trajectory() {}
std::shared_ptr<Trajectory> Trajectory::create() { return std::shared_ptr<Trajectory>(new Trajectory()); }
std::size_t Trajectory::get_nu() const { return platform_params_->n_rotors_ + (std::size_t)(robot_model_->nv() - 6); }
VectorXd Trajectory::zero_state() const {
  VectorXd x(get_nx(), 0.0);
  x[6] = 1.0;
  return x;
}

void Trajectory::autoSetup(const std::string& yaml_path) {  // :21-89
  ParserYaml parser(yaml_path);
  params_server_ = std::make_shared<ParamsServer>(parser.get_params());
  robot_model_path_ = getUrdfPath(params_server_->getParam<std::string>("robot/urdf"));
  robot_model_ = std::make_shared<RobotModel>(RobotModel::fromUrdf(robot_model_path_));
  platform_params_ = std::make_shared<MultiCopterBaseParams>();
  platform_params_->autoSetup("robot/platform/", params_server_, robot_model_);
  try {
    problem_params_.use_squash = params_server_->getParam<bool>("problem_params/use_squash");
    problem_params_.dt = (std::size_t)params_server_->getParam<int>("problem_params/dt");
    problem_params_.integrator = params_server_->getParam<std::string>("problem_params/integrator");
  } catch (const std::exception&) {
    problem_params_.use_squash = false;
    problem_params_.dt = 0;
    problem_params_.integrator = "";
  }
  try {
    initial_state_ = params_server_->getParam<VectorXd>("initial_state");
  } catch (const std::exception&) {
    initial_state_ = zero_state();
  }
  if (initial_state_.size() != get_nx())
    throw std::runtime_error("The specified initial state has wrong dimension. Should be " + std::to_string(get_nx()) +
                             " and it has " + std::to_string(initial_state_.size()));
  auto stages_params = params_server_->getParam<std::vector<std::map<std::string, std::string>>>("stages");
  std::size_t time = 0;
  bool stage_duration_0 = false;
  for (const auto& stage_param : stages_params) {
    std::shared_ptr<Stage> stage = Stage::create(shared_from_this());
    stage->autoSetup("stages/", stage_param, params_server_, time);
    if (!stage_duration_0 && stage->get_duration() == 0) {
      stage_duration_0 = true;
    } else if (stage_duration_0 && stage->get_duration() == 0) {
      throw std::runtime_error("Two consecutives stages cannot have duration 0. Please, unify them in a single stage.");
    } else {
      stage_duration_0 = false;
    }
    time += stage->get_duration();
    stages_.push_back(stage);
    if (!has_contact_) has_contact_ = stage->get_contacts()->get_contacts().size() != 0;
  }
  duration_ = time;
}

std::shared_ptr<ShootingProblem> Trajectory::createProblem() const {  // :91-100
  if (problem_params_.integrator == "")
    throw std::runtime_error(
        "Problem parameters not specified in the YAML file. Try calling createProblem() by passing the problem parameters.");
  return createProblem(problem_params_.dt, problem_params_.use_squash, problem_params_.integrator);
}

std::shared_ptr<ShootingProblem> Trajectory::createProblem(std::size_t dt, bool squash,
                                                           const std::string& integration_method) const {  // :102-143
  int integrator;
  if (integration_method == "IntegratedActionModelEuler")
    integrator = EMPC_INTEGRATOR_EULER;
  else if (integration_method == "IntegratedActionModelRK4")
    integrator = EMPC_INTEGRATOR_RK4;
  else
    throw std::out_of_range("map::at");  // IntegratedActionModelTypes_map.at(integration_method) (src/factory/int-action.cpp:24)
  if (dt == 0) throw std::runtime_error("dt must be positive");
  std::vector<EmpcCostSet> sets;
  std::vector<int> knot_set;
  constexpr std::size_t kMaxKnots = 1u << 20;  // far above any horizon the device buffers are sized for
  bool last_duration0 = false;
  int terminal_set = -1;
  for (std::size_t si = 0; si < stages_.size(); ++si) {
    const auto& stage = stages_[si];
    sets.push_back(makeCostSet(*stage->get_costs(), *stage->get_contacts()));
    std::size_t n_knots;
    if (stage->get_duration() / dt == 0 && si + 1 != stages_.size()) {
      n_knots = 1;
      last_duration0 = true;
    } else {
      n_knots = stage->get_duration() / dt;
      if (last_duration0) {
        // The reference subtracts in size_t: with n_knots == 0 (a stage shorter than dt after a zero-knot stage) that
        // wraps to SIZE_MAX and its std::vector(n_knots, iam) constructor throws length_error at once.  Fail the same
        // way, before any allocation.
        if (n_knots == 0)
          throw std::length_error("createProblem: stage '" + stage->get_name() + "' is shorter than dt = " + std::to_string(dt) +
                                  " ms and follows a zero-knot stage (cannot create std::vector larger than max_size())");
        n_knots -= 1;
      }
      last_duration0 = false;
    }
    terminal_set = (int)si;
    if (knot_set.size() + n_knots > kMaxKnots)
      throw std::length_error("createProblem: more than " + std::to_string(kMaxKnots) + " knots (duration / dt)");
    for (std::size_t k = 0; k < n_knots; ++k) knot_set.push_back((int)si);
  }
  knot_set.push_back(terminal_set);  // terminal model = the last stage's running model (:135)
  const EmpcModelDesc model = robot_model_->descWithFrames(frames_.ids());
  return std::make_shared<ShootingProblem>(initial_state_, model, sets, knot_set, platform_params_->tau_f_,
                                           platform_params_->u_lb, platform_params_->u_ub, double(dt) / 1000.0, has_contact_,
                                           squash, integrator);
}

void Trajectory::removeStage(std::size_t idx_stage) {
  if (idx_stage < stages_.size()) stages_.erase(stages_.begin() + (long)idx_stage);
}
void Trajectory::set_initial_state(const VectorXd& initial_state) {
  if (initial_state.size() == get_nx()) initial_state_ = initial_state;
}

}  // namespace eagle_mpc
