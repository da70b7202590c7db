// robot_model: URDF subset -> flattened kinematic tree with a free-flyer root.
// Stands in for pinocchio::urdf::buildModel(path, JointModelFreeFlyer(), model) (reference:
// src/trajectory.cpp:29-31, src/mpc-base.cpp:24-26): fixed joints are merged into their parent body, every
// link becomes an operational frame named after the link, revolute/continuous joints become bodies in
// depth-first URDF order, <limit effort> feeds MultiCopterBaseParams::setControlLimits.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/empc_types.h"

namespace eagle_mpc {

struct RobotFrame {
  std::string name;
  int body;
  double R[9];
  double p[3];
};

class RobotModel {
 public:
  // Build from a URDF file (throws std::runtime_error on unsupported content).
  static RobotModel fromUrdf(const std::string& path);
  static RobotModel fromUrdfString(const std::string& xml);

  int nq() const { return desc_.nq; }
  int nv() const { return desc_.nv; }
  int njoints() const { return desc_.nbodies - 1; }
  // pinocchio::Model::getFrameId semantics: returns frames().size() when the name is unknown
  std::size_t getFrameId(const std::string& name) const;
  const std::vector<RobotFrame>& frames() const { return frames_; }
  const std::vector<double>& effortLimit() const { return effort_limit_; }  // size nv (zeros for the base)
  const std::vector<std::string>& jointNames() const { return joint_names_; }
  double totalMass() const;

  // Model descriptor with no operational frames selected yet.
  const EmpcModelDesc& desc() const { return desc_; }
  // Copy of the descriptor with the given frames (indices into frames()) placed in its frame table.
  EmpcModelDesc descWithFrames(const std::vector<int>& frame_ids) const;

 private:
  EmpcModelDesc desc_;
  std::vector<RobotFrame> frames_;
  std::vector<double> effort_limit_;
  std::vector<std::string> joint_names_;
};

}  // namespace eagle_mpc
