// Counterpart of the reference's examples/python/trajectory.py with the C++ mirror classes (eagle_mpc namespace).
//   hipcc -std=c++17 -I include examples/cpp/trajectory.cpp -L eagle-mpc_amd -lempc -Wl,-rpath,$PWD/eagle-mpc_amd -o trajectory
#include <cstdio>
#include <string>

#include "../../eagle-mpc_amd/host/eagle_mpc.hpp"

int main(int argc, char** argv) {
  const std::string root = argc > 1 ? argv[1] : ".";
  eagle_mpc::set_yaml_dir(root + "/eagle-mpc_amd/data/yaml");
  eagle_mpc::set_robot_data_dir(root + "/eagle-mpc_amd/data/robots");
  try {
    auto trajectory = eagle_mpc::Trajectory::create();
    trajectory->autoSetup(eagle_mpc::yaml_dir() + "/hexacopter370_flying_arm_3/trajectories/displacement.yaml");
    auto problem = trajectory->createProblem(80, true, "IntegratedActionModelEuler");
    eagle_mpc::SolverSbFDDP solver(problem, trajectory->get_squash());  // the reference's constructor; batch_size = 1
    solver.setCallbacks({std::make_shared<eagle_mpc::CallbackVerbose>()});
    solver.solve({}, {}, 100);
    std::printf("iterations %zu cost %.6f\n", solver.get_iter(), solver.get_cost());
    const auto& xT = solver.get_xs().back();
    std::printf("final position %.4f %.4f %.4f\n", xT[0], xT[1], xT[2]);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
