// Counterpart of the reference's examples/python/mpc.py with the C++ mirror classes: plan once, then run the Carrot
// controller in closed loop against the device-resident RK4 plant (C ABI calls for the plant, include/empc.h).
//   hipcc -std=c++17 -I include examples/cpp/mpc.cpp -L eagle-mpc_amd -lempc -Wl,-rpath,$PWD/eagle-mpc_amd -o mpc
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
    eagle_mpc::SolverSbFDDP planner(problem);
    planner.solve({}, {}, 100);

    eagle_mpc::CarrotMpc mpc(trajectory, planner.get_xs(), 80,
                             eagle_mpc::yaml_dir() + "/hexacopter370_flying_arm_3/mpc/mpc.yaml");
    mpc.updateProblem(0);
    auto solver = mpc.get_solver(1);
    const std::size_t T = mpc.get_problem()->get_T();
    std::vector<eagle_mpc::VectorXd> xs0(planner.get_xs().begin(), planner.get_xs().begin() + T + 1);
    std::vector<eagle_mpc::VectorXd> us0(planner.get_us().begin(), planner.get_us().begin() + T);
    mpc.get_problem()->set_x0(planner.get_xs()[0]);
    solver->solve(xs0, us0, 100);
    solver->set_convergence_init(1e-3);
    if (empc_plant_set_state(solver->handle(), planner.get_xs()[0].data()) != EMPC_OK) throw std::runtime_error(empc_last_error());

    std::size_t t = 0;
    const std::size_t dt_simulator = 2;
    eagle_mpc::VectorXd x(planner.get_xs()[0].size());
    for (int i = 0; i < 100; ++i) {
      mpc.updateProblem(t);
      solver->syncProblem();
      if (empc_plant_get_state(solver->handle(), x.data()) != EMPC_OK) throw std::runtime_error(empc_last_error());
      mpc.get_problem()->set_x0(x);
      solver->solve(solver->get_xs(), solver->get_us(), mpc.get_iters());
      if (empc_plant_step(solver->handle(), dt_simulator / 1000.0, nullptr, 1) != EMPC_OK) throw std::runtime_error(empc_last_error());
      t += dt_simulator;
    }
    if (empc_plant_get_state(solver->handle(), x.data()) != EMPC_OK) throw std::runtime_error(empc_last_error());
    std::printf("t = %zu ms: plant position %.4f %.4f %.4f\n", t, x[0], x[1], x[2]);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
