// Host-side C++ (YAML / URDF readers, problem factory, Carrot / Rail / Weighted MPC) under AddressSanitizer + UBSan + LeakSanitizer:
// built and run by tests/test_host_sanitize.py.  The solver entry points are stubbed: nothing here touches a GPU.
#include <string>
#include <cstdio>
#include "../../eagle-mpc_amd/host/eagle_mpc.hpp"
// stubs for the solver entry points referenced by sbfddp.cpp (never called here)
extern "C" {
const char* empc_last_error(void) { return ""; }
void empc_solver_params_default(EmpcSolverParams*) {}
EmpcSolver* empc_solver_create(const EmpcProblemDesc*, const EmpcSolverParams*, int, int) { return nullptr; }
void empc_solver_destroy(EmpcSolver*) {}
int empc_solver_update_problem(EmpcSolver*, const EmpcProblemDesc*) { return -1; }
int empc_solver_set_x0(EmpcSolver*, const double*) { return -1; }
int empc_solver_set_warmstart(EmpcSolver*, const double*, const double*) { return -1; }
int empc_solver_set_convergence_init(EmpcSolver*, double) { return -1; }
int empc_solver_solve(EmpcSolver*, int, int) { return -1; }
int empc_solver_get_xs(EmpcSolver*, double*) { return -1; }
int empc_solver_get_us(EmpcSolver*, double*) { return -1; }
int empc_solver_get_us_squash(EmpcSolver*, double*) { return -1; }
int empc_solver_get_cost(EmpcSolver*, double*) { return -1; }
int empc_solver_get_stop(EmpcSolver*, double*) { return -1; }
int empc_solver_get_iters(EmpcSolver*, int*) { return -1; }
int empc_solver_get_status(EmpcSolver*, int*) { return -1; }
int empc_solver_enable_trace(EmpcSolver*, int) { return -1; }
int empc_solver_get_trace(EmpcSolver*, int, double*, int, int*) { return -1; }
}
using namespace eagle_mpc;
int main(int argc, char** argv) {
  const std::string root = argc > 1 ? argv[1] : ".";
  set_yaml_dir(root + "/eagle-mpc_amd/data/yaml");
  set_robot_data_dir(root + "/eagle-mpc_amd/data/robots");
  const char* files[] = {"hexacopter370/trajectories/hover.yaml", "hexacopter370_flying_arm_3/trajectories/displacement.yaml",
                         "hexacopter370_flying_arm_3/trajectories/eagle_catch.yaml", "hextilt_flying_arm_5/trajectories/push_slide.yaml"};
  const int dts[] = {40, 80, 32, 13};
  for (int i = 0; i < 4; ++i) {
    auto t = Trajectory::create();
    t->autoSetup(yaml_dir() + "/" + files[i]);
    auto p = t->createProblem(dts[i], true, "IntegratedActionModelEuler");
    const EmpcProblemDesc& d = p->desc();
    std::printf("%s: T %d sets %d nx %d\n", files[i], d.T, d.n_sets, d.nx);
    if (i == 1) {
      std::vector<VectorXd> ref(101, t->get_initial_state());
      CarrotMpc mpc(t, ref, 80, yaml_dir() + "/hexacopter370_flying_arm_3/mpc/mpc.yaml");
      for (std::size_t tm : {0, 1700, 2010, 8000, 9000}) mpc.updateProblem(tm);
      std::printf("mpc knots %zu T %zu\n", mpc.get_knots(), mpc.get_problem()->get_T());
      RailMpc rail(ref, 80, yaml_dir() + "/hexacopter370_flying_arm_3/mpc/mpc.yaml");
      for (std::size_t tm : {0, 79, 7990, 8000, 20000}) rail.updateProblem(tm);
      std::printf("rail knots %zu\n", rail.get_knots());
      // last: WeightedMpc edits the trajectory (merges its transition stages)
      WeightedMpc weighted(t, 80, yaml_dir() + "/hexacopter370_flying_arm_3/mpc/mpc.yaml");
      for (std::size_t tm : {0, 1990, 2000, 7990, 8000, 9000}) weighted.updateProblem(tm);
      std::printf("weighted stages %zu t_stages %zu\n", t->get_stages().size(), weighted.get_t_stages().size());
    }
  }
  // malformed inputs must throw, not crash
  const std::string bad[] = {"/nonexistent.yaml", root + "/README.md"};
  for (const std::string& b : bad) {
    try {
      auto t = Trajectory::create();
      t->autoSetup(b);
      std::printf("unexpected success on %s\n", b.c_str());
      return 2;
    } catch (const std::exception& e) {
      std::printf("rejected %s: %.60s\n", b.c_str(), e.what());
    }
  }
  return 0;
}
