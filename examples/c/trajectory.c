/* The drop-in boundary in plain C (C99): the calls a binding of ANY language makes -- what examples/cpp/trajectory.cpp and
 * examples/python/trajectory.py do through their mirrors (reference: examples/python/trajectory.py:11-26).
 *   gcc -std=c99 -I include examples/c/trajectory.c -L eagle-mpc_amd -lempc -Wl,-rpath,$PWD/eagle-mpc_amd -o trajectory_c
 *   ./trajectory_c [repository root] [trajectory yaml, relative to the yaml directory] [dt in ms]
 * Without a GPU it stops after the factory and the solver query (exit code 0, "no HIP device"): the host side of the boundary
 * needs none. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "empc.h"

static int fail(const char* what) {
  fprintf(stderr, "%s: %s\n", what, empc_last_error());
  return 1;
}

int main(int argc, char** argv) {
  const char* root = argc > 1 ? argv[1] : ".";
  const char* rel = argc > 2 ? argv[2] : "hexacopter370_flying_arm_3/trajectories/displacement.yaml";
  const int dt_ms = argc > 3 ? atoi(argv[3]) : 80;
  char yaml_dir[1024], robot_dir[1024], path[2048];
  snprintf(yaml_dir, sizeof yaml_dir, "%s/eagle-mpc_amd/data/yaml", root);
  snprintf(robot_dir, sizeof robot_dir, "%s/eagle-mpc_amd/data/robots", root);
  snprintf(path, sizeof path, "%s/%s", yaml_dir, rel);
  printf("libempc %s\n", empc_version());
  if (empc_set_data_dirs(yaml_dir, robot_dir) != EMPC_OK) return fail("set_data_dirs");

  /* Trajectory::create() + autoSetup(path) */
  EmpcTrajectory* traj = empc_trajectory_create(path);
  if (!traj) return fail("trajectory");
  int nx, ndx, nu, n_stages, has_contact, duration_ms;
  if (empc_trajectory_dims(traj, &nx, &ndx, &nu, &n_stages, &has_contact, &duration_ms) != EMPC_OK) return fail("dims");
  printf("nx %d ndx %d nu %d, %d stages, %d ms, contact dynamics: %s\n", nx, ndx, nu, n_stages, duration_ms, has_contact ? "yes" : "no");
  for (int s = 0; s < n_stages; ++s) {
    char name[64];
    int dur, transition, n_costs, n_contacts;
    if (empc_trajectory_stage_info(traj, s, name, sizeof name, &dur, &transition, &n_costs, &n_contacts) != EMPC_OK) return fail("stage");
    printf("  stage %-14s %5d ms  %2d costs  %d contacts%s\n", name, dur, n_costs, n_contacts, transition ? "  (transition)" : "");
  }

  /* trajectory->createProblem(dt, squash, "IntegratedActionModelEuler") */
  EmpcProblem* problem = empc_trajectory_create_problem(traj, dt_ms, 1, "IntegratedActionModelEuler");
  if (!problem) return fail("problem");
  const EmpcProblemDesc* desc = empc_problem_desc(problem);
  printf("shooting problem: T = %d knots of %g s\n", desc->T, desc->dt);

  /* SolverSbFDDP(problem, squash): which kernels would run it (needs no GPU) */
  EmpcSolverParams prm;
  empc_solver_params_default(&prm);
  if (!empc_solver_supported(desc, &prm))
    printf("no kernel instantiation: %s\n", empc_last_error());
  else
    printf("a kernel instantiation exists for this problem class\n");

  int rc = 0;
  if (empc_device_count() < 1) {
    printf("no HIP device: stopping before the solve (the product has no CPU path)\n");
  } else {
    EmpcSolver* solver = empc_solver_create(desc, &prm, 1, 0);
    if (!solver) {
      rc = fail("solver");
    } else {
      int iters = 0;
      double cost = 0.0;
      double* xs = (double*)malloc(sizeof(double) * (size_t)(desc->T + 1) * (size_t)nx);
      if (empc_solver_set_x0(solver, NULL) != EMPC_OK || empc_solver_set_warmstart(solver, NULL, NULL) != EMPC_OK ||
          empc_solver_solve(solver, 100, 0) != EMPC_OK || empc_solver_get_iters(solver, &iters) != EMPC_OK ||
          empc_solver_get_cost(solver, &cost) != EMPC_OK || empc_solver_get_xs(solver, xs) != EMPC_OK) {
        rc = fail("solve");
      } else {
        const double* xT = xs + (size_t)desc->T * (size_t)nx;
        printf("kernels: %s\niterations %d cost %.6f\nfinal position %.4f %.4f %.4f\n", empc_solver_kernel_family(solver), iters, cost,
               xT[0], xT[1], xT[2]);
      }
      free(xs);
      empc_solver_destroy(solver);
    }
  }
  empc_problem_destroy(problem);
  empc_trajectory_destroy(traj);
  return rc;
}
