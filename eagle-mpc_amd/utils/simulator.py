"""eagle_mpc.utils.simulator.AerialSimulator (bindings/python/eagle_mpc/utils/simulator.py:7-29): the plant of the
closed-loop examples -- IntegratedActionModelRK4 over the free forward dynamics with the multicopter actuation, fed with the
SQUASHED controls (`solver.us_squash[0]`, examples/python/mpc.py:60-61).

Here the step runs on the GPU (kernel k_plant_rk4 behind `empc_plant_step`; checked against the oracle's RK4 node at 1e-11,
tests/test_gpu_mpc.py::test_plant_rk4_parity).  The kinematic tree and the actuation come from the object the robot model
handle belongs to (`mpcController.robot_model` or `trajectory.robot_model`): the simulator steps the plant of that
object's solver, one plant per rollout of its batch (the reference's single plant is batch = 1)."""
import numpy as np


class AerialSimulator:
    def __init__(self, robotModel, platformParams, dt, x0):
        self.robotModel = robotModel
        self.platformParams = platformParams
        self.dt = dt / 1000.0
        self._dt_ms = dt
        owner = robotModel._owner
        self._solver = owner.solver if hasattr(owner, "solver") else None
        if self._solver is None:
            raise ValueError("AerialSimulator needs the robot model of a controller (mpcController.robot_model): the plant runs on "
                             "that controller's solver")
        x0 = np.asarray(x0, dtype=np.float64)
        self._batched = x0.ndim == 2
        self.x0 = x0
        self.states = [x0]
        self.controls = []
        self._solver.plant_states = x0 if self._batched else np.tile(x0, (self._solver.batch, 1))

    def simulateStep(self, u):
        u = np.asarray(u, dtype=np.float64)
        self.controls.append(np.copy(u))
        self._solver.plant_step(self._dt_ms, controls=u if u.ndim == 2 else np.tile(u, (self._solver.batch, 1)))
        x = self._solver.plant_states
        self.states.append(np.copy(x if self._batched else x[0]))
        return self.states[-1]
