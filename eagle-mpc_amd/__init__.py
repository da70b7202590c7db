"""eagle-mpc_amd: MI355X-native batched Squash-box FDDP (the hot path of PepMS/eagle-mpc).

Python is plumbing only: this package binds the C ABI of ``libempc.so`` (include/empc.h) with ctypes and mirrors the
names of the reference's Python bindings (bindings/python/eagle_mpc/{trajectory,sbfddp}.hpp).  There is no CPU
implementation of the solver here: constructing a solver without the HIP library or without a GPU raises.

The directory name contains a hyphen, so import it through ``load()`` in the repository-root helper
``empc_loader.py`` (``import empc_loader; empc = empc_loader.load()``).
"""
import ctypes as C
import os

import numpy as np

from . import ctypes_defs as T
from . import utils  # CallbackLogger / saveLogfile with the reference's layout

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EMPC_LIB_PATH") or os.path.join(_PKG_DIR, "libempc.so")  # EMPC_LIB_PATH: diagnostic builds (make stamps)
YAML_DIR = os.path.join(_PKG_DIR, "data", "yaml")  # the problem files the reference ships under yaml/ (unchanged: configuration data)
ROBOT_DIR = os.path.join(_PKG_DIR, "data", "robots")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class EmpcError(RuntimeError):
    pass


def _ptr(a, dtype=np.float64):
    if a is None:
        return None
    assert a.dtype == dtype and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp if dtype == np.float64 else _ip)


_lib = None


def lib():
    """Load libempc.so (built by __graft_entry__.build() / `make -C eagle-mpc_amd`). Fails loudly if missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EmpcError("libempc.so not found at %s: build it first (python -c 'import __graft_entry__ as g; g.build()'); "
                        "there is no Python/CPU fallback for the solver" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.empc_last_error.restype = C.c_char_p
    L.empc_version.restype = C.c_char_p
    L.empc_trajectory_create.restype = C.c_void_p
    L.empc_trajectory_create.argtypes = [C.c_char_p]
    L.empc_trajectory_destroy.argtypes = [C.c_void_p]
    L.empc_trajectory_dims.argtypes = [C.c_void_p] + [_ip] * 6
    L.empc_trajectory_stage_info.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, _ip, _ip, _ip, _ip]
    L.empc_trajectory_stage_t_ini.restype = C.c_longlong
    L.empc_trajectory_stage_t_ini.argtypes = [C.c_void_p, C.c_int]
    L.empc_trajectory_stage_cost.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int, _dp, _ip]
    L.empc_trajectory_get_initial_state.argtypes = [C.c_void_p, _dp]
    L.empc_trajectory_set_initial_state.argtypes = [C.c_void_p, _dp]
    L.empc_trajectory_get_platform.argtypes = [C.c_void_p, _dp, _dp, _dp, _ip]
    L.empc_trajectory_create_problem.restype = C.c_void_p
    L.empc_trajectory_create_problem.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
    L.empc_trajectory_get_param.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
    L.empc_trajectory_remove_stage.argtypes = [C.c_void_p, C.c_int]
    L.empc_trajectory_robot_model_path.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.empc_trajectory_stage_cost_type.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int]
    L.empc_trajectory_stage_contact.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    L.empc_trajectory_get_platform_params.argtypes = [C.c_void_p, _dp, C.c_char_p, C.c_int]
    L.empc_trajectory_get_rotor_pose.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _ip]
    L.empc_problem_destroy.argtypes = [C.c_void_p]
    L.empc_problem_desc.restype = C.POINTER(T.ProblemDesc)
    L.empc_problem_desc.argtypes = [C.c_void_p]
    L.empc_problem_set_x0.argtypes = [C.c_void_p, _dp]
    L.empc_solver_params_default.argtypes = [C.POINTER(T.SolverParams)]
    L.empc_solver_create.restype = C.c_void_p
    L.empc_solver_create.argtypes = [C.POINTER(T.ProblemDesc), C.POINTER(T.SolverParams), C.c_int, C.c_int]
    L.empc_solver_destroy.argtypes = [C.c_void_p]
    L.empc_mpc_solver_type.argtypes = [C.c_void_p, C.c_void_p]
    L.empc_solver_supported.argtypes = [C.POINTER(T.ProblemDesc), C.POINTER(T.SolverParams)]
    L.empc_solver_set_cost_refs.argtypes = [C.c_void_p, C.c_int, C.c_char_p, _dp, C.c_int, C.c_int, C.c_double]
    L.empc_solver_update_problem.argtypes = [C.c_void_p, C.POINTER(T.ProblemDesc)]
    L.empc_solver_kernel_family.restype = C.c_char_p
    L.empc_solver_kernel_family.argtypes = [C.c_void_p]
    L.empc_solver_set_x0.argtypes = [C.c_void_p, _dp]
    L.empc_solver_set_warmstart.argtypes = [C.c_void_p, _dp, _dp]
    L.empc_solver_set_convergence_init.argtypes = [C.c_void_p, C.c_double]
    L.empc_solver_solve.argtypes = [C.c_void_p, C.c_int, C.c_int]
    for name in ("xs", "us", "us_squash", "cost", "stop"):
        getattr(L, "empc_solver_get_" + name).argtypes = [C.c_void_p, _dp]
    L.empc_solver_get_iters.argtypes = [C.c_void_p, _ip]
    L.empc_solver_get_status.argtypes = [C.c_void_p, _ip]
    L.empc_solver_pack_results_device.argtypes = [C.c_void_p, C.c_void_p, _ip]
    L.empc_solver_get_stats.argtypes = [C.c_void_p, C.POINTER(T.SolveStats)]
    L.empc_solver_enable_trace.argtypes = [C.c_void_p, C.c_int]
    L.empc_solver_get_trace.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int, _ip]
    L.empc_solver_dims.argtypes = [C.c_void_p] + [_ip] * 6
    L.empc_tape_layout.argtypes = [C.c_void_p, C.POINTER(T.TapeLayout)]
    L.empc_linearize_batch.argtypes = [C.c_void_p, _dp, _dp, C.c_double, C.c_int, _dp, _dp, _dp]
    L.empc_backward_batch.argtypes = [C.c_void_p, C.c_double, C.c_int, _dp, _dp, _dp, _dp, _ip]
    L.empc_rollout_batch.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, _dp, _dp, _dp, _ip]
    L.empc_solver_get_states.argtypes = [C.c_void_p, C.POINTER(T.TrajState)]
    L.empc_solver_set_states.argtypes = [C.c_void_p, C.POINTER(T.TrajState)]
    L.empc_sweep_batch.argtypes = [C.c_void_p, C.c_int]
    L.empc_select_batch.argtypes = [C.c_void_p, _ip, _dp, _dp]
    L.empc_solver_get_trials.argtypes = [C.c_void_p, _dp, _dp, _ip]
    L.empc_solver_get_tape.argtypes = [C.c_void_p, _dp]
    L.empc_solver_get_gains.argtypes = [C.c_void_p, _dp, _dp, _dp]
    L.empc_solver_set_gains.argtypes = [C.c_void_p, _dp, _dp]
    L.empc_solver_stream_begin.argtypes = [C.c_void_p, C.c_int, _dp]
    L.empc_solver_stream_run.argtypes = [C.c_void_p, C.c_int]
    L.empc_solver_stream_results.argtypes = [C.c_void_p, _dp, _ip]
    L.empc_solver_stream_results_device.argtypes = [C.c_void_p, C.c_void_p]
    L.empc_solver_device_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_char_p, C.c_int]
    L.empc_plant_set_state.argtypes = [C.c_void_p, _dp]
    L.empc_plant_get_state.argtypes = [C.c_void_p, _dp]
    L.empc_plant_step.argtypes = [C.c_void_p, C.c_double, _dp, C.c_int]
    L.empc_solver_set_x0_from_plant.argtypes = [C.c_void_p]
    L.empc_carrot_mpc_create.restype = C.c_void_p
    L.empc_carrot_mpc_create.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int, C.c_char_p]
    L.empc_carrot_mpc_destroy.argtypes = [C.c_void_p]
    L.empc_carrot_mpc_params.argtypes = [C.c_void_p] + [_ip] * 7
    L.empc_carrot_mpc_t_stages.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    L.empc_carrot_mpc_update_problem.argtypes = [C.c_void_p, C.c_longlong]
    L.empc_carrot_mpc_state_reference.argtypes = [C.c_void_p, C.c_longlong, _dp]
    L.empc_carrot_mpc_set_x0.argtypes = [C.c_void_p, _dp]
    L.empc_carrot_mpc_problem_desc.restype = C.POINTER(T.ProblemDesc)
    L.empc_carrot_mpc_problem_desc.argtypes = [C.c_void_p]
    L.empc_rail_mpc_create.restype = C.c_void_p
    L.empc_rail_mpc_create.argtypes = [_dp, C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.empc_weighted_mpc_create.restype = C.c_void_p
    L.empc_weighted_mpc_create.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
    L.empc_mpc_destroy.argtypes = [C.c_void_p]
    L.empc_mpc_params.argtypes = [C.c_void_p] + [_ip] * 6
    L.empc_mpc_update_problem.argtypes = [C.c_void_p, C.c_longlong]
    L.empc_mpc_set_x0.argtypes = [C.c_void_p, _dp]
    L.empc_mpc_problem_desc.restype = C.POINTER(T.ProblemDesc)
    L.empc_mpc_problem_desc.argtypes = [C.c_void_p]
    L.empc_rail_mpc_state_reference.argtypes = [C.c_void_p, C.c_longlong, _dp]
    L.empc_weighted_mpc_t_stages.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
    L.empc_set_data_dirs(YAML_DIR.encode(), ROBOT_DIR.encode())
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise EmpcError(lib().empc_last_error().decode())


def device_count():
    return lib().empc_device_count()


def solver_supported(problem, params=None):
    """True when SolverSbFDDP(problem) has a kernel instantiation (needs no GPU); the reason otherwise is in last_error()."""
    prm = params if params is not None else default_params()
    return bool(lib().empc_solver_supported(C.byref(problem.desc), C.byref(prm)))


def last_error():
    return lib().empc_last_error().decode()


def yaml_path(rel):
    return os.path.join(YAML_DIR, rel)


class Problem:
    """ShootingProblem handle (flat EmpcProblemDesc)."""

    def __init__(self, handle, owner):
        self._h = C.c_void_p(handle)
        self._owner = owner

    @property
    def desc(self):
        p = lib().empc_problem_desc(self._h)
        if not p:
            raise EmpcError(lib().empc_last_error().decode())
        return p.contents

    @property
    def T(self):
        return self.desc.T

    @property
    def x0(self):
        d = self.desc
        return np.array([d.x0[i] for i in range(d.nx)])

    @x0.setter
    def x0(self, value):
        v = np.ascontiguousarray(value, dtype=np.float64)
        _check(lib().empc_problem_set_x0(self._h, _ptr(v)))

    def __del__(self):
        try:
            lib().empc_problem_destroy(self._h)
        except Exception:
            pass


class SquashingModelSmoothSat:
    """Mirror of crocoddyl.SquashingModelSmoothSat(u_lb, u_ub, ns) as the reference builds it (src/trajectory.cpp:49-50):
    plain data -- the limits of the smooth saturation sigma(s) and its smoothing.  The HIP kernels take the limits from the
    problem (EmpcProblemDesc.u_lb / u_ub, the same numbers); a solver given a squashing model checks that they agree."""

    def __init__(self, u_lb, u_ub, ns):
        self.u_lb = np.asarray(u_lb, dtype=np.float64).copy()
        self.u_ub = np.asarray(u_ub, dtype=np.float64).copy()
        assert self.u_lb.shape == self.u_ub.shape == (int(ns),)
        self.ns = int(ns)
        self.s_lb, self.s_ub = self.u_lb.copy(), self.u_ub.copy()  # SquashingModelSmoothSat: s_lb = u_lb, s_ub = u_ub
        self.smooth = 0.1


class RotorPose:
    """One entry of MultiCopterBaseParams.rotors_pose (a pinocchio::SE3 in the reference): rotation (3 x 3), translation (3)
    of the rotor frame in the base link, and the rotor's spin direction (the platform file's `spin_direction`)."""

    def __init__(self, rotation, translation, spin_direction):
        self.rotation, self.translation, self.spin_direction = rotation, translation, spin_direction


class PlatformParams:
    """MultiCopterBaseParams (bindings/python/eagle_mpc/multicopter-base-params.hpp:47-88): cf, cm, n_rotors, tau_f (6 x
    n_rotors), max_thrust / min_thrust, base_link_name, the control limits u_lb / u_ub (rotor thrusts, then arm joint torques:
    setControlLimits, src/multicopter-base-params.cpp:89-101), rotors_pose.  `max_torque` / `min_torque` are declared by the
    reference and never filled (include/eagle_mpc/multicopter-base-params.hpp:55-56); here they hold the arm part of the limits."""

    def __init__(self, tau_f, u_lb, u_ub, scalars=None, base_link_name="", rotors_pose=()):
        self.tau_f, self.u_lb, self.u_ub = tau_f, u_lb, u_ub
        self.n_rotors = tau_f.shape[1]
        if scalars is None:
            scalars = (0.0, 0.0, float(u_ub[0]), float(u_lb[0]), 0.0, 0.0)
        self.cf, self.cm, self.max_thrust, self.min_thrust, self.max_prop_speed, self.min_prop_speed = [float(v) for v in scalars]
        self.base_link_name = base_link_name
        self.rotors_pose = list(rotors_pose)
        self.max_torque, self.min_torque = u_ub[self.n_rotors:].copy(), u_lb[self.n_rotors:].copy()


class RobotModel:
    """What `trajectory.robot_model` / `mpcController.robot_model` hand to user code: name and dimensions.  `_owner` is the
    object whose kinematic tree this is (a Trajectory or a controller): utils.simulator.AerialSimulator runs its plant on
    the owner's device-side model."""

    def __init__(self, name, nq, nv, owner):
        self.name, self.nq, self.nv, self._owner = name, nq, nv, owner


class StageInfo:
    """One entry of Trajectory.stages: the read-only view of a Stage (bindings/python/eagle_mpc/stage.hpp:47-73: name, duration,
    t_ini, is_transition, is_terminal, costs, cost_types, contacts, contact_types).  `costs` is a list of {name, weight,
    active, type}; `cost_types` / `contact_types` map names to the factory's type names; `is_terminal` is false for every stage,
    as in the reference (src/stage.cpp:15 sets it once and nothing changes it)."""

    def __init__(self, name, duration, is_transition, n_costs, n_contacts, t_ini, costs, contacts=()):
        self.name, self.duration, self.is_transition, self.t_ini = name, duration, is_transition, t_ini
        self.n_costs, self.n_contacts, self.costs = n_costs, n_contacts, costs
        self.contacts = list(contacts)
        self.cost_types = {c["name"]: c.get("type") for c in costs}
        self.contact_types = {c["name"]: c["type"] for c in self.contacts}
        self.is_terminal = False

    def __repr__(self):
        return "StageInfo(%r, duration=%d ms, t_ini=%d ms, costs=%d, contacts=%d%s)" % (
            self.name, self.duration, self.t_ini, self.n_costs, self.n_contacts, ", transition" if self.is_transition else "")


class Trajectory:
    """Mirror of eagle_mpc.Trajectory (bindings/python/eagle_mpc/trajectory.hpp:24-63)."""

    def __init__(self):
        self._h = None

    def autoSetup(self, yaml_file):
        h = lib().empc_trajectory_create(os.fspath(yaml_file).encode())
        if not h:
            raise EmpcError(lib().empc_last_error().decode())
        self._h = C.c_void_p(h)
        v = [C.c_int() for _ in range(6)]
        _check(lib().empc_trajectory_dims(self._h, *[C.byref(x) for x in v]))
        self.nx, self.ndx, self.nu, _, hc, _ = [x.value for x in v]
        self.has_contact = bool(hc)

    @property
    def duration(self):
        """get_duration(): total duration in ms, as autoSetup summed it (src/trajectory.cpp:90-99)"""
        n = C.c_int()
        _check(lib().empc_trajectory_dims(self._h, None, None, None, None, None, C.byref(n)))
        return n.value

    @property
    def stages(self):
        """get_stages() by value: one StageInfo per stage, in order (name, duration, is_transition, t_ini, the names / weights /
        active flags of its costs, the number of its contacts).  A snapshot: edit the trajectory through its own methods."""
        return [StageInfo(**self.stage_info(i)) for i in range(self.n_stages)]

    @property
    def robot_model_path(self):
        """get_robot_model_path(): the URDF file the robot model was built from"""
        n = lib().empc_trajectory_robot_model_path(self._h, None, 0)
        if n < 0:
            raise EmpcError(lib().empc_last_error().decode())
        buf = C.create_string_buffer(n + 1)
        lib().empc_trajectory_robot_model_path(self._h, buf, n + 1)
        return buf.value.decode()

    def removeStage(self, idx_stage):
        """Trajectory::removeStage (src/trajectory.cpp:145-150): erases stage `idx_stage`; durations and start times of the
        other stages stay as they are, as in the reference"""
        if lib().empc_trajectory_remove_stage(self._h, int(idx_stage)) != 0:
            raise IndexError(lib().empc_last_error().decode())

    @property
    def n_stages(self):
        """len(get_stages()); live, because WeightedMpc removes the transition stages of the trajectory it is given"""
        n = C.c_int()
        _check(lib().empc_trajectory_dims(self._h, None, None, None, C.byref(n), None, None))
        return n.value

    def createProblem(self, dt=0, squash=True, integration_method="IntegratedActionModelEuler"):
        h = lib().empc_trajectory_create_problem(self._h, int(dt), int(bool(squash)), integration_method.encode())
        if not h:
            raise EmpcError(lib().empc_last_error().decode())
        return Problem(h, self)

    @property
    def initial_state(self):
        x = np.zeros(self.nx)
        _check(lib().empc_trajectory_get_initial_state(self._h, _ptr(x)))
        return x

    @initial_state.setter
    def initial_state(self, value):
        v = np.ascontiguousarray(value, dtype=np.float64)
        assert v.shape == (self.nx,)
        _check(lib().empc_trajectory_set_initial_state(self._h, _ptr(v)))

    def platform(self):
        n = C.c_int()
        _check(lib().empc_trajectory_get_platform(self._h, None, None, None, C.byref(n)))
        tau_f = np.zeros((6, n.value))
        lb = np.zeros(self.nu)
        ub = np.zeros(self.nu)
        _check(lib().empc_trajectory_get_platform(self._h, _ptr(tau_f), _ptr(lb), _ptr(ub), C.byref(n)))
        return tau_f, lb, ub

    @property
    def platform_params(self):
        """get_platform_params(): the fields of MultiCopterBaseParams the hot path uses (tau_f, u_lb, u_ub, n_rotors;
        bindings/python/eagle_mpc/multicopter-base-params.hpp)"""
        tau_f, lb, ub = self.platform()
        sc = np.zeros(6)
        name = C.create_string_buffer(128)
        _check(lib().empc_trajectory_get_platform_params(self._h, _ptr(sc), name, 128))
        poses = []
        for i in range(tau_f.shape[1]):
            R, p, spin = np.zeros((3, 3)), np.zeros(3), C.c_int()
            _check(lib().empc_trajectory_get_rotor_pose(self._h, i, _ptr(R), _ptr(p), C.byref(spin)))
            poses.append(RotorPose(R, p, spin.value))
        return PlatformParams(tau_f, lb, ub, sc, name.value.decode(), poses)

    @property
    def robot_model(self):
        """get_robot_model(): a handle naming the robot (the pinocchio model itself lives behind the C ABI as EmpcModelDesc)"""
        return RobotModel(self.get_param("robot/name").strip(chr(34)), self.nx - self.ndx // 2, self.ndx // 2, self)

    @property
    def squash(self):
        """get_squash(): SquashingModelSmoothSat(u_lb, u_ub, nu) of the platform (src/trajectory.cpp:49-50)"""
        _, lb, ub = self.platform()
        return SquashingModelSmoothSat(lb, ub, self.nu)

    def stage_info(self, i):
        name = C.create_string_buffer(64)
        v = [C.c_int() for _ in range(4)]
        _check(lib().empc_trajectory_stage_info(self._h, i, name, 64, *[C.byref(x) for x in v]))
        costs = []
        for k in range(v[2].value):
            cname, w, a = C.create_string_buffer(64), C.c_double(), C.c_int()
            _check(lib().empc_trajectory_stage_cost(self._h, i, k, cname, 64, C.byref(w), C.byref(a)))
            ctype = C.create_string_buffer(64)
            _check(lib().empc_trajectory_stage_cost_type(self._h, i, cname.value, ctype, 64))
            costs.append(dict(name=cname.value.decode(), weight=w.value, active=bool(a.value), type=ctype.value.decode()))
        contacts = []
        for k in range(v[3].value):
            cname, ctype = C.create_string_buffer(64), C.create_string_buffer(64)
            _check(lib().empc_trajectory_stage_contact(self._h, i, k, cname, 64, ctype, 64))
            contacts.append(dict(name=cname.value.decode(), type=ctype.value.decode()))
        return dict(name=name.value.decode(), duration=v[0].value, is_transition=bool(v[1].value), n_costs=v[2].value,
                    n_contacts=v[3].value, t_ini=int(lib().empc_trajectory_stage_t_ini(self._h, i)), costs=costs, contacts=contacts)

    def get_param(self, key):
        buf = C.create_string_buffer(4096)
        n = lib().empc_trajectory_get_param(self._h, key.encode(), buf, 4096)
        if n < 0:
            raise KeyError(lib().empc_last_error().decode())
        return buf.value.decode()

    def __del__(self):
        try:
            if self._h:
                lib().empc_trajectory_destroy(self._h)
        except Exception:
            pass


def default_params():
    p = T.SolverParams()
    lib().empc_solver_params_default(C.byref(p))
    return p


class IterationRecord:
    """What a crocoddyl callback reads from the solver after one iteration (names of crocoddyl.SolverAbstract)."""

    def __init__(self, r, solver):
        self.phase, self.iter = int(r[0]), int(r[1])
        self.cost, self.stop, self.x_reg, self.u_reg, self.stepLength = float(r[2]), float(r[3]), float(r[4]), float(r[4]), float(r[5])
        self.is_feasible = bool(r[6])
        self.dV, self.dVexp, self.gap_norm, self.d = float(r[7]), float(r[8]), float(r[9]), (float(r[10]), float(r[11]))
        self.solver = solver


class CallbackVerbose:
    """crocoddyl.CallbackVerbose: one line per iteration (iter, cost, stop, grad, xreg, ureg, step, feasibility).
    ``lines`` keeps the iteration lines only (one per iteration, what tests and log readers index); the header that is reprinted
    every ten iterations goes to the stream but not into ``lines``."""

    def __init__(self, stream=None):
        self.stream = stream
        self.lines = []

    def __call__(self, rec):
        if rec.iter % 10 == 0:  # crocoddyl reprints the header every ten iterations (as host/sbfddp.cpp does)
            print("iter \t cost \t      stop \t    grad \t  xreg \t      ureg \t step \t feas", file=self.stream)
        self._emit("%4d  %0.5e  %0.5e  %0.5e  %10.5e  %10.5e   %0.4f     %d" %
                   (rec.iter, rec.cost, rec.stop, -rec.d[1], rec.x_reg, rec.u_reg, rec.stepLength, int(rec.is_feasible)))

    def _emit(self, line):
        self.lines.append(line)
        print(line, file=self.stream)


class SolverSbFDDP:
    """Batched mirror of eagle_mpc.SolverSbFDDP (bindings/python/eagle_mpc/sbfddp.hpp:24-80).

    ``solve`` follows the reference signature; results are exposed both per batch (``xs_batch`` ...) and, for
    trajectory 0, under the reference's property names (``xs``, ``us``, ``us_squash``, ``iter``, ``cost``).
    """

    SOLVER_TYPE = T.SOLVER_SBFDDP
    TRACE_BUDGET_BYTES = 256 << 20  # device memory the callbacks' iteration trace may take without being asked for

    def __init__(self, problem, squashing_model=None, *, batch=1, device=0, params=None):
        """SolverSbFDDP(problem, squashing_model) as in the reference (include/eagle_mpc/sbfddp.hpp:39-40,
        examples/python/trajectory.py:20-21: ``SolverSbFDDP(problem, trajectory.squash)``); keyword-only extensions:
        ``batch`` rollouts of the problem on HIP device ``device``."""
        self.problem = problem
        self.batch = int(batch)
        if squashing_model is not None:
            if isinstance(squashing_model, (int, np.integer)) and not isinstance(squashing_model, bool):
                raise TypeError("the second argument of SolverSbFDDP is the squashing model (trajectory.squash), as in the "
                                "reference; rounds 1-2 of this mirror took the batch there: pass batch=%d by keyword" % squashing_model)
            if not isinstance(squashing_model, SquashingModelSmoothSat):
                raise TypeError("squashing_model must be a SquashingModelSmoothSat (e.g. trajectory.squash)")
            d = problem.desc
            lb = np.array([d.u_lb[i] for i in range(d.nu)])
            ub = np.array([d.u_ub[i] for i in range(d.nu)])
            if squashing_model.ns != d.nu or not (np.array_equal(squashing_model.u_lb, lb) and np.array_equal(squashing_model.u_ub, ub)):
                raise EmpcError("squashing model does not belong to this problem (control limits differ)")
        self.squashing_model = squashing_model
        self._callbacks = []
        prm = T.SolverParams.from_buffer_copy(params) if params is not None else default_params()
        prm.solver_type = self.SOLVER_TYPE
        h = lib().empc_solver_create(C.byref(problem.desc), C.byref(prm), self.batch, int(device))
        if not h:
            raise EmpcError(lib().empc_last_error().decode())
        self._h = C.c_void_p(h)
        v = [C.c_int() for _ in range(6)]
        _check(lib().empc_solver_dims(self._h, *[C.byref(x) for x in v]))
        _, self.T, self.nx, self.ndx, self.nu, self.rec = [x.value for x in v]
        self._convergence_init = prm.convergence_init
        self._n_alphas = prm.n_alphas

    # -- callbacks (SolverAbstract::setCallbacks; examples/python/trajectory.py:25) ----------------------------------------
    def setCallbacks(self, callbacks):
        """Callables invoked once per DDP iteration of trajectory 0 with a record of that iteration (attributes iter, cost,
        stop, x_reg, u_reg, stepLength, dV, dVexp, is_feasible, phase ...).  The solve runs entirely on the device, so the
        callbacks are replayed from the device iteration trace right after solve() returns, in iteration order -- the same
        information at the same granularity as crocoddyl's per-iteration hook (src/sbfddp.cpp:303-307, 381-385)."""
        self._callbacks = list(callbacks)
        if self._callbacks and not getattr(self, "_trace_cap", 0):
            self.enable_trace(512)

    def getCallbacks(self):
        return list(self._callbacks)

    def _size_callback_trace(self, maxiter):
        """callbacks replay the device trace: the ring must hold every iteration of the solve (two FDDP passes + the DDP
        clean-up, maxiter + 1 records each at most), or the first records of a long solve would be lost"""
        if self._callbacks:
            need = 3 * (int(maxiter) + 1) + 8
            if getattr(self, "_trace_cap", 0) < need:
                # the ring lives on the device: batch x records x 96 B.  Callbacks replay trajectory 0 only, but the ring is per
                # trajectory; beyond TRACE_BUDGET_BYTES the caller must size it (enable_trace) or drop the callbacks
                nbytes = self.batch * need * 96
                if nbytes > self.TRACE_BUDGET_BYTES:
                    raise EmpcError("callbacks on a batch of %d with maxiter %d need a %.0f MB iteration trace on the device "
                                    "(budget %.0f MB): call enable_trace(records) yourself, lower maxiter, or use a batch-1 solver "
                                    "for the verbose run" % (self.batch, maxiter, nbytes / 1e6, self.TRACE_BUDGET_BYTES / 1e6))
                self.enable_trace(need)

    def _replay_callbacks(self):
        if not self._callbacks:
            return
        xs, us = None, None
        for r in self.trace(0):
            rec = IterationRecord(r, self)
            for cb in self._callbacks:
                cb(rec)

    # -- reference API -------------------------------------------------------------------------------------
    def solve(self, init_xs=None, init_us=None, maxiter=100, is_feasible=False, regInit=1e-9, x0s=None):
        """solve(init_xs, init_us, maxiter, is_feasible, regInit).  Batched extensions: ``x0s`` (B x nx; the string
        "plant" takes the device-resident plant states) and ``init_xs = init_us = "previous"``, which warm-starts every
        trajectory from its own last solution without moving it through the host -- the MPC call
        ``solver.solve(solver.xs, solver.us, iters)`` of examples/python/mpc.py:55."""
        B, T_, nx, nu = self.batch, self.T, self.nx, self.nu
        if isinstance(x0s, str):
            if x0s != "plant":
                raise ValueError("x0s must be an array, None or 'plant'")
            _check(lib().empc_solver_set_x0_from_plant(self._h))
        else:
            if x0s is not None:
                x0s = np.ascontiguousarray(x0s, dtype=np.float64).reshape(B, nx)
            _check(lib().empc_solver_set_x0(self._h, _ptr(x0s)))
        if isinstance(init_xs, str) or isinstance(init_us, str):
            if init_xs != "previous" or init_us != "previous":
                raise ValueError("init_xs / init_us must both be 'previous' to reuse the last solution")
            self._size_callback_trace(maxiter)
            _check(lib().empc_solver_solve(self._h, int(maxiter), int(bool(is_feasible))))
            self._replay_callbacks()
            return True
        xs = us = None
        if init_xs is not None and len(init_xs):
            xs = np.ascontiguousarray(init_xs, dtype=np.float64)
            if xs.shape == (T_ + 1, nx):
                xs = np.ascontiguousarray(np.broadcast_to(xs, (B, T_ + 1, nx)))
            assert xs.shape == (B, T_ + 1, nx)
        if init_us is not None and len(init_us):
            us = np.ascontiguousarray(init_us, dtype=np.float64)
            if us.shape == (T_, nu):
                us = np.ascontiguousarray(np.broadcast_to(us, (B, T_, nu)))
            assert us.shape == (B, T_, nu)
        _check(lib().empc_solver_set_warmstart(self._h, _ptr(xs), _ptr(us)))
        self._size_callback_trace(maxiter)
        _check(lib().empc_solver_solve(self._h, int(maxiter), int(bool(is_feasible))))
        self._replay_callbacks()
        return True

    @property
    def convergence_init(self):
        return self._convergence_init

    @convergence_init.setter
    def convergence_init(self, v):
        _check(lib().empc_solver_set_convergence_init(self._h, float(v)))
        self._convergence_init = float(v)

    @property
    def kernel_family(self):
        """'runtime model' or 'baked <robot>...': which kernel instantiation serves this solver (include/empc.h)."""
        return lib().empc_solver_kernel_family(self._h).decode()

    def update_problem(self):
        _check(lib().empc_solver_update_problem(self._h, C.byref(self.problem.desc)))

    def set_cost_refs(self, knot, name, ref=None, active=None, weight=None):
        """Edit one cost entry of node `knot` in place -- residual.reference / cost.active / cost.weight of the reference's
        crocoddyl models, as MpcAbstract.updateProblem does (src/mpc-controllers/carrot-mpc.cpp:298-359).  None = keep."""
        r = None if ref is None else np.ascontiguousarray(ref, dtype=np.float64).ravel()
        _check(lib().empc_solver_set_cost_refs(self._h, int(knot), name.encode(), _ptr(r), 0 if r is None else r.size,
                                               -1 if active is None else int(bool(active)),
                                               float("nan") if weight is None else float(weight)))

    # -- plant of closed-loop runs (AerialSimulator, bindings/python/eagle_mpc/utils/simulator.py) ------------
    @property
    def plant_states(self):
        x = np.zeros((self.batch, self.nx))
        _check(lib().empc_plant_get_state(self._h, _ptr(x)))
        return x

    @plant_states.setter
    def plant_states(self, value):
        x = np.ascontiguousarray(value, dtype=np.float64)
        assert x.shape == (self.batch, self.nx)
        _check(lib().empc_plant_set_state(self._h, _ptr(x)))

    def plant_step(self, dt_ms, controls=None, substeps=1):
        """simulateStep(u) for every plant: RK4 over dt_ms milliseconds (x substeps).  controls=None applies each
        trajectory's own us_squash[0] of the last solve."""
        u = None
        if controls is not None:
            u = np.ascontiguousarray(controls, dtype=np.float64)
            if u.shape == (self.nu,):
                u = np.ascontiguousarray(np.broadcast_to(u, (self.batch, self.nu)))
            assert u.shape == (self.batch, self.nu)
        _check(lib().empc_plant_step(self._h, float(dt_ms) / 1000.0, _ptr(u), int(substeps)))

    def _get(self, name, shape):
        a = np.zeros(shape)
        _check(getattr(lib(), "empc_solver_get_" + name)(self._h, _ptr(a)))
        return a

    @property
    def xs_batch(self):
        return self._get("xs", (self.batch, self.T + 1, self.nx))

    @property
    def us_batch(self):
        return self._get("us", (self.batch, self.T, self.nu))

    @property
    def us_squash_batch(self):
        return self._get("us_squash", (self.batch, self.T, self.nu))

    @property
    def cost_batch(self):
        return self._get("cost", (self.batch,))

    @property
    def stop_batch(self):
        return self._get("stop", (self.batch,))

    @property
    def iter_batch(self):
        a = np.zeros(self.batch, dtype=np.int32)
        _check(lib().empc_solver_get_iters(self._h, _ptr(a, np.int32)))
        return a

    @property
    def status_batch(self):
        a = np.zeros(self.batch, dtype=np.int32)
        _check(lib().empc_solver_get_status(self._h, _ptr(a, np.int32)))
        return a

    xs = property(lambda self: list(self.xs_batch[0]))
    us = property(lambda self: list(self.us_batch[0]))
    us_squash = property(lambda self: list(self.us_squash_batch[0]))
    iter = property(lambda self: int(self.iter_batch[0]))
    cost = property(lambda self: float(self.cost_batch[0]))
    stop = property(lambda self: float(self.stop_batch[0]))

    def pack_results_device(self, device_ptr=None):
        """Rows xs | us_squash | cost | iters of every rollout written to device memory at `device_ptr` (e.g. a torch CUDA
        tensor's data_ptr()); returns the row length in doubles.  device_ptr=None only queries the length."""
        n = C.c_int()
        _check(lib().empc_solver_pack_results_device(self._h, C.c_void_p(device_ptr) if device_ptr else None, C.byref(n)))
        return n.value

    # -- per-iteration trace (the reference's callback hook: setCallbacks, src/sbfddp.cpp:303-307) -----------------
    TRACE_FIELDS = ("phase", "iter", "cost", "stop", "xreg", "steplength", "feasible", "dV", "dVexp", "gapnorm", "d0", "d1")

    def enable_trace(self, capacity=512):
        """Record {phase, iter, cost, stop, xreg, steplength, feasible, dV, dVexp, gapnorm, d0, d1} of every iteration of
        every trajectory in a device ring of `capacity` records each (0 = off); read it with ``trace(b)``."""
        _check(lib().empc_solver_enable_trace(self._h, int(capacity)))
        self._trace_cap = int(capacity)

    def trace(self, b=0):
        """Iteration records of trajectory b from the last solve, oldest first: array (n, 12), columns TRACE_FIELDS."""
        n = C.c_int()
        _check(lib().empc_solver_get_trace(self._h, int(b), None, 0, C.byref(n)))
        k = min(n.value, getattr(self, "_trace_cap", 0))
        out = np.zeros((k, len(self.TRACE_FIELDS)))
        if k:
            _check(lib().empc_solver_get_trace(self._h, int(b), _ptr(out), k, C.byref(n)))
        return out

    def stats_na(self):
        """step lengths tried per line search (SolverParams.n_alphas of this solver)"""
        return int(self._n_alphas)

    def stats(self):
        s = T.SolveStats()
        _check(lib().empc_solver_get_stats(self._h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in T.SolveStats._fields_}

    # -- phase-level kernels (parity tests, roofline) -------------------------------------------------------
    def tape_layout(self):
        l = T.TapeLayout()
        _check(lib().empc_tape_layout(self._h, C.byref(l)))
        return {n: getattr(l, n) for n, _ in T.TapeLayout._fields_}

    def tape_blocks(self, rec):
        """Split one tape record (1-D array of `rec` doubles) into its named blocks (matrices as 2-D arrays)."""
        l = self.tape_layout()
        n, m = self.ndx, self.nu

        def mat(off, ld, rows, cols):
            return np.array([rec[off + r * ld: off + r * ld + cols] for r in range(rows)])
        return {"Fx": mat(l["off_fx"], l["ld_fx"], n, n), "Fu": mat(l["off_fu"], l["ld_fu"], n, m),
                "Lxx": mat(l["off_lxx"], l["ld_lxx"], n, n), "Lxu": mat(l["off_lxu"], l["ld_lxu"], n, m),
                "Luu": mat(l["off_luu"], l["ld_luu"], m, m), "Lx": rec[l["off_lx"]:l["off_lx"] + n],
                "Lu": rec[l["off_lu"]:l["off_lu"] + m], "gap": rec[l["off_gap"]:l["off_gap"] + n],
                "cost": rec[l["off_cost"]:l["off_cost"] + 1]}

    def linearize(self, xs, us, smooth=0.1, is_feasible=False, x0s=None, fetch=True):
        B = self.batch
        if x0s is not None:
            _check(lib().empc_solver_set_x0(self._h, _ptr(np.ascontiguousarray(x0s, dtype=np.float64))))
        xs = None if xs is None else np.ascontiguousarray(xs, dtype=np.float64)
        us = None if us is None else np.ascontiguousarray(us, dtype=np.float64)
        tape = np.zeros((B, self.T + 1, self.rec)) if fetch else None
        _check(lib().empc_linearize_batch(self._h, _ptr(xs), _ptr(us), float(smooth), int(is_feasible), _ptr(tape), None, None))
        return tape

    def backward(self, xreg=1e-9, is_feasible=False):
        B, T_, n, m = self.batch, self.T, self.ndx, self.nu
        K = np.zeros((B, T_, m, n))
        k = np.zeros((B, T_, m))
        Vx = np.zeros((B, T_ + 1, n))
        dgdq = np.zeros((B, 2))
        ok = np.zeros(B, dtype=np.int32)
        _check(lib().empc_backward_batch(self._h, float(xreg), int(is_feasible), _ptr(K), _ptr(k), _ptr(Vx), _ptr(dgdq),
                                         _ptr(ok, np.int32)))
        return K, k, Vx, dgdq, ok

    def rollout(self, alpha, ddp=False, is_feasible=False):
        B, T_ = self.batch, self.T
        xs = np.zeros((B, T_ + 1, self.nx))
        us = np.zeros((B, T_, self.nu))
        cost = np.zeros(B)
        ok = np.zeros(B, dtype=np.int32)
        _check(lib().empc_rollout_batch(self._h, float(alpha), int(ddp), int(is_feasible), _ptr(xs), _ptr(us), _ptr(cost),
                                        _ptr(ok, np.int32)))
        return xs, us, cost, ok

    # -- step-wise entry points: one iteration from any iterate (teacher-forced parity) ----------------------
    def get_states(self):
        """the solver scalars of every trajectory (array of ctypes_defs.TrajState)"""
        st = (T.TrajState * self.batch)()
        _check(lib().empc_solver_get_states(self._h, st))
        return st

    def set_states(self, states):
        assert len(states) == self.batch
        _check(lib().empc_solver_set_states(self._h, states))

    def set_x0s(self, x0s):
        x0s = np.ascontiguousarray(x0s, dtype=np.float64).reshape(self.batch, self.nx)
        _check(lib().empc_solver_set_x0(self._h, _ptr(x0s)))

    def set_candidates(self, xs, us):
        xs = np.ascontiguousarray(xs, dtype=np.float64).reshape(self.batch, self.T + 1, self.nx)
        us = np.ascontiguousarray(us, dtype=np.float64).reshape(self.batch, self.T, self.nu)
        _check(lib().empc_solver_set_warmstart(self._h, _ptr(xs), _ptr(us)))

    def sweep(self, stages=T.STAGE_ALL):
        """run the named stages of ONE iteration (linearize, backward, rollout, select) over the batch from the current
        candidates and scalars"""
        _check(lib().empc_sweep_batch(self._h, int(stages)))

    def select(self, try_ok=None, try_cost=None, try_dv=None):
        """the line-search decision alone, on trial results given here (batch x n_alphas) or left by the last rollout"""
        def arr(a, dt):
            return None if a is None else np.ascontiguousarray(a, dtype=dt).reshape(self.batch, self._n_alphas)
        ok, c, dv = arr(try_ok, np.int32), arr(try_cost, np.float64), arr(try_dv, np.float64)
        _check(lib().empc_select_batch(self._h, _ptr(ok, np.int32), _ptr(c), _ptr(dv)))

    def trials(self):
        """(cost_try, dv, ok) of every step length from the last rollout, each batch x n_alphas"""
        c = np.zeros((self.batch, self._n_alphas))
        dv = np.zeros((self.batch, self._n_alphas))
        ok = np.zeros((self.batch, self._n_alphas), dtype=np.int32)
        _check(lib().empc_solver_get_trials(self._h, _ptr(c), _ptr(dv), _ptr(ok, np.int32)))
        return c, dv, ok

    def tape(self):
        t = np.zeros((self.batch, self.T + 1, self.rec))
        _check(lib().empc_solver_get_tape(self._h, _ptr(t)))
        return t

    def gains(self):
        K = np.zeros((self.batch, self.T, self.nu, self.ndx))
        k = np.zeros((self.batch, self.T, self.nu))
        Vx = np.zeros((self.batch, self.T + 1, self.ndx))
        _check(lib().empc_solver_get_gains(self._h, _ptr(K), _ptr(k), _ptr(Vx)))
        return K, k, Vx

    def set_gains(self, K=None, k=None):
        K = None if K is None else np.ascontiguousarray(K, dtype=np.float64).reshape(self.batch, self.T, self.nu, self.ndx)
        k = None if k is None else np.ascontiguousarray(k, dtype=np.float64).reshape(self.batch, self.T, self.nu)
        _check(lib().empc_solver_set_gains(self._h, _ptr(K), _ptr(k)))

    # -- streamed solves ("continuous batching") ----------------------------------------------------------------
    def stream_begin(self, x0s):
        """queue of initial states (n_jobs x nx) for solve_stream: copied to the device, result rows allocated there"""
        x0s = np.ascontiguousarray(x0s, dtype=np.float64).reshape(-1, self.nx)
        _check(lib().empc_solver_stream_begin(self._h, x0s.shape[0], _ptr(x0s)))
        self._stream_jobs = x0s.shape[0]

    def stream_run(self, maxiter=100):
        """solve([], [], maxiter) for every queued initial state through the solver's `batch` slots: a slot that finishes
        takes the next job in the same sweep, so the batch stays full until the queue is dry"""
        _check(lib().empc_solver_stream_run(self._h, int(maxiter)))

    def stream_results(self):
        """dict of per-job results: xs, us, us_squash, cost, iter, status"""
        n = C.c_int()
        _check(lib().empc_solver_stream_results(self._h, None, C.byref(n)))
        rows = np.zeros((self._stream_jobs, n.value))
        _check(lib().empc_solver_stream_results(self._h, _ptr(rows), C.byref(n)))
        T_, nx, nu = self.T, self.nx, self.nu
        nxs, nus = (T_ + 1) * nx, T_ * nu
        return dict(xs=rows[:, :nxs].reshape(-1, T_ + 1, nx).copy(), us=rows[:, nxs:nxs + nus].reshape(-1, T_, nu).copy(),
                    us_squash=rows[:, nxs + nus:nxs + 2 * nus].reshape(-1, T_, nu).copy(), cost=rows[:, nxs + 2 * nus].copy(),
                    iter=rows[:, nxs + 2 * nus + 1].astype(np.int32), status=rows[:, nxs + 2 * nus + 2].astype(np.int32))

    def stream_row_doubles(self):
        n = C.c_int()
        _check(lib().empc_solver_stream_results(self._h, None, C.byref(n)))
        return n.value

    def stream_results_device(self, device_ptr):
        """the result rows copied device to device to `device_ptr` (n_jobs x stream_row_doubles() doubles)"""
        _check(lib().empc_solver_stream_results_device(self._h, C.c_void_p(device_ptr)))

    def device_info(self):
        """(HIP device index the solver's memory lives on, its PCI bus id) -- asked of the library, not echoed from the constructor"""
        dev, bus = C.c_int(-1), C.create_string_buffer(32)
        _check(lib().empc_solver_device_info(self._h, C.byref(dev), bus, 32))
        return dev.value, bus.value.decode()

    def solve_stream(self, x0s, maxiter=100):
        self.stream_begin(x0s)
        self.stream_run(maxiter)
        return self.stream_results()

    def __del__(self):
        try:
            lib().empc_solver_destroy(self._h)
        except Exception:
            pass


class SolverBoxFDDP(SolverSbFDDP):
    """crocoddyl.SolverBoxFDDP on the same kernels (crocoddyl 1.8 core/solvers/box-fddp.cpp): the backward pass solves one
    box QP per knot for the feed-forward term, the gains live on the free subspace, the forward pass clamps the controls.
    Build the problem without squashing (``trajectory.createProblem(dt, False, ...)``), as
    src/mpc-controllers/carrot-mpc.cpp:188-193 does for this solver.  ``us_squash`` equals ``us``."""

    SOLVER_TYPE = T.SOLVER_BOXFDDP


class SolverBoxDDP(SolverSbFDDP):
    """crocoddyl.SolverBoxDDP (core/solvers/box-ddp.cpp): the box-QP backward pass with DDP's gap-free rollout."""

    SOLVER_TYPE = T.SOLVER_BOXDDP


_SOLVER_CLASSES = {T.SOLVER_SBFDDP: SolverSbFDDP, T.SOLVER_BOXFDDP: SolverBoxFDDP, T.SOLVER_BOXDDP: SolverBoxDDP}
SOLVER_NAMES = {T.SOLVER_SBFDDP: "SolverSbFDDP", T.SOLVER_BOXFDDP: "SolverBoxFDDP", T.SOLVER_BOXDDP: "SolverBoxDDP"}


class MpcProblem:
    """The controller's ShootingProblem (get_problem()); owned by the controller."""

    def __init__(self, mpc, prefix="empc_carrot_mpc"):
        self._mpc = mpc
        self._desc_fn = getattr(lib(), prefix + "_problem_desc")
        self._set_x0_fn = getattr(lib(), prefix + "_set_x0")

    @property
    def desc(self):
        p = self._desc_fn(self._mpc._h)
        if not p:
            raise EmpcError(lib().empc_last_error().decode())
        return p.contents

    @property
    def T(self):
        return self.desc.T

    @property
    def x0(self):
        d = self.desc
        return np.array([d.x0[i] for i in range(d.nx)])

    @x0.setter
    def x0(self, value):
        v = np.ascontiguousarray(value, dtype=np.float64)
        assert v.shape == (self._mpc.nx,)
        _check(self._set_x0_fn(self._mpc._h, _ptr(v)))


def _mpc_squash(self):
    """get_squash() of MpcAbstract (bindings/python/eagle_mpc/mpc-base.hpp:53): SquashingModelSmoothSat over the platform's limits"""
    pp = self.platform_params
    return SquashingModelSmoothSat(pp.u_lb, pp.u_ub, len(pp.u_lb))


def _mpc_robot_model_path(self):
    """get_robot_model_path() of MpcAbstract (bindings .../mpc-base.hpp:43): the URDF of the trajectory the controller was built
    from (RailMpc, which takes no trajectory, reads the robot from its own YAML: not recorded here -> None)"""
    tr = getattr(self, "trajectory", None)
    return tr.robot_model_path if tr is not None else None


def _mpc_create_problem(self):
    """MpcAbstract::createProblem() (bindings .../mpc-base.hpp:39): the controllers of this mirror build their problem in the
    constructor, as the reference's constructors do (src/mpc-controllers/carrot-mpc.cpp:48); returns it"""
    return self.problem



class CarrotMpc:
    """Mirror of eagle_mpc.CarrotMpc(trajectory, state_ref, dt_ref, yaml_path)
    (bindings/python/eagle_mpc/mpc-controllers/carrot-mpc.hpp, src/mpc-controllers/carrot-mpc.cpp).

    ``updateProblem(t)`` edits the cost tables on the host and, when the solver exists, uploads them;
    ``solver`` is created on first use with ``batch`` rollouts (the batched extension: one controller, B plants).
    """

    def __init__(self, trajectory, state_ref, dt_ref, yaml_path, batch=1, device=0, params=None):
        self.trajectory = trajectory
        ref = np.ascontiguousarray(np.asarray(state_ref, dtype=np.float64).reshape(-1, trajectory.nx))
        h = lib().empc_carrot_mpc_create(trajectory._h, _ptr(ref), ref.shape[0], int(dt_ref), os.fspath(yaml_path).encode())
        if not h:
            raise EmpcError(lib().empc_last_error().decode())
        self._h = C.c_void_p(h)
        v = [C.c_int() for _ in range(7)]
        _check(lib().empc_carrot_mpc_params(self._h, *[C.byref(x) for x in v]))
        self.knots, self.iters, self.dt, self.nx, self.ndx, self.nu, n_ts = [x.value for x in v]
        ts = (C.c_longlong * n_ts)()
        _check(lib().empc_carrot_mpc_t_stages(self._h, ts))
        self.t_stages = [int(x) for x in ts]
        self.problem = MpcProblem(self)
        self._batch, self._device, self._params = int(batch), int(device), params
        self._solver = None
        self.solver_type = SOLVER_NAMES[lib().empc_mpc_solver_type(self._h, None)]  # get_solver_type(), the YAML's `solver:`

    @property
    def solver(self):
        if self._solver is None:
            cls = _SOLVER_CLASSES[lib().empc_mpc_solver_type(self._h, None)]
            self._solver = cls(self.problem, batch=self._batch, device=self._device, params=self._params)
        return self._solver

    @property
    def robot_model(self):
        """get_robot_model() of MpcAbstract (include/eagle_mpc/mpc-base.hpp)"""
        m = self.problem.desc.model
        return RobotModel(self.trajectory.robot_model.name, m.nq, m.nv, self)

    @property
    def platform_params(self):
        return self.trajectory.platform_params

    def updateProblem(self, current_time):
        _check(lib().empc_carrot_mpc_update_problem(self._h, int(current_time)))
        if self._solver is not None:
            self._solver.update_problem()

    def computeStateReference(self, time):
        x = np.zeros(self.nx)
        _check(lib().empc_carrot_mpc_state_reference(self._h, int(time), _ptr(x)))
        return x

    def __del__(self):
        try:
            lib().empc_carrot_mpc_destroy(self._h)
        except Exception:
            pass


class _Mpc:
    """Common part of RailMpc / WeightedMpc (MpcAbstract, include/eagle_mpc/mpc-base.hpp:59-119)."""

    def _init_common(self, h, batch, device, params):
        if not h:
            raise EmpcError(lib().empc_last_error().decode())
        self._h = C.c_void_p(h)
        v = [C.c_int() for _ in range(6)]
        _check(lib().empc_mpc_params(self._h, *[C.byref(x) for x in v]))
        self.knots, self.iters, self.dt, self.nx, self.ndx, self.nu = [x.value for x in v]
        self.problem = MpcProblem(self, prefix="empc_mpc")
        self._batch, self._device, self._params = int(batch), int(device), params
        self._solver = None
        self.solver_type = SOLVER_NAMES[lib().empc_mpc_solver_type(None, self._h)]

    @property
    def solver(self):
        if self._solver is None:
            cls = _SOLVER_CLASSES[lib().empc_mpc_solver_type(None, self._h)]
            self._solver = cls(self.problem, batch=self._batch, device=self._device, params=self._params)
        return self._solver

    @property
    def robot_model(self):
        m = self.problem.desc.model
        tr = getattr(self, "trajectory", None)
        return RobotModel(tr.robot_model.name if tr is not None else "robot", m.nq, m.nv, self)

    @property
    def platform_params(self):
        d = self.problem.desc
        tau_f = np.array([[d.tau_f[r * d.n_rotors + c] for c in range(d.n_rotors)] for r in range(6)])
        return PlatformParams(tau_f, np.array([d.u_lb[i] for i in range(d.nu)]), np.array([d.u_ub[i] for i in range(d.nu)]))

    def updateProblem(self, current_time):
        _check(lib().empc_mpc_update_problem(self._h, int(current_time)))
        if self._solver is not None:
            self._solver.update_problem()

    def __del__(self):
        try:
            lib().empc_mpc_destroy(self._h)
        except Exception:
            pass


class RailMpc(_Mpc):
    """Mirror of eagle_mpc.RailMpc(state_ref, dt_ref, yaml_path) (src/mpc-controllers/rail-mpc.cpp): every knot
    tracks the planned state at its own time."""

    def __init__(self, state_ref, dt_ref, yaml_path, batch=1, device=0, params=None):
        ref = np.ascontiguousarray(np.asarray(state_ref, dtype=np.float64))
        if ref.ndim != 2:
            raise ValueError("state_ref must be (n_ref, nx)")
        self._init_common(lib().empc_rail_mpc_create(_ptr(ref), ref.shape[0], ref.shape[1], int(dt_ref),
                                                     os.fspath(yaml_path).encode()), batch, device, params)

    def computeStateReference(self, time):
        x = np.zeros(self.nx)
        _check(lib().empc_rail_mpc_state_reference(self._h, int(time), _ptr(x)))
        return x


class WeightedMpc(_Mpc):
    """Mirror of eagle_mpc.WeightedMpc(trajectory, dt_ref, yaml_path) (src/mpc-controllers/weighted-mpc.cpp).
    Like the reference, construction merges every transition stage of ``trajectory`` into the stage after it."""

    def __init__(self, trajectory, dt_ref, yaml_path, batch=1, device=0, params=None):
        self.trajectory = trajectory
        self._init_common(lib().empc_weighted_mpc_create(trajectory._h, int(dt_ref), os.fspath(yaml_path).encode()),
                          batch, device, params)
        n = lib().empc_weighted_mpc_t_stages(self._h, None, 0)
        ts = (C.c_longlong * max(n, 1))()
        _check(min(lib().empc_weighted_mpc_t_stages(self._h, ts, n), 0))
        self.t_stages = [int(ts[i]) for i in range(n)]


def perturbed_x0s(x0, batch, nq, seed=0, amplitude=0.05, joint_lb=None, joint_ub=None):
    """Batch of initial states: element 0 is x0, elements b >= 1 follow the reference's perturbation recipe
    x0 += 0.05 U(-1,1), quaternion renormalised (benchmark/utils/utils.hpp:15-27)."""
    x0 = np.asarray(x0, dtype=np.float64)
    out = np.tile(x0, (batch, 1))
    for b in range(1, batch):
        rng = np.random.default_rng(seed * 1000003 + b)
        out[b] += amplitude * rng.uniform(-1, 1, size=x0.shape)
        out[b, 3:7] /= np.linalg.norm(out[b, 3:7])
    return out


for _cls in (CarrotMpc, _Mpc):
    _cls.squash = property(_mpc_squash)
    _cls.robot_model_path = property(_mpc_robot_model_path)
    _cls.createProblem = _mpc_create_problem
