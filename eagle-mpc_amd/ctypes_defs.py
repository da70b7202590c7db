"""ctypes mirrors of include/empc_types.h and include/empc.h (plain data only)."""
import ctypes as C

MAX_BODIES = 8
MAX_NV = 14
MAX_NQ = 15
MAX_NX = 29
MAX_NDX = 28
MAX_ROTORS = 8
MAX_NU = 16
MAX_NR = 28
MAX_FRAMES = 8
MAX_COSTS = 20
MAX_CONTACTS = 2
NAME_LEN = 40

d = C.c_double
i32 = C.c_int32


class ModelDesc(C.Structure):
    _fields_ = [
        ("nbodies", i32), ("nq", i32), ("nv", i32), ("nframes", i32),
        ("parent", i32 * MAX_BODIES),
        ("jplace_R", (d * 9) * MAX_BODIES),
        ("jplace_p", (d * 3) * MAX_BODIES),
        ("axis", (d * 3) * MAX_BODIES),
        ("mass", d * MAX_BODIES),
        ("com", (d * 3) * MAX_BODIES),
        ("inertia", (d * 9) * MAX_BODIES),
        ("effort_limit", d * MAX_BODIES),
        ("frame_body", i32 * MAX_FRAMES),
        ("frame_R", (d * 9) * MAX_FRAMES),
        ("frame_p", (d * 3) * MAX_FRAMES),
        ("frame_name", (C.c_char * NAME_LEN) * MAX_FRAMES),
        ("gravity", d * 3),
    ]


class Cost(C.Structure):
    _fields_ = [
        ("name", C.c_char * NAME_LEN),
        ("type", i32), ("activation", i32), ("active", i32), ("frame", i32), ("nr", i32), ("is_barrier", i32),
        ("ref_share", i32), ("reserved", i32),
        ("weight", d),
        ("ref", d * MAX_NX),
        ("act_w", d * MAX_NR),
        ("lb", d * MAX_NR),
        ("ub", d * MAX_NR),
    ]


class Contact(C.Structure):
    _fields_ = [
        ("name", C.c_char * NAME_LEN),
        ("type", i32), ("frame", i32),
        ("ref_p", d * 3), ("ref_R", d * 9), ("gains", d * 2),
    ]


class CostSet(C.Structure):
    _fields_ = [
        ("ncosts", i32), ("ncontacts", i32),
        ("costs", Cost * MAX_COSTS),
        ("contacts", Contact * MAX_CONTACTS),
    ]


class ProblemDesc(C.Structure):
    _fields_ = [
        ("model", ModelDesc),
        ("nx", i32), ("ndx", i32), ("nu", i32), ("n_rotors", i32), ("T", i32), ("n_sets", i32),
        ("has_contact", i32), ("use_squash", i32), ("integrator", i32), ("reserved", i32),
        ("dt", d),
        ("tau_f", d * (6 * MAX_ROTORS)),
        ("u_lb", d * MAX_NU),
        ("u_ub", d * MAX_NU),
        ("x0", d * MAX_NX),
        ("sets", C.POINTER(CostSet)),
        ("knot_set", C.POINTER(i32)),
    ]


class SolverParams(C.Structure):
    _fields_ = [
        ("smooth_init", d), ("smooth_mult", d), ("barrier_weight", d),
        ("convergence_init", d), ("convergence_stop", d), ("convergence_mult", d),
        ("reg_init", d), ("th_acceptnegstep", d), ("th_stop_gaps", d),
        ("th_grad", d), ("th_acceptstep", d), ("th_stepdec", d), ("th_stepinc", d),
        ("reg_incfactor", d), ("reg_decfactor", d), ("reg_min", d), ("reg_max", d),
        ("th_gaptol", d),
        ("n_alphas", i32), ("stop_criteria", i32), ("gap_norm", i32), ("terminal_dt_scaling", i32),
        ("smoothsat_power", i32), ("solver_type", i32),
        ("box_th_stop", d), ("boxqp_th_acceptstep", d), ("boxqp_th_grad", d), ("boxqp_reg", d),
        ("boxqp_maxiter", i32), ("reserved", i32),
    ]


class SolveStats(C.Structure):
    _fields_ = [
        ("sweeps", C.c_int), ("max_iters", C.c_int),
        ("total_iters", C.c_longlong), ("linearize_units", C.c_longlong), ("rollout_units", C.c_longlong),
        ("backward_units", C.c_longlong),
        ("ms_total", d), ("ms_linearize", d), ("ms_backward", d), ("ms_rollout", d), ("ms_select", d), ("ms_calc", d),
        ("n_linearize", C.c_int), ("n_backward", C.c_int), ("n_rollout", C.c_int), ("n_select", C.c_int),
        ("n_calc", C.c_int), ("timing_every", C.c_int),
        ("linearize_units_all", C.c_longlong), ("rollout_units_all", C.c_longlong), ("backward_units_all", C.c_longlong),
    ]


class TrajState(C.Structure):
    """EmpcTrajState (include/empc_types.h): the solver scalars of one trajectory"""
    _fields_ = [(n, i32) for n in
                ("phase", "iter", "total_iters", "status", "is_feasible", "was_feasible", "need_calc", "need_lin", "maxiter",
                 "bwd_failed", "trace_count", "last_ok", "accepted_alpha", "last_alpha", "job", "reserved")] + \
               [(n, d) for n in
                ("smooth", "smooth_next", "convergence", "th_stop", "xreg", "ureg", "cost", "cost_prev", "stop", "steplength",
                 "dV", "dVexp", "d0", "d1", "dg_u", "dq_u", "dg_f", "dq_f", "gapnorm", "qu2")]


STAGE_LINEARIZE, STAGE_BACKWARD, STAGE_ROLLOUT, STAGE_SELECT, STAGE_ALL = 1, 2, 4, 8, 15
PHASE_DDP, PHASE_DONE = 100, 255


class TapeLayout(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("rec", "off_fx", "off_fu", "off_lxx", "off_lxu", "off_luu", "off_lx", "off_lu", "off_gap", "off_cost",
                 "ld_fx", "ld_fu", "ld_lxx", "ld_lxu", "ld_luu")]


SOLVER_SBFDDP, SOLVER_BOXFDDP, SOLVER_BOXDDP = 0, 1, 2
STATUS_CONVERGED = 1
STATUS_REG_MAX = 2
STATUS_MAXITER = 4
STATUS_DDP_CLEANUP = 8

COST_STATE, COST_CONTROL, COST_FRAME_PLACEMENT, COST_FRAME_ROTATION, COST_FRAME_VELOCITY, COST_FRAME_TRANSLATION, \
    COST_CONTACT_FRICTION_CONE = range(7)
ACT_QUAD, ACT_WEIGHTED_QUAD, ACT_QUADRATIC_BARRIER, ACT_WEIGHTED_QUADRATIC_BARRIER = range(4)
CONTACT_3D, CONTACT_6D = 0, 1  # EmpcContactType
