// empc_solver.hip -- host driver + solver part of the C ABI (include/empc.h); the kernels live in empc_launch.hpp and are
// instantiated per robot class in empc_inst_*.hip.
//
// One solve = a sequence of "sweeps"; in every sweep each still-active trajectory advances by exactly one DDP
// iteration of its current pass (FDDP pass, or the DDP clean-up):
//     calc (phase starts only) -> linearize -> backward -> rollout (all step lengths at once) -> select
// All per-trajectory solver state lives on the device (TrajState); the host only reads back the number of
// trajectories that are still active after each sweep.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/empc.h"
#include "empc_launch.hpp"

using namespace empc;

#define HIP_CHECK(expr)                                                                                   \
  do {                                                                                                    \
    hipError_t _e = (expr);                                                                               \
    if (_e != hipSuccess)                                                                                 \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr);        \
  } while (0)

// --------------------------------------------------------------------------------------------------------------------
// solver object
// --------------------------------------------------------------------------------------------------------------------
// contact_rows: 0 for problems built on the free dynamics; 3 / 6 for Contact dynamics (DifferentialActionModelContactFwdDynamics
// on every node) whose contact stages hold a ContactModel3D / ContactModel6D
// Does the problem's robot equal a baked table (tree + platform) bit for bit?  Only then may the instantiation over the
// baked constants stand in for the runtime-model one.
static bool baked_matches(const empc::BakedTree& t, const DevProblem& P) {
  const EmpcModelDesc& m = P.model;
  if (m.nbodies != t.nbodies || m.nq != t.nq || m.nv != t.nv || P.n_rotors != t.n_rotors) return false;
  auto same = [](const double* a, const double* b, int n) { return std::memcmp(a, b, sizeof(double) * n) == 0; };
  for (int b = 0; b < m.nbodies; ++b) {
    if (m.parent[b] != t.parent[b]) return false;
    if (!same(m.jplace_R[b], t.jplace_R[b], 9) || !same(m.jplace_p[b], t.jplace_p[b], 3) || !same(m.axis[b], t.axis[b], 3) ||
        !same(&m.mass[b], &t.mass[b], 1) || !same(m.com[b], t.com[b], 3) || !same(m.inertia[b], t.inertia[b], 9))
      return false;
  }
  return same(m.gravity, t.gravity, 3) && same(P.tau_f, t.tau_f, 6 * P.n_rotors) && same(P.u_lb, t.u_lb, P.nu) &&
         same(P.u_ub, t.u_ub, P.nu);
}
// EMPC_BAKED=0 keeps every problem on the runtime-model instantiations (tests compare the two)
static bool baked_enabled() {
  const char* e = std::getenv("EMPC_BAKED");
  return !(e && std::atoi(e) == 0);
}
static bool experimental_contact() {
  const char* e = std::getenv("EMPC_EXPERIMENTAL_CONTACT");
  return e && e[0] && e[0] != '0';
}
static thread_local const char* find_table_reason = nullptr;  // why the last find_table refused (nullptr: no instantiation exists)
static const char* no_table_message() {
  return find_table_reason ? find_table_reason : "no kernel instantiation for this (bodies, rotors, contact) combination";
}
static bool find_table(const DevProblem& P, int nb, int nrot, bool contact, int contact_rows, KernelTable& k, const char** which = nullptr) {
  if (which) *which = "runtime model";
  find_table_reason = nullptr;
  if (baked_enabled() && !(contact && nb == 4 && std::getenv("EMPC_FORCE_MIXED_CONTACT"))) {
    // the robots the library carries as compile-time tables (tools/bake_models.py), each with the dynamics it is instantiated for
    struct Baked {
      const empc::BakedTree* tree;
      KernelTable (*free_dynamics)();
      KernelTable (*contact3)();
      const char* free_name;
      const char* contact3_name;
    };
    static const Baked baked[] = {
        {&empc::kBakedHex370Arm3, empc_table_baked_arm3, empc_table_baked_arm3_contact, "baked hexacopter370_flying_arm_3",
         "baked hexacopter370_flying_arm_3, ContactModel3D"},
        {&empc::kBakedHextiltArm5, empc_table_baked_arm5, nullptr, "baked hextilt_flying_arm_5", nullptr},
        {&empc::kBakedHex680Arm2, empc_table_baked_arm2, nullptr, "baked hexacopter680_flying_arm_2", nullptr},
        {&empc::kBakedHex370, empc_table_baked_hex370, nullptr, "baked hexacopter370", nullptr},
        {&empc::kBakedHextilt, empc_table_baked_hextilt, nullptr, "baked hextilt", nullptr},
        {&empc::kBakedIris, empc_table_baked_iris, nullptr, "baked iris", nullptr},
        {&empc::kBakedIrisPx4, empc_table_baked_iris_px4, nullptr, "baked iris_px4", nullptr},
    };
    for (const Baked& e : baked) {
      if (!baked_matches(*e.tree, P)) continue;
      if (!contact && e.free_dynamics) {
        k = e.free_dynamics();
        if (which) *which = e.free_name;
        return true;
      }
      if (contact && contact_rows == 3 && e.contact3) {
        k = e.contact3();
        if (which) *which = e.contact3_name;
        return true;
      }
    }
  }
  // diagnostic: run a single-type contact problem through the mixed instantiation (tests: bitwise the same results)
  if (contact && nb == 4 && std::getenv("EMPC_FORCE_MIXED_CONTACT")) contact_rows = empc::CT_MIXED;
  if (nb == 1 && nrot == 4 && !contact) k = empc_table_1_4();
  else if (nb == 1 && nrot == 6 && !contact) k = empc_table_1_6();
  else if (nb == 3 && nrot == 6 && !contact) k = empc_table_3_6();
  // contact dynamics on the single-body and two-joint classes: one instantiation per class serves ContactModel3D, ContactModel6D
  // and problems mixing them (the branch on the node's contact type costs nothing next to the dynamics; no shipped file is here).
  // Opt-in (EMPC_EXPERIMENTAL_CONTACT=1) until tests/test_zz_gpu_contact_small_classes.py has passed on hardware: these three
  // instantiations have only run on the CPU lane emulator; without the switch the class is refused with that reason.
  else if ((nb == 1 || nb == 3) && contact && !experimental_contact()) {
    find_table_reason = "contact dynamics on the (1,4) / (1,6) / (3,6) robot classes: kernels not yet verified on hardware; "
                        "set EMPC_EXPERIMENTAL_CONTACT=1 to run them";
    return false;
  }
  else if (nb == 1 && nrot == 4 && contact) k = empc_table_1_4_contact();
  else if (nb == 1 && nrot == 6 && contact) k = empc_table_1_6_contact();
  else if (nb == 3 && nrot == 6 && contact) k = empc_table_3_6_contact();
  else if (nb == 4 && nrot == 6 && !contact) k = empc_table_4_6();
  // a stage with TWO ContactModel3D contacts (ContactModelMultiple, src/stage.cpp:38-48): the six-row instantiation of the
  // (4,6) and (6,6) classes.  Opt-in like the classes above: these kernels have only run on the CPU lane emulator.
  else if (contact && contact_rows == empc::CT_PAIR3 && !((nb == 4 || nb == 6) && nrot == 6)) {
    find_table_reason = "two contacts per stage: kernels exist for the (4,6) and (6,6) robot classes only";
    return false;
  }
  else if (contact && contact_rows == empc::CT_PAIR3 && !experimental_contact()) {
    find_table_reason = "two contacts per stage (two ContactModel3D, six constraint rows): kernels not yet verified on hardware; "
                        "set EMPC_EXPERIMENTAL_CONTACT=1 to run them";
    return false;
  }
  else if (nb == 4 && nrot == 6 && contact && contact_rows == empc::CT_PAIR3) k = empc_table_4_6_contact_pair();
  else if (nb == 6 && nrot == 6 && contact && contact_rows == empc::CT_PAIR3) k = empc_table_6_6_contact_pair();
  else if (nb == 4 && nrot == 6 && contact && contact_rows == empc::CT_MIXED) k = empc_table_4_6_contact_mixed();
  else if (nb == 4 && nrot == 6 && contact && contact_rows != 6) k = empc_table_4_6_contact();
  else if (nb == 4 && nrot == 6 && contact && contact_rows == 6) k = empc_table_4_6_contact6();
  else if (nb == 6 && nrot == 6 && !contact) k = empc_table_6_6();
  else if (nb == 6 && nrot == 6 && contact && contact_rows != 6 && contact_rows != empc::CT_MIXED) k = empc_table_6_6_contact();
  else if (nb == 6 && nrot == 6 && contact && contact_rows == 6) k = empc_table_6_6_contact6();
  else if (nb == 6 && nrot == 6 && contact && contact_rows == empc::CT_MIXED && !experimental_contact()) {
    find_table_reason = "stages of both contact types on the (6,6) robot class: kernels not yet verified on hardware; "
                        "set EMPC_EXPERIMENTAL_CONTACT=1 to run them";
    return false;
  }
  else if (nb == 6 && nrot == 6 && contact && contact_rows == empc::CT_MIXED) k = empc_table_6_6_contact_mixed();
  else return false;
  return true;
}
// problem classes the factory accepts but no kernel implements yet
static void check_device_support(const EmpcProblemDesc& d) {
  (void)d;  // every problem class prepare_problem lets through has kernels today
}

struct EmpcSolver {
  HostProblem H;
  KernelTable kt;
  const char* kernel_family = "runtime model";  // which instantiation find_table picked (empc_solver_kernel_family)
  int device = 0, B = 0, T = 0, NA = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[8] = {};
  static constexpr int MAX_STREAMS = 16;
  int n_streams = 1;
  hipStream_t streams[MAX_STREAMS] = {};
  hipEvent_t cev[MAX_STREAMS][2][7] = {};  // per chunk, per in-flight sweep slot: kernel boundaries + 'results on the host'
  DevBuffers D;
  DevProblem* dP = nullptr;
  EmpcCostSet* dsets = nullptr;
  SetInfo* dset_info = nullptr;
  int* dknot = nullptr;
  int* dlin_knots = nullptr;
  int* dlin_list = nullptr;  // [2][B] linearize lists of the two sweep slots
  int* dact_list = nullptr;  // [2][B] lists of the trajectories still iterating, same slots
  int* dcalc_list = nullptr; // [2][B] lists of the trajectories that start a pass (need_calc), same slots
  double* dscratch = nullptr;  // output staging (squashed controls)
  double* dplant_x = nullptr;  // [B][NX] plant states of closed-loop runs (empc_plant_*)
  double* dplant_u = nullptr;  // [B][NU] staging of caller-supplied plant controls
  Rk4Buffers R4 = {};          // stage batch of IntegratedActionModelRK4 problems (empty otherwise)
  double* dtrace = nullptr;    // [B][trace_cap][EMPC_TRACE_WORDS] iteration records (empc_solver_enable_trace)
  int trace_cap = 0;
  hipEvent_t t_begin = nullptr, t_end = nullptr;  // brackets of one solve
  int* h_active = nullptr;     // pinned, written by the select kernel itself (no copy command in the stream)
  int* h_active_dev = nullptr; // the same memory as the device sees it
  int* dticket = nullptr;      // [MAX_STREAMS] completion tickets of select
  std::vector<TrajState> h_st;
  bool have_state = false;
  // the problem as the caller described it (before barrier injection and the other preparations), kept so that single
  // cost entries can be edited in place (empc_solver_set_cost_refs); re-prepared and uploaded lazily before the next launch
  EmpcProblemDesc user_desc;
  std::vector<EmpcCostSet> user_sets;
  std::vector<int32_t> user_knot_set;
  bool problem_dirty = false;
  EmpcSolveStats stats;
  // streamed solves (empc_solver_stream_*): the queue of initial states and the result rows, resident on the device
  double* dq_x0 = nullptr;
  double* dq_rows = nullptr;
  int* dq_head = nullptr;                  // [0] next job; the summed iterations live behind it (8-byte aligned)
  unsigned long long* dq_iters = nullptr;
  int q_njobs = 0;
  std::vector<void*> allocs;

  template <class Tt>
  Tt* dalloc(size_t n) {
    void* p = nullptr;
    HIP_CHECK(hipMalloc(&p, n * sizeof(Tt)));
    allocs.push_back(p);
    return static_cast<Tt*>(p);
  }
  void use() { HIP_CHECK(hipSetDevice(device)); }
  ~EmpcSolver() {
    if (hipSetDevice(device) != hipSuccess) return;
    for (void* p : allocs) (void)hipFree(p);
    if (h_active) (void)hipHostFree(h_active);
    if (dtrace) (void)hipFree(dtrace);
    if (dq_x0) (void)hipFree(dq_x0);
    if (dq_rows) (void)hipFree(dq_rows);
    if (dq_head) (void)hipFree(dq_head);
    if (t_begin) (void)hipEventDestroy(t_begin);
    if (t_end) (void)hipEventDestroy(t_end);
    for (auto& e : ev)
      if (e) (void)hipEventDestroy(e);
    for (int c = 0; c < MAX_STREAMS; ++c) {
      for (auto& slot : cev[c])
        for (auto& e : slot)
          if (e) (void)hipEventDestroy(e);
      if (streams[c]) (void)hipStreamDestroy(streams[c]);
    }
    if (stream) (void)hipStreamDestroy(stream);
  }
};

static void upload_problem(EmpcSolver* s) {
  HIP_CHECK(hipMemcpyAsync(s->dP, &s->H.P, sizeof(DevProblem), hipMemcpyHostToDevice, s->stream));
  HIP_CHECK(hipMemcpyAsync(s->dsets, s->H.sets.data(), sizeof(EmpcCostSet) * s->H.sets.size(), hipMemcpyHostToDevice, s->stream));
  HIP_CHECK(hipMemcpyAsync(s->dset_info, s->H.set_info.data(), sizeof(SetInfo) * s->H.set_info.size(), hipMemcpyHostToDevice, s->stream));
  HIP_CHECK(hipMemcpyAsync(s->dknot, s->H.knot_set.data(), sizeof(int) * s->H.knot_set.size(), hipMemcpyHostToDevice, s->stream));
  {
    std::vector<int> order;
    s->D.n_lean = group_linearize_knots(s->H, order);
    HIP_CHECK(hipMemcpyAsync(s->dlin_knots, order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice, s->stream));
    HIP_CHECK(hipStreamSynchronize(s->stream));  // `order` is a local
  }
  HIP_CHECK(hipStreamSynchronize(s->stream));
}

static void remember_problem(EmpcSolver* s, const EmpcProblemDesc& d) {
  s->user_desc = d;
  s->user_sets.assign(d.sets, d.sets + d.n_sets);
  s->user_knot_set.assign(d.knot_set, d.knot_set + d.T + 1);
  s->user_desc.sets = s->user_sets.data();
  s->user_desc.knot_set = s->user_knot_set.data();
  s->problem_dirty = false;
}
// pending edits of empc_solver_set_cost_refs reach the device here: before anything that launches a kernel
static void flush_problem(EmpcSolver* s) {
  if (!s->problem_dirty) return;
  const EmpcSolverParams prm = s->H.P.prm;
  HostProblem N;
  prepare_problem(s->user_desc, prm, N);
  if (N.contact_rows != s->H.contact_rows || N.sets.size() != s->H.sets.size())
    throw std::invalid_argument("set_cost_refs: the edited problem no longer matches the solver's kernel class");
  s->H = std::move(N);
  upload_problem(s);
  s->problem_dirty = false;
}

static void upload_states(EmpcSolver* s) {
  HIP_CHECK(hipMemcpyAsync(s->D.st, s->h_st.data(), sizeof(TrajState) * s->B, hipMemcpyHostToDevice, s->stream));
}
static void download_states(EmpcSolver* s) {
  HIP_CHECK(hipMemcpyAsync(s->h_st.data(), s->D.st, sizeof(TrajState) * s->B, hipMemcpyDeviceToHost, s->stream));
  HIP_CHECK(hipStreamSynchronize(s->stream));
}

#define EMPC_TRY try {
#define EMPC_CATCH(ret)                          \
  }                                              \
  catch (const std::invalid_argument& e) {       \
    empc::set_last_error(e.what());              \
    return ret(EMPC_ERR_INVALID);                \
  }                                              \
  catch (const std::exception& e) {              \
    empc::set_last_error(e.what());              \
    return ret(EMPC_ERR_RUNTIME);                \
  }                                              \
  catch (...) {                                  \
    empc::set_last_error("unknown exception");   \
    return ret(EMPC_ERR_RUNTIME);                \
  }
#define RET_INT(x) (x)
#define RET_NULL(x) nullptr

extern "C" {

int empc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int empc_solver_supported(const EmpcProblemDesc* problem, const EmpcSolverParams* params) {
  try {
    if (!problem) throw std::invalid_argument("problem is NULL");
    EmpcSolverParams prm;
    if (params)
      prm = *params;
    else
      empc_solver_params_default(&prm);
    HostProblem H;
    prepare_problem(*problem, prm, H);  // dimension / chain / integrator / contact-type limits of the kernels
    check_device_support(*problem);
    KernelTable kt;
    if (!find_table(H.P, problem->model.nbodies, problem->n_rotors, problem->has_contact != 0, H.contact_rows, kt))
      throw std::runtime_error(no_table_message());
    return 1;
  } catch (const std::exception& e) {
    empc::set_last_error(e.what());
    return 0;
  }
}

EmpcSolver* empc_solver_create(const EmpcProblemDesc* problem, const EmpcSolverParams* params, int batch, int device) {
  EmpcSolver* s = nullptr;
  EMPC_TRY
  if (!problem) throw std::invalid_argument("problem is NULL");
  if (batch < 1) throw std::invalid_argument("batch must be >= 1");
  EmpcSolverParams prm;
  if (params)
    prm = *params;
  else
    empc_solver_params_default(&prm);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    throw std::runtime_error("no HIP device available: the batched SbFDDP solver has no CPU fallback");
  if (device < 0 || device >= ndev) throw std::invalid_argument("device index out of range");
  s = new EmpcSolver();
  s->device = device;
  prepare_problem(*problem, prm, s->H);
  remember_problem(s, *problem);
  check_device_support(*problem);
  if (!find_table(s->H.P, problem->model.nbodies, problem->n_rotors, problem->has_contact != 0, s->H.contact_rows, s->kt, &s->kernel_family)) {
    delete s;
    empc::set_last_error(no_table_message());
    return nullptr;
  }
  s->use();
  s->B = batch;
  s->T = problem->T;
  s->NA = prm.n_alphas;
  HIP_CHECK(hipStreamCreate(&s->stream));
  for (auto& e : s->ev) HIP_CHECK(hipEventCreate(&e));
  HIP_CHECK(hipEventCreate(&s->t_begin));
  HIP_CHECK(hipEventCreate(&s->t_end));
  HIP_CHECK(hipHostMalloc((void**)&s->h_active, sizeof(int) * 4 * EmpcSolver::MAX_STREAMS, hipHostMallocMapped));  // per chunk and sweep slot: {active, re-linearizing}
  HIP_CHECK(hipHostGetDevicePointer((void**)&s->h_active_dev, s->h_active, 0));
  {
    // independent chunks of the batch run on separate streams (EMPC_STREAMS overrides; 1 = single stream)
    int ns = 1;  // measured on MI355X (profiles/r01_streams.txt): lock-stepped chunks do not overlap usefully
    if (const char* e = std::getenv("EMPC_STREAMS")) ns = std::atoi(e);
    if (ns < 1) ns = 1;
    if (ns > EmpcSolver::MAX_STREAMS) ns = EmpcSolver::MAX_STREAMS;
    s->n_streams = ns;
    for (int c = 0; c < ns; ++c) {
      HIP_CHECK(hipStreamCreate(&s->streams[c]));
      for (auto& slot : s->cev[c])
        for (auto& e : slot) HIP_CHECK(hipEventCreate(&e));
    }
  }
  const size_t B = batch, T = s->T, NA = s->NA;
  const KernelTable& k = s->kt;
  s->dP = s->dalloc<DevProblem>(1);
  s->dsets = s->dalloc<EmpcCostSet>(s->H.sets.size());
  s->dset_info = s->dalloc<SetInfo>(s->H.sets.size());
  s->dknot = s->dalloc<int>(T + 1);
  DevBuffers& D = s->D;
  D.P = s->dP;
  D.sets = s->dsets;
  D.set_info = s->dset_info;
  D.knot_set = s->dknot;
  s->dlin_knots = s->dalloc<int>(T + 1);
  D.lin_knots = s->dlin_knots;
  D.n_lean = T + 1;
  D.st = s->dalloc<TrajState>(B);
  D.x0 = s->dalloc<double>(B * k.nx);
  D.xs = s->dalloc<double>(B * (T + 1) * k.nx);
  D.us = s->dalloc<double>(B * T * k.nu);
  D.acc = s->dalloc<double>(B * (T + 1) * k.nacc);
  D.tape = s->dalloc<double>(B * (T + 1) * k.rec + 128);  // + slack: the backward pass prefetches whole 128-double rows
  D.K = s->dalloc<double>(B * T * k.nu * k.ndx);
  D.kff = s->dalloc<double>(B * T * k.nu);
  D.Vx = s->dalloc<double>(B * (T + 1) * k.ndx);
  D.Vf = s->dalloc<double>(B * (T + 1) * k.ndx);
  D.xs_try = s->dalloc<double>(B * NA * (T + 1) * k.nx);
  D.us_try = s->dalloc<double>(B * NA * T * k.nu);
  D.acc_try = s->dalloc<double>(B * NA * (T + 1) * k.nacc);
  D.try_cost = s->dalloc<double>(B * NA);
  D.try_dv = s->dalloc<double>(B * NA);
  D.try_ok = s->dalloc<int>(B * NA);
  D.try_ncalc = s->dalloc<int>(B * NA);
  D.us_last = s->dalloc<double>(B * T * k.nu);
  D.n_active = s->dalloc<int>(6 * EmpcSolver::MAX_STREAMS);  // per chunk and sweep slot: {active trajectories, entries of the linearize list, of the calc list}
  s->dticket = s->dalloc<int>(EmpcSolver::MAX_STREAMS);
  HIP_CHECK(hipMemsetAsync(s->dticket, 0, sizeof(int) * EmpcSolver::MAX_STREAMS, s->stream));
  s->dlin_list = s->dalloc<int>(2 * (size_t)batch);          // per sweep slot: the linearize list of every chunk, chunk after chunk
  s->dact_list = s->dalloc<int>(2 * (size_t)batch);
  s->dcalc_list = s->dalloc<int>(2 * (size_t)batch);
  D.dbg = s->dalloc<unsigned long long>(128);
  HIP_CHECK(hipMemsetAsync(D.dbg, 0, 128 * sizeof(unsigned long long), s->stream));
  D.B = batch;
  D.T = s->T;
  D.NA = s->NA;
  D.gaptol = std::max(prm.th_gaptol, 1e-13);
  D.integrator = problem->integrator;
  D.solver_type = prm.solver_type;
  if (prm.solver_type < EMPC_SOLVER_SBFDDP || prm.solver_type > EMPC_SOLVER_BOXDDP) throw std::invalid_argument("unknown solver_type");
  HIP_CHECK(hipMemsetAsync(D.kff, 0, sizeof(double) * B * T * k.nu, s->stream));
  HIP_CHECK(hipMemsetAsync(D.K, 0, sizeof(double) * B * T * k.nu * k.ndx, s->stream));
  if (problem->integrator == EMPC_INTEGRATOR_RK4) {
    s->R4.ys = s->dalloc<double>(4 * B * (T + 1) * k.nx);
    s->R4.accs = s->dalloc<double>(4 * B * (T + 1) * k.nacc);
    s->R4.us4 = s->dalloc<double>(4 * B * T * k.nu);
    s->R4.tape4 = s->dalloc<double>(4 * B * (T + 1) * k.rec + 128);
    s->R4.st4 = s->dalloc<TrajState>(4 * B);
    HIP_CHECK(hipMemsetAsync(s->R4.tape4, 0, sizeof(double) * 4 * B * (T + 1) * k.rec, s->stream));
    HIP_CHECK(hipMemsetAsync(s->R4.accs, 0, sizeof(double) * 4 * B * (T + 1) * k.nacc, s->stream));
    HIP_CHECK(hipMemsetAsync(s->R4.ys, 0, sizeof(double) * 4 * B * (T + 1) * k.nx, s->stream));
    HIP_CHECK(hipMemsetAsync(s->R4.us4, 0, sizeof(double) * 4 * B * T * k.nu, s->stream));
  }
  s->dscratch = s->dalloc<double>(B * T * k.nu);
  s->dplant_x = s->dalloc<double>(B * k.nx);
  s->dplant_u = s->dalloc<double>(B * k.nu);
  HIP_CHECK(hipMemsetAsync(D.tape, 0, sizeof(double) * B * (T + 1) * k.rec, s->stream));
  HIP_CHECK(hipMemsetAsync(D.us_last, 0, sizeof(double) * B * T * k.nu, s->stream));
  HIP_CHECK(hipMemsetAsync(D.acc, 0, sizeof(double) * B * (T + 1) * k.nacc, s->stream));
  upload_problem(s);
  s->h_st.assign(B, TrajState());
  std::memset(s->h_st.data(), 0, sizeof(TrajState) * B);
  std::memset(&s->stats, 0, sizeof(s->stats));
  if (empc_solver_set_x0(s, nullptr) != EMPC_OK || empc_solver_set_warmstart(s, nullptr, nullptr) != EMPC_OK)
    throw std::runtime_error(empc_last_error());
  return s;
  }
  catch (const std::exception& e) {
    empc::set_last_error(e.what());
    delete s;
    return nullptr;
  }
}

void empc_solver_destroy(EmpcSolver* s) { delete s; }

int empc_solver_dims(const EmpcSolver* s, int* batch, int* T, int* nx, int* ndx, int* nu, int* rec_doubles) {
  if (!s) return EMPC_ERR_INVALID;
  if (batch) *batch = s->B;
  if (T) *T = s->T;
  if (nx) *nx = s->kt.nx;
  if (ndx) *ndx = s->kt.ndx;
  if (nu) *nu = s->kt.nu;
  if (rec_doubles) *rec_doubles = s->kt.rec_full;
  return EMPC_OK;
}
int empc_tape_layout(const EmpcSolver* s, EmpcTapeLayout* l) {
  if (!s || !l) return EMPC_ERR_INVALID;
  l->rec = s->kt.rec_full;
  l->off_fx = s->kt.off[0];
  l->off_fu = s->kt.off[1];
  l->off_lxx = s->kt.off[2];
  l->off_lxu = s->kt.off[3];
  l->off_luu = s->kt.off[4];
  l->off_lx = s->kt.off[5];
  l->off_lu = s->kt.off[6];
  l->off_gap = s->kt.off[7];
  l->off_cost = s->kt.off[8];
  l->ld_fx = s->kt.ld[0];
  l->ld_fu = s->kt.ld[1];
  l->ld_lxx = s->kt.ld[2];
  l->ld_lxu = s->kt.ld[3];
  l->ld_luu = s->kt.ld[4];
  return EMPC_OK;
}

// The kernel table, the buffers and the host copies of integrator / solver type are fixed when the solver is created: a new
// problem image may change cost tables, references, weights, x0 and dt, not the class of the problem.
static void check_same_class(const EmpcSolver* s, const HostProblem& N) {
  const HostProblem& O = s->H;
  auto fail = [](const char* what) {
    throw std::invalid_argument(std::string("update_problem: the new problem differs from the one the solver was created with in ") + what);
  };
  if (N.P.T != s->T || N.P.nx != s->kt.nx || N.P.nu != s->kt.nu || N.sets.size() != O.sets.size()) fail("its shapes (T, nx, nu, number of cost sets)");
  if (N.P.integrator != O.P.integrator) fail("its integrator (Euler / RK4)");
  if ((N.P.has_contact != 0) != (O.P.has_contact != 0)) fail("its dynamics (free / contact)");
  if (N.contact_rows != O.contact_rows) fail("its contact type (ContactModel3D / ContactModel6D)");
  if ((N.P.use_squash != 0) != (O.P.use_squash != 0)) fail("use_squash");
  if (N.P.model.nbodies != O.P.model.nbodies || N.P.n_rotors != O.P.n_rotors) fail("its robot class (bodies, rotors)");
}

const char* empc_solver_kernel_family(const EmpcSolver* s) { return s ? s->kernel_family : ""; }

int empc_solver_update_problem(EmpcSolver* s, const EmpcProblemDesc* problem) {
  EMPC_TRY
  if (!s || !problem) throw std::invalid_argument("NULL argument");
  s->use();
  const EmpcSolverParams prm = s->H.P.prm;
  HostProblem N;  // prepared aside: s->H keeps matching the device image if anything below throws
  prepare_problem(*problem, prm, N);
  check_same_class(s, N);
  // same class, possibly another robot of it: the instantiation over baked constants serves only the robot it was baked from
  KernelTable kt;
  const char* family = nullptr;
  if (!find_table(N.P, N.P.model.nbodies, N.P.n_rotors, N.P.has_contact != 0, N.contact_rows, kt, &family))
    throw std::runtime_error("update_problem: no kernel instantiation for the new problem");
  s->kt = kt;
  s->kernel_family = family;
  s->H = std::move(N);
  remember_problem(s, *problem);
  upload_problem(s);
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}

// One cost entry of the node's model, edited in place: what MpcAbstract::updateProblem does to the crocoddyl models between
// two solves (src/mpc-controllers/carrot-mpc.cpp:298-359: residual->set_reference, costs->get_costs().at(name)->weight / active).
// Nodes that share a cost set (all knots of a trajectory stage) see the edit together, like the reference's shared models.
int empc_solver_set_cost_refs(EmpcSolver* s, int knot, const char* cost_name, const double* ref, int nref, int active, double weight) {
  EMPC_TRY
  if (!s || !cost_name) throw std::invalid_argument("NULL argument");
  if (knot < 0 || knot > s->T) throw std::invalid_argument("set_cost_refs: knot out of range");
  if ((int)s->user_knot_set.size() != s->T + 1) throw std::invalid_argument("set_cost_refs: the solver holds no problem");
  const int set_index = s->user_knot_set[knot];
  if (set_index < 0 || set_index >= (int)s->user_sets.size()) throw std::invalid_argument("set_cost_refs: knot refers to a missing cost set");
  EmpcCostSet& set = s->user_sets[set_index];
  EmpcCost* c = nullptr;
  for (int i = 0; i < set.ncosts; ++i)
    if (std::strncmp(set.costs[i].name, cost_name, EMPC_NAME_LEN) == 0) c = &set.costs[i];
  if (!c) throw std::invalid_argument(std::string("set_cost_refs: node has no cost named '") + cost_name + "'");
  if (ref) {
    if (nref < 1 || nref > EMPC_MAX_NX) throw std::invalid_argument("set_cost_refs: reference length out of range");
    std::memcpy(c->ref, ref, sizeof(double) * nref);
  }
  if (active >= 0) c->active = active ? 1 : 0;
  if (weight == weight) c->weight = weight;  // NaN = leave
  s->problem_dirty = true;
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}

int empc_solver_set_x0(EmpcSolver* s, const double* x0s) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  s->use();
  const size_t nx = s->kt.nx;
  if (x0s) {
    HIP_CHECK(hipMemcpyAsync(s->D.x0, x0s, sizeof(double) * s->B * nx, hipMemcpyHostToDevice, s->stream));
  } else {
    std::vector<double> tmp((size_t)s->B * nx);
    for (int b = 0; b < s->B; ++b) std::memcpy(&tmp[b * nx], s->H.x0.data(), sizeof(double) * nx);
    HIP_CHECK(hipMemcpyAsync(s->D.x0, tmp.data(), sizeof(double) * tmp.size(), hipMemcpyHostToDevice, s->stream));
  }
  HIP_CHECK(hipStreamSynchronize(s->stream));
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}

// ---- plant of closed-loop MPC runs (bindings/python/eagle_mpc/utils/simulator.py) ------------------------------
int empc_plant_set_state(EmpcSolver* s, const double* x) {
  EMPC_TRY
  if (!s || !x) throw std::invalid_argument("NULL argument");
  s->use();
  HIP_CHECK(hipMemcpyAsync(s->dplant_x, x, sizeof(double) * s->B * s->kt.nx, hipMemcpyHostToDevice, s->stream));
  HIP_CHECK(hipStreamSynchronize(s->stream));
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}
int empc_plant_get_state(EmpcSolver* s, double* x) {
  EMPC_TRY
  if (!s || !x) throw std::invalid_argument("NULL argument");
  s->use();
  HIP_CHECK(hipMemcpyAsync(x, s->dplant_x, sizeof(double) * s->B * s->kt.nx, hipMemcpyDeviceToHost, s->stream));
  HIP_CHECK(hipStreamSynchronize(s->stream));
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}
int empc_plant_step(EmpcSolver* s, double dt_s, const double* u, int n_substeps) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  if (!(dt_s > 0) || n_substeps < 1) throw std::invalid_argument("plant step: dt must be positive and n_substeps >= 1");
  if (!u && !s->have_state) throw std::invalid_argument("plant step: no solve has run yet, pass the controls explicitly");
  s->use();
  const double* du = nullptr;
  if (u) {
    HIP_CHECK(hipMemcpyAsync(s->dplant_u, u, sizeof(double) * s->B * s->kt.nu, hipMemcpyHostToDevice, s->stream));
    du = s->dplant_u;
  }
  s->kt.plant(s->D, s->dplant_x, du, dt_s, n_substeps, s->stream);
  HIP_CHECK(hipStreamSynchronize(s->stream));
  HIP_CHECK(hipGetLastError());
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}
int empc_solver_set_x0_from_plant(EmpcSolver* s) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  s->use();
  HIP_CHECK(hipMemcpyAsync(s->D.x0, s->dplant_x, sizeof(double) * s->B * s->kt.nx, hipMemcpyDeviceToDevice, s->stream));
  HIP_CHECK(hipStreamSynchronize(s->stream));
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}

int empc_solver_set_warmstart(EmpcSolver* s, const double* xs, const double* us) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  s->use();
  const size_t nx = s->kt.nx, nu = s->kt.nu, T = s->T, B = s->B;
  if (xs) {
    HIP_CHECK(hipMemcpyAsync(s->D.xs, xs, sizeof(double) * B * (T + 1) * nx, hipMemcpyHostToDevice, s->stream));
  } else {
    std::vector<double> tmp(B * (T + 1) * nx, 0.0);  // state->zero() at every node (crocoddyl setCandidate)
    for (size_t i = 0; i < B * (T + 1); ++i) tmp[i * nx + 6] = 1.0;
    HIP_CHECK(hipMemcpyAsync(s->D.xs, tmp.data(), sizeof(double) * tmp.size(), hipMemcpyHostToDevice, s->stream));
    HIP_CHECK(hipStreamSynchronize(s->stream));
  }
  if (us)
    HIP_CHECK(hipMemcpyAsync(s->D.us, us, sizeof(double) * B * T * nu, hipMemcpyHostToDevice, s->stream));
  else
    HIP_CHECK(hipMemsetAsync(s->D.us, 0, sizeof(double) * B * T * nu, s->stream));
  HIP_CHECK(hipStreamSynchronize(s->stream));
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}

int empc_solver_set_convergence_init(EmpcSolver* s, double c) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  s->use();
  s->H.P.prm.convergence_init = c;
  HIP_CHECK(hipMemcpyAsync(s->dP, &s->H.P, sizeof(DevProblem), hipMemcpyHostToDevice, s->stream));
  HIP_CHECK(hipStreamSynchronize(s->stream));
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}

static int copy_out_fwd(EmpcSolver* s, const void* dsrc, void* hdst, size_t bytes);
static void timed(EmpcSolver* s, int slot, double& acc_ms) {
  float ms = 0;
  if (hipEventElapsedTime(&ms, s->ev[slot], s->ev[slot + 1]) == hipSuccess) acc_ms += ms;
}

// One chunk of the batch = one HIP stream.  The three hot kernels are latency bound at batch 1024 (backward: one
// wavefront per SIMD; rollout: 160 wavefronts), so independent chunks running on separate streams overlap on the
// device and fill the idle SIMDs; there is no dependency between trajectories, hence none between chunks.
struct Chunk {
  DevBuffers D;
  hipStream_t stream;
  hipEvent_t ev[2][7];
  int b0, nb, active, idx;
  int lin;  // trajectories on the linearize list of the chunk's oldest in-flight sweep
};

static DevBuffers chunk_view(const EmpcSolver* s, int b0, int nb, int idx) {
  DevBuffers D = s->D;
  const size_t T = s->T, NA = s->NA;
  const KernelTable& k = s->kt;
  D.st += b0;
  D.x0 += (size_t)b0 * k.nx;
  D.xs += (size_t)b0 * (T + 1) * k.nx;
  D.us += (size_t)b0 * T * k.nu;
  D.acc += (size_t)b0 * (T + 1) * k.nacc;
  D.tape += (size_t)b0 * (T + 1) * k.rec;
  D.K += (size_t)b0 * T * k.nu * k.ndx;
  D.kff += (size_t)b0 * T * k.nu;
  D.Vx += (size_t)b0 * (T + 1) * k.ndx;
  D.Vf += (size_t)b0 * (T + 1) * k.ndx;
  D.xs_try += (size_t)b0 * NA * (T + 1) * k.nx;
  D.us_try += (size_t)b0 * NA * T * k.nu;
  D.acc_try += (size_t)b0 * NA * (T + 1) * k.nacc;
  D.try_cost += (size_t)b0 * NA;
  D.try_dv += (size_t)b0 * NA;
  D.try_ok += (size_t)b0 * NA;
  D.try_ncalc += (size_t)b0 * NA;
  D.us_last += (size_t)b0 * T * k.nu;
  if (D.trace) D.trace += (size_t)b0 * D.trace_cap * EMPC_TRACE_WORDS;
  D.n_active = s->D.n_active + 6 * idx;
  D.B = nb;
  return D;
}

static Rk4Buffers chunk_rk4(const EmpcSolver* s, int b0) {
  Rk4Buffers R = s->R4;
  const size_t T = s->T, o = (size_t)4 * b0;
  const KernelTable& k = s->kt;
  R.ys += o * (T + 1) * k.nx;
  R.accs += o * (T + 1) * k.nacc;
  R.us4 += o * T * k.nu;
  R.tape4 += o * (T + 1) * k.rec;
  R.st4 += o;
  return R;
}

// The sweep loop of one solve: every chunk of the batch runs calc -> linearize -> backward -> rollout -> select on its
// stream until no trajectory of it is active any more.  `hard_cap` bounds the sweeps of a chunk.
static void run_sweeps(EmpcSolver* s, int hard_cap) {
  EmpcSolveStats& S = s->stats;
  std::memset(&S, 0, sizeof(S));
  const KernelTable& k = s->kt;
  // chunking
  int nchunks = s->n_streams;
  if (nchunks > s->B) nchunks = s->B;
  if (nchunks < 1) nchunks = 1;
  std::vector<Chunk> chunks(nchunks);
  for (int c = 0; c < nchunks; ++c) {
    const int lo = (int)((long long)s->B * c / nchunks), hi = (int)((long long)s->B * (c + 1) / nchunks);
    chunks[c].b0 = lo;
    chunks[c].nb = hi - lo;
    chunks[c].idx = c;
    chunks[c].active = hi - lo;
    chunks[c].lin = hi - lo;  // the first sweep linearizes every trajectory
    chunks[c].D = chunk_view(s, lo, hi - lo, c);
    chunks[c].stream = s->streams[c];
    for (int q = 0; q < 2; ++q)
      for (int e = 0; e < 7; ++e) chunks[c].ev[q][e] = s->cev[c][q][e];
  }
  const hipEvent_t t_begin = s->t_begin, t_end = s->t_end;  // owned by the solver: nothing to leak on an error path
  HIP_CHECK(hipEventRecord(t_begin, s->stream));
  // Every chunk runs its own sweep loop on its own stream, two sweeps deep: sweep k + 1 is queued before the host has
  // seen the active count of sweep k (its kernels return at once for finished trajectories), so the device never waits
  // for the host round trip between sweeps; the one surplus sweep at the end is empty and is not counted.
  // Chunks start staggered -- chunk c waits for chunk c-1's first backward pass.
  std::vector<int> queued(nchunks, 0), retired(nchunks, 0);
  std::vector<std::array<bool, 2>> timed_slot(nchunks, std::array<bool, 2>{false, false});
  static const int timing_every = [] {
    const char* e = std::getenv("EMPC_TIMING_EVERY");  // 1 = every sweep (default 4); the per-kernel times of EmpcSolveStats
    const int v = e ? std::atoi(e) : 4;                 // are sums over the TIMED launches, n_* counts them
    return v < 1 ? 1 : v;
  }();
  S.timing_every = timing_every;
  auto enqueue = [&](Chunk& c) {
    const int q = queued[c.idx] & 1;
    DevBuffers Dq = c.D;
    Dq.n_active = c.D.n_active + 3 * q;  // per sweep slot: {active trajectories, linearize-list entries, calc-list entries}
    Dq.lin_count_out = Dq.n_active + 1;
    Dq.calc_count_out = Dq.n_active + 2;
    Dq.lin_list_out = s->dlin_list + (size_t)q * s->B + c.b0;
    Dq.act_list_out = s->dact_list + (size_t)q * s->B + c.b0;
    Dq.calc_list_out = s->dcalc_list + (size_t)q * s->B + c.b0;
    if (queued[c.idx] > 0) {  // the lists written by the previous sweep's select; the first sweep takes every trajectory
      Dq.lin_count = c.D.n_active + 3 * (1 - q) + 1;
      Dq.calc_count = c.D.n_active + 3 * (1 - q) + 2;
      Dq.calc_list = s->dcalc_list + (size_t)(1 - q) * s->B + c.b0;
      Dq.lin_list = s->dlin_list + (size_t)(1 - q) * s->B + c.b0;
      Dq.act_count = c.D.n_active + 3 * (1 - q);
      Dq.act_list = s->dact_list + (size_t)(1 - q) * s->B + c.b0;
      Dq.lin_bound = c.active > 0 ? c.active : 1;  // the active set only shrinks: the last count the host saw bounds the list
    }
    Dq.counters_next = c.D.n_active + 3 * (1 - q);  // zeroed by this sweep's select for the next sweep
    Dq.done_ticket = s->dticket + c.idx;
    Dq.host_active = s->h_active_dev + 2 * (2 * c.idx + q);
    // kernel-boundary events only on the sweeps that are timed (every `timing_every`-th: each record is a command in the
    // stream, ~10 us of dead time between two kernels); the end-of-sweep event is always there (the host waits on it)
    const bool timed = (queued[c.idx] % timing_every) == 0;
    timed_slot[c.idx][q] = timed;
    if (timed) HIP_CHECK(hipEventRecord(c.ev[q][0], c.stream));
    k.calc(Dq, c.stream);
    if (timed) HIP_CHECK(hipEventRecord(c.ev[q][1], c.stream));
    if (s->D.integrator == EMPC_INTEGRATOR_RK4)
      k.rk4_linearize(Dq, chunk_rk4(s, c.b0), c.stream);
    else
      k.linearize(Dq, c.stream);
    if (timed) HIP_CHECK(hipEventRecord(c.ev[q][2], c.stream));
    k.backward(Dq, c.stream);
    if (timed) HIP_CHECK(hipEventRecord(c.ev[q][3], c.stream));
    k.rollout(Dq, c.stream);
    if (timed) HIP_CHECK(hipEventRecord(c.ev[q][4], c.stream));
    k.select(Dq, c.stream);
    HIP_CHECK(hipEventRecord(c.ev[q][5], c.stream));  // the active count is in pinned memory once select is done
    queued[c.idx]++;
  };
  auto retire = [&](Chunk& c) {  // oldest in-flight sweep of the chunk
    const int q = retired[c.idx] & 1;
    HIP_CHECK(hipEventSynchronize(c.ev[q][5]));
    if (timed_slot[c.idx][q]) {
      auto el = [&](int a) {
        float ms = 0;
        return hipEventElapsedTime(&ms, c.ev[q][a], c.ev[q][a + 1]) == hipSuccess ? (double)ms : 0.0;
      };
      S.ms_calc += el(0);
      S.ms_linearize += el(1);
      S.ms_backward += el(2);
      S.ms_rollout += el(3);
      S.ms_select += el(4);
      S.n_calc++;
      S.n_linearize++;
      S.n_backward++;
      S.n_rollout++;
      S.n_select++;
      S.backward_units += (long long)c.active * s->T;
      S.rollout_units += (long long)c.active * s->NA * (s->T + 1);
      S.linearize_units += (long long)c.lin * (s->T + 1);  // the trajectories on this sweep's linearize list
    }
    // (units of EVERY sweep, timed or not: total work of the solve)
    S.backward_units_all += (long long)c.active * s->T;
    S.rollout_units_all += (long long)c.active * s->NA * (s->T + 1);
    S.linearize_units_all += (long long)c.lin * (s->T + 1);
    c.active = s->h_active[2 * (2 * c.idx + q)];
    c.lin = s->h_active[2 * (2 * c.idx + q) + 1];  // what select put on the next sweep's linearize list
    retired[c.idx]++;
  };
  for (auto& c : chunks) {
    // counters of both sweep slots and the completion ticket start at zero; from then on select keeps them (no memset
    // and no copy command per sweep)
    HIP_CHECK(hipMemsetAsync(c.D.n_active, 0, sizeof(int) * 6, c.stream));
    HIP_CHECK(hipMemsetAsync(s->dticket + c.idx, 0, sizeof(int), c.stream));
    if (c.idx > 0) HIP_CHECK(hipStreamWaitEvent(c.stream, chunks[c.idx - 1].ev[0][3], 0));  // stagger the first sweep
    enqueue(c);
    if (hard_cap > 1) enqueue(c);
  }
  int total_active = s->B;
  bool any_live = true;
  while (any_live) {
    any_live = false;
    for (auto& c : chunks) {
      if (retired[c.idx] == queued[c.idx] || c.active <= 0) continue;
      retire(c);
      if (c.active > 0 && queued[c.idx] < hard_cap) enqueue(c);
      if (c.active > 0 && retired[c.idx] < queued[c.idx]) any_live = true;
    }
    HIP_CHECK(hipGetLastError());
  }
  total_active = 0;
  for (auto& c : chunks) {
    total_active += c.active;
    S.sweeps = std::max(S.sweeps, retired[c.idx]);
    HIP_CHECK(hipStreamSynchronize(c.stream));  // drains the surplus (empty) sweep ...
    while (retired[c.idx] < queued[c.idx]) {    // ... whose launches still count as launches (they are in any profile)
      const int act = c.active, lin = c.lin;
      c.active = c.lin = 0;  // no units processed
      retire(c);
      c.active = act;
      c.lin = lin;
    }
  }
  for (int c = 1; c < nchunks; ++c) HIP_CHECK(hipStreamSynchronize(chunks[c].stream));
  HIP_CHECK(hipEventRecord(t_end, s->stream));
  download_states(s);
  float ms = 0;
  HIP_CHECK(hipEventElapsedTime(&ms, t_begin, t_end));
  S.ms_total = ms;
  for (int b = 0; b < s->B; ++b) {
    S.total_iters += s->h_st[b].total_iters;
    S.max_iters = std::max(S.max_iters, s->h_st[b].total_iters);
  }
  // a solve that ran into the sweep cap leaves no state a later "previous" warm start may build on
  s->have_state = (total_active == 0);
  if (total_active > 0) throw std::runtime_error("solve did not terminate within the sweep cap (internal error)");
}

int empc_solver_solve(EmpcSolver* s, int maxiter, int is_feasible) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  if (maxiter < 1) throw std::invalid_argument("maxiter must be >= 1");
  s->use();
  flush_problem(s);
  for (int b = 0; b < s->B; ++b) {
    TrajState prev = s->h_st[b];
    init_traj_state(s->h_st[b], s->H.P.prm, maxiter, is_feasible != 0, s->have_state ? &prev : nullptr);
  }
  upload_states(s);
  if (s->D.solver_type != EMPC_SOLVER_SBFDDP)  // the BoxQP of knot t is warm-started at k_[t]: zeros at the start of a solve
    HIP_CHECK(hipMemsetAsync(s->D.kff, 0, sizeof(double) * s->B * s->T * s->kt.nu, s->stream));
  HIP_CHECK(hipStreamSynchronize(s->stream));
  run_sweeps(s, 4 * (3 * (maxiter + 1) + 8));  // (passes + clean-up) x maxiter can never be exceeded
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}

int empc_solver_set_gains(EmpcSolver* s, const double* K, const double* k) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  s->use();
  const size_t B = s->B, T = s->T, n = s->kt.ndx, m = s->kt.nu;
  if (K) HIP_CHECK(hipMemcpy(s->D.K, K, sizeof(double) * B * T * m * n, hipMemcpyHostToDevice));
  if (k) HIP_CHECK(hipMemcpy(s->D.kff, k, sizeof(double) * B * T * m, hipMemcpyHostToDevice));
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}

// ---- streamed solves ("continuous batching", include/empc.h) -----------------------------------------------------------
static size_t stream_row(const EmpcSolver* s) { return (size_t)(s->T + 1) * s->kt.nx + 2 * (size_t)s->T * s->kt.nu + 3; }

int empc_solver_stream_begin(EmpcSolver* s, int n_jobs, const double* x0s) {
  EMPC_TRY
  if (!s || !x0s) throw std::invalid_argument("NULL argument");
  if (n_jobs < 1) throw std::invalid_argument("stream: n_jobs must be >= 1");
  s->use();
  // (each pointer is forgotten before anything else can throw: the destructor frees what is still set)
  s->q_njobs = 0;
  if (double* p = s->dq_x0) {
    s->dq_x0 = nullptr;
    HIP_CHECK(hipFree(p));
  }
  if (double* p = s->dq_rows) {
    s->dq_rows = nullptr;
    HIP_CHECK(hipFree(p));
  }
  if (!s->dq_head) {
    HIP_CHECK(hipMalloc((void**)&s->dq_head, 16));
    s->dq_iters = reinterpret_cast<unsigned long long*>(s->dq_head + 2);
  }
  HIP_CHECK(hipMalloc((void**)&s->dq_x0, sizeof(double) * (size_t)n_jobs * s->kt.nx));
  HIP_CHECK(hipMalloc((void**)&s->dq_rows, sizeof(double) * (size_t)n_jobs * stream_row(s)));
  HIP_CHECK(hipMemcpy(s->dq_x0, x0s, sizeof(double) * (size_t)n_jobs * s->kt.nx, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemset(s->dq_rows, 0, sizeof(double) * (size_t)n_jobs * stream_row(s)));
  s->q_njobs = n_jobs;
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}

int empc_solver_stream_run(EmpcSolver* s, int maxiter) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  if (maxiter < 1) throw std::invalid_argument("maxiter must be >= 1");
  if (s->q_njobs < 1) throw std::invalid_argument("stream: call empc_solver_stream_begin first");
  s->use();
  flush_problem(s);
  // the first jobs go to the slots directly: setCandidate([], []) + problem.x0 of the job
  const int nfirst = std::min(s->B, s->q_njobs);
  if (empc_solver_set_warmstart(s, nullptr, nullptr) != EMPC_OK) throw std::runtime_error(empc_last_error());
  HIP_CHECK(hipMemcpyAsync(s->D.x0, s->dq_x0, sizeof(double) * (size_t)nfirst * s->kt.nx, hipMemcpyDeviceToDevice, s->stream));
  for (int b = 0; b < s->B; ++b) {
    init_traj_state(s->h_st[b], s->H.P.prm, maxiter, false, nullptr);
    s->h_st[b].job = b < nfirst ? b : -1;
    if (b >= nfirst) s->h_st[b].phase = PHASE_DONE;
  }
  upload_states(s);
  HIP_CHECK(hipMemsetAsync(s->D.kff, 0, sizeof(double) * s->B * s->T * s->kt.nu, s->stream));
  HIP_CHECK(hipMemsetAsync(s->dq_head, 0, 16, s->stream));
  HIP_CHECK(hipMemcpyAsync(s->dq_head, &nfirst, sizeof(int), hipMemcpyHostToDevice, s->stream));
  HIP_CHECK(hipStreamSynchronize(s->stream));
  const DevBuffers keep = s->D;
  s->D.trace = nullptr;  // the iteration trace (callbacks of the mirrors) records plain solves: a slot of a stream works
  s->D.trace_cap = 0;    // on many jobs in a row; it is switched off for the duration of the stream and back on after
  s->D.q_x0 = s->dq_x0;
  s->D.q_rows = s->dq_rows;
  s->D.q_head = s->dq_head;
  s->D.q_iters = s->dq_iters;
  s->D.q_njobs = s->q_njobs;
  s->D.q_maxiter = maxiter;
  const long long per_solve = 4LL * (3 * (maxiter + 1) + 8);
  const long long rounds = (s->q_njobs + s->B - 1) / s->B + 1;
  try {
    run_sweeps(s, (int)std::min<long long>(per_solve * rounds, 1LL << 30));
  } catch (...) {
    s->D = keep;
    throw;
  }
  s->D = keep;
  unsigned long long it = 0;
  HIP_CHECK(hipMemcpy(&it, s->dq_iters, sizeof(it), hipMemcpyDeviceToHost));
  s->stats.total_iters = (long long)it;
  int mx = 0;  // largest iteration count of any JOB (select keeps it next to the queue head), not of the slots' last jobs
  HIP_CHECK(hipMemcpy(&mx, s->dq_head + 1, sizeof(mx), hipMemcpyDeviceToHost));
  s->stats.max_iters = mx;
  s->have_state = false;  // the slots hold the last jobs they worked on, not one solve of the batch
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}

int empc_solver_stream_results(EmpcSolver* s, double* rows, int* row_doubles) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  if (row_doubles) *row_doubles = (int)stream_row(s);
  if (!rows) return EMPC_OK;
  if (s->q_njobs < 1 || !s->dq_rows) throw std::invalid_argument("stream: nothing to fetch");
  s->use();
  HIP_CHECK(hipMemcpy(rows, s->dq_rows, sizeof(double) * (size_t)s->q_njobs * stream_row(s), hipMemcpyDeviceToHost));
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}

int empc_solver_stream_results_device(EmpcSolver* s, double* dst_device) {
  EMPC_TRY
  if (!s || !dst_device) throw std::invalid_argument("NULL argument");
  if (s->q_njobs < 1 || !s->dq_rows) throw std::invalid_argument("stream: nothing to fetch");
  s->use();
  HIP_CHECK(hipMemcpyAsync(dst_device, s->dq_rows, sizeof(double) * (size_t)s->q_njobs * stream_row(s), hipMemcpyDeviceToDevice, s->stream));
  HIP_CHECK(hipStreamSynchronize(s->stream));
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}

int empc_solver_device_info(EmpcSolver* s, int* device_index, char* pci_bus_id, int pci_len) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("NULL argument");
  // where the solver's device memory really lives (not what the caller asked for): attributes of its problem image
  hipPointerAttribute_t at;
  HIP_CHECK(hipPointerGetAttributes(&at, s->dP));
  if (device_index) *device_index = at.device;
  if (pci_bus_id && pci_len > 0) HIP_CHECK(hipDeviceGetPCIBusId(pci_bus_id, pci_len, at.device));
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}

// ---- step-wise entry points (include/empc.h): one iteration from any iterate -------------------------------------------
int empc_solver_get_states(EmpcSolver* s, EmpcTrajState* states) {
  EMPC_TRY
  if (!s || !states) throw std::invalid_argument("NULL argument");
  s->use();
  download_states(s);
  std::memcpy(states, s->h_st.data(), sizeof(TrajState) * s->B);
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}
int empc_solver_set_states(EmpcSolver* s, const EmpcTrajState* states) {
  EMPC_TRY
  if (!s || !states) throw std::invalid_argument("NULL argument");
  s->use();
  std::memcpy(s->h_st.data(), states, sizeof(TrajState) * s->B);
  upload_states(s);
  HIP_CHECK(hipStreamSynchronize(s->stream));
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}
int empc_sweep_batch(EmpcSolver* s, int stages) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  if (stages <= 0 || (stages & ~EMPC_STAGE_ALL)) throw std::invalid_argument("sweep: unknown stage bits");
  s->use();
  flush_problem(s);
  DevBuffers D = s->D;  // every trajectory, no work lists, no host hand-over
  HIP_CHECK(hipMemsetAsync(D.n_active, 0, sizeof(int) * 6, s->stream));
  std::memset(&s->stats, 0, sizeof(s->stats));
  HIP_CHECK(hipEventRecord(s->ev[0], s->stream));
  if (stages & EMPC_STAGE_LINEARIZE) {
    s->kt.calc(D, s->stream);
    if (s->D.integrator == EMPC_INTEGRATOR_RK4)
      s->kt.rk4_linearize(D, s->R4, s->stream);
    else
      s->kt.linearize(D, s->stream);
  }
  HIP_CHECK(hipEventRecord(s->ev[1], s->stream));
  if (stages & EMPC_STAGE_BACKWARD) s->kt.backward(D, s->stream);
  HIP_CHECK(hipEventRecord(s->ev[2], s->stream));
  if (stages & EMPC_STAGE_ROLLOUT) s->kt.rollout(D, s->stream);
  HIP_CHECK(hipEventRecord(s->ev[3], s->stream));
  if (stages & EMPC_STAGE_SELECT) s->kt.select(D, s->stream);
  HIP_CHECK(hipEventRecord(s->ev[4], s->stream));
  download_states(s);
  HIP_CHECK(hipGetLastError());
  timed(s, 0, s->stats.ms_linearize);
  timed(s, 1, s->stats.ms_backward);
  timed(s, 2, s->stats.ms_rollout);
  timed(s, 3, s->stats.ms_select);
  s->stats.sweeps = 1;
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}
int empc_select_batch(EmpcSolver* s, const int* try_ok, const double* try_cost, const double* try_dv) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  s->use();
  const size_t n = (size_t)s->B * s->NA;
  if (try_ok) HIP_CHECK(hipMemcpyAsync(s->D.try_ok, try_ok, sizeof(int) * n, hipMemcpyHostToDevice, s->stream));
  if (try_cost) HIP_CHECK(hipMemcpyAsync(s->D.try_cost, try_cost, sizeof(double) * n, hipMemcpyHostToDevice, s->stream));
  if (try_dv) HIP_CHECK(hipMemcpyAsync(s->D.try_dv, try_dv, sizeof(double) * n, hipMemcpyHostToDevice, s->stream));
  HIP_CHECK(hipStreamSynchronize(s->stream));  // the host arrays are the caller's
  return empc_sweep_batch(s, EMPC_STAGE_SELECT);
  EMPC_CATCH(RET_INT)
}
int empc_solver_get_trials(EmpcSolver* s, double* try_cost, double* try_dv, int* try_ok) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  s->use();
  const size_t n = (size_t)s->B * s->NA;
  if (try_cost) HIP_CHECK(hipMemcpy(try_cost, s->D.try_cost, sizeof(double) * n, hipMemcpyDeviceToHost));
  if (try_dv) HIP_CHECK(hipMemcpy(try_dv, s->D.try_dv, sizeof(double) * n, hipMemcpyDeviceToHost));
  if (try_ok) HIP_CHECK(hipMemcpy(try_ok, s->D.try_ok, sizeof(int) * n, hipMemcpyDeviceToHost));
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}
// the device tape as the ABI describes it: records in the full layout (with EMPC_REC_TRI off the device layout is that layout)
static int tape_out(EmpcSolver* s, double* tape) {
  const size_t nrec = (size_t)s->B * (s->T + 1);
  if (s->kt.rec == s->kt.rec_full) return copy_out_fwd(s, s->D.tape, tape, sizeof(double) * nrec * s->kt.rec);
  std::vector<double> h(nrec * s->kt.rec);
  const int rc = copy_out_fwd(s, s->D.tape, h.data(), sizeof(double) * h.size());
  if (rc != EMPC_OK) return rc;
  for (size_t i = 0; i < nrec; ++i) s->kt.unpack(h.data() + i * s->kt.rec, tape + i * s->kt.rec_full);
  return EMPC_OK;
}
int empc_solver_get_tape(EmpcSolver* s, double* tape) {
  if (!s) return EMPC_ERR_INVALID;
  return tape_out(s, tape);
}
int empc_solver_get_gains(EmpcSolver* s, double* K, double* k, double* Vx) {
  if (!s) return EMPC_ERR_INVALID;
  const size_t B = s->B, T = s->T, n = s->kt.ndx, m = s->kt.nu;
  int rc = EMPC_OK;
  if (K && rc == EMPC_OK) rc = copy_out_fwd(s, s->D.K, K, sizeof(double) * B * T * m * n);
  if (k && rc == EMPC_OK) rc = copy_out_fwd(s, s->D.kff, k, sizeof(double) * B * T * m);
  if (Vx && rc == EMPC_OK) rc = copy_out_fwd(s, s->D.Vx, Vx, sizeof(double) * B * (T + 1) * n);
  return rc;
}

static int copy_out(EmpcSolver* s, const void* dsrc, void* hdst, size_t bytes);
static int copy_out_fwd(EmpcSolver* s, const void* dsrc, void* hdst, size_t bytes) { return copy_out(s, dsrc, hdst, bytes); }
static int copy_out(EmpcSolver* s, const void* dsrc, void* hdst, size_t bytes) {
  EMPC_TRY
  if (!s || !hdst) throw std::invalid_argument("NULL argument");
  s->use();
  HIP_CHECK(hipMemcpyAsync(hdst, dsrc, bytes, hipMemcpyDeviceToHost, s->stream));
  HIP_CHECK(hipStreamSynchronize(s->stream));
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}
int empc_solver_get_xs(EmpcSolver* s, double* xs) {
  if (!s) return EMPC_ERR_INVALID;
  return copy_out(s, s->D.xs, xs, sizeof(double) * s->B * (s->T + 1) * s->kt.nx);
}
int empc_solver_get_us(EmpcSolver* s, double* us) {
  if (!s) return EMPC_ERR_INVALID;
  return copy_out(s, s->D.us, us, sizeof(double) * s->B * s->T * s->kt.nu);
}
int empc_solver_get_us_squash(EmpcSolver* s, double* out) {
  EMPC_TRY
  if (!s || !out) throw std::invalid_argument("NULL argument");
  s->use();
  s->kt.squash_out(s->D, s->dscratch, s->stream);
  HIP_CHECK(hipMemcpyAsync(out, s->dscratch, sizeof(double) * s->B * s->T * s->kt.nu, hipMemcpyDeviceToHost, s->stream));
  HIP_CHECK(hipStreamSynchronize(s->stream));
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}
// One row per rollout: xs | us_squash | cost | iters (as double), written to a DEVICE buffer of batch x row doubles --
// the payload of the multi-GPU result gather, packed without a host round trip.
int empc_solver_pack_results_device(EmpcSolver* s, double* dst_device, int* row_doubles) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  s->use();
  const size_t nx = s->kt.nx, nu = s->kt.nu, T = s->T, B = s->B;
  const size_t nxs = (T + 1) * nx, nus = T * nu, row = nxs + nus + 2;
  if (row_doubles) *row_doubles = (int)row;
  if (!dst_device) return EMPC_OK;  // size query
  s->kt.pack_rows(s->D, dst_device, s->stream);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(s->stream));
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}
int empc_solver_get_cost(EmpcSolver* s, double* cost) {
  if (!s || !cost) return EMPC_ERR_INVALID;
  for (int b = 0; b < s->B; ++b) cost[b] = s->h_st[b].cost;
  return EMPC_OK;
}
int empc_solver_get_stop(EmpcSolver* s, double* stop) {
  if (!s || !stop) return EMPC_ERR_INVALID;
  for (int b = 0; b < s->B; ++b) stop[b] = s->h_st[b].stop;
  return EMPC_OK;
}
int empc_solver_get_iters(EmpcSolver* s, int* iters) {
  if (!s || !iters) return EMPC_ERR_INVALID;
  for (int b = 0; b < s->B; ++b) iters[b] = s->h_st[b].iter;
  return EMPC_OK;
}
int empc_solver_get_status(EmpcSolver* s, int* status) {
  if (!s || !status) return EMPC_ERR_INVALID;
  for (int b = 0; b < s->B; ++b) status[b] = s->h_st[b].status;
  return EMPC_OK;
}
int empc_solver_enable_trace(EmpcSolver* s, int capacity) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  if (capacity < 0) throw std::invalid_argument("trace capacity must be >= 0");
  s->use();
  if (s->dtrace) {
    HIP_CHECK(hipFree(s->dtrace));
    s->dtrace = nullptr;
  }
  s->trace_cap = 0;
  s->D.trace = nullptr;
  s->D.trace_cap = 0;
  if (capacity > 0) {
    const size_t n = (size_t)s->B * capacity * EMPC_TRACE_WORDS;
    HIP_CHECK(hipMalloc((void**)&s->dtrace, n * sizeof(double)));
    HIP_CHECK(hipMemset(s->dtrace, 0, n * sizeof(double)));
    s->trace_cap = capacity;
    s->D.trace = s->dtrace;
    s->D.trace_cap = capacity;
  }
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}
int empc_solver_get_trace(EmpcSolver* s, int b, double* records, int max_records, int* n_records) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  if (b < 0 || b >= s->B) throw std::invalid_argument("trajectory index out of range");
  if (!s->dtrace) throw std::invalid_argument("trace is off: call empc_solver_enable_trace first");
  s->use();
  const int written = s->h_st[b].trace_count, cap = s->trace_cap;
  if (n_records) *n_records = written;
  if (!records || max_records <= 0) return EMPC_OK;
  const int avail = std::min(written, cap), n = std::min(avail, max_records);
  std::vector<double> ring((size_t)cap * EMPC_TRACE_WORDS);
  HIP_CHECK(hipMemcpy(ring.data(), s->dtrace + (size_t)b * cap * EMPC_TRACE_WORDS, ring.size() * sizeof(double), hipMemcpyDeviceToHost));
  const int first = written - avail;  // oldest record still in the ring
  for (int i = 0; i < n; ++i)
    std::memcpy(records + (size_t)i * EMPC_TRACE_WORDS, ring.data() + (size_t)((first + i) % cap) * EMPC_TRACE_WORDS,
                sizeof(double) * EMPC_TRACE_WORDS);
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}
// diagnostic builds only (-DEMPC_STAMPS): per-stage cycle counters written by trajectory 0
int empc_solver_debug_counters(EmpcSolver* s, unsigned long long* out, int n) {
  if (!s || !out || n > 128) return EMPC_ERR_INVALID;
  if (hipMemcpy(out, s->D.dbg, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost) != hipSuccess) return EMPC_ERR_RUNTIME;
  return EMPC_OK;
}
int empc_solver_get_stats(EmpcSolver* s, EmpcSolveStats* stats) {
  if (!s || !stats) return EMPC_ERR_INVALID;
  *stats = s->stats;
  return EMPC_OK;
}

// ---- phase-level entry points --------------------------------------------------------------------------------
static void phase_setup(EmpcSolver* s, double smooth, int is_feasible, double xreg, bool ddp, bool need_lin) {
  for (int b = 0; b < s->B; ++b) {
    TrajState& st = s->h_st[b];
    init_traj_state(st, s->H.P.prm, 100, false, nullptr);
    st.smooth = smooth;
    st.is_feasible = is_feasible;
    st.xreg = st.ureg = xreg;
    st.phase = ddp ? PHASE_DDP : 0;
    st.need_lin = need_lin ? 1 : 0;
    st.need_calc = need_lin ? 1 : 0;
  }
  upload_states(s);
}

int empc_linearize_batch(EmpcSolver* s, const double* xs, const double* us, double smooth, int is_feasible, double* tape,
                         double* cost, double* xnext) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  (void)xnext;
  s->use();
  flush_problem(s);
  if (xs || us) {
    if (empc_solver_set_warmstart(s, xs, us) != EMPC_OK) throw std::runtime_error(empc_last_error());
  }
  phase_setup(s, smooth, is_feasible, s->H.P.prm.reg_init, false, true);
  HIP_CHECK(hipEventRecord(s->ev[0], s->stream));
  s->kt.calc(s->D, s->stream);
  HIP_CHECK(hipEventRecord(s->ev[1], s->stream));
  if (s->D.integrator == EMPC_INTEGRATOR_RK4)
    s->kt.rk4_linearize(s->D, s->R4, s->stream);
  else
    s->kt.linearize(s->D, s->stream);
  HIP_CHECK(hipEventRecord(s->ev[2], s->stream));
  HIP_CHECK(hipStreamSynchronize(s->stream));
  HIP_CHECK(hipGetLastError());
  std::memset(&s->stats, 0, sizeof(s->stats));
  timed(s, 0, s->stats.ms_calc);
  timed(s, 1, s->stats.ms_linearize);
  s->stats.n_linearize = 1;
  s->stats.linearize_units = (long long)s->B * (s->T + 1);
  const size_t n = (size_t)s->B * (s->T + 1) * s->kt.rec;
  if (tape) {
    const int rc = tape_out(s, tape);
    if (rc != EMPC_OK) return rc;
  }
  if (cost) {
    std::vector<double> h(n);
    HIP_CHECK(hipMemcpy(h.data(), s->D.tape, sizeof(double) * n, hipMemcpyDeviceToHost));
    for (int b = 0; b < s->B; ++b) {
      double c = 0;
      for (int t = 0; t <= s->T; ++t) c += h[((size_t)b * (s->T + 1) + t) * s->kt.rec + s->kt.dev_off_cost];
      cost[b] = c;
    }
  }
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}

int empc_backward_batch(EmpcSolver* s, double xreg, int is_feasible, double* K, double* k, double* Vx, double* dgdq, int* ok) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  s->use();
  flush_problem(s);
  const double smooth = s->h_st.empty() ? s->H.P.prm.smooth_init : s->h_st[0].smooth;
  phase_setup(s, smooth > 0 ? smooth : s->H.P.prm.smooth_init, is_feasible, xreg, false, true);
  HIP_CHECK(hipEventRecord(s->ev[0], s->stream));
  s->kt.backward(s->D, s->stream);
  HIP_CHECK(hipEventRecord(s->ev[1], s->stream));
  download_states(s);
  HIP_CHECK(hipGetLastError());
  std::memset(&s->stats, 0, sizeof(s->stats));
  timed(s, 0, s->stats.ms_backward);
  s->stats.n_backward = 1;
  s->stats.backward_units = (long long)s->B * s->T;
  const size_t B = s->B, T = s->T, n = s->kt.ndx, m = s->kt.nu;
  if (K) HIP_CHECK(hipMemcpy(K, s->D.K, sizeof(double) * B * T * m * n, hipMemcpyDeviceToHost));
  if (k) HIP_CHECK(hipMemcpy(k, s->D.kff, sizeof(double) * B * T * m, hipMemcpyDeviceToHost));
  if (Vx) HIP_CHECK(hipMemcpy(Vx, s->D.Vx, sizeof(double) * B * (T + 1) * n, hipMemcpyDeviceToHost));
  for (int b = 0; b < s->B; ++b) {
    const TrajState& st = s->h_st[b];
    if (dgdq) {
      dgdq[2 * b] = st.dg_u + (st.is_feasible ? 0.0 : st.dg_f);
      dgdq[2 * b + 1] = st.dq_u + (st.is_feasible ? 0.0 : st.dq_f);
    }
    if (ok) ok[b] = st.bwd_failed ? 0 : 1;
  }
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}

int empc_rollout_batch(EmpcSolver* s, double alpha, int ddp, int is_feasible, double* xs_try, double* us_try,
                       double* cost_try, int* ok) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  s->use();
  flush_problem(s);
  int ai = -1;
  for (int i = 0; i < s->NA; ++i)
    if (std::ldexp(1.0, -i) == alpha) ai = i;
  if (ai < 0) throw std::invalid_argument("alpha must be one of 2^-n, n < n_alphas");
  // keep the scalars the backward pass left on the device; only flip the flags the rollout reads
  download_states(s);
  for (int b = 0; b < s->B; ++b) {
    TrajState& st = s->h_st[b];
    st.phase = ddp ? PHASE_DDP : 0;
    st.is_feasible = is_feasible;
    st.need_lin = 0;
    st.need_calc = 0;
    st.bwd_failed = 0;
  }
  upload_states(s);
  HIP_CHECK(hipEventRecord(s->ev[0], s->stream));
  s->kt.rollout(s->D, s->stream);
  HIP_CHECK(hipEventRecord(s->ev[1], s->stream));
  HIP_CHECK(hipStreamSynchronize(s->stream));
  HIP_CHECK(hipGetLastError());
  std::memset(&s->stats, 0, sizeof(s->stats));
  timed(s, 0, s->stats.ms_rollout);
  s->stats.n_rollout = 1;
  s->stats.rollout_units = (long long)s->B * s->NA * (s->T + 1);
  const size_t B = s->B, T = s->T, NA = s->NA, nx = s->kt.nx, nu = s->kt.nu;
  std::vector<double> c(B * NA);
  std::vector<int> o(B * NA);
  HIP_CHECK(hipMemcpy(c.data(), s->D.try_cost, sizeof(double) * B * NA, hipMemcpyDeviceToHost));
  HIP_CHECK(hipMemcpy(o.data(), s->D.try_ok, sizeof(int) * B * NA, hipMemcpyDeviceToHost));
  for (size_t b = 0; b < B; ++b) {
    const size_t slot = b * NA + ai;
    if (xs_try)
      HIP_CHECK(hipMemcpy(xs_try + b * (T + 1) * nx, s->D.xs_try + slot * (T + 1) * nx, sizeof(double) * (T + 1) * nx, hipMemcpyDeviceToHost));
    if (us_try)
      HIP_CHECK(hipMemcpy(us_try + b * T * nu, s->D.us_try + slot * T * nu, sizeof(double) * T * nu, hipMemcpyDeviceToHost));
    if (cost_try) cost_try[b] = c[slot];
    if (ok) ok[b] = o[slot];
  }
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}

}  // extern "C"
