// empc_solver.hip -- HIP kernels (gfx950) + host driver + solver part of the C ABI (include/empc.h).
//
// One solve = a sequence of "sweeps"; in every sweep each still-active trajectory advances by exactly one DDP
// iteration of its current pass (FDDP pass, or the DDP clean-up):
//     calc (phase starts only) -> linearize -> backward -> rollout (all step lengths at once) -> select
// All per-trajectory solver state lives on the device (TrajState); the host only reads back the number of
// trajectories that are still active after each sweep.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/empc.h"
#include "empc_internal.hpp"
#include "empc_prep.hpp"
#include "empc_linearize2.hpp"
#include "empc_backward2.hpp"
#include "empc_backward3.hpp"

using namespace empc;

#define HIP_CHECK(expr)                                                                                   \
  do {                                                                                                    \
    hipError_t _e = (expr);                                                                               \
    if (_e != hipSuccess)                                                                                 \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr);        \
  } while (0)

// --------------------------------------------------------------------------------------------------------------------
// kernels
// --------------------------------------------------------------------------------------------------------------------
template <class DM, bool CT>
__global__ void __launch_bounds__(64) k_calc(DevBuffers D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = D.B * (D.T + 1);
  if (idx >= n) return;
  // consecutive lanes = consecutive trajectories of the same node (same cost set -> no divergence)
  const int t = idx / D.B, b = idx % D.B;
  calc_thread<DM, CT>(D, b, t);
}

template <class DM, bool CT>
__global__ void __launch_bounds__(64) k_rollout(DevBuffers D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D.B * D.NA) return;
  const int b = idx / D.NA, ai = idx % D.NA;
  rollout_thread<DM, CT>(D, b, ai);
}

// the shipped rollout: one wavefront per trajectory, lanes = step lengths, feedback product by the whole wave
template <class DM, bool CT>
__global__ void __launch_bounds__(64) k_rollout5(DevBuffers D) {
  extern __shared__ double smem_roll5[];
  static_assert(DM::NU <= 64, "feedback rows must fit one wavefront");
  LaneExec ex{(int)threadIdx.x};
  rollout_wave5<DM, CT>(ex, D, blockIdx.x, 64, smem_roll5);
}

template <class DM, bool CT, int LPU, int BLK, bool FR>
// Two wavefronts per SIMD: at the compiler's own choice (326 registers, one wavefront per SIMD) the kernel sits at
// ~1 resident wave per SIMD with 37 % of its time in waits; capping the budget at 256 registers costs ~250 spilled
// values but doubles the resident waves: 2.02 -> 1.42 ms per launch (profiles/README.md).
#ifndef EMPC_LIN_WAVES
#define EMPC_LIN_WAVES 2
#endif
__global__ void __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(DM::NV > 9 ? 1 : EMPC_LIN_WAVES, DM::NV > 9 ? 1 : EMPC_LIN_WAVES)))
k_linearize(DevBuffers D) {
  extern __shared__ double smem_lin[];
  constexpr int UPB = BLK / LPU;  // units per block
  constexpr int USZ = CT ? Lin2Smem<DM>::SIZE : Lin2Smem<DM>::SIZE_NC;
#ifdef EMPC_LIN_NO_ROLES
  constexpr int RW = 0;
#else
  constexpr int RW = (BLK / 64 >= 3) ? 3 : (BLK / 64 == 2 ? 2 : 0);  // wavefronts that share the single-lane sections
#endif
  // this body's knots: the lean group or the rest of the sorted knot list; a block holds UPB trajectories of ONE knot
  const int k0 = FR ? D.n_lean : 0, nk = FR ? (D.T + 1 - D.n_lean) : D.n_lean;
  const int bpk = (D.B + UPB - 1) / UPB;
  const int kn = blockIdx.x / bpk;
  if (kn >= nk) return;
  const int t = EMPC_KPTR(int, D.lin_knots)[k0 + kn];
  const int i0 = (blockIdx.x % bpk) * UPB;  // position in the list of trajectories that linearize in this sweep
  const int nlist = D.lin_list ? *D.lin_count : D.B;
  if (i0 >= nlist) return;
  const int u = threadIdx.x / LPU, lane = threadIdx.x % LPU;
  bool active = i0 + u < nlist;
  const int b = active ? (D.lin_list ? D.lin_list[i0 + u] : i0 + u) : (D.lin_list ? D.lin_list[i0] : i0);
  if (active) {
    const TrajState& st = D.st[b];
    active = !(st.phase == PHASE_DONE || !st.need_lin);
  }
  LaneExec ex{lane};
  if constexpr (RW > 0) {
    if (!__syncthreads_or(active ? 1 : 0)) return;  // nothing to do in the whole block
    const LinRole R{(int)threadIdx.x, UPB, USZ, i0, nlist, D.lin_list, smem_lin, active};
    linearize_unit2<DM, CT, FR, LaneExec, RW>(ex, D, b, t, LPU, smem_lin + (size_t)u * USZ, &R);
  } else {
    if (!active) return;
    linearize_unit2<DM, CT, FR>(ex, D, b, t, LPU, smem_lin + (size_t)u * USZ);
  }
}

// workgroup-wide executor: barriers are real workgroup barriers
struct BlockExec {
  int lane;
  static constexpr int SLOTS = 1;
  template <class F>
  __device__ __forceinline__ void each(F&& f) {
    __builtin_amdgcn_sched_barrier(0);  // keep the scheduler from interleaving stages (register pressure)
    f(lane, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void sync() {
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  }
  template <class F>
  __device__ __forceinline__ bool any(F&& f) {
    return __syncthreads_or(f(lane, 0) ? 1 : 0) != 0;
  }
  // one v_mfma_f64_16x16x4_f64 of the wavefront: acc[im][in] += A-operand a[ia] x B-operand b[ib] (per-lane values)
  template <class A, class B, class C>
  __device__ __forceinline__ void mfma(A& a, int ia, B& b, int ib, C& c, int im, int in) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 v = {c[0][im][in][0], c[0][im][in][1], c[0][im][in][2], c[0][im][in][3]};
    v = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][ia], b[0][ib], v, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) c[0][im][in][r] = v[r];
  }
};

#ifndef EMPC_BWD_NL
#define EMPC_BWD_NL 64
#endif
template <class DM>
__global__ void __launch_bounds__(EMPC_BWD_NL) k_backward(DevBuffers D) {
  extern __shared__ double smem_bwd[];
  const int b = blockIdx.x;
  BlockExec ex{(int)threadIdx.x};
  backward_traj2<DM, EMPC_BWD_NL>(ex, D, b, smem_bwd);
}

// matrix-core form of the backward pass (one wavefront per trajectory)
template <class DM>
__global__ void __launch_bounds__(64) k_backward3(DevBuffers D) {
  extern __shared__ double smem_bwd3[];
  BlockExec ex{(int)threadIdx.x};
  backward_traj3<DM>(ex, D, blockIdx.x, smem_bwd3);
}

template <class DM>
__global__ void __launch_bounds__(64) k_select(DevBuffers D) {
  __shared__ int sh[2];
  const int b = blockIdx.x;
  if (threadIdx.x == 0) {
    int acc_ai, last_ai;
    select_decide<DM>(D, b, acc_ai, last_ai);
    sh[0] = acc_ai;
    sh[1] = last_ai;
    if (D.st[b].phase != PHASE_DONE) {
      atomicAdd(D.n_active, 1);
      if (D.lin_count_out && D.st[b].need_lin) D.lin_list_out[atomicAdd(D.lin_count_out, 1)] = b;
    }
  }
  __syncthreads();
  select_copy<DM>(D, b, sh[0], sh[1], threadIdx.x, blockDim.x);
}

// us_squash[b][t] = sigma(us_last[b][t]) with the trajectory's final smooth (fillSquashedOutputs, src/sbfddp.cpp:479-486)
template <class DM>
__global__ void k_squash_out(DevBuffers D, double* out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = D.B * D.T * DM::NU;
  if (idx >= n) return;
  const int b = idx / (D.T * DM::NU), i = idx % DM::NU;
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  double u = D.us_last[idx], du;
  if (P.use_squash) squash1(D.us_last[idx], P.u_lb[i], P.u_ub[i], D.st[b].smooth, P.prm.smoothsat_power, u, du);
  out[idx] = u;
}

// Plant of the closed-loop MPC runs: x[b] <- RK4(x[b], u[b], dt) repeated nsub times, one lane per plant.
// u == nullptr takes the squashed first control of the last solve (control = solver.us_squash[0], examples/python/mpc.py:60).
template <class DM>
__global__ void __launch_bounds__(64) k_plant_rk4(DevBuffers D, double* x, const double* u, double dt, int nsub) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= D.B) return;
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  double uu[DM::NU], xa[DM::NX], xb[DM::NX];
#pragma unroll
  for (int i = 0; i < DM::NU; ++i) {
    if (u) {
      uu[i] = u[(size_t)b * DM::NU + i];
    } else {
      const double s = D.us_last[(size_t)b * D.T * DM::NU + i];
      double du;
      uu[i] = s;
      if (P.use_squash) squash1(s, P.u_lb[i], P.u_ub[i], D.st[b].smooth, P.prm.smoothsat_power, uu[i], du);
    }
  }
#pragma unroll
  for (int i = 0; i < DM::NX; ++i) xa[i] = x[(size_t)b * DM::NX + i];
  for (int k = 0; k < nsub; ++k) {
    plant_rk4_step<DM>(P, xa, uu, dt, xb);
#pragma unroll
    for (int i = 0; i < DM::NX; ++i) xa[i] = xb[i];
  }
#pragma unroll
  for (int i = 0; i < DM::NX; ++i) x[(size_t)b * DM::NX + i] = xa[i];
}

// --------------------------------------------------------------------------------------------------------------------
// solver object
// --------------------------------------------------------------------------------------------------------------------
struct KernelTable {
  void (*calc)(DevBuffers, hipStream_t);
  void (*linearize)(DevBuffers, hipStream_t);
  void (*backward)(DevBuffers, hipStream_t);
  void (*rollout)(DevBuffers, hipStream_t);
  void (*select)(DevBuffers, hipStream_t);
  void (*squash_out)(DevBuffers, double*, hipStream_t);
  void (*plant)(DevBuffers, double*, const double*, double, int, hipStream_t);
  int nx, ndx, nu, nv, nacc, rec;
  int off[9], ld[5];
};

template <class DM, bool CT>
static void launch_calc(DevBuffers D, hipStream_t s) {
  const int n = D.B * (D.T + 1);
  hipLaunchKernelGGL((k_calc<DM, CT>), dim3((n + 63) / 64), dim3(64), 0, s, D);
}
template <class DM, bool CT, int BLK>
static void launch_linearize_blk(DevBuffers D, hipStream_t s) {
  constexpr int LPU = (3 * DM::NV <= 32) ? 32 : 64;
  constexpr int UPB = BLK / LPU;
  constexpr int USZ = CT ? Lin2Smem<DM>::SIZE : Lin2Smem<DM>::SIZE_NC;
  const int n = D.B * (D.T + 1);
  const size_t smem = sizeof(double) * USZ * UPB;
  static const bool once = [&] {
    if (getenv("EMPC_DEBUG_OCC")) {
      int nb = -1;
      hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_linearize<DM, CT, LPU, BLK, false>, BLK, smem);
      hipFuncAttributes fa;
      hipError_t e2 = hipFuncGetAttributes(&fa, (const void*)k_linearize<DM, CT, LPU, BLK, false>);
      fprintf(stderr, "[empc] k_linearize BLK=%d dyn smem=%zu B: max active blocks/CU=%d (%s); regs=%d static smem=%zu local=%zu (%s)\n", BLK,
              smem, nb, hipGetErrorString(e), fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes, hipGetErrorString(e2));
    }
    return true;
  }();
  (void)once;
  // lean body over the knots without operational frames, full body over the rest; every unit runs in exactly one of them
  const int bpk = (D.B + UPB - 1) / UPB;  // blocks per knot: a block never straddles two knots
  const int n_lean = bpk * D.n_lean, n_full = bpk * (D.T + 1 - D.n_lean);
  if (n_lean > 0) hipLaunchKernelGGL((k_linearize<DM, CT, LPU, BLK, false>), dim3(n_lean), dim3(BLK), smem, s, D);
  if (n_full > 0) hipLaunchKernelGGL((k_linearize<DM, CT, LPU, BLK, true>), dim3(n_full), dim3(BLK), smem, s, D);
}
template <class DM, bool CT>
static void launch_linearize(DevBuffers D, hipStream_t s) {
  static const int blk = [] {
    const char* e = getenv("EMPC_LIN_BLOCK");
    return e ? atoi(e) : 256;  // 4 wavefronts: chain | Euler step | state differences on their own wavefronts (LinRole)
  }();
  if (blk == 64)
    launch_linearize_blk<DM, CT, 64>(D, s);
  else if (blk == 128)
    launch_linearize_blk<DM, CT, 128>(D, s);
  else
    launch_linearize_blk<DM, CT, 256>(D, s);
}
template <class DM>
static void launch_backward(DevBuffers D, hipStream_t s) {
  static const int version = [] {
    const char* e = getenv("EMPC_BACKWARD");  // 2 = vector form, 3 = matrix-core form
    return e ? atoi(e) : 3;
  }();
  if (version == 2)
    hipLaunchKernelGGL(k_backward<DM>, dim3(D.B), dim3(EMPC_BWD_NL), sizeof(double) * Bwd2Smem<DM>::SIZE, s, D);
  else
    hipLaunchKernelGGL(k_backward3<DM>, dim3(D.B), dim3(64), sizeof(double) * Bwd3Smem<DM>::SIZE, s, D);
}
template <class DM, bool CT>
static void launch_rollout(DevBuffers D, hipStream_t s) {
  const int n = D.B * D.NA;
  static const int version = [] {
    const char* e = getenv("EMPC_ROLLOUT");  // 1 = per-lane form (also the fallback for more than MAX_ALPHAS step lengths)
    return e ? atoi(e) : 5;
  }();
  if (version == 1 || D.NA > MAX_ALPHAS) {
    hipLaunchKernelGGL((k_rollout<DM, CT>), dim3((n + 63) / 64), dim3(64), 0, s, D);
  } else {
    hipLaunchKernelGGL((k_rollout5<DM, CT>), dim3(D.B), dim3(64), sizeof(double) * Roll5Smem<DM>::SIZE, s, D);
  }
}
template <class DM>
static void launch_select(DevBuffers D, hipStream_t s) {
  hipLaunchKernelGGL(k_select<DM>, dim3(D.B), dim3(64), 0, s, D);
}
template <class DM>
static void launch_squash_out(DevBuffers D, double* out, hipStream_t s) {
  const int n = D.B * D.T * DM::NU;
  hipLaunchKernelGGL(k_squash_out<DM>, dim3((n + 255) / 256), dim3(256), 0, s, D, out);
}
template <class DM>
static void launch_plant(DevBuffers D, double* x, const double* u, double dt, int nsub, hipStream_t s) {
  hipLaunchKernelGGL(k_plant_rk4<DM>, dim3((D.B + 63) / 64), dim3(64), 0, s, D, x, u, dt, nsub);
}
template <class DM, bool CT>
static KernelTable make_table() {
  KernelTable k;
  k.calc = launch_calc<DM, CT>;
  k.linearize = launch_linearize<DM, CT>;
  k.backward = launch_backward<DM>;
  k.rollout = launch_rollout<DM, CT>;
  k.select = launch_select<DM>;
  k.squash_out = launch_squash_out<DM>;
  k.plant = launch_plant<DM>;
  k.nx = DM::NX;
  k.ndx = DM::NDX;
  k.nu = DM::NU;
  k.nv = DM::NV;
  k.nacc = DM::NACC;
  k.rec = DM::REC;
  const int off[9] = {DM::OFF_FX, DM::OFF_FU, DM::OFF_LXX, DM::OFF_LXU, DM::OFF_LUU, DM::OFF_LX, DM::OFF_LU, DM::OFF_GAP, DM::OFF_COST};
  std::memcpy(k.off, off, sizeof(off));
  const int ld[5] = {DM::NM, DM::NM, DM::NM, DM::NM, DM::NU};  // Fx, Fu, Lxx, Lxu, Luu leading dimensions
  std::memcpy(k.ld, ld, sizeof(ld));
  return k;
}
static bool find_table(int nb, int nrot, bool contact, KernelTable& k) {
  if (nb == 1 && nrot == 6 && !contact) k = make_table<Dims<1, 6>, false>();
  else if (nb == 4 && nrot == 6 && !contact) k = make_table<Dims<4, 6>, false>();
  else if (nb == 4 && nrot == 6 && contact) k = make_table<Dims<4, 6>, true>();
  else if (nb == 6 && nrot == 6 && !contact) k = make_table<Dims<6, 6>, false>();
  else return false;
  return true;
}

struct EmpcSolver {
  HostProblem H;
  KernelTable kt;
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
  int* dknot = nullptr;
  int* dlin_knots = nullptr;
  int* dlin_list = nullptr;  // [2][B] linearize lists of the two sweep slots
  double* dscratch = nullptr;  // output staging (squashed controls)
  double* dplant_x = nullptr;  // [B][NX] plant states of closed-loop runs (empc_plant_*)
  double* dplant_u = nullptr;  // [B][NU] staging of caller-supplied plant controls
  int* h_active = nullptr;     // pinned
  std::vector<TrajState> h_st;
  bool have_state = false;
  EmpcSolveStats stats;
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
  HIP_CHECK(hipMemcpyAsync(s->dknot, s->H.knot_set.data(), sizeof(int) * s->H.knot_set.size(), hipMemcpyHostToDevice, s->stream));
  {
    std::vector<int> order;
    s->D.n_lean = group_linearize_knots(s->H, order);
    HIP_CHECK(hipMemcpyAsync(s->dlin_knots, order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice, s->stream));
    HIP_CHECK(hipStreamSynchronize(s->stream));  // `order` is a local
  }
  HIP_CHECK(hipStreamSynchronize(s->stream));
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
  if (problem->has_contact)
    for (const auto& cs : s->H.sets)
      for (int i = 0; i < cs.ncontacts; ++i)
        if (cs.contacts[i].type != EMPC_CONTACT_3D)
          throw std::runtime_error("only ContactModel3D is implemented on the device");
  if (!find_table(problem->model.nbodies, problem->n_rotors, problem->has_contact != 0, s->kt)) {
    delete s;
    empc::set_last_error("no kernel instantiation for this (bodies, rotors, contact) combination");
    return nullptr;
  }
  s->use();
  s->B = batch;
  s->T = problem->T;
  s->NA = prm.n_alphas;
  HIP_CHECK(hipStreamCreate(&s->stream));
  for (auto& e : s->ev) HIP_CHECK(hipEventCreate(&e));
  HIP_CHECK(hipHostMalloc((void**)&s->h_active, sizeof(int) * 2 * EmpcSolver::MAX_STREAMS));
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
  s->dknot = s->dalloc<int>(T + 1);
  DevBuffers& D = s->D;
  D.P = s->dP;
  D.sets = s->dsets;
  D.knot_set = s->dknot;
  s->dlin_knots = s->dalloc<int>(T + 1);
  D.lin_knots = s->dlin_knots;
  D.n_lean = T + 1;
  D.st = s->dalloc<TrajState>(B);
  D.x0 = s->dalloc<double>(B * k.nx);
  D.xs = s->dalloc<double>(B * (T + 1) * k.nx);
  D.us = s->dalloc<double>(B * T * k.nu);
  D.acc = s->dalloc<double>(B * (T + 1) * k.nacc);
  D.tape = s->dalloc<double>(B * (T + 1) * k.rec + 64);  // + one wavefront of slack: the backward pass prefetches whole 64-double rows
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
  D.us_last = s->dalloc<double>(B * T * k.nu);
  D.n_active = s->dalloc<int>(4 * EmpcSolver::MAX_STREAMS);  // per chunk and sweep slot: {active trajectories, entries of the linearize list}
  s->dlin_list = s->dalloc<int>(2 * (size_t)batch);          // per sweep slot: the linearize list of every chunk, chunk after chunk
  D.dbg = s->dalloc<unsigned long long>(64);
  HIP_CHECK(hipMemsetAsync(D.dbg, 0, 64 * sizeof(unsigned long long), s->stream));
  D.B = batch;
  D.T = s->T;
  D.NA = s->NA;
  D.gaptol = std::max(prm.th_gaptol, 1e-13);
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
  if (rec_doubles) *rec_doubles = s->kt.rec;
  return EMPC_OK;
}
int empc_tape_layout(const EmpcSolver* s, EmpcTapeLayout* l) {
  if (!s || !l) return EMPC_ERR_INVALID;
  l->rec = s->kt.rec;
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

int empc_solver_update_problem(EmpcSolver* s, const EmpcProblemDesc* problem) {
  EMPC_TRY
  if (!s || !problem) throw std::invalid_argument("NULL argument");
  if (problem->T != s->T || problem->nx != s->kt.nx || problem->nu != s->kt.nu || problem->n_sets != (int)s->H.sets.size())
    throw std::invalid_argument("update_problem: shapes differ from the problem the solver was created with");
  s->use();
  const EmpcSolverParams prm = s->H.P.prm;
  prepare_problem(*problem, prm, s->H);
  upload_problem(s);
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
  D.us_last += (size_t)b0 * T * k.nu;
  D.n_active = s->D.n_active + 4 * idx;
  D.B = nb;
  return D;
}

int empc_solver_solve(EmpcSolver* s, int maxiter, int is_feasible) {
  EMPC_TRY
  if (!s) throw std::invalid_argument("solver is NULL");
  if (maxiter < 1) throw std::invalid_argument("maxiter must be >= 1");
  s->use();
  for (int b = 0; b < s->B; ++b) {
    TrajState prev = s->h_st[b];
    init_traj_state(s->h_st[b], s->H.P.prm, maxiter, is_feasible != 0, s->have_state ? &prev : nullptr);
  }
  upload_states(s);
  HIP_CHECK(hipStreamSynchronize(s->stream));
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
    chunks[c].D = chunk_view(s, lo, hi - lo, c);
    chunks[c].stream = s->streams[c];
    for (int q = 0; q < 2; ++q)
      for (int e = 0; e < 7; ++e) chunks[c].ev[q][e] = s->cev[c][q][e];
  }
  hipEvent_t t_begin, t_end;
  HIP_CHECK(hipEventCreate(&t_begin));
  HIP_CHECK(hipEventCreate(&t_end));
  HIP_CHECK(hipEventRecord(t_begin, s->stream));
  const int hard_cap = 4 * (3 * (maxiter + 1) + 8);  // (passes + clean-up) x maxiter can never be exceeded
  // Every chunk runs its own sweep loop on its own stream, two sweeps deep: sweep k + 1 is queued before the host has
  // seen the active count of sweep k (its kernels return at once for finished trajectories), so the device never waits
  // for the host round trip between sweeps; the one surplus sweep at the end is empty and is not counted.
  // Chunks start staggered -- chunk c waits for chunk c-1's first backward pass.
  std::vector<int> queued(nchunks, 0), retired(nchunks, 0);
  auto enqueue = [&](Chunk& c) {
    const int q = queued[c.idx] & 1;
    DevBuffers Dq = c.D;
    Dq.n_active = c.D.n_active + 2 * q;
    Dq.lin_count_out = Dq.n_active + 1;
    Dq.lin_list_out = s->dlin_list + (size_t)q * s->B + c.b0;
    if (queued[c.idx] > 0) {  // the list written by the previous sweep's select; the first sweep takes every trajectory
      Dq.lin_count = c.D.n_active + 2 * (1 - q) + 1;
      Dq.lin_list = s->dlin_list + (size_t)(1 - q) * s->B + c.b0;
    }
    HIP_CHECK(hipMemsetAsync(Dq.n_active, 0, 2 * sizeof(int), c.stream));
    HIP_CHECK(hipEventRecord(c.ev[q][0], c.stream));
    k.calc(Dq, c.stream);
    HIP_CHECK(hipEventRecord(c.ev[q][1], c.stream));
    k.linearize(Dq, c.stream);
    HIP_CHECK(hipEventRecord(c.ev[q][2], c.stream));
    k.backward(Dq, c.stream);
    HIP_CHECK(hipEventRecord(c.ev[q][3], c.stream));
    k.rollout(Dq, c.stream);
    HIP_CHECK(hipEventRecord(c.ev[q][4], c.stream));
    k.select(Dq, c.stream);
    HIP_CHECK(hipEventRecord(c.ev[q][5], c.stream));
    HIP_CHECK(hipMemcpyAsync(s->h_active + 2 * c.idx + q, Dq.n_active, sizeof(int), hipMemcpyDeviceToHost, c.stream));
    HIP_CHECK(hipEventRecord(c.ev[q][6], c.stream));
    queued[c.idx]++;
  };
  auto retire = [&](Chunk& c) {  // oldest in-flight sweep of the chunk
    const int q = retired[c.idx] & 1;
    HIP_CHECK(hipEventSynchronize(c.ev[q][6]));
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
    S.linearize_units += (long long)c.active * (s->T + 1);  // upper bound: trajectories that re-linearize this sweep
    c.active = s->h_active[2 * c.idx + q];
    retired[c.idx]++;
  };
  for (auto& c : chunks) {
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
      const int act = c.active;
      c.active = 0;  // no units processed
      retire(c);
      c.active = act;
    }
  }
  for (int c = 1; c < nchunks; ++c) HIP_CHECK(hipStreamSynchronize(chunks[c].stream));
  HIP_CHECK(hipEventRecord(t_end, s->stream));
  download_states(s);
  float ms = 0;
  HIP_CHECK(hipEventElapsedTime(&ms, t_begin, t_end));
  S.ms_total = ms;
  (void)hipEventDestroy(t_begin);
  (void)hipEventDestroy(t_end);
  s->have_state = true;
  for (int b = 0; b < s->B; ++b) {
    S.total_iters += s->h_st[b].total_iters;
    S.max_iters = std::max(S.max_iters, s->h_st[b].total_iters);
  }
  if (total_active > 0) throw std::runtime_error("solve did not terminate within the sweep cap (internal error)");
  return EMPC_OK;
  EMPC_CATCH(RET_INT)
}

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
  s->kt.squash_out(s->D, s->dscratch, s->stream);
  HIP_CHECK(hipMemcpy2DAsync(dst_device, row * sizeof(double), s->D.xs, nxs * sizeof(double), nxs * sizeof(double), B,
                             hipMemcpyDeviceToDevice, s->stream));
  HIP_CHECK(hipMemcpy2DAsync(dst_device + nxs, row * sizeof(double), s->dscratch, nus * sizeof(double), nus * sizeof(double), B,
                             hipMemcpyDeviceToDevice, s->stream));
  std::vector<double> tail(2 * B);
  for (size_t b = 0; b < B; ++b) {
    tail[2 * b] = s->h_st[b].cost;
    tail[2 * b + 1] = (double)s->h_st[b].iter;
  }
  HIP_CHECK(hipMemcpy2DAsync(dst_device + nxs + nus, row * sizeof(double), tail.data(), 2 * sizeof(double), 2 * sizeof(double), B,
                             hipMemcpyHostToDevice, s->stream));
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
// diagnostic builds only (-DEMPC_STAMPS): per-stage cycle counters written by trajectory 0
int empc_solver_debug_counters(EmpcSolver* s, unsigned long long* out, int n) {
  if (!s || !out || n > 64) return EMPC_ERR_INVALID;
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
  if (xs || us) {
    if (empc_solver_set_warmstart(s, xs, us) != EMPC_OK) throw std::runtime_error(empc_last_error());
  }
  phase_setup(s, smooth, is_feasible, s->H.P.prm.reg_init, false, true);
  HIP_CHECK(hipEventRecord(s->ev[0], s->stream));
  s->kt.calc(s->D, s->stream);
  HIP_CHECK(hipEventRecord(s->ev[1], s->stream));
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
    HIP_CHECK(hipMemcpyAsync(tape, s->D.tape, sizeof(double) * n, hipMemcpyDeviceToHost, s->stream));
    HIP_CHECK(hipStreamSynchronize(s->stream));
  }
  if (cost) {
    std::vector<double> h(n);
    HIP_CHECK(hipMemcpy(h.data(), s->D.tape, sizeof(double) * n, hipMemcpyDeviceToHost));
    for (int b = 0; b < s->B; ++b) {
      double c = 0;
      for (int t = 0; t <= s->T; ++t) c += h[((size_t)b * (s->T + 1) + t) * s->kt.rec + s->kt.off[8]];
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
