// empc_launch.hpp -- the HIP kernels (gfx950), their launchers and the per-robot kernel table.
//
// Every (bodies, rotors, contact) combination is instantiated in its own translation unit (empc_inst_*.hip) so that the
// instantiations compile in parallel; empc_solver.hip (host driver + C ABI) only sees the table accessors declared
// below.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/empc.h"
#include "empc_internal.hpp"
#include "empc_prep.hpp"
#ifdef EMPC_INSTANTIATE  // kernel bodies are only needed where a table is instantiated
#include "empc_linearize2.hpp"
#include "empc_backward4.hpp"
#include "empc_rollout6.hpp"
#endif
#include "empc_rk4.hpp"

using namespace empc;

// launchers of one (bodies, rotors, contact) instantiation
struct KernelTable {
  void (*calc)(DevBuffers, hipStream_t);
  void (*linearize)(DevBuffers, hipStream_t);
  void (*backward)(DevBuffers, hipStream_t);
  void (*rollout)(DevBuffers, hipStream_t);
  void (*rk4_linearize)(DevBuffers, Rk4Buffers, hipStream_t);  // IntegratedActionModelRK4 nodes: stages, raw records, assembly
  void (*select)(DevBuffers, hipStream_t);
  void (*squash_out)(DevBuffers, double*, hipStream_t);
  void (*pack_rows)(DevBuffers, double*, hipStream_t);
  void (*plant)(DevBuffers, double*, const double*, double, int, hipStream_t);
  int nx, ndx, nu, nv, nacc, rec;
  int off[9], ld[5];  // the FULL layout: what a record looks like outside the device (empc_tape_layout)
  int rec_full, dev_off_cost;  // EMPC_REC_TRI: `rec` is the device stride, records leave the device through `unpack` in rec_full doubles
  void (*unpack)(const double*, double*);
};


// (bodies, rotors) = (1,4) iris | (1,6) hexacopter370 / hextilt | (3,6) hexacopter680_flying_arm_2 |
// (4,6) hexacopter370_flying_arm_3, free and contact dynamics | (6,6) hextilt_flying_arm_5, free and contact dynamics
KernelTable empc_table_1_4();
KernelTable empc_table_1_6();
KernelTable empc_table_3_6();
KernelTable empc_table_1_4_contact();  // (both contact types behind a branch: CT_MIXED)
KernelTable empc_table_1_6_contact();
KernelTable empc_table_3_6_contact();
KernelTable empc_table_4_6();
KernelTable empc_table_4_6_contact();
KernelTable empc_table_4_6_contact6();
KernelTable empc_table_4_6_contact_mixed();
KernelTable empc_table_6_6();
KernelTable empc_table_6_6_contact();
KernelTable empc_table_6_6_contact6();
KernelTable empc_table_6_6_contact_mixed();  // (opt-in: EMPC_EXPERIMENTAL_CONTACT)
KernelTable empc_table_4_6_contact_pair();   // two ContactModel3D per stage (CT_PAIR3; opt-in: EMPC_EXPERIMENTAL_CONTACT)
KernelTable empc_table_6_6_contact_pair();
// instantiations over the baked constants of a shipped robot (csrc/baked/, tools/bake_models.py): picked by find_table when
// the problem's model and platform equal the baked tables bit for bit
KernelTable empc_table_baked_arm3();
KernelTable empc_table_baked_arm3_contact();
KernelTable empc_table_baked_arm5();
KernelTable empc_table_baked_arm2();
KernelTable empc_table_baked_hex370();
KernelTable empc_table_baked_hextilt();
KernelTable empc_table_baked_iris();
KernelTable empc_table_baked_iris_px4();

#ifdef EMPC_INSTANTIATE
// --------------------------------------------------------------------------------------------------------------------
// kernels
// --------------------------------------------------------------------------------------------------------------------
template <class DM, int CT>
__global__ void __launch_bounds__(64) k_calc(DevBuffers D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  // consecutive lanes = consecutive trajectories of the same node (same cost set -> no divergence); with the list of the
  // trajectories that start a pass (written by the previous sweep's select) only those are walked
  const int nb = D.calc_list ? *D.calc_count : D.B;
  if (idx >= nb * (D.T + 1)) return;
  const int t = idx / nb, i = idx % nb;
  calc_thread<DM, CT>(D, D.calc_list ? D.calc_list[i] : i, t);
}

template <class DM, int CT>
__global__ void __launch_bounds__(64) k_rollout(DevBuffers D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D.B * D.NA) return;
  const int b = idx / D.NA, ai = idx % D.NA;
  rollout_thread<DM, CT>(D, b, ai);
}

// the shipped rollout: G trajectories x NA step lengths per wavefront, four role wavefronts per workgroup (empc_rollout6.hpp)
struct RoleExec {
  int lane;
  static constexpr int SLOTS = 1;
  template <class F>
  __device__ __forceinline__ void each(F&& f) {
    f(lane, 0);
  }
  // Workgroup barrier that orders LDS traffic only: the roles hand everything over through LDS, while their global
  // loads (next knot's nominal data) and stores (trial trajectories) stay in flight across it -- __syncthreads() would
  // drain them (vmcnt(0)) twice per knot.
  __device__ __forceinline__ void sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
};
// EMPC_ROLL_WAVES / EMPC_BWD_WAVES (build-time, default 1): resident wavefronts per SIMD the chain kernels are compiled for.
// 2 caps their register budget at 256 (they take 280-400 at the compiler's choice): the occupancy experiment of
// profiles/r04_slots_sweep.md -- measured slower at 1024 slots and no better per 1024 at 2048 / 4096, so 1 is shipped.
#ifndef EMPC_ROLL_WAVES
#define EMPC_ROLL_WAVES 1
#endif
#ifndef EMPC_BWD_WAVES
#define EMPC_BWD_WAVES 1
#endif
template <class DM, int CT, bool RK4>
__global__ void __launch_bounds__(64 * R6_WAVES) __attribute__((amdgpu_waves_per_eu(EMPC_ROLL_WAVES, EMPC_ROLL_WAVES))) k_rollout6(DevBuffers D) {
  extern __shared__ double smem_roll6[];
  RoleExec ex{(int)(threadIdx.x & 63)};
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: the role switch is a scalar branch
  if (wave == R6_A)
    rollout_group6<DM, CT, R6_A, RoleExec, RK4>(ex, D, blockIdx.x, smem_roll6);
  else if (wave == R6_B)
    rollout_group6<DM, CT, R6_B, RoleExec, RK4>(ex, D, blockIdx.x, smem_roll6);
  else if (wave == R6_C)
    rollout_group6<DM, CT, R6_C, RoleExec, RK4>(ex, D, blockIdx.x, smem_roll6);
  else
    rollout_group6<DM, CT, R6_D, RoleExec, RK4>(ex, D, blockIdx.x, smem_roll6);
}

// constraint rows of the contact of knot t (3 when the knot has none)
__device__ __forceinline__ int knot_contact_rows(const DevBuffers& D, int t) {
  const EMPC_K EmpcCostSet& set = EMPC_KPTR(EmpcCostSet, D.sets)[EMPC_KPTR(int, D.knot_set)[t]];
  return (set.ncontacts > 0 && set.contacts[0].type == EMPC_CONTACT_6D) ? 6 : 3;
}
// CT_PAIR3 problems: does knot t hold two contacts (the six-row body) or one ContactModel3D / none (the three-row body)
__device__ __forceinline__ bool knot_contact_pair(const DevBuffers& D, int t) {
  const EMPC_K EmpcCostSet& set = EMPC_KPTR(EmpcCostSet, D.sets)[EMPC_KPTR(int, D.knot_set)[t]];
  return set.ncontacts > 1;
}
template <class DM, int CT, int LPU, int BLK, bool FR>
// Two wavefronts per SIMD: at the compiler's own choice (326 registers, one wavefront per SIMD) the kernel sits at
// ~1 resident wave per SIMD with 37 % of its time in waits; capping the budget at 256 registers costs ~250 spilled
// values but doubles the resident waves: 2.02 -> 1.42 ms per launch (profiles/README.md).
#ifndef EMPC_LIN_WAVES
#define EMPC_LIN_WAVES 2
#endif
#ifndef EMPC_LIN_WAVES_BIG
#define EMPC_LIN_WAVES_BIG 1
#endif
__device__ __forceinline__ void lin_block(const DevBuffers& D, const int block, double* smem_lin) {
  constexpr int UPB = BLK / LPU;  // units per block
  constexpr int USZ = Lin2Smem<DM>::size_for(CT);
#ifdef EMPC_LIN_NO_ROLES
  constexpr int RW = 0;
#else
  constexpr int RW = (BLK / 64 >= 3) ? 3 : (BLK / 64 == 2 ? 2 : 0);  // wavefronts that share the single-lane sections
#endif
  // this body's knots: the lean group or the rest of the sorted knot list; a block holds UPB trajectories of ONE knot
  const int k0 = FR ? D.n_lean : 0, nk = FR ? (D.T + 1 - D.n_lean) : D.n_lean;
  const int bpk = ((D.lin_bound > 0 ? D.lin_bound : D.B) + UPB - 1) / UPB;
  const int kn = block / bpk;
  if (kn >= nk) return;
  const int t = EMPC_KPTR(int, D.lin_knots)[k0 + kn];
  const int i0 = (block % bpk) * UPB;  // position in the list of trajectories that linearize in this sweep
  const int nlist = D.lin_list ? *D.lin_count : D.B;
  if (i0 >= nlist) return;
  int u = threadIdx.x / LPU;
  const int lane = threadIdx.x % LPU;
  // A unit that fills its wavefront (LPU = 64: the 11-dof class with contact dynamics) has a wave-uniform unit index: say so,
  // and the trajectory index, the record pointer and the unit's LDS block live in scalar registers.  (Besides the registers
  // it saves, this keeps those long-lived values out of the vector-register live-range splitting that, under the pressure
  // of these instantiations, left the record pointer defined in lane 0 only -- LABNOTES.md, round 4, "GPU memory fault".)
  if constexpr (LPU == 64) u = __builtin_amdgcn_readfirstlane(u);
  bool active = i0 + u < nlist;
  int b = active ? (D.lin_list ? D.lin_list[i0 + u] : i0 + u) : (D.lin_list ? D.lin_list[i0] : i0);
  if constexpr (LPU == 64) b = __builtin_amdgcn_readfirstlane(b);
  if (active) {
    const TrajState& st = D.st[b];
    active = !(st.phase == PHASE_DONE || !st.need_lin);
  }
  LaneExec ex{lane};
  if constexpr (RW > 0) {
    if (!__syncthreads_or(active ? 1 : 0)) return;  // nothing to do in the whole block
    const LinRole R{(int)threadIdx.x, UPB, USZ, i0, nlist, D.lin_list, smem_lin, active};
    if constexpr (CT == CT_MIXED) {
      // stages of both contact types in one problem: all units of a block are trajectories of ONE knot, so the type of
      // its contact is uniform over the workgroup (the barriers inside stay in uniform control flow)
      if (knot_contact_rows(D, t) == 6)
        linearize_unit2<DM, 6, FR, LaneExec, RW>(ex, D, b, t, LPU, smem_lin + (size_t)u * USZ, &R);
      else
        linearize_unit2<DM, 3, FR, LaneExec, RW>(ex, D, b, t, LPU, smem_lin + (size_t)u * USZ, &R);
    } else if constexpr (CT == CT_PAIR3) {
      // (the same argument: one knot per block, so the number of its contacts is uniform over the workgroup)
      if (knot_contact_pair(D, t))
        linearize_unit2<DM, CT_PAIR3, FR, LaneExec, RW>(ex, D, b, t, LPU, smem_lin + (size_t)u * USZ, &R);
      else
        linearize_unit2<DM, 3, FR, LaneExec, RW>(ex, D, b, t, LPU, smem_lin + (size_t)u * USZ, &R);
    } else
      linearize_unit2<DM, CT, FR, LaneExec, RW>(ex, D, b, t, LPU, smem_lin + (size_t)u * USZ, &R);
  } else {
    if (!active) return;
    if constexpr (CT == CT_MIXED) {
      if (knot_contact_rows(D, t) == 6)
        linearize_unit2<DM, 6, FR>(ex, D, b, t, LPU, smem_lin + (size_t)u * USZ);
      else
        linearize_unit2<DM, 3, FR>(ex, D, b, t, LPU, smem_lin + (size_t)u * USZ);
    } else if constexpr (CT == CT_PAIR3) {
      if (knot_contact_pair(D, t))
        linearize_unit2<DM, CT_PAIR3, FR>(ex, D, b, t, LPU, smem_lin + (size_t)u * USZ);
      else
        linearize_unit2<DM, 3, FR>(ex, D, b, t, LPU, smem_lin + (size_t)u * USZ);
    } else
      linearize_unit2<DM, CT, FR>(ex, D, b, t, LPU, smem_lin + (size_t)u * USZ);
  }
}
#define EMPC_LIN_ATTR __attribute__((amdgpu_waves_per_eu(DM::NV > 9 ? EMPC_LIN_WAVES_BIG : EMPC_LIN_WAVES, DM::NV > 9 ? EMPC_LIN_WAVES_BIG : EMPC_LIN_WAVES)))
template <class DM, int CT, int LPU, int BLK, bool FR>
__global__ void __launch_bounds__(BLK) EMPC_LIN_ATTR k_linearize(DevBuffers D) {
  extern __shared__ double smem_lin[];
  lin_block<DM, CT, LPU, BLK, FR>(D, blockIdx.x, smem_lin);
}
// Both bodies in one launch: the first `n_lean_blocks` workgroups run the lean body over the knots without operational
// frames, the rest the full body over the other knots.  One launch instead of two: the short full-body launch no longer
// waits for the last lean workgroup (straggler sweeps: two unit latencies per sweep become one).
template <class DM, int CT, int LPU, int BLK>
__global__ void __launch_bounds__(BLK) EMPC_LIN_ATTR k_linearize_all(DevBuffers D, int n_full_blocks) {
  extern __shared__ double smem_lin[];
  // the full-body units (frame costs, contacts: few and ~1.5x as long) come first in the grid, the lean ones fill in behind them
  if ((int)blockIdx.x < n_full_blocks)
    lin_block<DM, CT, LPU, BLK, true>(D, blockIdx.x, smem_lin);
  else
    lin_block<DM, CT, LPU, BLK, false>(D, blockIdx.x - n_full_blocks, smem_lin);
}

// workgroup-wide executor: barriers are real workgroup barriers
// EMPC_BWD_GLDS: ROWS pieces of 1 KiB (128 doubles) from global memory straight into LDS (LDS-DMA, no register destination): lane l
// of piece q moves the 16 bytes at src_lane + 128 q (src_lane = src + 2 l) to dst + 128 q + 2 l -- the hardware adds lane x 16 to
// the wave-uniform LDS base in M0.  Inline assembly on purpose: hipcc tracks the builtin form (__builtin_amdgcn_global_load_lds) as
// an LDS write and waits vmcnt(0) at the next ds_read of ANY LDS address (checked in the ISA) -- the exposed HBM latency this
// variant exists to hide; an asm load is absent from its bookkeeping and the kernel retires it with s_waitcnt vmcnt(0) one knot
// later.  One statement per record: M0 (compiler-reserved) is saved, advanced by 1 KiB per piece and restored inside it, the
// s_nop separates each M0 write from the instruction that reads it; the source address advances in the same registers.
// operands: %0 saved M0, %1 the lane's source address (advanced in place), %2 scratch SGPR pair holding 0x400, %3 LDS byte address
#define EMPC_GLDS_HEAD "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_mov_b64 %2, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
#define EMPC_GLDS_NEXT "v_lshl_add_u64 %1, %1, 0, %2\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
#define EMPC_GLDS_TAIL "s_mov_b32 m0, %0"
#define EMPC_GLDS_R0 ""
#define EMPC_GLDS_R1 EMPC_GLDS_R0 EMPC_GLDS_NEXT
#define EMPC_GLDS_R2 EMPC_GLDS_R1 EMPC_GLDS_NEXT
#define EMPC_GLDS_R3 EMPC_GLDS_R2 EMPC_GLDS_NEXT
#define EMPC_GLDS_R4 EMPC_GLDS_R3 EMPC_GLDS_NEXT
#define EMPC_GLDS_R5 EMPC_GLDS_R4 EMPC_GLDS_NEXT
#define EMPC_GLDS_R6 EMPC_GLDS_R5 EMPC_GLDS_NEXT
#define EMPC_GLDS_R7 EMPC_GLDS_R6 EMPC_GLDS_NEXT
#define EMPC_GLDS_R8 EMPC_GLDS_R7 EMPC_GLDS_NEXT
#define EMPC_GLDS_R9 EMPC_GLDS_R8 EMPC_GLDS_NEXT
#define EMPC_GLDS_R10 EMPC_GLDS_R9 EMPC_GLDS_NEXT
#define EMPC_GLDS_R11 EMPC_GLDS_R10 EMPC_GLDS_NEXT
#define EMPC_GLDS_R12 EMPC_GLDS_R11 EMPC_GLDS_NEXT
template <int ROWS>
__device__ __forceinline__ void glds_rows(double* dst, const double* src_lane) {
  static_assert(ROWS >= 1 && ROWS <= 13, "pieces of 1 KiB per record (9 for the 9-DoF arm, 8 with EMPC_REC_TRI)");
  const double* g = src_lane;
  const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)dst);
  unsigned keep;
  unsigned long long inc;
  if constexpr (ROWS == 1)
    asm volatile(EMPC_GLDS_HEAD EMPC_GLDS_R0 EMPC_GLDS_TAIL : "=&s"(keep), "+v"(g), "=&s"(inc) : "s"(l) : "memory", "scc");
  else if constexpr (ROWS == 2)
    asm volatile(EMPC_GLDS_HEAD EMPC_GLDS_R1 EMPC_GLDS_TAIL : "=&s"(keep), "+v"(g), "=&s"(inc) : "s"(l) : "memory", "scc");
  else if constexpr (ROWS == 3)
    asm volatile(EMPC_GLDS_HEAD EMPC_GLDS_R2 EMPC_GLDS_TAIL : "=&s"(keep), "+v"(g), "=&s"(inc) : "s"(l) : "memory", "scc");
  else if constexpr (ROWS == 4)
    asm volatile(EMPC_GLDS_HEAD EMPC_GLDS_R3 EMPC_GLDS_TAIL : "=&s"(keep), "+v"(g), "=&s"(inc) : "s"(l) : "memory", "scc");
  else if constexpr (ROWS == 5)
    asm volatile(EMPC_GLDS_HEAD EMPC_GLDS_R4 EMPC_GLDS_TAIL : "=&s"(keep), "+v"(g), "=&s"(inc) : "s"(l) : "memory", "scc");
  else if constexpr (ROWS == 6)
    asm volatile(EMPC_GLDS_HEAD EMPC_GLDS_R5 EMPC_GLDS_TAIL : "=&s"(keep), "+v"(g), "=&s"(inc) : "s"(l) : "memory", "scc");
  else if constexpr (ROWS == 7)
    asm volatile(EMPC_GLDS_HEAD EMPC_GLDS_R6 EMPC_GLDS_TAIL : "=&s"(keep), "+v"(g), "=&s"(inc) : "s"(l) : "memory", "scc");
  else if constexpr (ROWS == 8)
    asm volatile(EMPC_GLDS_HEAD EMPC_GLDS_R7 EMPC_GLDS_TAIL : "=&s"(keep), "+v"(g), "=&s"(inc) : "s"(l) : "memory", "scc");
  else if constexpr (ROWS == 9)
    asm volatile(EMPC_GLDS_HEAD EMPC_GLDS_R8 EMPC_GLDS_TAIL : "=&s"(keep), "+v"(g), "=&s"(inc) : "s"(l) : "memory", "scc");
  else if constexpr (ROWS == 10)
    asm volatile(EMPC_GLDS_HEAD EMPC_GLDS_R9 EMPC_GLDS_TAIL : "=&s"(keep), "+v"(g), "=&s"(inc) : "s"(l) : "memory", "scc");
  else if constexpr (ROWS == 11)
    asm volatile(EMPC_GLDS_HEAD EMPC_GLDS_R10 EMPC_GLDS_TAIL : "=&s"(keep), "+v"(g), "=&s"(inc) : "s"(l) : "memory", "scc");
  else if constexpr (ROWS == 12)
    asm volatile(EMPC_GLDS_HEAD EMPC_GLDS_R11 EMPC_GLDS_TAIL : "=&s"(keep), "+v"(g), "=&s"(inc) : "s"(l) : "memory", "scc");
  else if constexpr (ROWS == 13)
    asm volatile(EMPC_GLDS_HEAD EMPC_GLDS_R12 EMPC_GLDS_TAIL : "=&s"(keep), "+v"(g), "=&s"(inc) : "s"(l) : "memory", "scc");
}

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
  // EMPC_ANY_BALLOT (off: not yet run on hardware): the workgroups of this executor are ONE wavefront, so a ballot answers
  // "any lane?" without the LDS reduction and the barrier __syncthreads_or compiles to (300 cycles per knot of the backward
  // pass).  __syncthreads_or also kept the compiler from moving LDS traffic across it; the ballot form states that itself.
  template <class F>
  __device__ __forceinline__ bool any(F&& f) {
#if EMPC_ANY_BALLOT
    __builtin_amdgcn_sched_barrier(0);
    const bool r = __builtin_amdgcn_ballot_w64(f(lane, 0)) != 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_sched_barrier(0);
    return r;
#else
    return __syncthreads_or(f(lane, 0) ? 1 : 0) != 0;
#endif
  }
  // EMPC_BWD_GLDS: ROWS pieces of 1 KiB from global memory straight into LDS (glds_rows above); async_wait() retires them
  template <int ROWS>
  __device__ __forceinline__ void async_rows(double* dst, const double* src) {
    __builtin_amdgcn_sched_barrier(0);
    glds_rows<ROWS>(dst, src + 2 * lane);
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void async_wait() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
    __builtin_amdgcn_sched_barrier(0);
  }
  // EMPC_BWD_FUSE: wave broadcasts.  bcast(a, j, L): a[.][j] of lane L (a compile-time lane after unrolling) as a wave-uniform
  // value (two v_readlane_b32 -> a scalar register pair); bcast1(a, L): the same for one value per lane; first(a): lane 0's flag
  template <class A>
  __device__ __forceinline__ double bcast(A& a, int j, int L) {
    return rdlane(a[0][j], L);
  }
  template <class A>
  __device__ __forceinline__ double bcast1(A& a, int L) {
    return rdlane(a[0], L);
  }
  template <class A>
  __device__ __forceinline__ bool first(A& a) {
    return __builtin_amdgcn_readfirstlane(a[0] ? 1 : 0) != 0;
  }
  static __device__ __forceinline__ double rdlane(double v, int L) {
    unsigned long long u;
    __builtin_memcpy(&u, &v, 8);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, L), hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), L);
    u = ((unsigned long long)hi << 32) | lo;
    double r;
    __builtin_memcpy(&r, &u, 8);
    return r;
  }
  // EMPC_BWD_MFMA4: one v_mfma_f64_4x4x4_4b_f64: register r of accumulator tile (im, in) += A-operand a[ia] x B-operand b[ib], four
  // independent 4 x 4 x 4 products (block = (lane % 16) / 4)
  template <class A, class B, class C>
  __device__ __forceinline__ void mfma4(A& a, int ia, B& b, int ib, C& c, int im, int in, int r) {
    c[0][im][in][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0][ia], b[0][ib], c[0][im][in][r], 0, 0, 0);
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

// the shipped backward pass: matrix cores, zero-padded LDS tiles (empc_backward4.hpp)
template <class DM, bool BOX>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(EMPC_BWD_WAVES, EMPC_BWD_WAVES))) k_backward4(DevBuffers D) {
  extern __shared__ double smem_bwd4[];
  BlockExec ex{(int)threadIdx.x};
  backward_traj4<DM, BOX>(ex, D, blockIdx.x, smem_bwd4);
}

// The same pass with WPB trajectories per workgroup, one per wavefront and SIMD (EMPC_BWD_WPB=4): the wavefronts share
// nothing (own LDS slice, wavefront-level ordering instead of workgroup barriers), but the workgroup takes a whole CU, so a
// half batch occupies half of the CUs completely instead of half of every CU's SIMDs -- which leaves whole CUs to the
// rollout workgroups (four role wavefronts, one per SIMD) of ANOTHER chunk of the batch running on its own stream.
struct WaveExec {
  int lane;
  static constexpr int SLOTS = 1;
  template <class F>
  __device__ __forceinline__ void each(F&& f) {
    __builtin_amdgcn_sched_barrier(0);
    f(lane, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void sync() {
#if defined(__HIP_DEVICE_COMPILE__)
    wave_sync();  // LDS operations of one wavefront complete in order: only the compiler has to keep them in order
#endif
    __builtin_amdgcn_sched_barrier(0);
  }
  template <class F>
  __device__ __forceinline__ bool any(F&& f) {
    return __builtin_amdgcn_ballot_w64(f(lane, 0)) != 0;
  }
  // EMPC_BWD_GLDS: ROWS pieces of 1 KiB from global memory straight into LDS (glds_rows above); async_wait() retires them
  template <int ROWS>
  __device__ __forceinline__ void async_rows(double* dst, const double* src) {
    __builtin_amdgcn_sched_barrier(0);
    glds_rows<ROWS>(dst, src + 2 * lane);
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void async_wait() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
    __builtin_amdgcn_sched_barrier(0);
  }
  // EMPC_BWD_FUSE: wave broadcasts.  bcast(a, j, L): a[.][j] of lane L (a compile-time lane after unrolling) as a wave-uniform
  // value (two v_readlane_b32 -> a scalar register pair); bcast1(a, L): the same for one value per lane; first(a): lane 0's flag
  template <class A>
  __device__ __forceinline__ double bcast(A& a, int j, int L) {
    return rdlane(a[0][j], L);
  }
  template <class A>
  __device__ __forceinline__ double bcast1(A& a, int L) {
    return rdlane(a[0], L);
  }
  template <class A>
  __device__ __forceinline__ bool first(A& a) {
    return __builtin_amdgcn_readfirstlane(a[0] ? 1 : 0) != 0;
  }
  static __device__ __forceinline__ double rdlane(double v, int L) {
    unsigned long long u;
    __builtin_memcpy(&u, &v, 8);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, L), hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), L);
    u = ((unsigned long long)hi << 32) | lo;
    double r;
    __builtin_memcpy(&r, &u, 8);
    return r;
  }
  // EMPC_BWD_MFMA4: one v_mfma_f64_4x4x4_4b_f64: register r of accumulator tile (im, in) += A-operand a[ia] x B-operand b[ib], four
  // independent 4 x 4 x 4 products (block = (lane % 16) / 4)
  template <class A, class B, class C>
  __device__ __forceinline__ void mfma4(A& a, int ia, B& b, int ib, C& c, int im, int in, int r) {
    c[0][im][in][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0][ia], b[0][ib], c[0][im][in][r], 0, 0, 0);
  }
  template <class A, class B, class C>
  __device__ __forceinline__ void mfma(A& a, int ia, B& b, int ib, C& c, int im, int in) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 v = {c[0][im][in][0], c[0][im][in][1], c[0][im][in][2], c[0][im][in][3]};
    v = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][ia], b[0][ib], v, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) c[0][im][in][r] = v[r];
  }
};
template <class DM, bool BOX, int WPB>
__global__ void __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(1, 1))) k_backward4w(DevBuffers D) {
  extern __shared__ double smem_bwd4w[];
  const int w = (int)threadIdx.x / 64;
  const int b = (int)blockIdx.x * WPB + w;
  if (b >= D.B) return;
  WaveExec ex{(int)threadIdx.x % 64};
  backward_traj4<DM, BOX>(ex, D, b, smem_bwd4w + (size_t)w * Bwd4Smem<DM>::SIZE);
}

template <class DM>
__global__ void __launch_bounds__(256) k_select(DevBuffers D) {
  __shared__ int sh[3];
  const int b = blockIdx.x;
  if (threadIdx.x == 0) {
    if (b == 0 && D.counters_next) {  // every reader of the other slot's counters ran before this kernel
      D.counters_next[0] = 0;
      D.counters_next[1] = 0;
      D.counters_next[2] = 0;
    }
    int acc_ai, last_ai;
    select_decide<DM>(D, b, acc_ai, last_ai);
    sh[0] = acc_ai;
    sh[1] = last_ai;
  }
  __syncthreads();
  select_copy<DM>(D, b, sh[0], sh[1], threadIdx.x, blockDim.x);
  if (D.q_rows) {
    // streamed solves: a trajectory that has just finished hands its row over and the slot takes the next job of the queue
    // (the whole workgroup sees the same state: it was written before the barrier above)
    TrajState& st = D.st[b];
    if (st.phase == PHASE_DONE && st.job >= 0) {
      __syncthreads();  // the accepted candidate is complete in global memory
      stream_write_row<DM>(D, b, threadIdx.x, blockDim.x);
      if (threadIdx.x == 0) {
        atomicAdd(D.q_iters, (unsigned long long)st.total_iters);
        atomicMax(D.q_head + 1, st.total_iters);  // largest iteration count of any job (EmpcSolveStats::max_iters of a stream)
        const int j = atomicAdd(D.q_head, 1);
        sh[2] = j < D.q_njobs ? j : -1;
      }
      __syncthreads();  // the row is out; the next job is known to everybody
      const int job = sh[2];
      if (job >= 0) stream_refill<DM>(D, b, job, threadIdx.x, blockDim.x);
      if (threadIdx.x == 0) {
        if (job >= 0) traj_state_init(st, EMPC_KREF(DevProblem, D.P).prm, D.q_maxiter, false, (const TrajState*)nullptr);
        st.job = job;
      }
    }
  }
  if (threadIdx.x == 0) {
    if (D.st[b].phase != PHASE_DONE) {
      const int pos = atomicAdd(D.n_active, 1);
      if (D.act_list_out) D.act_list_out[pos] = b;
      if (D.lin_count_out && D.st[b].need_lin) D.lin_list_out[atomicAdd(D.lin_count_out, 1)] = b;
      if (D.calc_count_out && D.st[b].need_calc) D.calc_list_out[atomicAdd(D.calc_count_out, 1)] = b;
    }
    if (D.host_active) {
      __threadfence();
      if (atomicAdd(D.done_ticket, 1) == (int)gridDim.x - 1) {  // last workgroup: all counts are in
        D.host_active[0] = atomicAdd(D.n_active, 0);
        D.host_active[1] = D.lin_count_out ? atomicAdd(D.lin_count_out, 0) : 0;
        *D.done_ticket = 0;
        __threadfence_system();
      }
    }
  }
}

// us_squash[b][t] = sigma(us_last[b][t]) with the trajectory's final smooth (fillSquashedOutputs, src/sbfddp.cpp:479-486)
template <class DM>
__global__ void k_squash_out(DevBuffers D, double* out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = D.B * D.T * DM::NU;
  if (idx >= n) return;
  const int b = idx / (D.T * DM::NU), i = idx % DM::NU;
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  // without squashing (the box solvers' problems) there is no squashing data to report: us_squash is the accepted us
  double u = D.us[idx], du;
  if (P.use_squash) squash1(D.us_last[idx], P.u_lb[i], P.u_ub[i], D.st[b].smooth, P.prm.smoothsat_power, u, du);
  out[idx] = u;
}

// One row per rollout: xs | us_squash | cost | iters (as double) -- the payload of the multi-GPU result gather, packed
// entirely on the device (costs and iteration counts come from the device-resident TrajState, not from the host copy).
template <class DM>
__global__ void k_pack_rows(DevBuffers D, double* out) {
  const size_t nxs = (size_t)(D.T + 1) * DM::NX, nus = (size_t)D.T * DM::NU, row = nxs + nus + 2;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)D.B * row) return;
  const int b = (int)(idx / row);
  const size_t j = idx % row;
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  double v;
  if (j < nxs) {
    v = D.xs[(size_t)b * nxs + j];
  } else if (j < nxs + nus) {
    const size_t k = j - nxs;
    const int i = (int)(k % DM::NU);
    double du;
    v = D.us[(size_t)b * nus + k];
    if (P.use_squash) squash1(D.us_last[(size_t)b * nus + k], P.u_lb[i], P.u_ub[i], D.st[b].smooth, P.prm.smoothsat_power, v, du);
  } else {
    v = (j == nxs + nus) ? D.st[b].cost : (double)D.st[b].iter;
  }
  out[idx] = v;
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
      uu[i] = D.us[(size_t)b * D.T * DM::NU + i];
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

template <class DM, int CT>
static void launch_calc(DevBuffers D, hipStream_t s) {
  const int n = D.B * (D.T + 1);
  hipLaunchKernelGGL((k_calc<DM, CT>), dim3((n + 63) / 64), dim3(64), 0, s, D);
}
template <class DM, int CT, int BLK>
static void launch_linearize_blk(DevBuffers D, hipStream_t s) {
  constexpr int LPU = lin_lanes_per_unit<DM, CT>();  // 32; 64 for the 11-dof class with contact dynamics
  constexpr int UPB = BLK / LPU;
  constexpr int USZ = Lin2Smem<DM>::size_for(CT);
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
  // blocks per knot: a block never straddles two knots.  The list of trajectories that linearize is at most lin_bound long
  // (what the host last saw of the active count), so a sweep with few stragglers launches few workgroups instead of
  // B / UPB per knot that return at once
  const int bpk = ((D.lin_bound > 0 ? D.lin_bound : D.B) + UPB - 1) / UPB;
  const int n_lean = bpk * D.n_lean, n_full = bpk * (D.T + 1 - D.n_lean);
  static const int merged = [] {
    const char* e = getenv("EMPC_LIN_MERGED");  // 0 = one launch per body, 1 = one launch when few trajectories are left, 2 = always one launch
    return e ? atoi(e) : 1;
  }();
  if constexpr (BLK == 256) {
    // one launch holding both bodies: always (2), or only when few trajectories are left (1), where the second launch is
    // mostly latency
    if (n_lean > 0 && n_full > 0 && (merged == 2 || (merged == 1 && D.lin_bound > 0 && 2 * D.lin_bound <= D.B))) {
      hipLaunchKernelGGL((k_linearize_all<DM, CT, LPU, BLK>), dim3(n_lean + n_full), dim3(BLK), smem, s, D, n_full);
      return;
    }
  }
  if (n_lean > 0) hipLaunchKernelGGL((k_linearize<DM, CT, LPU, BLK, false>), dim3(n_lean), dim3(BLK), smem, s, D);
  if (n_full > 0) hipLaunchKernelGGL((k_linearize<DM, CT, LPU, BLK, true>), dim3(n_full), dim3(BLK), smem, s, D);
}
template <class DM, int CT>
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
// IntegratedActionModelRK4: stage states -> differential-model records of the 4 B stage trajectories (the linearize kernel
// in RAW mode) -> chain rule per node (empc_rk4.hpp)
template <class DM, int CT>
__global__ void __launch_bounds__(64) k_rk4_stages(DevBuffers D, Rk4Buffers R) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D.B * (D.T + 1)) return;
  const int t = idx / D.B, b = idx % D.B;  // consecutive lanes = trajectories of one node: same cost set
  rk4_stage_thread<DM, CT>(D, R, b, t);
}
template <class DM>
__global__ void __launch_bounds__(64) k_rk4_assemble(DevBuffers D, Rk4Buffers R) {
  extern __shared__ double smem_rk4[];
  BlockExec ex{(int)threadIdx.x};  // one wavefront per node; the products run on the matrix cores
  const int u = blockIdx.x;
  rk4_assemble_unit<DM>(ex, D, R, u % D.B, u / D.B, 64, smem_rk4);
}
template <class DM, int CT>
static void launch_rk4_linearize(DevBuffers D, Rk4Buffers R, hipStream_t s) {
  const int n = D.B * (D.T + 1);
  hipLaunchKernelGGL((k_rk4_stages<DM, CT>), dim3((n + 63) / 64), dim3(64), 0, s, D, R);
  DevBuffers Dv = D;  // the stage batch as the linearize kernel sees it
  Dv.B = 4 * D.B;
  Dv.st = R.st4;
  Dv.xs = R.ys;
  Dv.us = R.us4;
  Dv.acc = R.accs;
  Dv.tape = R.tape4;
  Dv.x0 = R.ys;  // never read in RAW mode
  Dv.lin_list = nullptr;
  Dv.lin_count = nullptr;
  Dv.lin_bound = 0;  // the stage batch is not compacted
  Dv.raw = 1;
  launch_linearize<DM, CT>(Dv, s);
  hipLaunchKernelGGL(k_rk4_assemble<DM>, dim3(n), dim3(64), sizeof(double) * Rk4Smem<DM>::SIZE, s, D, R);
}

template <class DM>
static void launch_backward(DevBuffers D, hipStream_t s) {
  constexpr int WPB = 4;
  constexpr size_t smem4 = sizeof(double) * Bwd4Smem<DM>::SIZE * WPB;
  static const bool cu_exclusive = [] {
    const char* e = getenv("EMPC_BWD_WPB");  // 4 = one workgroup per CU (four trajectories), 1 = one wavefront per workgroup
    const bool on = e && atoi(e) == WPB && smem4 <= 160 * 1024;
    if (on) (void)hipFuncSetAttribute((const void*)k_backward4w<DM, false, WPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem4);
    return on;
  }();
  if (cu_exclusive && D.solver_type == EMPC_SOLVER_SBFDDP) {
    hipLaunchKernelGGL((k_backward4w<DM, false, WPB>), dim3((D.B + WPB - 1) / WPB), dim3(64 * WPB), smem4, s, D);
    return;
  }
  if (D.solver_type != EMPC_SOLVER_SBFDDP)  // the BoxQP gains: their own instantiation
    hipLaunchKernelGGL((k_backward4<DM, true>), dim3(D.B), dim3(64), sizeof(double) * Bwd4Smem<DM>::SIZE, s, D);
  else
    hipLaunchKernelGGL((k_backward4<DM, false>), dim3(D.B), dim3(64), sizeof(double) * Bwd4Smem<DM>::SIZE, s, D);
}
template <class DM, int CT>
static void launch_rollout(DevBuffers D, hipStream_t s) {
  const int n = D.B * D.NA;
  static const int version = [] {
    const char* e = getenv("EMPC_ROLLOUT");  // 6 = packed role-split form (default), 1 = per-lane form
    return e ? atoi(e) : 6;
  }();
  if (version == 1 || D.NA > MAX_ALPHAS) {  // > 16 step lengths: the per-lane form
    hipLaunchKernelGGL((k_rollout<DM, CT>), dim3((n + 63) / 64), dim3(64), 0, s, D);
  } else {
    const size_t smem = sizeof(double) * Roll6Smem<DM>::size_for(CT);
    static const bool once = [&] {  // more than 64 KB of dynamic LDS needs the opt-in
      (void)hipFuncSetAttribute((const void*)k_rollout6<DM, CT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      (void)hipFuncSetAttribute((const void*)k_rollout6<DM, CT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      return true;
    }();
    (void)once;
    const int G = roll6_group_size(D.NA);
    if (D.integrator == EMPC_INTEGRATOR_RK4)
      hipLaunchKernelGGL((k_rollout6<DM, CT, true>), dim3((D.B + G - 1) / G), dim3(64 * R6_WAVES), smem, s, D);
    else
      hipLaunchKernelGGL((k_rollout6<DM, CT, false>), dim3((D.B + G - 1) / G), dim3(64 * R6_WAVES), smem, s, D);
  }
}
template <class DM>
static void launch_select(DevBuffers D, hipStream_t s) {
  hipLaunchKernelGGL(k_select<DM>, dim3(D.B), dim3(256), 0, s, D);
}
template <class DM>
static void launch_squash_out(DevBuffers D, double* out, hipStream_t s) {
  const int n = D.B * D.T * DM::NU;
  hipLaunchKernelGGL(k_squash_out<DM>, dim3((n + 255) / 256), dim3(256), 0, s, D, out);
}
template <class DM>
static void launch_pack_rows(DevBuffers D, double* out, hipStream_t s) {
  const size_t n = (size_t)D.B * ((size_t)(D.T + 1) * DM::NX + (size_t)D.T * DM::NU + 2);
  hipLaunchKernelGGL(k_pack_rows<DM>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, D, out);
}
template <class DM>
static void launch_plant(DevBuffers D, double* x, const double* u, double dt, int nsub, hipStream_t s) {
  hipLaunchKernelGGL(k_plant_rk4<DM>, dim3((D.B + 63) / 64), dim3(64), 0, s, D, x, u, dt, nsub);
}
template <class DM, int CT>
static KernelTable make_table() {
  KernelTable k;
  k.calc = launch_calc<DM, CT>;
  k.linearize = launch_linearize<DM, CT>;
  k.backward = launch_backward<DM>;
  k.rollout = launch_rollout<DM, CT>;
  k.rk4_linearize = launch_rk4_linearize<DM, CT>;
  k.select = launch_select<DM>;
  k.squash_out = launch_squash_out<DM>;
  k.pack_rows = launch_pack_rows<DM>;
  k.plant = launch_plant<DM>;
  k.nx = DM::NX;
  k.ndx = DM::NDX;
  k.nu = DM::NU;
  k.nv = DM::NV;
  k.nacc = DM::NACC;
  k.rec = DM::REC;
  k.rec_full = DM::REC_FULL;
  k.dev_off_cost = DM::OFF_COST;
  k.unpack = unpack_record<DM>;
  const int off[9] = {DM::OFF_FX, DM::OFF_FU, DM::FULL_LXX, DM::FULL_LXU, DM::FULL_LUU, DM::FULL_LX, DM::FULL_LU, DM::FULL_GAP, DM::FULL_COST};
  std::memcpy(k.off, off, sizeof(off));
  const int ld[5] = {DM::NM, DM::NM, DM::NM, DM::NM, DM::NU};  // Fx, Fu, Lxx, Lxu, Luu leading dimensions
  std::memcpy(k.ld, ld, sizeof(ld));
  return k;
}
// Table of a BAKED robot: the launchers whose kernels touch the rigid-body model (calc, linearize, rollout, the RK4 stage
// kernels, the plant) come from the instantiation over the baked constants; everything else (backward, select, packing:
// no model inside) is the robot class's own table, passed in.  The baked translation units are the only ones built with
// -fno-honor-nans -fno-signed-zeros (Makefile), so the solver's NaN-driven control flow keeps strict IEEE semantics.
template <class DM, int CT>
static KernelTable make_baked_table(KernelTable k) {
  k.calc = launch_calc<DM, CT>;
  k.linearize = launch_linearize<DM, CT>;
  k.rollout = launch_rollout<DM, CT>;
  k.rk4_linearize = launch_rk4_linearize<DM, CT>;
  k.plant = launch_plant<DM>;
  return k;
}
#endif  // EMPC_INSTANTIATE
