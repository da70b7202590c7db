// empc_rollout6.hpp -- HOT-C kernel body, role-split form (SolverFDDP::forwardPass / SolverSbFDDP::forwardPassDDP for all
// step lengths at once; reference call sites src/sbfddp.cpp:264, 352, 416-460).
//
// Why this shape.  The forward pass is a chain over the knots; at batch 1024 the wave-per-trajectory form (rollout_wave5)
// runs one wavefront per SIMD with 10 of 64 lanes busy and ~4.6k dependent vector instructions per knot: its time is the
// length of that instruction chain, not memory (profiles/r01_rollout_ablation.txt).  Two changes shorten the chain:
//   * packing: one wavefront carries G = 64 / NA trajectories x NA step lengths (6 x 10 = 60 busy lanes), so a batch of
//     1024 needs 171 workgroups instead of 1024 wavefronts, and the SIMDs that frees run
//   * roles: the four wavefronts of a workgroup each execute ONE independent part of a node for the same 60 trials, side
//     by side on the four SIMDs of a CU, and hand their results over through LDS between two workgroup barriers per knot:
//       A  feedback + actuation   dx = x_try (-) xs[t], dv term, u = us - alpha k - K dx, squash, tau = B sigma(u)
//       B  bias + frames          RNEA bias forces h(q, v) with the operational-frame captures, frame costs, contact frame
//       C  inertia + integration  CRBA + Cholesky of M(q) beside A and B; after the barrier a = M^-1 (tau - h), contact KKT,
//                                 semi-implicit Euler step, gap contraction -> x_try of the next knot
//       D  costs + memory         State / Control / friction-cone cost values and the node's cost sum (one knot behind),
//                                 stores of xs_try / us_try / acc_try, staging of the next knot's nominal data
//     The chain of a knot becomes max(A, C) + the join instead of the sum of everything.
// The arithmetic of a trial is the arithmetic of node_nominal / rollout_wave5, operation for operation (same helper
// functions, same summation order of the costs), so the step lengths accepted by select do not depend on the form.
#pragma once
#include "empc_kernels.hpp"

namespace empc {

constexpr int R6_A = 0, R6_B = 1, R6_C = 2, R6_D = 3, R6_WAVES = 4;
constexpr int R6_GMAX = 8;  // most trajectories packed into one wavefront (LDS staging is sized for it)

template <class DM>
struct Roll6Smem {
  static constexpr int NX = DM::NX, NU = DM::NU, NV = DM::NV, NDX = DM::NDX, NL = 64;
  // nominal data of the packed trajectories at one knot, one array per quantity: [trajectory][stride].  A read by the
  // 60 trial lanes touches G addresses (one per trajectory, broadcast to its step lengths); odd strides (in doubles) put
  // those in distinct banks; the gain rows need an even stride (16-byte stores) = 2 x odd.
  static constexpr int XS = NX | 1, GS = NDX | 1, US = NU | 1;
  static constexpr int KS = ((NU * NDX) % 4 == 2) ? NU * NDX : NU * NDX + 2;
  static constexpr int NOM_K = 0;                               // K[t]      (staged by B)
  static constexpr int NOM_X = NOM_K + R6_GMAX * KS;            // xs[t]     (the rest staged by D)
  static constexpr int NOM_GAP = NOM_X + R6_GMAX * XS;          // fs[t]
  static constexpr int NOM_VF = NOM_GAP + R6_GMAX * GS;         // Vxx[t] fs[t]
  static constexpr int NOM_US = NOM_VF + R6_GMAX * GS;          // us[t]
  static constexpr int NOM_KF = NOM_US + R6_GMAX * US;          // k[t]
  static constexpr int NOMSZ = (NOM_KF + R6_GMAX * US + 1) / 2 * 2;
  // staging work items per lane (compile-time upper bounds for GMAX trajectories)
  static constexpr int KE = NU * NDX / 2;                       // 16-byte elements of one K[t]
  static constexpr int NKI = (R6_GMAX * KE + NL - 1) / NL;
  static constexpr int NXI = (R6_GMAX * NX + NL - 1) / NL, NGI = (R6_GMAX * NDX + NL - 1) / NL, NUI = (R6_GMAX * NU + NL - 1) / NL;
  // per-lane exchange slots, laid out [item][lane]
  static constexpr int OFF_XT = 0;                            // x_try of the current knot            (C -> A, B, C, D)
  static constexpr int OFF_UT = OFF_XT + NX * NL;             // control of the trial (unsquashed s)  (A -> D)
  static constexpr int OFF_TAU = OFF_UT + NU * NL;            // generalized force                    (A -> C)
  static constexpr int OFF_H = OFF_TAU + NV * NL;             // bias forces                          (B -> C)
  static constexpr int OFF_CAP = OFF_H + NV * NL;             // contact frame capture, 24 doubles    (B -> C)
#if EMPC_ROLL_CAP_LDS
  static constexpr int OFF_ACC = OFF_CAP + NCAP * 24 * NL;    // (one 24-double slot per captured frame: B -> B's frame costs, C)
#else
  static constexpr int OFF_ACC = OFF_CAP + 24 * NL;           // acceleration | contact force         (C -> D)
#endif
  static constexpr int OFF_ELLF = OFF_ACC + DM::NACC * NL;    // frame-cost sum, by knot parity       (B -> D)
  static constexpr int OFF_VAL = OFF_ELLF + 2 * NL;           // activation value per cost            (D -> D)
  static constexpr int OFF_FLAG = OFF_VAL + EMPC_MAX_COSTS * NL;  // ok flag of role C
  static constexpr int OFF_DUMP = OFF_FLAG + NL;              // write-only: staging items without work land here (2 per lane)
  static constexpr int OFF_TB = OFF_DUMP + 2 * NL;            // trajectory index of each packed slot (ints)
  static constexpr int OFF_NOM = OFF_TB + R6_GMAX;            // [2][NOMSZ], 16-byte aligned
  static constexpr int SIZE = (OFF_NOM + 2 * NOMSZ + 1) / 2 * 2;
  // CT_PAIR3 (two ContactModel3D of one stage): the capture of the second contact frame (B -> C) behind everything else
  static constexpr int OFF_CAPB = SIZE;
  static constexpr int size_for(int CT) { return CT == CT_PAIR3 ? SIZE + 24 * NL : SIZE; }
  static_assert(OFF_NOM % 2 == 0 && OFF_DUMP % 2 == 0 && KS % 2 == 0 && (NU * NDX) % 2 == 0, "gain rows are staged in 16-byte pieces");
};

// keeps the instruction scheduler from moving anything across this point (it otherwise sinks every staging load next to
// its LDS write: load, wait, write, load, wait, write ... -- one exposed memory latency per item)
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define R6_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define R6_SCHED_FENCE() \
  do {                   \
  } while (0)
#endif

// 16-byte piece of a gain row; a native vector type so that an array of them lives in registers (a struct array stayed in
// scratch memory: load, wait, scratch store, ..., scratch load, wait, LDS write)
#if defined(__HIPCC__)
typedef double R6Pair __attribute__((ext_vector_type(2)));
#else
typedef double R6Pair __attribute__((vector_size(16)));
#endif

// trajectories per wavefront for NA step lengths
EMPC_HD int roll6_group_size(int NA) {
  const int g = 64 / NA;
  return g > R6_GMAX ? R6_GMAX : g;
}

// per-lane view of its trial: which trajectory / step length, and the trajectory-level switches of the forward pass
struct Roll6Lane {
  int b, g, ai, live;
  double alpha, smooth;
  bool plain, need_dv;
};
EMPC_HD Roll6Lane roll6_lane(const DevBuffers& D, const int* TB, int lane, int G) {
  Roll6Lane L;
  const int NA = D.NA;
  L.g = lane / NA;
  L.ai = lane % NA;
  L.b = (L.g < G) ? TB[L.g] : -1;
  L.live = 0;
  L.alpha = ldexp(1.0, -L.ai);
  L.smooth = 0.1;
  L.plain = true;
  L.need_dv = false;
  if (L.b >= 0) {
    const TrajState& st = D.st[L.b];
    const bool ddp = (st.phase == PHASE_DDP), feas = st.is_feasible != 0;
    L.live = (st.phase == PHASE_DONE || st.bwd_failed) ? 0 : 1;
    L.smooth = st.smooth;
    L.plain = ddp || feas || (L.ai == 0);
    L.need_dv = !ddp && !feas;
  }
  return L;
}

// One instantiation per role: a wavefront runs the whole rollout of ITS role (warp specialisation), so the registers of a
// role hold that role's state only; the four instantiations execute the same sequence of workgroup barriers.
// Exec concept: ex.each(f) runs f(lane, slot) on the lanes of the calling wavefront; ex.sync() is the workgroup barrier.
// RK4 = true: IntegratedActionModelRK4 nodes (src/factory/int-action.cpp:29-31).  A knot becomes four stages of the same two
// phases -- the differential model at y_i = x (+) c_i dt k_{i-1}: bias forces, inertia, acceleration, costs -- with the
// feedback control, the squashing and the generalized force of stage 0 kept for all four; role C carries the knot's state and
// the weighted sum of the k_i, role D the weighted sum of the stage costs (node_nominal_rk4, operation for operation).
template <class DM, int CT, int ROLE, class Exec, bool RK4 = false>
EMPC_HD void rollout_group6(Exec& ex, const DevBuffers& D, int group, double* N) {
  typedef Roll6Smem<DM> SM;
  constexpr int NX = DM::NX, NU = DM::NU, NV = DM::NV, NQ = DM::NQ, NDX = DM::NDX, REC = DM::REC, NB = DM::NB, NROT = DM::NROT;
  constexpr int NL = SM::NL;
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  const auto& m = model_of<DM>(P);
  const auto PL = platform_of<DM>(P);
  const int T = D.T, NA = D.NA;
  const int G = roll6_group_size(NA);
  const int nlist = D.act_list ? *D.act_count : D.B;
  if (group * G >= nlist) return;  // whole workgroup: nothing packed here
  int* TB = reinterpret_cast<int*>(N + SM::OFF_TB);
  double* XT = N + SM::OFF_XT;
  double* UT = N + SM::OFF_UT;
  double* TAU = N + SM::OFF_TAU;
  double* HB = N + SM::OFF_H;
  double* CAP = N + SM::OFF_CAP;
  double* ACC = N + SM::OFF_ACC;
  double* ELLF = N + SM::OFF_ELLF;
  double* VAL = N + SM::OFF_VAL;
  double* FLAG = N + SM::OFF_FLAG;
  double* NOM = N + SM::OFF_NOM;
  const double dt = P.dt;

  // ---- packed slots -> trajectories ---------------------------------------------------------------------------------
  if constexpr (ROLE == R6_A) ex.each([&](int lane, int sl) {
    if (lane < R6_GMAX) {
      const int i = group * G + lane;
      TB[lane] = (lane < G && i < nlist) ? (D.act_list ? D.act_list[i] : i) : -1;
    }
  });
  ex.sync();

  // ---- staging of the nominal data: work items of this lane, fixed for the whole rollout ----------------------------------
  // item = (element offset at knot 0 in its global array, offset inside one staging buffer); -1 = nothing to do.
  // B moves the gain rows (16 bytes per item), D everything else (8 bytes per item).
  // (global offsets are 64-bit: b * (T + 1) * REC passes 2^31 at ~19k trajectories of the arm robots)
  typedef long long goff;
  goff kI[Exec::SLOTS][SM::NKI], xI[Exec::SLOTS][SM::NXI], gI[Exec::SLOTS][SM::NGI], vI[Exec::SLOTS][SM::NGI], uI[Exec::SLOTS][SM::NUI];
  int kO[Exec::SLOTS][SM::NKI], xO[Exec::SLOTS][SM::NXI], gO[Exec::SLOTS][SM::NGI], uO[Exec::SLOTS][SM::NUI];
  if constexpr (ROLE == R6_B) ex.each([&](int lane, int sl) {
#pragma unroll
    for (int k = 0; k < SM::NKI; ++k) {
      const int idx = lane + NL * k, g = idx / SM::KE, e = idx % SM::KE;
      const int b = (g < G) ? TB[g] : -1;
      kI[sl][k] = (b >= 0) ? ((goff)b * T * NU * NDX + 2 * e) : -1;
      kO[sl][k] = SM::NOM_K + g * SM::KS + 2 * e;  // inside a staging buffer
    }
  });
  if constexpr (ROLE == R6_A || ROLE == R6_D) ex.each([&](int lane, int sl) {
#pragma unroll
    for (int k = 0; k < SM::NXI; ++k) {
      const int idx = lane + NL * k, g = idx / NX, e = idx % NX;
      const int b = (g < G) ? TB[g] : -1;
      xI[sl][k] = (b >= 0) ? ((goff)b * (T + 1) * NX + e) : -1;
      xO[sl][k] = SM::NOM_X + g * SM::XS + e;
    }
#pragma unroll
    for (int k = 0; k < SM::NGI; ++k) {
      const int idx = lane + NL * k, g = idx / NDX, e = idx % NDX;
      const int b = (g < G) ? TB[g] : -1;
      gI[sl][k] = (b >= 0) ? ((goff)b * (T + 1) * REC + DM::OFF_GAP + e) : -1;
      vI[sl][k] = (b >= 0) ? ((goff)b * (T + 1) * NDX + e) : -1;
      gO[sl][k] = g * SM::GS + e;  // relative to NOM_GAP resp. NOM_VF
    }
#pragma unroll
    for (int k = 0; k < SM::NUI; ++k) {
      const int idx = lane + NL * k, g = idx / NU, e = idx % NU;
      const int b = (g < G) ? TB[g] : -1;
      uI[sl][k] = (b >= 0) ? ((goff)b * T * NU + e) : -1;
      uO[sl][k] = g * SM::US + e;  // relative to NOM_US resp. NOM_KF
    }
  });
  // the two halves of a transfer: loads into registers (issued first, in flight while the role computes), LDS writes last
  R6Pair kV[Exec::SLOTS][SM::NKI];
  double xV[Exec::SLOTS][SM::NXI], gV[Exec::SLOTS][SM::NGI], vV[Exec::SLOTS][SM::NGI], uV[Exec::SLOTS][SM::NUI], fV[Exec::SLOTS][SM::NUI];
  // (loads AND LDS writes are unconditional -- an item without work reads element 0 of its array and writes to a dump
  // slot -- so that the compiler issues all loads back to back; a predicated write makes it sink each load into the
  // predicated block and wait for it there, one memory latency per item: measured 9k cycles per knot)
  double* const dumpl = N + SM::OFF_DUMP;
  auto fetch_gains = [&](int t, int sl) {  // K[t], t < T
#pragma unroll
    for (int k = 0; k < SM::NKI; ++k) {
      const size_t o = (kI[sl][k] >= 0) ? (size_t)kI[sl][k] + (size_t)t * NU * NDX : 0;
      kV[sl][k] = *reinterpret_cast<const R6Pair*>(D.K + o);
    }
  };
  auto put_gains = [&](int t, int lane, int sl) {
    double* dst = NOM + (size_t)(t & 1) * SM::NOMSZ;
#pragma unroll
    for (int k = 0; k < SM::NKI; ++k)
      *reinterpret_cast<R6Pair*>((kI[sl][k] >= 0) ? dst + kO[sl][k] : dumpl + 2 * lane) = kV[sl][k];
  };
  // A moves xs, Vxx f, us and k of the next knot while it waits for C in phase II; D moves the gaps in phase I (C contracts
  // the next trial state towards them right after the barrier)
  auto fetch_nom = [&](int t, int sl) {
#pragma unroll
    for (int k = 0; k < SM::NXI; ++k) xV[sl][k] = D.xs[(xI[sl][k] >= 0) ? (size_t)xI[sl][k] + (size_t)t * NX : 0];
#pragma unroll
    for (int k = 0; k < SM::NGI; ++k) vV[sl][k] = D.Vf[(gI[sl][k] >= 0) ? (size_t)vI[sl][k] + (size_t)t * NDX : 0];
    if (t < T) {
#pragma unroll
      for (int k = 0; k < SM::NUI; ++k) {
        const size_t o = (uI[sl][k] >= 0) ? (size_t)uI[sl][k] + (size_t)t * NU : 0;
        uV[sl][k] = D.us[o];
        fV[sl][k] = D.kff[o];
      }
    }
  };
  auto put_nom = [&](int t, int lane, int sl) {
    double* dst = NOM + (size_t)(t & 1) * SM::NOMSZ;
#pragma unroll
    for (int k = 0; k < SM::NXI; ++k) *((xI[sl][k] >= 0) ? dst + xO[sl][k] : dumpl + lane) = xV[sl][k];
#pragma unroll
    for (int k = 0; k < SM::NGI; ++k) *((gI[sl][k] >= 0) ? dst + SM::NOM_VF + gO[sl][k] : dumpl + lane) = vV[sl][k];
    if (t < T) {
#pragma unroll
      for (int k = 0; k < SM::NUI; ++k) {
        *((uI[sl][k] >= 0) ? dst + SM::NOM_US + uO[sl][k] : dumpl + lane) = uV[sl][k];
        *((uI[sl][k] >= 0) ? dst + SM::NOM_KF + uO[sl][k] : dumpl + lane) = fV[sl][k];
      }
    }
  };
  auto fetch_gap = [&](int t, int sl) {
#pragma unroll
    for (int k = 0; k < SM::NGI; ++k) gV[sl][k] = D.tape[(gI[sl][k] >= 0) ? (size_t)gI[sl][k] + (size_t)t * REC : 0];
  };
  auto put_gap = [&](int t, int lane, int sl) {
    double* dst = NOM + (size_t)(t & 1) * SM::NOMSZ;
#pragma unroll
    for (int k = 0; k < SM::NGI; ++k) *((gI[sl][k] >= 0) ? dst + SM::NOM_GAP + gO[sl][k] : dumpl + lane) = gV[sl][k];
  };
  if constexpr (ROLE == R6_B) ex.each([&](int lane, int sl) {
    if (T > 0) {
      fetch_gains(0, sl);
      put_gains(0, lane, sl);
    }
  });
  if constexpr (ROLE == R6_A) ex.each([&](int lane, int sl) {
    fetch_nom(0, sl);
    put_nom(0, lane, sl);
  });
  if constexpr (ROLE == R6_D) ex.each([&](int lane, int sl) {
    fetch_gap(0, sl);
    put_gap(0, lane, sl);
  });
  ex.sync();

  // every wavefront keeps the same lane -> (trajectory, step length) view in registers
  Roll6Lane LL[Exec::SLOTS];
  ex.each([&](int lane, int sl) { LL[sl] = roll6_lane(D, TB, lane, G); });

  // ---- per-role state that lives across the knots -----------------------------------------------------------------------
  double dvA[Exec::SLOTS];                    // A: -f^T Vxx (xs (-) xs_try) summed over the knots
  double Lc[Exec::SLOTS][DM::NTRI];           // C: Cholesky factor of M, from phase I to phase II of a knot
#ifdef EMPC_ROLL_GAP_EARLY
  // C: the gap correction of the NEXT trial state, (alpha - 1) fs[t+1], and exp6 of its first six entries -- known before the
  // knot's dynamics are, so taken in phase I (where C waits for A and D) instead of at the end of phase II (the knot's critical
  // path).  Build-time experiment, off by default: prepared while the GPU pool was closed, bit-identical on the lane emulator.
  double stepC[Exec::SLOTS][NDX], qeC[Exec::SLOTS][4], peC[Exec::SLOTS][3];
#endif
  int okC[Exec::SLOTS];                       // C: trial state stayed finite
  int ncC[Exec::SLOTS], ncD[Exec::SLOTS];     // C / D: running nodes reached before the first failure (T: none)
  double costD[Exec::SLOTS];                  // D: cost of the trial
  int okD[Exec::SLOTS];
  double xkC[Exec::SLOTS][RK4 ? NX : 1], ksC[Exec::SLOTS][RK4 ? NDX : 1];  // C, RK4 nodes: state of the knot, sum w_i k_i
  double ellD[Exec::SLOTS];                   // D, RK4 nodes: sum w_i l_i of the knot being finished
  double xD[Exec::SLOTS][NX], rsD[Exec::SLOTS][NDX];  // D: trial state and last state difference, from phase I to phase II of a knot
  int rsofD[Exec::SLOTS];                     // D: which reference rsD belongs to
  constexpr int NST = RK4 ? 4 : 1;

  // x_try of knot 0 (C): x0, contracted towards the nominal start by the gap when the pass keeps gaps
  if constexpr (ROLE == R6_C) ex.each([&](int lane, int sl) {
    const Roll6Lane& L = LL[sl];
    okC[sl] = 1;
    ncC[sl] = T;
    if (!L.live) return;
    const double* gap0 = NOM + SM::NOM_GAP + L.g * SM::GS;  // staging buffer 0
    double xn[NX], xt[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xn[i] = D.x0[(size_t)L.b * NX + i];
    if (L.plain) {
#pragma unroll
      for (int i = 0; i < NX; ++i) xt[i] = xn[i];
    } else {
      double step[NDX];
#pragma unroll
      for (int i = 0; i < NDX; ++i) step[i] = gap0[i] * (L.alpha - 1.0);
      state_integrate<DM>(xn, step, xt, nullptr);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) XT[i * NL + lane] = xt[i];
  });
  if constexpr (ROLE == R6_A) ex.each([&](int lane, int sl) { dvA[sl] = 0.0; });
  if constexpr (ROLE == R6_D) ex.each([&](int lane, int sl) {
    costD[sl] = 0.0;
    okD[sl] = 1;
    ncD[sl] = T;
  });
  ex.sync();

  // cost of knot tp for D's lane, from the values D left in VAL, the contact force in ACC and B's frame-cost sum
  // (RK4 nodes: stage stp of knot tp; the knot's cost closes with its last stage)
  auto finish_cost = [&](int tp, int stp, int lane, int sl, const Roll6Lane& L) {
    const int kset = EMPC_KPTR(int, D.knot_set)[tp];
    const EMPC_K EmpcCostSet& set = EMPC_KPTR(EmpcCostSet, D.sets)[kset];
    const EMPC_K SetInfo& si = EMPC_KPTR(SetInfo, D.set_info)[kset];
    const bool terminal = (tp == T);
    const bool use_contact = CT && P.has_contact && set.ncontacts > 0;
    double ell = 0;
    for (int k = 0; k < si.n_sc; ++k) {
      const int ci = si.sc_ci[k];
      ell += set.costs[ci].weight * VAL[ci * NL + lane];
    }
    for (int k = 0; k < si.n_cone; ++k) {
      const auto& c = set.costs[si.cone_ci[k]];
      double lam[3], r[6] = {0, 0, 0, 0, 0, 0};
      const int fo = cone_force_offset<CT>(set, c.frame);  // CT_PAIR3: the contact on the cost's frame; 0 otherwise
#pragma unroll
      for (int i = 0; i < 3; ++i) lam[i] = ACC[(NV + fo + i) * NL + lane];
#pragma unroll
      for (int i = 0; i < 5; ++i)
        r[i] = use_contact ? (c.ref[4 + 3 * i] * lam[0] + c.ref[5 + 3 * i] * lam[1] + c.ref[6 + 3 * i] * lam[2]) : 0.0;
      ell += c.weight * activation_value<6>(c, r, 5);
    }
    ell += ELLF[((RK4 ? 4 * tp + stp : tp) & 1) * NL + lane];
    bool closes = true;
    if constexpr (RK4) {
      const double w = (stp == 0 || stp == 3) ? 1.0 : 2.0;
      ellD[sl] = (stp == 0) ? ell : ellD[sl] + w * ell;
      closes = (stp == 3);
      if (closes) {
        const double cscale = (terminal && !P.prm.terminal_dt_scaling) ? 1.0 / 6.0 : dt / 6.0;
        costD[sl] += ellD[sl] * cscale;
      }
    } else {
      const double cscale = (terminal && !P.prm.terminal_dt_scaling) ? 1.0 : dt;
      costD[sl] += cscale * ell;
    }
    if (closes && bad_number(costD[sl])) {
      if (okD[sl]) ncD[sl] = (tp + 1 < T) ? tp + 1 : T;
      okD[sl] = 0;
    }
    // acceleration | contact force of the knot (RK4: of its stage 0) -> acc_try (linearize reuses them for the accepted trial)
    if (stp == 0) {
      double* ac_o = D.acc_try + ((size_t)L.b * NA + L.ai) * (T + 1) * DM::NACC + (size_t)tp * DM::NACC;
#pragma unroll
      for (int i = 0; i < DM::NACC; ++i) ac_o[i] = ACC[i * NL + lane];
    }
  };

#if defined(EMPC_STAMPS) && defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
  // diagnostic builds: cycles of this role in {phase I work, wait at barrier 1, phase II work, wait at barrier 2}
  unsigned long long r6st[5] = {0, 0, 0, 0, __builtin_readcyclecounter()};
#define R6_STAMP(i)                                                 \
  do {                                                              \
    const unsigned long long now_ = __builtin_readcyclecounter();   \
    r6st[i] += now_ - r6st[4];                                      \
    r6st[4] = now_;                                                 \
  } while (0)
  unsigned long long r6b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define R6_SUB(i)                                                   \
  do {                                                              \
    __builtin_amdgcn_sched_barrier(0);                              \
    const unsigned long long now_ = __builtin_readcyclecounter();   \
    r6b[i] += now_ - r6b[7];                                        \
    r6b[7] = now_;                                                  \
    __builtin_amdgcn_sched_barrier(0);                              \
  } while (0)
#else
#define R6_STAMP(i) \
  do {              \
  } while (0)
#define R6_SUB(i) \
  do {            \
  } while (0)
#endif
  for (int t = 0; t <= T; ++t) {
    const int kset = EMPC_KPTR(int, D.knot_set)[t];
    const EMPC_K EmpcCostSet& set = EMPC_KPTR(EmpcCostSet, D.sets)[kset];
    const EMPC_K SetInfo& si = EMPC_KPTR(SetInfo, D.set_info)[kset];
    const bool terminal = (t == T);
    const bool use_contact = CT && P.has_contact && set.ncontacts > 0;
    for (int st = 0; st < NST; ++st) {
    const int stepq = RK4 ? 4 * t + st : t;  // step: what B's frame-cost slot and D's one-step-behind bookkeeping count in
    const bool last_stage = (st == NST - 1);
    // =============================================== phase I ===========================================================
    // ---- A: state difference to the nominal trajectory, feedback, squashing, generalized force ----------------------------
    if constexpr (ROLE == R6_A) ex.each([&](int lane, int sl) {
      const Roll6Lane& L = LL[sl];
      if (!L.live || st > 0) return;
      const double* nom = NOM + (size_t)(t & 1) * SM::NOMSZ;
      const double* n_x = nom + SM::NOM_X + L.g * SM::XS;
      const double* n_vf = nom + SM::NOM_VF + L.g * SM::GS;
      const double* n_us = nom + SM::NOM_US + L.g * SM::US;
      const double* n_kf = nom + SM::NOM_KF + L.g * SM::US;
      const double* n_K = nom + SM::NOM_K + L.g * SM::KS;
      // Every LDS operand of this role is requested in blocks ahead of its use (one wait per block): left to itself the
      // compiler walks the 180 gain entries as read, wait, multiply-add, read, wait, ... with one read in flight -- one LDS
      // round trip per entry, ~7k of the role's 8.5k cycles per knot.  Same operations in the same order.
      double x[NX], xn[NX], dx[NDX], s[NU], u[NU], tau[NV];
      R6_SUB(6);
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        x[i] = XT[i * NL + lane];
        xn[i] = n_x[i];
      }
      R6Pair kr[2][NDX / 2];  // rows of K[t], two in flight
      auto load_row = [&](int i, int buf) {
#pragma unroll
        for (int j = 0; j < NDX / 2; ++j) kr[buf][j] = *reinterpret_cast<const R6Pair*>(n_K + i * NDX + 2 * j);
      };
      double usv[NU], kfv[NU];
      if (!terminal) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          usv[i] = n_us[i];
          kfv[i] = n_kf[i];
        }
        load_row(0, 0);
      }
      R6_SCHED_FENCE();
      R6_SUB(0);
      state_diff<DM>(xn, x, dx, nullptr);
      R6_SUB(1);
      if (L.need_dv) {
        double vf[NDX];
#pragma unroll
        for (int i = 0; i < NDX; ++i) vf[i] = n_vf[i];
        R6_SCHED_FENCE();
        double dv = dvA[sl];
#pragma unroll
        for (int i = 0; i < NDX; ++i) dv += vf[i] * dx[i];  // +(Vxx f).(xs_try (-) xs)
        dvA[sl] = dv;
      }
      R6_SUB(2);
      if (!terminal) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          if (i + 1 < NU) load_row(i + 1, (i + 1) & 1);
          R6_SCHED_FENCE();
          double a_ = usv[i] - kfv[i] * L.alpha;
#pragma unroll
          for (int j = 0; j < NDX; ++j) a_ -= kr[i & 1][j / 2][j % 2] * dx[j];
          // SolverBox{DDP,FDDP}::forwardPass clamp the trial control to the limits of the model
          s[i] = (P.prm.solver_type != EMPC_SOLVER_SBFDDP) ? fmin(fmax(a_, PL.u_lb[i]), PL.u_ub[i]) : a_;
          R6_SCHED_FENCE();
        }
      } else {
#pragma unroll
        for (int i = 0; i < NU; ++i) s[i] = 0.0;
      }
      R6_SUB(3);
      if (P.use_squash) {
        double lbv[NU], ubv[NU];
        const int power = P.prm.smoothsat_power;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          lbv[i] = PL.u_lb[i];
          ubv[i] = PL.u_ub[i];
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          double du;
          squash1(s[i], lbv[i], ubv[i], L.smooth, power, u[i], du);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = s[i];
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double a_ = 0;
#pragma unroll
        for (int c = 0; c < NROT; ++c) a_ += PL.tau_f[r * NROT + c] * u[c];
        tau[r] = a_;
      }
#pragma unroll
      for (int i = 6; i < NV; ++i) tau[i] = u[NROT + i - 6];
#pragma unroll
      for (int i = 0; i < NV; ++i) TAU[i * NL + lane] = tau[i];
#pragma unroll
      for (int i = 0; i < NU; ++i) UT[i * NL + lane] = s[i];
      R6_SUB(4);
    });
    // ---- B: bias forces with the frame captures; frame costs; contact frame for C ------------------------------------------
    if constexpr (ROLE == R6_B) ex.each([&](int lane, int sl) {
      const Roll6Lane& L = LL[sl];
      R6_SUB(6);
      if (L.live) {
      double x[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = XT[i * NL + lane];
      const double* q = x;
      const double* v = x + NQ;
      double R0[9], cs[NB], sn[NB];
      quat_to_R(q + 3, R0);
#pragma unroll
      for (int b = 1; b < NB; ++b) fsincos(q[7 + b - 1], &sn[b - 1], &cs[b - 1]);
      // operational frames this node captures (frame costs, contact): the host made the list (SetInfo), same order as the
      // table scan of node_nominal
      int capf[NCAP];
#pragma unroll
      for (int k = 0; k < NCAP; ++k) capf[k] = si.capf[k];
      // (CT_PAIR3 problems: si.ccap = slot of contact 0's frame | slot of contact 1's frame << 8, see prepare_problem)
      const int ncap = si.ncap, ccap = (CT == CT_PAIR3) ? (si.ccap & 0xff) : si.ccap;
#if EMPC_ROLL_CAP_LDS
      double zero[NV], h[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) zero[i] = 0.0;
      R6_SUB(1);
      rnea_chain_f<NB, double>(m, R0, q, cs, sn, v, zero, true, -1, nullptr, h, ncap, capf,
                               [&](int c, int f, const double* Rw, const double* pw, const double* vb, const double* ab) {
                                 FrameCap<double> fk;
                                 frame_capture<double>(m, f, Rw, pw, vb, ab, fk);
                                 double* slot = CAP + (size_t)c * 24 * NL + lane;
#pragma unroll
                                 for (int i = 0; i < 9; ++i) slot[i * NL] = fk.R[i];
#pragma unroll
                                 for (int i = 0; i < 3; ++i) slot[(9 + i) * NL] = fk.p[i];
#pragma unroll
                                 for (int i = 0; i < 6; ++i) slot[(12 + i) * NL] = fk.v[i];
#pragma unroll
                                 for (int i = 0; i < 6; ++i) slot[(18 + i) * NL] = fk.a[i];
                               });
      R6_SUB(2);
#pragma unroll
      for (int i = 0; i < NV; ++i) HB[i * NL + lane] = h[i];
#else
      FrameCap<double> caps[NCAP];
      double zero[NV], h[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) zero[i] = 0.0;
      R6_SUB(1);
      rnea_chain<NB, double>(m, R0, q, cs, sn, v, zero, true, -1, nullptr, h, ncap, capf, caps);
      R6_SUB(2);
#pragma unroll
      for (int i = 0; i < NV; ++i) HB[i * NL + lane] = h[i];
      if constexpr (CT) {
        if (use_contact) {
          FrameCap<double> ck = caps[0];
#pragma unroll
          for (int kk = 1; kk < NCAP; ++kk)
            if (kk == ccap) ck = caps[kk];
#pragma unroll
          for (int i = 0; i < 9; ++i) CAP[i * NL + lane] = ck.R[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) CAP[(9 + i) * NL + lane] = ck.p[i];
#pragma unroll
          for (int i = 0; i < 6; ++i) CAP[(12 + i) * NL + lane] = ck.v[i];
#pragma unroll
          for (int i = 0; i < 6; ++i) CAP[(18 + i) * NL + lane] = ck.a[i];
          if constexpr (CT == CT_PAIR3) {
            if (set.ncontacts > 1) {
              const int ccap2 = si.ccap >> 8;
              double* CAPB = N + SM::OFF_CAPB;
              FrameCap<double> ck2 = caps[0];
#pragma unroll
              for (int kk = 1; kk < NCAP; ++kk)
                if (kk == ccap2) ck2 = caps[kk];
#pragma unroll
              for (int i = 0; i < 9; ++i) CAPB[i * NL + lane] = ck2.R[i];
#pragma unroll
              for (int i = 0; i < 3; ++i) CAPB[(9 + i) * NL + lane] = ck2.p[i];
#pragma unroll
              for (int i = 0; i < 6; ++i) CAPB[(12 + i) * NL + lane] = ck2.v[i];
#pragma unroll
              for (int i = 0; i < 6; ++i) CAPB[(18 + i) * NL + lane] = ck2.a[i];
            }
          }
        }
      }
#endif
      R6_SUB(3);
      // frame costs (value only), summed in cost order
      double ell_frames = 0;
      for (int kf = 0; kf < si.n_frame; ++kf) {
        const auto& c = set.costs[si.frame_ci[kf]];
        double cval = 0;
        {
#if EMPC_ROLL_CAP_LDS
          int fslot = 0;  // (same choice as the register form: slot 0 unless a later slot holds the cost's frame)
#pragma unroll
          for (int kk = 1; kk < NCAP; ++kk)
            if (kk < ncap && capf[kk] == c.frame) fslot = kk;
          FrameCap<double> fk;
          {
            const double* slot = CAP + (size_t)fslot * 24 * NL + lane;
#pragma unroll
            for (int i = 0; i < 9; ++i) fk.R[i] = slot[i * NL];
#pragma unroll
            for (int i = 0; i < 3; ++i) fk.p[i] = slot[(9 + i) * NL];
#pragma unroll
            for (int i = 0; i < 6; ++i) fk.v[i] = slot[(12 + i) * NL];
          }
#else
          FrameCap<double> fk = caps[0];
#pragma unroll
          for (int kk = 1; kk < NCAP; ++kk)
            if (kk < ncap && capf[kk] == c.frame) fk = caps[kk];
#endif
          double r[6];
          int nr = 6;
          if (c.type == EMPC_COST_FRAME_PLACEMENT) {
            double rR[9], dp[3], rp[3], qq[4];
            matTmul3<double>(c.ref + 3, fk.R, rR);
#pragma unroll
            for (int i = 0; i < 3; ++i) dp[i] = fk.p[i] - c.ref[i];
            matTvec3<double>(c.ref + 3, dp, rp);
            R_to_quat(rR, qq);
            log6_quat(qq, rp, r);
          } else if (c.type == EMPC_COST_FRAME_ROTATION) {
            double rR[9], qq[4];
            matTmul3<double>(c.ref, fk.R, rR);
            R_to_quat(rR, qq);
            quat_log3(qq, r);
            nr = 3;
          } else if (c.type == EMPC_COST_FRAME_TRANSLATION) {
#pragma unroll
            for (int i = 0; i < 3; ++i) r[i] = fk.p[i] - c.ref[i];
            nr = 3;
          } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) r[i] = fk.v[i] - c.ref[i];
          }
          if (nr == 3) {
            r[3] = r[4] = r[5] = 0.0;
          }
          cval = activation_value<6>(c, r, nr);
        }
        ell_frames += c.weight * cval;
      }
      ELLF[(stepq & 1) * NL + lane] = ell_frames;
      R6_SUB(4);
      }
    });
    // ---- C: joint-space inertia and its Cholesky factor ------------------------------------------------------------------------
    if constexpr (ROLE == R6_C) ex.each([&](int lane, int sl) {
      const Roll6Lane& L = LL[sl];
      if (!L.live) return;
      R6_SUB(6);
      double cs[NB], sn[NB];
#pragma unroll
      for (int b = 1; b < NB; ++b) fsincos(XT[(7 + b - 1) * NL + lane], &sn[b - 1], &cs[b - 1]);
      crba_chain<NB>(m, cs, sn, Lc[sl]);
      chol_packed<NV>(Lc[sl]);
#ifdef EMPC_ROLL_GAP_EARLY
      if (!terminal && last_stage && !L.plain) {
        const double* gp = D.tape + ((size_t)L.b * (T + 1) + (size_t)(t + 1)) * REC + DM::OFF_GAP;  // fs[t+1] of this lane's trajectory
#pragma unroll
        for (int i = 0; i < NDX; ++i) stepC[sl][i] = gp[i] * (L.alpha - 1.0);
        exp6_quat(stepC[sl], qeC[sl], peC[sl]);
      }
#endif
      R6_SUB(5);
    });
    // ---- D: cost of the previous knot, State costs of this one, stores, staging of the next knot's nominal data ----------------
    if constexpr (ROLE == R6_D) ex.each([&](int lane, int sl) {
      const Roll6Lane& L = LL[sl];
      if (t < T && last_stage) fetch_gap(t + 1, sl);  // in flight behind the State costs
      R6_SCHED_FENCE();
      if (L.live) {
        R6_SUB(6);
        if (stepq > 0) finish_cost(st > 0 ? t : t - 1, st > 0 ? st - 1 : NST - 1, lane, sl, L);
        R6_SUB(0);
        double x[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = XT[i * NL + lane];
        if (st == 0) {
          double* xs_o = D.xs_try + ((size_t)L.b * NA + L.ai) * (T + 1) * NX + (size_t)t * NX;
#pragma unroll
          for (int i = 0; i < NX; ++i) xs_o[i] = x[i];
        }
        R6_SUB(1);
        // State costs: the first half of the set's list here, the rest after the barrier (D's phase I -- 7k cycles per knot with
        // all of them -- was the longest of the four roles, its phase II the shortest beside C's 5.5k; the state and the last
        // state difference travel across the barrier in registers, XT itself is overwritten by C in phase II).  Same
        // operations, same order per cost.
        rsofD[sl] = -1;
        const int n_first = (si.n_state + 1) / 2;
        for (int ks = 0; ks < n_first; ++ks) {
          const int ci = si.state_ci[ks];
          const auto& c = set.costs[ci];
          if (!(c.ref_share >= 0 && c.ref_share == rsofD[sl])) {
            state_diff<DM>(c.ref, x, rsD[sl], nullptr);
            rsofD[sl] = (c.ref_share >= 0) ? c.ref_share : ci;
          }
          VAL[ci * NL + lane] = activation_value<NDX>(c, rsD[sl], NDX);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) xD[sl][i] = x[i];
        R6_SUB(2);
      }
      R6_SCHED_FENCE();
      if (t < T && last_stage) put_gap(t + 1, lane, sl);
    });
    R6_STAMP(0);
    ex.sync();
    R6_STAMP(1);
    // =============================================== phase II ==========================================================
    // ---- C: acceleration, contact, Euler step, next trial state -------------------------------------------------------------
    if constexpr (ROLE == R6_C) ex.each([&](int lane, int sl) {
      const Roll6Lane& L = LL[sl];
      if (!L.live) return;
      R6_SUB(6);
      double x[NX], a[NV], lam[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = XT[i * NL + lane];
      {
        // both operands of every row requested before the first subtraction (the compiler otherwise pairs each read with
        // its own wait: one exposed LDS round trip per row on the critical path of the knot)
        double tv[NV], hv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          tv[i] = TAU[i * NL + lane];
          hv[i] = HB[i * NL + lane];
        }
        R6_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < NV; ++i) a[i] = tv[i] - hv[i];
      }
      R6_SUB(0);
      chol_solve_packed<NV>(Lc[sl], a);
      R6_SUB(1);
      if constexpr (CT) {
        if (use_contact) {
          FrameCap<double> ck;
#if EMPC_ROLL_CAP_LDS
          // the slot role B's register form copied from: caps[0] unless a later slot is the contact frame's (si.ccap)
          const int ccapc = (CT == CT_PAIR3) ? (si.ccap & 0xff) : si.ccap;
          const double* CAPc = CAP + (size_t)((ccapc >= 1 && ccapc < NCAP) ? ccapc : 0) * 24 * NL;
#else
          const double* CAPc = CAP;
#endif
#pragma unroll
          for (int i = 0; i < 9; ++i) ck.R[i] = CAPc[i * NL + lane];
#pragma unroll
          for (int i = 0; i < 3; ++i) ck.p[i] = CAPc[(9 + i) * NL + lane];
#pragma unroll
          for (int i = 0; i < 6; ++i) ck.v[i] = CAPc[(12 + i) * NL + lane];
#pragma unroll
          for (int i = 0; i < 6; ++i) ck.a[i] = CAPc[(18 + i) * NL + lane];
          double R0[9], cs[NB], sn[NB];
          quat_to_R(x + 3, R0);
#pragma unroll
          for (int b = 1; b < NB; ++b) fsincos(x[7 + b - 1], &sn[b - 1], &cs[b - 1]);
          if constexpr (CT == CT_PAIR3) {
            if (set.ncontacts > 1) {
              FrameCap<double> ck2;
#if EMPC_ROLL_CAP_LDS
              const int ccap2 = si.ccap >> 8;
              const double* CAPd = CAP + (size_t)((ccap2 >= 1 && ccap2 < NCAP) ? ccap2 : 0) * 24 * NL;
#else
              const double* CAPd = N + SM::OFF_CAPB;
#endif
#pragma unroll
              for (int i = 0; i < 9; ++i) ck2.R[i] = CAPd[i * NL + lane];
#pragma unroll
              for (int i = 0; i < 3; ++i) ck2.p[i] = CAPd[(9 + i) * NL + lane];
#pragma unroll
              for (int i = 0; i < 6; ++i) ck2.v[i] = CAPd[(12 + i) * NL + lane];
#pragma unroll
              for (int i = 0; i < 6; ++i) ck2.a[i] = CAPd[(18 + i) * NL + lane];
              contact_forward_pair3<DM>(m, set.contacts[0], set.contacts[1], ck, ck2, R0, x, cs, sn, Lc[sl], a, lam);
            } else {
              contact_forward<DM, 3>(m, set.contacts[0], ck, R0, x, cs, sn, Lc[sl], a, lam);
            }
          } else {
            contact_forward<DM, CT>(m, set.contacts[0], ck, R0, x, cs, sn, Lc[sl], a, lam);
          }
        }
      }
      R6_SUB(2);
#pragma unroll
      for (int i = 0; i < NV; ++i) ACC[i * NL + lane] = a[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) ACC[(NV + i) * NL + lane] = lam[i];
      bool final_step = !terminal;
      if constexpr (RK4) {
        // k_i = [v(y_i); a_i]; sum w_i k_i; the next stage state y_{i+1} = x (+) c_{i+1} dt k_i (the terminal node evaluates its
        // four stages too: its cost is their weighted sum)
        const double rk4_c[4] = {0.0, 0.5, 0.5, 1.0};
        const double w = (st == 0 || st == 3) ? 1.0 : 2.0;
        double kst[NDX];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          kst[i] = x[NQ + i];
          kst[NV + i] = a[i];
        }
        if (st == 0) {
#pragma unroll
          for (int i = 0; i < NX; ++i) xkC[sl][i] = x[i];
#pragma unroll
          for (int i = 0; i < NDX; ++i) ksC[sl][i] = kst[i];
        } else {
#pragma unroll
          for (int i = 0; i < NDX; ++i) ksC[sl][i] = ksC[sl][i] + w * kst[i];
        }
        if (st < 3) {
          double dxr[NDX], y[NX];
#pragma unroll
          for (int i = 0; i < NDX; ++i) dxr[i] = rk4_c[st + 1] * dt * kst[i];
          state_integrate<DM>(xkC[sl], dxr, y, nullptr);
#pragma unroll
          for (int i = 0; i < NX; ++i) XT[i * NL + lane] = y[i];
          final_step = false;
        }
      }
      if (final_step) {
        const double* v = x + NQ;
        double dxe[NDX], xn[NX], xt[NX];
        if constexpr (RK4) {
#pragma unroll
          for (int i = 0; i < NDX; ++i) dxe[i] = ksC[sl][i] * dt / 6.0;
          state_integrate<DM>(xkC[sl], dxe, xn, nullptr);
        } else {
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            dxe[i] = v[i] * dt + a[i] * dt * dt;
            dxe[NV + i] = a[i] * dt;
          }
          state_integrate<DM>(x, dxe, xn, nullptr);
        }
        double mx = 0;
        bool isn = false;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          mx = fmax(mx, fabs(xn[i]));
          isn = isn || is_nan(xn[i]);
        }
        if (isn || bad_number(mx)) {
          if (okC[sl]) ncC[sl] = (t + 1 < T) ? t + 1 : T;
          okC[sl] = 0;
        }
        R6_SUB(3);
        if (L.plain) {
#pragma unroll
          for (int i = 0; i < NX; ++i) xt[i] = xn[i];
        } else {
#ifdef EMPC_ROLL_GAP_EARLY
          state_integrate_pre<DM>(xn, stepC[sl], qeC[sl], peC[sl], xt);
#else
          const double* gapn = NOM + (size_t)((t + 1) & 1) * SM::NOMSZ + SM::NOM_GAP + L.g * SM::GS;
          double step[NDX];
#pragma unroll
          for (int i = 0; i < NDX; ++i) step[i] = gapn[i] * (L.alpha - 1.0);
          state_integrate<DM>(xn, step, xt, nullptr);
#endif
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) XT[i * NL + lane] = xt[i];
      }
      R6_SUB(4);
    });
    // ---- B: next knot's gain rows into the staging buffer (A reads them in phase I of the next knot) ---------------------------
    if constexpr (ROLE == R6_B) ex.each([&](int lane, int sl) {
      if (t + 1 < T && last_stage) {
        fetch_gains(t + 1, sl);
        R6_SCHED_FENCE();
        put_gains(t + 1, lane, sl);
      }
    });
    // ---- A: the rest of the next knot's nominal data (xs, Vxx f, us, k) -------------------------------------------------------
    if constexpr (ROLE == R6_A) ex.each([&](int lane, int sl) {
      if (t < T && last_stage) {
        fetch_nom(t + 1, sl);
        R6_SCHED_FENCE();
        put_nom(t + 1, lane, sl);
      }
    });
    // ---- D: Control costs of this knot, control of the trial to memory -----------------------------------------------------------
    if constexpr (ROLE == R6_D) ex.each([&](int lane, int sl) {
      const Roll6Lane& L = LL[sl];
      if (!L.live) return;
      R6_SUB(6);
      // the rest of the State costs of this knot / stage (see phase I)
      for (int ks = (si.n_state + 1) / 2; ks < si.n_state; ++ks) {
        const int ci = si.state_ci[ks];
        const auto& c = set.costs[ci];
        if (!(c.ref_share >= 0 && c.ref_share == rsofD[sl])) {
          state_diff<DM>(c.ref, xD[sl], rsD[sl], nullptr);
          rsofD[sl] = (c.ref_share >= 0) ? c.ref_share : ci;
        }
        VAL[ci * NL + lane] = activation_value<NDX>(c, rsD[sl], NDX);
      }
      if (st > 0) return;  // (RK4 nodes: the control and its cost values are those of stage 0 for all four stages)
      double s[NU];
#pragma unroll
      for (int i = 0; i < NU; ++i) s[i] = terminal ? 0.0 : UT[i * NL + lane];
      for (int kc = 0; kc < si.n_ctrl; ++kc) {
        const int ci = si.ctrl_ci[kc];
        VAL[ci * NL + lane] = control_cost_value<NU>(set.costs[ci], s, L.smooth, PL);
      }
      if (!terminal) {
        double* us_o = D.us_try + ((size_t)L.b * NA + L.ai) * T * NU + (size_t)t * NU;
#pragma unroll
        for (int i = 0; i < NU; ++i) us_o[i] = s[i];
      }
      R6_SUB(3);
    });
    R6_STAMP(2);
    ex.sync();
    R6_STAMP(3);
    }  // stages of the knot
  }
#if defined(EMPC_STAMPS) && defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
  if (group == 0 && D.dbg)
    ex.each([&](int lane, int sl) {
      if (lane == 0)
        for (int i = 0; i < 4; ++i) D.dbg[48 + ROLE * 4 + i] = r6st[i];
      if (lane == 0 && ROLE == R6_B)
        for (int i = 0; i < 7; ++i) D.dbg[i] = r6b[i];
      if (lane == 0)
        for (int i = 0; i < 7; ++i) D.dbg[64 + ROLE * 8 + i] = r6b[i];
    });
#endif
  // ---- results of the trials -------------------------------------------------------------------------------------------------------
  // (C hands D the number of running nodes it reached, T + 1 = the rollout stayed finite)
  if constexpr (ROLE == R6_C) ex.each([&](int lane, int sl) { FLAG[lane] = okC[sl] ? (double)(T + 1) : (double)ncC[sl]; });
  if constexpr (ROLE == R6_A) ex.each([&](int lane, int sl) {
    const Roll6Lane& L = LL[sl];
    if (L.live) D.try_dv[(size_t)L.b * NA + L.ai] = dvA[sl];
  });
  ex.sync();
  if constexpr (ROLE == R6_D) ex.each([&](int lane, int sl) {
    const Roll6Lane& L = LL[sl];
    if (!L.live) return;
    finish_cost(T, NST - 1, lane, sl, L);
    const size_t slot = (size_t)L.b * NA + L.ai;
    D.try_cost[slot] = costD[sl];
    const int ncc = (int)FLAG[lane];
    D.try_ok[slot] = (okD[sl] && ncc > T) ? 1 : 0;
    D.try_ncalc[slot] = (ncc < ncD[sl]) ? ncc : ncD[sl];
  });
}

}  // namespace empc
