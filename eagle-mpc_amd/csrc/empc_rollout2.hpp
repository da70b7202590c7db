// empc_rollout2.hpp -- HOT-C kernel body, second generation.
//
// One wavefront = TPB trajectories x LPT step-length slots (4 x 16).  Each lane still integrates its own rollout
// (SolverFDDP::forwardPass(alpha) / SolverSbFDDP::forwardPassDDP), but the per-knot inputs every step length of a
// trajectory needs -- gains K, k, the candidate (xs, us), Vxx f and the gap -- are staged once per knot in LDS by the
// whole wavefront (coalesced), double-buffered, with the next knot's block prefetched into registers while the current
// knot is integrated.  The feedback product K dx then reads K with broadcast LDS reads instead of 162 global loads
// per lane and knot.
#pragma once
#include "empc_kernels.hpp"

namespace empc {

template <class DM>
struct Roll2Smem {
  static constexpr int n = DM::NDX, m = DM::NU, NX = DM::NX;
  static constexpr int S_K = 0;
  static constexpr int S_KF = S_K + m * n;
  static constexpr int S_US = S_KF + m;
  static constexpr int S_XS = S_US + m;
  static constexpr int S_VF = S_XS + NX;
  static constexpr int S_GAP = S_VF + n;
  static constexpr int SLOT = S_GAP + n;
  static constexpr int TPB = 4;   // trajectories per wavefront
  static constexpr int LPT = 16;  // step-length slots per trajectory (>= n_alphas)
  static constexpr int BLK = TPB * SLOT;
  static constexpr int SIZE = 2 * BLK;
  static constexpr int PRE = (BLK + 63) / 64;
};

template <class DM>
EMPC_HD double roll2_fetch(const DevBuffers& D, int b, int t, int e) {
  typedef Roll2Smem<DM> SM;
  constexpr int n = DM::NDX, m = DM::NU, NX = DM::NX, REC = DM::REC;
  const int T = D.T;
  if (b >= D.B) return 0.0;
  if (e < SM::S_KF) return (t < T) ? D.K[((size_t)b * T + t) * m * n + e] : 0.0;
  if (e < SM::S_US) return (t < T) ? D.kff[((size_t)b * T + t) * m + (e - SM::S_KF)] : 0.0;
  if (e < SM::S_XS) return (t < T) ? D.us[((size_t)b * T + t) * m + (e - SM::S_US)] : 0.0;
  if (e < SM::S_VF) return D.xs[((size_t)b * (T + 1) + t) * NX + (e - SM::S_XS)];
  if (e < SM::S_GAP) return D.Vf[((size_t)b * (T + 1) + t) * n + (e - SM::S_VF)];
  return D.tape[((size_t)b * (T + 1) + t) * REC + DM::OFF_GAP + (e - SM::S_GAP)];
}

template <class DM, bool CT, class Exec>
EMPC_HD void rollout_block2(Exec& ex, const DevBuffers& D, int b0, double* smem) {
  typedef Roll2Smem<DM> SM;
  constexpr int NX = DM::NX, NU = DM::NU, NV = DM::NV, NDX = DM::NDX;
  const int T = D.T, NA = D.NA;
  // whole-wavefront early exit when none of its trajectories rolls out this sweep
  bool any_active = false;
  for (int tb = 0; tb < SM::TPB; ++tb) {
    const int b = b0 + tb;
    if (b < D.B && D.st[b].phase != PHASE_DONE && !D.st[b].bwd_failed) any_active = true;
  }
  if (!any_active) return;

  struct LaneSt {
    double xnext[NX];
    double cost_try, dv;
    int ok, live;
  };
  LaneSt LS[Exec::SLOTS];
  double pre[Exec::SLOTS][SM::PRE];

  ex.each([&](int lane, int sl) {
    const int tb = lane / SM::LPT, ai = lane % SM::LPT, b = b0 + tb;
    LaneSt& L = LS[sl];
    L.live = (b < D.B && ai < NA && D.st[b < D.B ? b : 0].phase != PHASE_DONE && !D.st[b < D.B ? b : 0].bwd_failed) ? 1 : 0;
    L.ok = 1;
    L.cost_try = 0;
    L.dv = 0;
    for (int i = 0; i < NX; ++i) L.xnext[i] = (b < D.B) ? D.x0[(size_t)b * NX + i] : 0.0;
    // knot 0 straight into buffer 0
    for (int idx = lane; idx < SM::BLK; idx += 64) smem[idx] = roll2_fetch<DM>(D, b0 + idx / SM::SLOT, 0, idx % SM::SLOT);
  });
  ex.sync();
  for (int t = 0; t <= T; ++t) {
    const double* buf = smem + (t & 1) * SM::BLK;
    double* nbuf = smem + ((t + 1) & 1) * SM::BLK;
    ex.each([&](int lane, int sl) {
      if (t < T) {
#pragma unroll
        for (int q = 0; q < SM::PRE; ++q) {
          const int idx = lane + q * 64;
          pre[sl][q] = (idx < SM::BLK) ? roll2_fetch<DM>(D, b0 + idx / SM::SLOT, t + 1, idx % SM::SLOT) : 0.0;
        }
      }
      LaneSt& L = LS[sl];
      if (L.live && L.ok) {
        const int tb = lane / SM::LPT, ai = lane % SM::LPT, b = b0 + tb;
        const TrajState& st = D.st[b];
        const bool ddp = (st.phase == PHASE_DDP);
        const bool feas = st.is_feasible != 0;
        const double alpha = ldexp(1.0, -ai);
        const bool plain = ddp || feas || (ai == 0);
        const double* S = buf + tb * SM::SLOT;
        double xtry[NX], dx[NDX], utry[NU], acc[NV], usq[NU], lam[6];
        if (plain) {
#pragma unroll
          for (int i = 0; i < NX; ++i) xtry[i] = L.xnext[i];
        } else {
          double step[NDX];
#pragma unroll
          for (int i = 0; i < NDX; ++i) step[i] = S[SM::S_GAP + i] * (alpha - 1.0);
          state_integrate<DM>(L.xnext, step, xtry, nullptr);
        }
        state_diff<DM>(S + SM::S_XS, xtry, dx, nullptr);
        if (!ddp && !feas) {
          double a_ = 0;
#pragma unroll
          for (int i = 0; i < NDX; ++i) a_ += S[SM::S_VF + i] * dx[i];
          L.dv += a_;
        }
        const size_t slot = (size_t)b * NA + ai;
        double* xs_o = D.xs_try + slot * (T + 1) * NX;
        double* us_o = D.us_try + slot * T * NU;
        double* ac_o = D.acc_try + slot * (T + 1) * DM::NACC;
        double cost;
        const EMPC_K EmpcCostSet& set = EMPC_KPTR(EmpcCostSet, D.sets)[EMPC_KPTR(int, D.knot_set)[t]];
        if (t < T) {
#pragma unroll
          for (int i = 0; i < NU; ++i) {
            double a_ = S[SM::S_US + i] - S[SM::S_KF + i] * alpha;
#pragma unroll
            for (int j = 0; j < NDX; ++j) a_ -= S[SM::S_K + i * NDX + j] * dx[j];
            utry[i] = a_;
          }
          node_nominal<DM, CT>(EMPC_KREF(DevProblem, D.P), set, st.smooth, xtry, utry, false, L.xnext, acc, cost, usq, lam);
#pragma unroll
          for (int i = 0; i < NU; ++i) us_o[(size_t)t * NU + i] = utry[i];
        } else {
          double xn2[NX];
          node_nominal<DM, CT>(EMPC_KREF(DevProblem, D.P), set, st.smooth, xtry, nullptr, true, xn2, acc, cost, usq, lam);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) xs_o[(size_t)t * NX + i] = xtry[i];
#pragma unroll
        for (int i = 0; i < NV; ++i) ac_o[(size_t)t * DM::NACC + i] = acc[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) ac_o[(size_t)t * DM::NACC + NV + i] = lam[i];
        L.cost_try += cost;
        if (bad_number(L.cost_try)) L.ok = 0;
        if (t < T) {
          double mx = 0;
          bool isn = false;
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            mx = fmax(mx, fabs(L.xnext[i]));
            isn = isn || (L.xnext[i] != L.xnext[i]);
          }
          if (isn || bad_number(mx)) L.ok = 0;
        }
      }
      if (t < T) {
#pragma unroll
        for (int q = 0; q < SM::PRE; ++q) {
          const int idx = lane + q * 64;
          if (idx < SM::BLK) nbuf[idx] = pre[sl][q];
        }
      }
    });
    ex.sync();
  }
  ex.each([&](int lane, int sl) {
    const LaneSt& L = LS[sl];
    if (!L.live) return;
    const int tb = lane / SM::LPT, ai = lane % SM::LPT, b = b0 + tb;
    const size_t slot = (size_t)b * NA + ai;
    D.try_cost[slot] = L.cost_try;
    D.try_dv[slot] = L.dv;
    D.try_ok[slot] = L.ok;
  });
}

}  // namespace empc
