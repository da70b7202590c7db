// TEST INFRASTRUCTURE (superseded kernel form: the r01 wave-per-trajectory rollout, kept as a cross-check of the shipped
// packed form through the CPU lane emulator; not compiled into libempc.so)
#pragma once
#include "../../../eagle-mpc_amd/csrc/empc_kernels.hpp"

namespace empc {

// ---------------------------------------------------------------------------------------------------------------------
// rollout, cooperative feedback (r01): one wavefront per trajectory, lanes 0..NA-1 own the step lengths.
// Measured on the per-lane form (profiles/r01_rollout_ablation.txt): 37 % of the time is the feedback product
// K[t] (xs_try (-) xs) -- every lane streams the same 162 doubles of K through too few registers, one exposed memory
// latency per chunk.  Here the whole wave does that product: lane g*NU + i keeps row i of K[t] in registers (fetched one
// knot ahead, 1 coalesced row per lane) and produces u_i for step lengths g, g + G, ...; the trial lanes only publish
// their dx to LDS and read their u back.  xs[t], the gap and Vxx f travel the same way (1 double per lane, one knot
// ahead, through LDS).
// ---------------------------------------------------------------------------------------------------------------------
template <class DM>
struct Roll5Smem {
  static constexpr int NX = DM::NX, NU = DM::NU, NDX = DM::NDX;
  static constexpr int IN_X = 0;                 // xs[t]
  static constexpr int IN_GAP = IN_X + NX;       // fs[t]
  static constexpr int IN_VF = IN_GAP + NDX;     // Vxx[t] fs[t]
  static constexpr int NIN = IN_VF + NDX;
  static constexpr int NPRE = (NIN + 63) / 64;
  static constexpr int OFF_IN = 0;               // two buffers of NIN
  static constexpr int OFF_DX = OFF_IN + 2 * NIN;                 // [MAX_ALPHAS][NDX]
  static constexpr int OFF_U = OFF_DX + MAX_ALPHAS * NDX;         // [MAX_ALPHAS][NU]
  static constexpr int SIZE = (OFF_U + MAX_ALPHAS * NU + 1) / 2 * 2;
};

template <class DM, int CT, class Exec>
EMPC_HD void rollout_wave5(Exec& ex, const DevBuffers& D, int b, int nl, double* N) {
  typedef Roll5Smem<DM> SM;
  const TrajState& st = D.st[b];
  if (st.phase == PHASE_DONE || st.bwd_failed) return;
  constexpr int NX = DM::NX, NU = DM::NU, NV = DM::NV, NDX = DM::NDX, REC = DM::REC;
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  const int T = D.T, NA = D.NA;
  const bool ddp = (st.phase == PHASE_DDP);
  const bool feas = st.is_feasible != 0;
  const bool need_dv = !ddp && !feas;
  const double smooth = st.smooth;
  const int G = nl / NU;  // step lengths served per pass of the feedback product
  RollLane<DM> L[Exec::SLOTS];
  double xtry_l[Exec::SLOTS][NX];
  double krow[Exec::SLOTS][NDX], ku[Exec::SLOTS], kk[Exec::SLOTS];  // row i of K[t], us[t][i], k[t][i]
  double pre[Exec::SLOTS][SM::NPRE];
  int alive[Exec::SLOTS];
  int ncalc_l[Exec::SLOTS];

  auto fetch_in = [&](int t, int i) -> double {
    if (i < SM::IN_GAP) return D.xs[((size_t)b * (T + 1) + t) * NX + (i - SM::IN_X)];
    if (i < SM::IN_VF) return D.tape[((size_t)b * (T + 1) + t) * REC + DM::OFF_GAP + (i - SM::IN_GAP)];
    return D.Vf[((size_t)b * (T + 1) + t) * NDX + (i - SM::IN_VF)];
  };
  auto fetch_row = [&](int t, int lane, int sl) {
    const int i = lane % NU;
    if (lane < G * NU && t < T) {
      const double* Kr = D.K + (((size_t)b * T + t) * NU + i) * NDX;
#pragma unroll
      for (int j = 0; j < NDX; ++j) krow[sl][j] = Kr[j];
      ku[sl] = D.us[((size_t)b * T + t) * NU + i];
      kk[sl] = D.kff[((size_t)b * T + t) * NU + i];
    }
  };
  ex.each([&](int lane, int sl) {
#pragma unroll
    for (int i = 0; i < NX; ++i) L[sl].xnext[i] = D.x0[(size_t)b * NX + i];
    L[sl].cost_try = 0;
    L[sl].dv = 0;
    L[sl].ok = 1;
    ncalc_l[sl] = T;
    alive[sl] = (lane < NA) ? 1 : 0;
#pragma unroll
    for (int k = 0; k < SM::NPRE; ++k) {
      const int i = lane + nl * k;
      if (i < SM::NIN) N[SM::OFF_IN + i] = fetch_in(0, i);
    }
    fetch_row(0, lane, sl);
  });
  ex.sync();
  for (int t = 0; t <= T; ++t) {
    const double* in = N + SM::OFF_IN + (t & 1) * SM::NIN;
    const EMPC_K EmpcCostSet& set = EMPC_KPTR(EmpcCostSet, D.sets)[EMPC_KPTR(int, D.knot_set)[t]];
    // ---- S1: trial state and its difference to the nominal one (trial lanes); next knot's staged inputs requested ----
    ex.each([&](int lane, int sl) {
      if (t < T) {
#pragma unroll
        for (int k = 0; k < SM::NPRE; ++k) {
          const int i = lane + nl * k;
          pre[sl][k] = (i < SM::NIN) ? fetch_in(t + 1, i) : 0.0;
        }
      }
      if (!alive[sl]) return;
      const int ai = lane;
      const double alpha = ldexp(1.0, -ai);
      const bool plain = ddp || feas || (ai == 0);
      double dx[NDX];
      if (plain) {
#pragma unroll
        for (int i = 0; i < NX; ++i) xtry_l[sl][i] = L[sl].xnext[i];
      } else {
        double step[NDX];
#pragma unroll
        for (int i = 0; i < NDX; ++i) step[i] = in[SM::IN_GAP + i] * (alpha - 1.0);
        state_integrate<DM>(L[sl].xnext, step, xtry_l[sl], nullptr);
      }
      state_diff<DM>(in + SM::IN_X, xtry_l[sl], dx, nullptr);
      if (need_dv) {
        double dv = L[sl].dv;
#pragma unroll
        for (int i = 0; i < NDX; ++i) dv += in[SM::IN_VF + i] * dx[i];  // +(Vxx f).(xs_try (-) xs)
        L[sl].dv = dv;
      }
#pragma unroll
      for (int i = 0; i < NDX; ++i) N[SM::OFF_DX + ai * NDX + i] = dx[i];
    });
    ex.sync();
    // ---- S2: feedback product by the whole wave: u_i = us_i - alpha k_i - K_i . dx(alpha) --------------------------
    if (t < T) {
      ex.each([&](int lane, int sl) {
        if (lane < G * NU) {
          const int g = lane / NU, i = lane % NU;
          for (int a = g; a < NA; a += G) {
            const double* dxa = N + SM::OFF_DX + a * NDX;
            double a_ = ku[sl] - kk[sl] * ldexp(1.0, -a);
#pragma unroll
            for (int j = 0; j < NDX; ++j) a_ -= krow[sl][j] * dxa[j];
            N[SM::OFF_U + a * NU + i] = a_;
          }
        }
        fetch_row(t + 1, lane, sl);  // consumed at S2 of the next knot; in flight during S3
      });
      ex.sync();
    }
    // ---- S3: the node itself (trial lanes) ------------------------------------------------------------------------------
    ex.each([&](int lane, int sl) {
      if (alive[sl]) {
        const int ai = lane;
        const size_t slot = (size_t)b * NA + ai;
        double* xs_o = D.xs_try + slot * (T + 1) * NX;
        double* us_o = D.us_try + slot * T * NU;
        double* ac_o = D.acc_try + slot * (T + 1) * DM::NACC;
        double utry[NU], acc[NV], usq[NU], lam[6], cost;
        if (t < T) {
#pragma unroll
          for (int i = 0; i < NU; ++i) utry[i] = N[SM::OFF_U + ai * NU + i];
          node_nominal<DM, CT>(P, set, smooth, xtry_l[sl], utry, false, L[sl].xnext, acc, cost, usq, lam);
#pragma unroll
          for (int i = 0; i < NU; ++i) us_o[(size_t)t * NU + i] = utry[i];
        } else {
          double xn2[NX];
          node_nominal<DM, CT>(P, set, smooth, xtry_l[sl], (const double*)nullptr, true, xn2, acc, cost, usq, lam);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) xs_o[(size_t)t * NX + i] = xtry_l[sl][i];
#pragma unroll
        for (int i = 0; i < NV; ++i) ac_o[(size_t)t * DM::NACC + i] = acc[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) ac_o[(size_t)t * DM::NACC + NV + i] = lam[i];
        L[sl].cost_try += cost;
        if (bad_number(L[sl].cost_try)) {
          if (L[sl].ok) ncalc_l[sl] = (t + 1 < T) ? t + 1 : T;
          L[sl].ok = 0;
          alive[sl] = 0;
        } else if (t < T) {
          double mx = 0;
          bool isn = false;
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            mx = fmax(mx, fabs(L[sl].xnext[i]));
            isn = isn || (L[sl].xnext[i] != L[sl].xnext[i]);
          }
          if (isn || bad_number(mx)) {
            if (L[sl].ok) ncalc_l[sl] = (t + 1 < T) ? t + 1 : T;
            L[sl].ok = 0;
            alive[sl] = 0;
          }
        }
      }
      if (t < T) {
        double* nxt = N + SM::OFF_IN + ((t + 1) & 1) * SM::NIN;
#pragma unroll
        for (int k = 0; k < SM::NPRE; ++k) {
          const int i = lane + nl * k;
          if (i < SM::NIN) nxt[i] = pre[sl][k];
        }
      }
    });
    ex.sync();
  }
  ex.each([&](int lane, int sl) {
    if (lane >= NA) return;
    const size_t slot = (size_t)b * NA + lane;
    D.try_cost[slot] = L[sl].cost_try;
    D.try_dv[slot] = L[sl].dv;
    D.try_ok[slot] = L[sl].ok;
    D.try_ncalc[slot] = L[sl].ok ? T : ncalc_l[sl];
  });
}


}  // namespace empc
